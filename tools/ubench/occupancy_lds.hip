// How many 1024-thread workgroups with a given amount of dynamic LDS share a CU?  512 workgroups that each spin for a fixed
// number of cycles: the kernel's duration / one workgroup's duration = "rounds".   hipcc --offload-arch=gfx950 -O2 occupancy_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void spin(unsigned long long cycles, unsigned* sink) {
  extern __shared__ unsigned smem[];
  smem[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned acc = 0;
  while (__builtin_readcyclecounter() - t0 < cycles) acc += smem[(threadIdx.x + acc) & 1023];
  if (acc == 0xFFFFFFFFu) *sink = acc;
}
int main() {
  unsigned* d; hipMalloc(&d, 4);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int sizes[] = {16, 48, 64, 76, 78, 80, 82, 90, 100};
  for (int threads : {1024, 512, 256}) for (int kb : sizes) {
    const size_t lds = (size_t)kb * 1024;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(spin, dim3(512), dim3(threads), lds, 0, 5000ull /* 100 MHz ticks = 50 us */, d);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep) printf("threads %4d LDS %3d KB: %.3f ms (hipGetLastError %d)\n", threads, kb, ms, (int)hipGetLastError());
    }
  }
  return 0;
}
