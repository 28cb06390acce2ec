// Micro-benchmark: issue rate of the VALU ops k6_grid_cost is made of (gfx950).
// Each kernel runs ITER x 32 independent ops of one kind per lane; 256 CUs x 8 blocks x 256 thr.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; b[i] = seed * 0.5f + i; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) { asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_add_f32 %0, %1, %0" : "+v"(b[i]) : "v"(a[i])); }
      if (KIND == 1) { asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(b[i]) : "v"(a[i])); }
      if (KIND == 2) { asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_min_f32 %0, %1, %0" : "+v"(b[i]) : "v"(a[i])); }
      if (KIND == 3) { asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(b[i]) : "v"(a[i])); }
      if (KIND == 4) { asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i])); asm volatile("v_rndne_f32 %0, %0" : "+v"(b[i])); }
      if (KIND == 5) { asm volatile("v_add_f32_e64 %0, |%1|, |%0|" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_add_f32_e64 %0, |%1|, |%0|" : "+v"(b[i]) : "v"(a[i])); }
      if (KIND == 8) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b[i]) : "vcc"); asm volatile("v_cmp_lt_f32 vcc, %1, %0" :: "v"(a[i]), "v"(b[i]) : "vcc"); }
      if (KIND == 9) { asm volatile("v_cndmask_b32_e64 %0, %1, %0, s[20:21]" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_cndmask_b32_e64 %0, %1, %0, s[20:21]" : "+v"(b[i]) : "v"(a[i])); }
    }
    if (KIND == 6 || KIND == 7) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 x = {a[i], a[i + 1]}, y = {b[i], b[i + 1]};
        if (KIND == 6) { asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(x) : "v"(y)); asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(y) : "v"(x));
                         asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(x) : "v"(y)); asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(y) : "v"(x)); }
        if (KIND == 7) { asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(x) : "v"(y)); asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(y) : "v"(x));
                         asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(x) : "v"(y)); asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(y) : "v"(x)); }
        a[i] = x.x; a[i + 1] = x.y; b[i] = y.x; b[i + 1] = y.y;
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, float* d, int blocks, double lane_ops_per_instr) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)blocks * 256 * ITER * 32;   // per-lane instructions
  const double rate = instr / (ms * 1e-3);                  // lane-instr/s
  printf("%-22s %8.3f ms  %7.2f T lane-instr/s  = %6.1f%% of 78.6T (256CU*4SIMD*32lanes*2.4GHz)  [%.2f T elem-ops/s]\n", name, ms, rate / 1e12,
         100.0 * rate / 78.6e12, rate * lane_ops_per_instr / 1e12);
}

int main() {
  float* d; hipMalloc(&d, 256 * 64 * 256 * sizeof(float));
  for (int blocks : {256 * 4, 256 * 8}) {
    printf("blocks=%d (x256 threads)\n", blocks);
    run<0>("v_add_f32", d, blocks, 1);
    run<1>("v_fma_f32", d, blocks, 1);
    run<2>("v_min_f32", d, blocks, 1);
    run<3>("v_cndmask_b32 vcc", d, blocks, 1);
    run<9>("v_cndmask_b32 sgpr", d, blocks, 1);
    run<4>("v_rndne_f32", d, blocks, 1);
    run<5>("v_add_f32 |a|,|b|", d, blocks, 1);
    run<8>("v_cmp_lt_f32", d, blocks, 1);
    run<6>("v_pk_add_f32", d, blocks, 2);
    run<7>("v_pk_fma_f32", d, blocks, 2);
  }
  return 0;
}
