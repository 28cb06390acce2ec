// Micro-benchmark 2: which VALU ops of k6_grid_cost issue at full rate on gfx950, and whether a
// "slow" op overlaps with a "fast" one from the same wave / other waves.
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 4096
#define OP2(str) asm volatile(str : "+v"(a[i]) : "v"(b[i])); asm volatile(str : "+v"(b[i]) : "v"(a[i]));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; b[i] = seed * 0.5f + i; }
  asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[0]), "v"(b[0]) : "vcc");
  asm volatile("s_mov_b64 s[20:21], vcc" ::: "s20", "s21");
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) { OP2("v_add_f32 %0, %1, %0") }
      if (KIND == 1) { OP2("v_cndmask_b32 %0, %1, %0, vcc") }
      if (KIND == 2) { OP2("v_cndmask_b32_e64 %0, %1, %0, s[20:21]") }
      if (KIND == 3) { OP2("v_min_f32 %0, %1, %0") }
      if (KIND == 4) { OP2("v_max_f32 %0, %1, %0") }
      if (KIND == 5) { OP2("v_and_b32 %0, %1, %0") }
      if (KIND == 6) { OP2("v_xor_b32 %0, %1, %0") }
      if (KIND == 7) { OP2("v_mul_f32 %0, %1, %0") }
      if (KIND == 8) { OP2("v_sub_f32 %0, %1, %0") }
      if (KIND == 9) { asm volatile("v_fract_f32 %0, %0" : "+v"(a[i])); asm volatile("v_fract_f32 %0, %0" : "+v"(b[i])); }
      if (KIND == 10) { asm volatile("v_floor_f32 %0, %0" : "+v"(a[i])); asm volatile("v_floor_f32 %0, %0" : "+v"(b[i])); }
      if (KIND == 11) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b[i]) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b[i]) : "vcc"); }
      if (KIND == 12) { asm volatile("v_cmp_lt_f32_e64 s[22:23], %0, %1" :: "v"(a[i]), "v"(b[i]) : "s22", "s23"); asm volatile("v_cmp_gt_f32_e64 s[24:25], %0, %1" :: "v"(a[i]), "v"(b[i]) : "s24", "s25"); }
      if (KIND == 13) { asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_min_f32 %0, %1, %0" : "+v"(b[i]) : "v"(a[i])); }   // alternate fast/slow, dependent
      if (KIND == 14) { asm volatile("v_med3_f32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_med3_f32 %0, %1, %0, %1" : "+v"(b[i]) : "v"(a[i])); }
      if (KIND == 15) { asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(b[i]) : "v"(a[i])); }
      if (KIND == 16) { OP2("v_fmac_f32 %0, %1, %0") }
      if (KIND == 17) { asm volatile("v_add_f32_e64 %0, |%1|, |%0|" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_add_f32_e64 %0, |%1|, |%0|" : "+v"(b[i]) : "v"(a[i])); }
      if (KIND == 18) { asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i])); asm volatile("v_rndne_f32 %0, %0" : "+v"(b[i])); }
      if (KIND == 19) { OP2("v_mov_b32 %0, %1") }
      if (KIND == 20) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_add_u32 %0, %1, %0" : "+v"(b[i]) : "v"(a[i])); }
      if (KIND == 21) { asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(b[i]) : "v"(a[i])); }
      // round 3: the ops a re-formulated K6 term would use
      if (KIND == 22) { OP2("v_ashrrev_i32 %0, 31, %0") }
      if (KIND == 23) { asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(b[i]) : "v"(a[i])); }
      if (KIND == 24) { asm volatile("v_cvt_flr_i32_f32 %0, %0" : "+v"(a[i])); asm volatile("v_cvt_flr_i32_f32 %0, %0" : "+v"(b[i])); }
      if (KIND == 25) { asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(b[i]) : "v"(a[i])); }
      if (KIND == 26) { asm volatile("v_sub_f32_e64 %0, |%1|, %0" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_sub_f32_e64 %0, |%1|, %0" : "+v"(b[i]) : "v"(a[i])); }
      if (KIND == 27) { OP2("v_lshlrev_b32 %0, 1, %0") }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// packed fp32 (VOP3P): two floats per lane and instruction.  lane-instr counted per INSTRUCTION (x2 for flops)
template <int KIND>
__global__ __launch_bounds__(256) void kp(float* out, float seed) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = v2f{seed + threadIdx.x * 1e-3f + i, seed + i}; b[i] = v2f{seed * 0.5f + i, seed * 0.25f + i}; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) { asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(b[i]) : "v"(a[i])); }
        if (KIND == 1) { asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(b[i]) : "v"(a[i])); }
        if (KIND == 2) { asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(b[i])); asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(b[i]) : "v"(a[i])); }
      }
    }
  }
  v2f s = {0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <int KIND>
void runp(const char* name, float* d, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kp<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kp<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)blocks * 256 * ITER * 32;
  printf("%-34s %8.3f ms  %7.2f T lane-instr/s (x2 floats each)\n", name, ms, instr / (ms * 1e-3) / 1e12);
}

template <int KIND>
void run(const char* name, float* d, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)blocks * 256 * ITER * 32;
  printf("%-34s %8.3f ms  %7.2f T lane-instr/s\n", name, ms, instr / (ms * 1e-3) / 1e12);
}

int main() {
  float* d; hipMalloc(&d, 256 * 64 * 256 * sizeof(float));
  const int blocks = 2048;
  run<0>("v_add_f32", d, blocks);  run<21>("v_fma_f32", d, blocks); run<7>("v_mul_f32", d, blocks); run<8>("v_sub_f32", d, blocks);
  run<16>("v_fmac_f32", d, blocks); run<17>("v_add_f32_e64 |a|,|b|", d, blocks);
  run<1>("v_cndmask_b32 (vcc)", d, blocks); run<2>("v_cndmask_b32_e64 (sgpr pair)", d, blocks);
  run<3>("v_min_f32", d, blocks); run<4>("v_max_f32", d, blocks); run<14>("v_med3_f32", d, blocks);
  run<5>("v_and_b32", d, blocks); run<6>("v_xor_b32", d, blocks); run<19>("v_mov_b32", d, blocks); run<20>("v_add_u32", d, blocks);
  run<9>("v_fract_f32", d, blocks); run<10>("v_floor_f32", d, blocks); run<18>("v_rndne_f32", d, blocks);
  run<11>("v_cmp_f32 (vcc)", d, blocks); run<12>("v_cmp_f32_e64 (sgpr pair)", d, blocks);
  run<22>("v_ashrrev_i32", d, blocks); run<23>("v_bfi_b32", d, blocks); run<24>("v_cvt_flr_i32_f32", d, blocks); run<25>("v_add_f32_dpp quad_perm", d, blocks);
  run<26>("v_sub_f32_e64 |a|, b", d, blocks); run<27>("v_lshlrev_b32", d, blocks);
  runp<0>("v_pk_add_f32", d, blocks); runp<1>("v_pk_mul_f32", d, blocks); runp<2>("v_pk_fma_f32", d, blocks);
  run<13>("alternate v_add / v_min", d, blocks); run<15>("alternate v_fma / v_cndmask", d, blocks);
  return 0;
}
