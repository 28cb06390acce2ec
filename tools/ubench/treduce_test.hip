// checks the register-only transposed reduction of k6_grid_cost against a host sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define ILCC_WAVE 64
constexpr int kAcc = 32;
__device__ __forceinline__ float swap32_add(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
template <int VARIANT>
__device__ __forceinline__ float dpp_xor4(float v) {
  if (VARIANT == 0) {
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x124, 0xf, 0x5, false);
    return __uint_as_float(__builtin_amdgcn_update_dpp(lo, __float_as_uint(v), 0x12C, 0xf, 0xa, false));
  } else {
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x124, 0xf, 0xa, false);
    return __uint_as_float(__builtin_amdgcn_update_dpp(lo, __float_as_uint(v), 0x12C, 0xf, 0x5, false));
  }
}
template <int VARIANT>
__global__ void k(const float* in, float* out, float* probe) {
  const int lane = threadIdx.x;
  float acc[kAcc];
  for (int k = 0; k < kAcc; ++k) acc[k] = in[k * 64 + lane];
  probe[lane] = dpp_xor4<VARIANT>((float)lane);          // should be lane^4
  probe[64 + lane] = dpp<0x128>((float)lane);            // lane^8
  probe[128 + lane] = dpp<0x4E>((float)lane);            // lane^2
  probe[192 + lane] = dpp<0xB1>((float)lane);            // lane^1
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = swap32_add(acc[k], acc[k + 16]);
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = swap16_add(acc[k], acc[k + 8]);
  { const bool up = (lane & 8) != 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float s0 = acc[k] + dpp<0x128>(acc[k]); const float s1 = acc[k + 4] + dpp<0x128>(acc[k + 4]); acc[k] = up ? s1 : s0; } }
  { const bool up = (lane & 4) != 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) { const float s0 = acc[k] + dpp_xor4<VARIANT>(acc[k]); const float s1 = acc[k + 2] + dpp_xor4<VARIANT>(acc[k + 2]); acc[k] = up ? s1 : s0; } }
  { const bool up = (lane & 2) != 0; const float s0 = acc[0] + dpp<0x4E>(acc[0]); const float s1 = acc[1] + dpp<0x4E>(acc[1]); acc[0] = up ? s1 : s0; }
  acc[0] = acc[0] + dpp<0xB1>(acc[0]);
  out[lane] = acc[0];
}
int main() {
  float h[32 * 64], *d, *o, *p, ho[64], hp[256];
  for (int i = 0; i < 32 * 64; ++i) h[i] = (float)((i * 37) % 101) * 0.25f;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 256); hipMalloc(&p, 1024);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int variant = 0; variant < 2; ++variant) {
    if (variant == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, o, p); else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, o, p);
    hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost); hipMemcpy(hp, p, 1024, hipMemcpyDeviceToHost);
    int bad = 0, badx4 = 0, badx8 = 0, badx2 = 0, badx1 = 0;
    for (int l = 0; l < 64; ++l) {
      double s = 0; for (int j = 0; j < 64; ++j) s += h[(l >> 1) * 64 + j];
      if (std::fabs(ho[l] - s) > 1e-3) ++bad;
      badx4 += hp[l] != (float)(l ^ 4); badx8 += hp[64 + l] != (float)(l ^ 8); badx2 += hp[128 + l] != (float)(l ^ 2); badx1 += hp[192 + l] != (float)(l ^ 1);
    }
    printf("variant %d: bad totals %d/64; xor4 wrong %d, xor8 wrong %d, xor2 wrong %d, xor1 wrong %d\n", variant, bad, badx4, badx8, badx2, badx1);
  }
  return 0;
}
