// K0 unpack_pointcloud2 -- replaces pcl::fromROSMsg(msg, pcl::PointCloud<pcl::PointXYZI>)
// (/root/reference/ilcc2/test/get_lidar_corners.cpp:163-164) for the device-resident path:
// sensor_msgs/PointCloud2 data[] (point_step bytes per point, row_step bytes per row) ->
// packed float4 {x, y, z, intensity}.  Unmatched fields stay 0 (pcl::PointXYZI's constructor zeroes them).
//
// HBM-bound: point_step bytes read + 16 bytes written per point (velodyne: 32 + 16).
//  * fast path (point_step % 16 == 0, field offsets % 4 == 0, base 16-byte aligned, rows packed):
//    a workgroup stages kTile points through LDS with 16-byte loads in memory order -- every lane of a
//    wavefront reads consecutive 16-byte pieces (1 KiB per wave-load, fully coalesced) -- then each
//    lane picks its point's four dwords out of LDS and writes one float4 (coalesced).
//  * generic path: four 4-byte reads per point assembled from bytes (any alignment, any padding).
// Launch: one workgroup of 256 threads per 1024 points -> 29 workgroups per VLP-16 message; a batch of
// messages is one launch (height := messages * height), >= 3.6k workgroups at 128 frames.
#include <hip/hip_runtime.h>

#include "ilcc_ingest.h"
#include "ilcc_internal.h"

namespace ilcc {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kUnpackThreads = 256;
constexpr int kUnpackTile = 1024;   // points per workgroup

struct UnpackArgs {
  const uint8_t* src;
  float4* dst;
  uint64_t n_points;
  uint32_t width, point_step, row_step;
  uint32_t off[4];   // x y z intensity, ILCC_FIELD_ABSENT -> 0
};

__device__ __forceinline__ float load_f32_bytes(const uint8_t* p) {
  const uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
  return __uint_as_float(v);
}

__global__ __launch_bounds__(kUnpackThreads) void k0_unpack_generic(UnpackArgs a) {
  const uint64_t i = (uint64_t)blockIdx.x * kUnpackThreads + threadIdx.x;
  if (i >= a.n_points) return;
  const uint64_t row = i / a.width, col = i - row * a.width;
  const uint8_t* p = a.src + row * a.row_step + col * a.point_step;
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (a.off[k] == ILCC_FIELD_ABSENT) ? 0.f : load_f32_bytes(p + a.off[k]);
  a.dst[i] = make_float4(v[0], v[1], v[2], v[3]);
}

// STEP16 = point_step / 16 (1..4): points are packed back to back (row_step == width * point_step)
template <int STEP16>
__global__ __launch_bounds__(kUnpackThreads) void k0_unpack_tiled(UnpackArgs a) {
  __shared__ u32x4 tile[kUnpackTile * STEP16];
  const uint64_t p0 = (uint64_t)blockIdx.x * kUnpackTile;
  const uint64_t left = a.n_points - p0;
  const uint32_t pts = left < (uint64_t)kUnpackTile ? (uint32_t)left : (uint32_t)kUnpackTile;
  const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(a.src) + p0 * STEP16;
  const uint32_t pieces = pts * STEP16;
#pragma unroll
  for (int k = 0; k < (kUnpackTile * STEP16) / kUnpackThreads; ++k) {
    const uint32_t j = k * kUnpackThreads + threadIdx.x;
    if (j < pieces) tile[j] = __builtin_nontemporal_load(src + j);
  }
  __syncthreads();
  const uint32_t* words = reinterpret_cast<const uint32_t*>(tile);
#pragma unroll
  for (int k = 0; k < kUnpackTile / kUnpackThreads; ++k) {
    const uint32_t j = k * kUnpackThreads + threadIdx.x;
    if (j < pts) {
      const uint32_t* w = words + j * (STEP16 * 4);
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = (a.off[c] == ILCC_FIELD_ABSENT) ? 0.f : __uint_as_float(w[a.off[c] >> 2]);
      a.dst[p0 + j] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

void set_global_error(const std::string& s);

}  // namespace ilcc

extern "C" int32_t ilcc_pointcloud2_unpack_device(const void* d_data, const ilcc_pointcloud2_layout* L, void* d_xyzi,
                                                  void* hip_stream) {
  using namespace ilcc;
  if (!L || (!d_data && L->data_bytes) || !d_xyzi) {
    set_global_error("null argument");
    return ILCC_BAD_ARGUMENT;
  }
  if (L->is_bigendian) {   // pcl::fromROSMsg memcpy's fields: it is wrong on such data too; refuse instead
    set_global_error("big-endian PointCloud2 is not supported");
    return ILCC_BAD_ARGUMENT;
  }
  UnpackArgs a;
  a.src = (const uint8_t*)d_data;
  a.dst = (float4*)d_xyzi;
  a.n_points = (uint64_t)L->height * L->width;
  a.width = L->width;
  a.point_step = L->point_step;
  a.row_step = L->row_step;
  a.off[0] = L->off_x;
  a.off[1] = L->off_y;
  a.off[2] = L->off_z;
  a.off[3] = L->off_intensity;
  if (a.n_points == 0) return ILCC_OK;
  hipStream_t s = (hipStream_t)hip_stream;
  bool aligned = (L->point_step % 16 == 0) && L->point_step <= 64 && ((uintptr_t)d_data % 16 == 0) &&
                 (L->height == 1 || L->row_step == (uint64_t)L->width * L->point_step);
  for (int k = 0; k < 4; ++k) aligned = aligned && (a.off[k] == ILCC_FIELD_ABSENT || a.off[k] % 4 == 0);
  if (aligned) {
    const uint32_t blocks = (uint32_t)((a.n_points + kUnpackTile - 1) / kUnpackTile);
    switch (L->point_step / 16) {
      case 1: hipLaunchKernelGGL(k0_unpack_tiled<1>, dim3(blocks), dim3(kUnpackThreads), 0, s, a); break;
      case 2: hipLaunchKernelGGL(k0_unpack_tiled<2>, dim3(blocks), dim3(kUnpackThreads), 0, s, a); break;
      case 3: hipLaunchKernelGGL(k0_unpack_tiled<3>, dim3(blocks), dim3(kUnpackThreads), 0, s, a); break;
      default: hipLaunchKernelGGL(k0_unpack_tiled<4>, dim3(blocks), dim3(kUnpackThreads), 0, s, a); break;
    }
  } else {
    const uint32_t blocks = (uint32_t)((a.n_points + kUnpackThreads - 1) / kUnpackThreads);
    hipLaunchKernelGGL(k0_unpack_generic, dim3(blocks), dim3(kUnpackThreads), 0, s, a);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_global_error(std::string("k0 launch: ") + hipGetErrorString(e));
    return ILCC_HIP_ERROR;
  }
  return ILCC_OK;
}

extern "C" int32_t ilcc_bag_first_cloud(int32_t device, const char* bag_path, const char* topic, float* xyzi,
                                        uint32_t cap_points, uint32_t* n_points) {
  using namespace ilcc;
  if (!n_points || (!xyzi && cap_points)) {
    set_global_error("null argument");
    return ILCC_BAD_ARGUMENT;
  }
  *n_points = 0;
  uint64_t bytes = 0;
  int32_t st = ilcc_bag_first_message(bag_path, topic, nullptr, nullptr, 0, &bytes);
  if (st != ILCC_CAPACITY && st != ILCC_OK) return st;
  std::vector<uint8_t> msg;
  try {
    msg.resize(bytes);
  } catch (...) {   // no exception crosses the C-ABI
    set_global_error("out of memory for the bag's message");
    return ILCC_IO_ERROR;
  }
  st = ilcc_bag_first_message(bag_path, topic, nullptr, msg.data(), bytes, &bytes);
  if (st != ILCC_OK) return st;
  ilcc_pointcloud2_layout L;
  st = ilcc_pointcloud2_parse(msg.data(), bytes, &L);
  if (st != ILCC_OK) return st;
  const uint64_t pts = (uint64_t)L.height * L.width;
  *n_points = (uint32_t)pts;
  if (pts > cap_points) {
    set_global_error("cloud larger than the buffer");
    return ILCC_CAPACITY;
  }
  if (pts == 0) return ILCC_OK;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
    set_global_error("no HIP device: libilcc_hip has no CPU fallback");
    return ILCC_HIP_ERROR;
  }
  void *d_in = nullptr, *d_out = nullptr;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMalloc(&d_in, L.data_bytes);
  if (e == hipSuccess) e = hipMalloc(&d_out, pts * 16);
  if (e == hipSuccess) e = hipMemcpy(d_in, msg.data() + L.data_offset, L.data_bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    st = ilcc_pointcloud2_unpack_device(d_in, &L, d_out, nullptr);
    if (st == ILCC_OK) e = hipMemcpy(xyzi, d_out, pts * 16, hipMemcpyDeviceToHost);
  }
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (e != hipSuccess) {
    set_global_error(std::string("hip: ") + hipGetErrorString(e));
    return ILCC_HIP_ERROR;
  }
  return st;
}
