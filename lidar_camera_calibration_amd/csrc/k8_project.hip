// K8 project -- LiDAR -> image projection per point (include/ilcc_project.h):
// ImageCornersEst::spaceToPlane (/root/reference/ilcc2/src/ImageCornersEst.cpp:135-155) fused with
// the per-point colour of pcd2image (test/pcd2image.cpp:61-75, HSVtoRGB src/ImageCornersEst.cpp:373-428)
// or the image sample of rgblidar (test/rgblidar.cpp:55-74), and an order-preserving compaction.
//
// HBM-bound: 16 B read per point, 16 B written per surviving point.  Two launches like K1: count per
// 4096-point chunk, then scatter with the chunk's prefix (the second read of the cloud comes out of
// L2 / Infinity Cache for sensor-sized clouds).  The projection itself is fp64 (the reference uses
// Eigen::Vector3d / Matrix3d), compiled with -ffp-contract=off so that pixel truncation matches bit
// for bit; with three fp64 divisions per point the kernels sit near the fp64-VALU / HBM crossover.
// A single-pass variant (decoupled look-back over published chunk counts, ticketed chunk order) was
// built and measured: bit-identical but 82 us instead of 39 us for 3.7 M points -- device-scope
// atomics and acquire loads leave the XCD's own L2 on this 8-XCD part, so ~900 chained workgroups
// serialise on fabric latency.  Two passes with plain loads are the better MI355X shape.
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

#include "ilcc_internal.h"
#include "ilcc_project.h"

namespace ilcc {

void set_global_error(const std::string& s);

constexpr int kProjThreads = 256;
constexpr int kProjChunk = 4096;

struct ProjArgs {
  const float4* pts;
  uint32_t n;
  ilcc_camera_model cam;
  double dis, lo, hi;
  const uint8_t* image;
  uint32_t image_step;
  uint32_t* counts;
  uint32_t chunks;
  void* out;
  uint32_t* total;
};

// spaceToPlane (:135-155); px, py = the (int) truncation the callers apply
__device__ __forceinline__ bool space_to_plane(const ilcc_camera_model& c, const float4 q, double dis, int32_t& px, int32_t& py) {
  const double X = (double)q.x, Y = (double)q.y, Z = (double)q.z;
  const double pc0 = c.R[0] * X + c.R[1] * Y + c.R[2] * Z + c.t[0];
  const double pc1 = c.R[3] * X + c.R[4] * Y + c.R[5] * Z + c.t[1];
  const double pc2 = c.R[6] * X + c.R[7] * Y + c.R[8] * Z + c.t[2];
  if (pc2 < 0 || pc2 > dis) return false;   // NaN depth passes this test, like the reference, and fails below
  const double u = pc0 / pc2, v = pc1 / pc2;
  const double cu = c.fx * u + c.cx, cv = c.fy * v + c.cy;
  if (cu > 0 && cu < (double)c.width && cv > 0 && cv < (double)c.height) {
    px = (int32_t)cu;
    py = (int32_t)cv;
    return true;
  }
  return false;
}

// float -> unsigned char the way an x86-64 build converts (cvttss2si, low byte)
__device__ __forceinline__ uint8_t to_u8(float v) { return (uint8_t)(int32_t)v; }
// double -> int as cvttsd2si: NaN / out of range give INT_MIN ("integer indefinite"), not saturation
__device__ __forceinline__ int32_t x86_d2i(double v) {
  return (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : (int32_t)0x80000000;
}

// HSVtoRGB (:373-428) with s = v = 100
__device__ __forceinline__ void hsv_to_rgb(int32_t h, int32_t s, int32_t v, uint8_t& r, uint8_t& g, uint8_t& b) {
  const float rgb_max = v * 2.55f;
  const float rgb_min = rgb_max * (100 - s) / 100.0f;
  const int32_t i = h / 60;
  const int32_t difs = h % 60;
  const float adj = (rgb_max - rgb_min) * difs / 60.0f;
  switch (i) {
    case 0: r = to_u8(rgb_max); g = to_u8(rgb_min + adj); b = to_u8(rgb_min); break;
    case 1: r = to_u8(rgb_max - adj); g = to_u8(rgb_max); b = to_u8(rgb_min); break;
    case 2: r = to_u8(rgb_min); g = to_u8(rgb_max); b = to_u8(rgb_min + adj); break;
    case 3: r = to_u8(rgb_min); g = to_u8(rgb_max - adj); b = to_u8(rgb_max); break;
    case 4: r = to_u8(rgb_min + adj); g = to_u8(rgb_min); b = to_u8(rgb_max); break;
    default: r = to_u8(rgb_max); g = to_u8(rgb_min); b = to_u8(rgb_max - adj); break;
  }
}

__global__ __launch_bounds__(kProjThreads) void k8_count(ProjArgs a) {
  const uint32_t beg = blockIdx.x * kProjChunk;
  const uint32_t end = (beg + kProjChunk < a.n) ? beg + kProjChunk : a.n;
  uint32_t cnt = 0;
#pragma unroll 4
  for (uint32_t i = beg + threadIdx.x; i < end; i += kProjThreads) {
    int32_t px, py;
    cnt += space_to_plane(a.cam, a.pts[i], a.dis, px, py) ? 1u : 0u;
  }
  __shared__ uint32_t sc[17];
  const uint32_t total = block_sum<uint32_t>(cnt, sc);
  if (threadIdx.x == 0) a.counts[blockIdx.x] = total;
}

template <bool SAMPLE_IMAGE>
__global__ __launch_bounds__(kProjThreads) void k8_scatter(ProjArgs a) {
  __shared__ uint32_t sc[17];
  __shared__ uint32_t s_base;
  // prefix of the chunk counts (<= a few thousand chunks): strided partial sums, then block_sum
  uint32_t part = 0, all = 0;
  for (uint32_t k = threadIdx.x; k < a.chunks; k += kProjThreads) {
    const uint32_t v = a.counts[k];
    if (k < blockIdx.x) part += v;
    all += v;
  }
  const uint32_t base0 = block_sum<uint32_t>(part, sc);
  if (blockIdx.x == 0) {
    const uint32_t tot = block_sum<uint32_t>(all, sc);
    if (threadIdx.x == 0) *a.total = tot;
  }
  if (threadIdx.x == 0) s_base = base0;
  __syncthreads();
  uint32_t running = s_base;
  const uint32_t beg = blockIdx.x * kProjChunk;
  const uint32_t end = (beg + kProjChunk < a.n) ? beg + kProjChunk : a.n;
  for (uint32_t t = beg; t < end; t += kProjThreads) {   // uniform trip count per workgroup
    const uint32_t i = t + threadIdx.x;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    int32_t px = 0, py = 0;
    bool keep = false;
    if (i < end) {
      q = a.pts[i];
      keep = space_to_plane(a.cam, q, a.dis, px, py);
    }
    uint32_t tot;
    const uint32_t rank = block_rank(keep, sc, tot);
    if (keep) {
      if (SAMPLE_IMAGE) {
        const uint8_t* p = a.image + (uint64_t)py * a.image_step + (uint32_t)px * 3u;
        const uint32_t rgb = ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | (uint32_t)p[0];   // rgblidar.cpp:63-65
        reinterpret_cast<float4*>(a.out)[running + rank] = make_float4(q.x, q.y, q.z, __uint_as_float(rgb));
      } else {
        const double h = ((double)q.w - a.lo) / (a.hi - a.lo) * 255;   // pcd2image.cpp:71
        uint8_t r, g, b;
        hsv_to_rgb(x86_d2i(h), 100, 100, r, g, b);
        uint4 rec;
        rec.x = (uint32_t)px;
        rec.y = (uint32_t)py;
        rec.z = (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16);
        rec.w = i;
        reinterpret_cast<uint4*>(a.out)[running + rank] = rec;
      }
    }
    running += tot;
  }
}

namespace {
struct Scratch {
  uint32_t* counts = nullptr;
  uint32_t cap = 0;
  uint32_t* total = nullptr;
};
std::mutex g_mu;
Scratch g_scratch[16];
}  // namespace

static int32_t run_project(bool sample, const void* d_xyzi, uint32_t n, const ilcc_camera_model* cam, double dis, double lo,
                           double hi, const void* image, uint32_t step, void* out, uint32_t* n_out, void* stream) {
  if (!cam || !n_out || (n && (!d_xyzi || !out)) || (sample && !image) || cam->width <= 0 || cam->height <= 0) {
    set_global_error("bad argument");
    return ILCC_BAD_ARGUMENT;
  }
  if (sample && step < (uint32_t)cam->width * 3u) {
    set_global_error("image_step smaller than a BGR row");
    return ILCC_BAD_ARGUMENT;
  }
  *n_out = 0;
  if (n == 0) return ILCC_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
    set_global_error("no HIP device: libilcc_hip has no CPU fallback");
    return ILCC_HIP_ERROR;
  }
  std::lock_guard<std::mutex> lock(g_mu);
  Scratch& sc = g_scratch[dev];
  const uint32_t chunks = (n + kProjChunk - 1) / kProjChunk;
  hipError_t e = hipSuccess;
  if (sc.cap < chunks) {
    if (sc.counts) (void)hipFree(sc.counts);
    sc.counts = nullptr;
    sc.cap = 0;
    const uint32_t cap = chunks < 1024 ? 1024 : chunks;
    e = hipMalloc((void**)&sc.counts, sizeof(uint32_t) * (size_t)cap);
    if (e == hipSuccess) sc.cap = cap;
  }
  if (e == hipSuccess && !sc.total) e = hipHostMalloc((void**)&sc.total, sizeof(uint32_t));
  if (e != hipSuccess) {
    set_global_error(std::string("hip: ") + hipGetErrorString(e));
    return ILCC_HIP_ERROR;
  }
  ProjArgs a;
  a.pts = (const float4*)d_xyzi;
  a.n = n;
  a.cam = *cam;
  a.dis = dis;
  a.lo = lo;
  a.hi = hi;
  a.image = (const uint8_t*)image;
  a.image_step = step;
  a.counts = sc.counts;
  a.chunks = chunks;
  a.out = out;
  a.total = sc.total;   // pinned host word, written by workgroup 0 of the scatter
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k8_count, dim3(chunks), dim3(kProjThreads), 0, s, a);
  if (sample) hipLaunchKernelGGL(k8_scatter<true>, dim3(chunks), dim3(kProjThreads), 0, s, a);
  else hipLaunchKernelGGL(k8_scatter<false>, dim3(chunks), dim3(kProjThreads), 0, s, a);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    set_global_error(std::string("k8: ") + hipGetErrorString(e));
    return ILCC_HIP_ERROR;
  }
  *n_out = *sc.total;
  return ILCC_OK;
}

}  // namespace ilcc

extern "C" int32_t ilcc_project_intensity_device(const void* d_xyzi, uint32_t n_points, const ilcc_camera_model* cam,
                                                 double distance_valid, double inten_low, double inten_high, void* d_hits,
                                                 uint32_t* n_hits, void* hip_stream) {
  return ilcc::run_project(false, d_xyzi, n_points, cam, distance_valid, inten_low, inten_high, nullptr, 0, d_hits, n_hits,
                           hip_stream);
}

extern "C" int32_t ilcc_colourise_device(const void* d_xyzi, uint32_t n_points, const ilcc_camera_model* cam,
                                         double distance_valid, const void* d_image_bgr, uint32_t image_step, void* d_xyzrgb,
                                         uint32_t* n_out, void* hip_stream) {
  return ilcc::run_project(true, d_xyzi, n_points, cam, distance_valid, 0.0, 1.0, d_image_bgr, image_step, d_xyzrgb, n_out,
                           hip_stream);
}
