// K1 roi_crop_compact -- replaces the three pcl::PassThrough passes of LidarCornersEst::setROI
// (/root/reference/ilcc2/src/LidarCornersEst.cpp:48-70).
//
// HBM-bound: every input point (16 B XYZI, float4, coalesced 1 KiB per wavefront load) is read
// ONCE; survivors (a few thousand per frame) are written in input order.
// Two kernels: the count pass tests every point and leaves, per 4096-point chunk, the survivor
// count and the 64 keep-masks of its wavefront loads (512 B); the order-preserving scatter works
// from the masks alone -- ranks are popcounts, no workgroup barriers -- and re-reads only the ~5 %
// of the points that survive.  Reading the cloud once is what lets the kernel be fed straight
// from pinned host memory over PCIe at link rate (PCIe-inclusive rate).
// Launch: grid = (chunks, frames) -> >= 1024 workgroups for a 128-frame VLP-16 batch.
#include "ilcc_internal.h"

namespace ilcc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Box {
  float lo[3], hi[3];
};

// pcl::PassThrough::setFilterLimits narrows to float: (float)(double(click) -+ half)
__device__ __forceinline__ Box make_box(const Ctx& c, uint32_t f) {
  Box b;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double cl = (double)c.clicks[3 * f + a];
    b.lo[a] = (float)(cl - c.p.roi_half[a]);
    b.hi[a] = (float)(cl + c.p.roi_half[a]);
  }
  return b;
}

template <bool FINITE_BOX>
__device__ __forceinline__ bool keep_point(const float4 q, const Box& b) {
  // non-finite x/y/z are dropped by every PassThrough; limits are inclusive (:54,59,64).  Against a box of finite limits the
  // inclusive comparisons alone say so: a NaN fails every one of them, an infinity the one against the finite limit beyond it
  // -- the same truth table as "finite and not (below or above)" at half the instructions (K1 is HBM-bound alone, but beside the
  // other batches' kernels its instructions count like everyone's).
  if (FINITE_BOX)
    return q.z >= b.lo[2] && q.z <= b.hi[2] && q.x >= b.lo[0] && q.x <= b.hi[0] && q.y >= b.lo[1] && q.y <= b.hi[1];
  const bool fin = isfinite(q.x) && isfinite(q.y) && isfinite(q.z);
  const bool in = !(q.z < b.lo[2] || q.z > b.hi[2]) && !(q.x < b.lo[0] || q.x > b.hi[0]) &&
                  !(q.y < b.lo[1] || q.y > b.hi[1]);
  return fin && in;
}

__global__ __launch_bounds__(kCropThreads) void k1_roi_count(Ctx c) {
  const uint32_t f = blockIdx.y, s = blockIdx.x;
  if (c.online_tier == 2u && c.frame_flags[f] == 0u) return;   // second tier of the online caller: the frames the first could not vouch for
  const uint64_t beg = c.off[f], end = c.off[f + 1];
  const uint64_t n = end - beg;
  // the batch's scratch words are reset here instead of by separate memset launches (each costs ~5 us plus a
  // gap on the batch's critical path): the frame's whole result record (no stale fields in failed frames), its
  // near-tie counter, and -- once per batch -- the K6 work counters
  if (s == 0) {
    uint32_t* w = reinterpret_cast<uint32_t*>(&c.res[f]);
    for (uint32_t k = threadIdx.x; k < sizeof(ilcc_result) / 4; k += kCropThreads) w[k] = 0u;
    if (f == 0 && threadIdx.x < kBatchWords && c.online_tier != 2u) c.grid_iters[threadIdx.x] = 0ull;   // (kBatchWords <= kCropThreads)
    __syncthreads();
  }
  if (s == 0 && threadIdx.x == 0) {
    // fresh per-frame record
    if (c.tie_count_all) c.tie_count_all[f] = 0u;
    ilcc_result* r = &c.res[f];
    r->status = ILCC_OK;
    r->n_points = (int32_t)n;
    r->n_roi = r->n_cluster = r->n_plane = 0;
    r->n_black = r->n_gray = r->n_white = 0;
    r->n_corners = 0;
    r->phase = 0;
    r->iters_a = r->iters_b = 0;
    r->grid_index = -1;
    r->cells_hit = r->n_oob = 0;
    r->grid_cost = 0.f;
    r->cost_a = r->cost_b = r->sel_cost = 0.0;
    r->theta_t[0] = r->theta_t[1] = r->theta_t[2] = 0.0;
    c.n_lab[f] = 0;
    c.grid_bound[f] = 0x7f800000u;   // +inf
    c.grid_bound_sub[f] = 0x7f800000u;
    if (c.online_tier == 1u) c.frame_flags[f] = 0u;
  }
  const uint64_t cbeg = (uint64_t)s * kCropChunk;
  uint32_t cnt = 0, fin = 0;
  unsigned long long* masks = c.crop_masks + ((uint64_t)f * c.crop_chunks + s) * (kCropChunk / ILCC_WAVE);
  if (cbeg < n) {
    const Box b = make_box(c, f);
    const uint64_t cend = (cbeg + kCropChunk < n) ? cbeg + kCropChunk : n;
    const float4* __restrict__ src = c.xyzi + beg;
    constexpr int kTrips = kCropChunk / kCropThreads;   // 16: all loads of a thread are independent
    float4 q[kTrips];
#pragma unroll
    for (int k = 0; k < kTrips; ++k) {
      const uint64_t i = cbeg + (uint64_t)k * kCropThreads + threadIdx.x;
      if (i < cend) {   // streamed once: keep it out of the way of the later stages' working set
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + i));
        q[k] = make_float4(v.x, v.y, v.z, v.w);
      } else {
        q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const bool finite_box = isfinite(b.lo[0]) && isfinite(b.hi[0]) && isfinite(b.lo[1]) && isfinite(b.hi[1]) && isfinite(b.lo[2]) && isfinite(b.hi[2]);
    const uint32_t left = (uint32_t)(cend - cbeg);   // points of this chunk (<= kCropChunk)
    if (finite_box && c.online_tier != 1u) {   // (uniform) the offline path: a finite ROI box, no count of the finite points
#pragma unroll
      for (int k = 0; k < kTrips; ++k) {
        const bool keep = ((uint32_t)k * kCropThreads + threadIdx.x < left) && keep_point<true>(q[k], b);
        const unsigned long long m = __ballot(keep);
        if (lane_id() == 0) masks[k * (kCropThreads / ILCC_WAVE) + wave_id()] = m;   // order (trip, wavefront) = input order
        cnt += keep ? 1u : 0u;
      }
    } else {
#pragma unroll
      for (int k = 0; k < kTrips; ++k) {
        const bool mine = (uint32_t)k * kCropThreads + threadIdx.x < left;
        const bool keep = mine && keep_point<false>(q[k], b);
        const unsigned long long m = __ballot(keep);
        if (lane_id() == 0) masks[k * (kCropThreads / ILCC_WAVE) + wave_id()] = m;
        cnt += keep ? 1u : 0u;
        fin += (mine && isfinite(q[k].x) && isfinite(q[k].y) && isfinite(q[k].z)) ? 1u : 0u;
      }
    }
  }
  __shared__ uint32_t sc[17];
  const uint32_t total = block_sum<uint32_t>(cnt, sc);
  if (threadIdx.x == 0) c.crop_counts[(uint64_t)f * c.crop_chunks + s] = total;
  if (c.online_tier == 1u) {   // (uniform) what an unbounded crop would keep: the online caller's n_roi
    const uint32_t total_fin = block_sum<uint32_t>(fin, sc);
    if (threadIdx.x == 0) c.crop_fin[(uint64_t)f * c.crop_chunks + s] = total_fin;
  }
}

__global__ __launch_bounds__(kCropThreads) void k1_roi_scatter(Ctx c) {
  const uint32_t f = blockIdx.y, s = blockIdx.x;
  if (c.online_tier == 2u && c.frame_flags[f] == 0u) return;
  const uint64_t beg = c.off[f], end = c.off[f + 1];
  const uint64_t n = end - beg;
  const uint32_t* counts = c.crop_counts + (uint64_t)f * c.crop_chunks;
  uint32_t base = 0, all = 0;
  for (uint32_t k = 0; k < c.crop_chunks; ++k) {
    const uint32_t v = counts[k];
    if (k < s) base += v;
    all += v;
  }
  if (s == 0 && threadIdx.x == 0 && c.online_tier == 1u) {
    uint32_t nf = 0;
    for (uint32_t k = 0; k < c.crop_chunks; ++k) nf += c.crop_fin[(uint64_t)f * c.crop_chunks + k];
    c.n_finite[f] = nf;
  }
  if (s == 0 && threadIdx.x == 0) {
    c.res[f].n_roi = (int32_t)all;
    if (all == 0) {
      if (c.online_tier == 1u)
        c.frame_flags[f] = 1u;   // nothing in the window says nothing about the cloud: second tier
      else
        c.res[f].status = ILCC_NO_ROI_POINTS;
    }
  }
  const uint64_t cbeg = (uint64_t)s * kCropChunk;
  if (cbeg >= n || counts[s] == 0) return;
  const float4* __restrict__ src = c.xyzi + beg;
  float4* __restrict__ dst = c.roi + beg;
  const unsigned long long* masks = c.crop_masks + ((uint64_t)f * c.crop_chunks + s) * (kCropChunk / ILCC_WAVE);
  constexpr int kWaves = kCropThreads / ILCC_WAVE, kMasks = kCropChunk / ILCC_WAVE;
  static_assert(kMasks == ILCC_WAVE, "one keep-mask per lane of the first wavefront");
  const int lane = lane_id(), w = wave_id();
  // rank of a survivor = survivors in all earlier masks (input order = mask order) + survivors below it in
  // its own mask: the first wavefront turns the 64 mask popcounts into an exclusive prefix in LDS
  __shared__ unsigned long long s_mask[kMasks];
  __shared__ uint32_t s_before[kMasks];
  if (w == 0) {
    const unsigned long long mk = masks[lane];
    uint32_t incl = (uint32_t)__popcll(mk);
    const uint32_t own = incl;
#pragma unroll
    for (int o = 1; o < ILCC_WAVE; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o, ILCC_WAVE);
      if (lane >= o) incl += t;
    }
    s_mask[lane] = mk;
    s_before[lane] = base + incl - own;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kCropChunk / kCropThreads; ++k) {
    const int mine = k * kWaves + w;
    const unsigned long long mk = s_mask[mine];
    if ((mk >> lane) & 1ull) {
      const uint64_t i = cbeg + (uint64_t)k * kCropThreads + threadIdx.x;
      dst[s_before[mine] + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = src[i];
    }
  }
}

void launch_roi_crop(const Ctx& c, hipStream_t s, hipEvent_t after_count) {
  const dim3 grid(c.crop_chunks, c.n_frames);
  hipLaunchKernelGGL(k1_roi_count, grid, dim3(kCropThreads), 0, s, c);
  if (after_count) (void)hipEventRecord(after_count, s);
  hipLaunchKernelGGL(k1_roi_scatter, grid, dim3(kCropThreads), 0, s, c);
}

}  // namespace ilcc
