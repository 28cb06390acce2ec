// K2 seeded_cluster -- replaces pcl::EuclideanClusterExtraction + nearestKSearch of
// LidarCornersEst::EuclideanCluster (/root/reference/ilcc2/src/LidarCornersEst.cpp:124-153).
//
// Single-linkage components of the radius graph (squared float distance dx*dx+dy*dy+dz*dz < (float)(tol*tol),
// FLANN's strict test) are built with a lock-free union-find.  Roots are always the smallest member index, so the
// labelling is deterministic.  Neighbour search, four ways with identical results:
//   frames of <= cluster_lds_points ROI points (the handle's LDS capacity, <= 4096): ONE 1024-thread workgroup per
//   frame, parents in LDS:
//     <= ILCC_K2_ALLPAIRS_MAX (256) points  tiled all-pairs (1024 "j" points per LDS tile, broadcast reads);
//     else (the ROI case)                   wave-cooperative search on a direct cell grid in LDS: points
//                                           counting-sorted by cell, a wavefront per occupied cell, candidates in
//                                           the lanes, own points through v_readlane, unions queued and executed
//                                           64 at a time -- or, when the bounding box needs more than 16384 cells,
//                                           per-point scans of counting-sorted hashed buckets in LDS;
//   larger frames (dense clouds: BASELINE config 5's ~20 k ROI points; un-cropped clouds of the online caller
//   get_chessboard_by_point, LidarCornersEst.cpp:72-115): SEVERAL workgroups per frame.  The frame's own workgroup only
//   resets the parents and the frame's hash table and puts the frame on the batch's list; k2l_insert / k2l_search
//   (persistent grids over (frame, 256-point chunk) items of the listed frames) build a spatial hash with chained
//   buckets in global memory and unite neighbours with device-scope atomics; k2l_finish (one workgroup per listed
//   frame) labels, sizes, picks and compacts exactly like the LDS path.  (Round 2 ran this search inside the frame's one
//   workgroup: 64 workgroups on 256 CUs, 16 ms per 64-frame config-5 batch.)
// Cluster choice follows the reference: components with
// cluster_min <= size <= cluster_max, sorted by size (largest = index 0); the one containing
// the exact 1-NN of the click wins, otherwise index 0.  Members are emitted in index order.
#include "ilcc_internal.h"

namespace ilcc {

#ifdef ILCC_K2_TIMING
#define K2_MARK(k) do { __syncthreads(); if (f == 0 && threadIdx.x == 0) tmark[k] = __builtin_readcyclecounter(); } while (0)
#else
#define K2_MARK(k) do {} while (0)
#endif

// find with path halving.  Parents only ever point to smaller indices (the larger root is hooked
// under the smaller), so replacing parent[x] by its grandparent keeps it an ancestor: safe against
// concurrent hooks, which only touch roots.
template <typename P>
__device__ __forceinline__ uint32_t uf_find(P* parent, uint32_t x) {
  for (;;) {
    const uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == x) return x;
    const uint32_t gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x = p;
  }
}

template <typename P>
__device__ __forceinline__ void uf_unite(P* parent, uint32_t a, uint32_t b) {
  for (;;) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) {
      const uint32_t t = a;
      a = b;
      b = t;
    }
    // hook the larger root under the smaller one
    const uint32_t old = atomicCAS(&parent[a], a, b);
    if (old == a) return;
  }
}

// unite, then point both endpoints straight at the common root (only non-roots are rewritten: a root's parent
// word belongs to the hooks).  Makes the cheap "parents equal?" test of the batched search effective.
template <typename P>
__device__ __forceinline__ void uf_unite_compress(P* parent, uint32_t a, uint32_t b) {
  uf_unite(parent, a, b);
  const uint32_t r = uf_find(parent, a);
  if (__hip_atomic_load(&parent[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a)
    __hip_atomic_store(&parent[a], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__hip_atomic_load(&parent[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != b)
    __hip_atomic_store(&parent[b], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bucket of an integer cell; different cells may share a bucket (a far cell's points then simply
// fail the distance test)
__device__ __forceinline__ uint32_t cell_hash(int cx, int cy, int cz) {
  const uint32_t h = (uint32_t)cx * 73856093u ^ (uint32_t)cy * 19349663u ^ (uint32_t)cz * 83492791u;
  return h & (uint32_t)(kClusterHashSize - 1);
}

struct NnKey {
  float d2;
  uint32_t idx;
};
__device__ __forceinline__ bool nn_less(const NnKey& x, const NnKey& y) {
  return x.d2 < y.d2 || (x.d2 == y.d2 && x.idx < y.idx);
}

#ifndef ILCC_K2_GRID_MAX
#define ILCC_K2_GRID_MAX 4096
#endif
constexpr int kClusterGridMax = ILCC_K2_GRID_MAX;     // upper bound of the LDS path's capacity (= kClusterLdsPointsMax)
static_assert(kClusterGridMax == kClusterLdsPointsMax, "one constant");
constexpr int kClusterGridBuckets = 8192;
constexpr int kClusterCells = 16384;       // direct cell grid of the wave-cooperative search (u16 run ends: 32 KiB)

template <bool LDS_PARENT>
__device__ void cluster_finish(const Ctx& c, uint32_t f, uint32_t* parent, uint32_t* sc);

// one workgroup, parents in LDS: frames of at most `cap` = c.cluster_lds_points ROI points
__device__ void cluster_frame(const Ctx& c, uint32_t f, uint32_t* lds_parent, float4* tile,
                              uint32_t* sc, float4* s_pts, uint32_t cap) {
  ilcc_result* r = &c.res[f];
  const uint32_t M = (uint32_t)r->n_roi;
  const uint64_t beg = c.off[f];
  const float4* __restrict__ P = c.roi + beg;
  uint32_t* parent = lds_parent;
  uint32_t* count = c.uf_count + beg;   // zeroed below, together with the parents
  const float tol2 = (float)(c.p.cluster_tol * c.p.cluster_tol);
  const uint32_t tid = threadIdx.x;

#ifdef ILCC_K2_TIMING
  __shared__ unsigned long long tmark[12];
#endif
  K2_MARK(0);
  for (uint32_t i = tid; i < M; i += kFrameThreads) {
    parent[i] = i;
    count[i] = 0u;   // component sizes, accumulated on the roots further down
  }
  __syncthreads();
  K2_MARK(1);

  // bounding box of the ROI points (every path below the all-pairs threshold skips it)
  bool direct = false;
  float3 blo = make_float3(0.f, 0.f, 0.f);
  int gnx = 1, gny = 1, gnz = 1;
  const float inv_cell_d = 1.0f / ((float)c.p.cluster_tol * 1.001f);
  if (M > (uint32_t)kClusterAllPairsMax) {
    float3 lo = make_float3(3.0e38f, 3.0e38f, 3.0e38f), hi = make_float3(-3.0e38f, -3.0e38f, -3.0e38f);
    for (uint32_t i = tid; i < M; i += kFrameThreads) {
      const float4 q = P[i];
      lo.x = fminf(lo.x, q.x); lo.y = fminf(lo.y, q.y); lo.z = fminf(lo.z, q.z);
      hi.x = fmaxf(hi.x, q.x); hi.y = fmaxf(hi.y, q.y); hi.z = fmaxf(hi.z, q.z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo.x = fminf(lo.x, __shfl_xor(lo.x, o, ILCC_WAVE)); lo.y = fminf(lo.y, __shfl_xor(lo.y, o, ILCC_WAVE));
      lo.z = fminf(lo.z, __shfl_xor(lo.z, o, ILCC_WAVE));
      hi.x = fmaxf(hi.x, __shfl_xor(hi.x, o, ILCC_WAVE)); hi.y = fmaxf(hi.y, __shfl_xor(hi.y, o, ILCC_WAVE));
      hi.z = fmaxf(hi.z, __shfl_xor(hi.z, o, ILCC_WAVE));
    }
    float* scf = reinterpret_cast<float*>(sc);
    __syncthreads();
    if (lane_id() == 0) {
      scf[wave_id()] = lo.x; scf[16 + wave_id()] = lo.y; scf[32 + wave_id()] = lo.z;
      scf[64 + wave_id()] = hi.x; scf[80 + wave_id()] = hi.y; scf[96 + wave_id()] = hi.z;
    }
    __syncthreads();
    for (int w = 0; w < kFrameThreads / ILCC_WAVE; ++w) {
      lo.x = fminf(lo.x, scf[w]); lo.y = fminf(lo.y, scf[16 + w]); lo.z = fminf(lo.z, scf[32 + w]);
      hi.x = fmaxf(hi.x, scf[64 + w]); hi.y = fmaxf(hi.y, scf[80 + w]); hi.z = fmaxf(hi.z, scf[96 + w]);
    }
    __syncthreads();
    blo = lo;
    gnx = (int)floorf((hi.x - lo.x) * inv_cell_d) + 1;
    gny = (int)floorf((hi.y - lo.y) * inv_cell_d) + 1;
    gnz = (int)floorf((hi.z - lo.z) * inv_cell_d) + 1;
    direct = (long long)gnx * gny * gnz <= (long long)kClusterCells;   // uniform: every thread holds the same box
  }
  if (direct) {
    // ---- wave-cooperative search on a direct cell grid in LDS (the ROI case: a 2 x 3 x 4 m box at 0.12 m
    // cells is ~15 k cells).  Points are counting-sorted by cell into LDS (xyz + original index); a wavefront
    // takes one occupied cell at a time: its 64 LANES HOLD THE CANDIDATES (the points of the 27 surrounding
    // cells, one 16-byte LDS read per lane and 64 candidates) and the cell's own points are broadcast one after
    // the other -- every lane busy, a fifth of the LDS traffic of a per-point scan.  Pairs that pass the cheap
    // parent test are queued and united 64 at a time (a union is two chains of dependent LDS atomics; done in
    // place it would stall the wavefront for one lane).  Same pairs, same distance arithmetic, same partition.
    uint32_t* key = lds_parent + cap;                               // cell of point i
    uint16_t* cur16 = reinterpret_cast<uint16_t*>(lds_parent + 2 * cap);   // kClusterCells run ends
    uint32_t* cur32 = lds_parent + 2 * cap;                          // the same words, two cells each
    uint16_t* occ = reinterpret_cast<uint16_t*>(s_pts + cap);       // occupied cells (<= M)
    uint32_t* n_occ = sc + 120;
    if (tid == 0) *n_occ = 0;
    for (uint32_t k = tid; k < (uint32_t)kClusterCells / 2; k += kFrameThreads) cur32[k] = 0u;
    __syncthreads();
    for (uint32_t i = tid; i < M; i += kFrameThreads) {
      const float4 q = P[i];
      const int cx = (int)floorf((q.x - blo.x) * inv_cell_d), cy = (int)floorf((q.y - blo.y) * inv_cell_d),
                cz = (int)floorf((q.z - blo.z) * inv_cell_d);
      const uint32_t cell = (uint32_t)(cx + gnx * (cy + gny * cz));
      key[i] = cell;
      const uint32_t sh = (cell & 1u) * 16u;
      const uint32_t old = atomicAdd(&cur32[cell >> 1], 1u << sh);
      if (((old >> sh) & 0xFFFFu) == 0u) occ[atomicAdd(n_occ, 1u)] = (uint16_t)cell;   // first point of the cell
    }
    __syncthreads();
    {   // exclusive scan of the 16384 counts (16 consecutive cells per thread): cur16[c] := start of run c
      constexpr int kPer = kClusterCells / kFrameThreads;
      uint32_t v[kPer], sum = 0;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        v[k] = cur16[tid * kPer + k];
        sum += v[k];
      }
      uint32_t incl = sum;
#pragma unroll
      for (int o = 1; o < ILCC_WAVE; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, ILCC_WAVE);
        if (lane_id() >= o) incl += t;
      }
      if (lane_id() == ILCC_WAVE - 1) sc[wave_id()] = incl;
      __syncthreads();
      uint32_t base = 0;
      for (int w = 0; w < wave_id(); ++w) base += sc[w];
      uint32_t run = base + incl - sum;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        cur16[tid * kPer + k] = (uint16_t)run;
        run += v[k];
      }
    }
    __syncthreads();
    for (uint32_t i = tid; i < M; i += kFrameThreads) {   // placement: afterwards cur16[c] = END of run c
      const float4 q = P[i];
      const uint32_t cell = key[i], sh = (cell & 1u) * 16u;
      const uint32_t at = (atomicAdd(&cur32[cell >> 1], 1u << sh) >> sh) & 0xFFFFu;
      s_pts[at] = make_float4(q.x, q.y, q.z, __uint_as_float(i));
    }
    __syncthreads();
    K2_MARK(8);
    uint2* wq = reinterpret_cast<uint2*>(tile) + wave_id() * 128;   // 16 wavefronts x 128 entries = the 16 KiB tile
    const int lane = lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t qn = 0;
#ifdef ILCC_K2_TIMING
    unsigned long long n_push = 0, n_flush = 0, n_iter = 0, t_flush = 0;
#endif
    const uint32_t cells_occ = *n_occ;
    for (uint32_t t = wave_id(); t < cells_occ; t += kFrameThreads / ILCC_WAVE) {
      const uint32_t cell = occ[t];
      const int cz = (int)(cell / (uint32_t)(gnx * gny)), rem = (int)(cell - (uint32_t)cz * (uint32_t)(gnx * gny));
      const int cy = rem / gnx, cx = rem - cy * gnx;
      const uint32_t cbeg = cell ? cur16[cell - 1] : 0u, clen = cur16[cell] - cbeg;
      // lanes 0..26: run (start, length) of neighbour cell r = lane; inclusive scan of the lengths over the lanes
      // Only the cell itself (r = 13) and its 13 FORWARD neighbours (r = 14..26: offsets after (0,0,0) in (dz,dy,dx) order)
      // are searched: a pair of points in two adjacent cells is met exactly once, from the cell that comes first -- the
      // full 27-cell search met every such pair from both sides and threw one of the two distance tests away (j < i).
      uint32_t rst = 0, rln = 0;
      if (lane >= 13 && lane < 27) {
        const int dz = lane / 9 - 1, dy = (lane / 3) % 3 - 1, dx = lane % 3 - 1;
        const int nx = cx + dx, ny = cy + dy, nz = cz + dz;
        if (nx >= 0 && nx < gnx && ny >= 0 && ny < gny && nz >= 0 && nz < gnz) {
          const uint32_t nc = (uint32_t)(nx + gnx * (ny + gny * nz));
          rst = nc ? cur16[nc - 1] : 0u;
          rln = cur16[nc] - rst;
        }
      }
      uint32_t incl = rln;
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o, ILCC_WAVE);
        if (lane >= o) incl += u;
      }
      const uint32_t total = __shfl(incl, 26, ILCC_WAVE);
      // the cell's own points, 64 at a time, one per lane; they are handed to all lanes with v_readlane
      // (the inner loop touches LDS only to queue a pair)
      for (uint32_t qb = 0; qb < clen; qb += ILCC_WAVE) {
        const uint32_t nq = (clen - qb < (uint32_t)ILCC_WAVE) ? clen - qb : (uint32_t)ILCC_WAVE;
        float4 own = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t own_par = 0;
        if ((uint32_t)lane < nq) {
          own = s_pts[cbeg + qb + lane];
          own_par = __hip_atomic_load(&parent[__float_as_uint(own.w)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        for (uint32_t g0 = 0; g0 < total; g0 += ILCC_WAVE) {
          const uint32_t g = g0 + (uint32_t)lane;
          const bool valid = g < total;
          // run r of candidate g: the first r with incl[r] > g (binary search over lanes 0..26 through shuffles)
          int r = 0;
#pragma unroll
          for (int step = 16; step > 0; step >>= 1) {
            const int probe = r + step - 1;
            const uint32_t e = __shfl(incl, probe < 26 ? probe : 26, ILCC_WAVE);
            if (probe <= 26 && e <= g) r += step;
          }
          r = r < 26 ? r : 26;
          const uint32_t r_incl = __shfl(incl, r, ILCC_WAVE), r_len = __shfl(rln, r, ILCC_WAVE), r_st = __shfl(rst, r, ILCC_WAVE);
          float4 cand = make_float4(0.f, 0.f, 0.f, 0.f);
          uint32_t j = 0xFFFFFFFFu, pj = 0;
          if (valid) {
            cand = s_pts[r_st + (g - (r_incl - r_len))];
            j = __float_as_uint(cand.w);
            pj = __hip_atomic_load(&parent[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          for (uint32_t qk = 0; qk < nq; ++qk) {
            const float px = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(own.x), qk));
            const float py = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(own.y), qk));
            const float pz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(own.z), qk));
            const uint32_t i = __builtin_amdgcn_readlane(__float_as_uint(own.w), qk);
            const uint32_t qi = __builtin_amdgcn_readlane(own_par, qk);
            const float ex = cand.x - px, ey = cand.y - py, ez = cand.z - pz;
            float d2 = ex * ex;
            d2 = d2 + ey * ey;
            d2 = d2 + ez * ez;
            const bool need = valid && (r != 13 || j < i) && d2 < tol2 && qi != pj;   // own cell: each pair once; equal parents = same set for good
#ifdef ILCC_K2_TIMING
            ++n_iter;
#endif
            const unsigned long long m = __ballot(need);
            if (m) {
              if (need) wq[qn + (uint32_t)__popcll(m & lt)] = make_uint2(i, j);
              qn += (uint32_t)__popcll(m);
#ifdef ILCC_K2_TIMING
              n_push += __popcll(m);
#endif
              if (qn >= 64u) {   // unite the newest 64 pairs, one per lane
                qn -= 64u;
                const uint2 e = wq[qn + lane];
#ifdef ILCC_K2_TIMING
                ++n_flush;
                const unsigned long long tf0 = __builtin_readcyclecounter();
#endif
                uf_unite(parent, e.x, e.y);
#ifdef ILCC_K2_TIMING
                t_flush += __builtin_readcyclecounter() - tf0;
#endif
                if (valid) pj = __hip_atomic_load(&parent[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((uint32_t)lane < nq)
                  own_par = __hip_atomic_load(&parent[__float_as_uint(own.w)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          }
        }
      }
    }
    if ((uint32_t)lane < qn) {
      const uint2 e = wq[lane];
      uf_unite(parent, e.x, e.y);
    }
#ifdef ILCC_K2_TIMING
    K2_MARK(9);
    if (f == 0 && lane == 0) printf("K2 f0 wave %d: iterations %llu pushes %llu flushes %llu cycles in flushes %llu\n", (int)wave_id(), n_iter, n_push, n_flush, t_flush);
    if (f == 0 && tid == 0) printf("K2 f0 direct: box+build %llu walk %llu cycles, occupied cells %u, grid %d x %d x %d\n", tmark[8] - tmark[1], tmark[9] - tmark[8], cells_occ, gnx, gny, gnz);
#endif
  } else if (M > (uint32_t)kClusterAllPairsMax) {
    // ---- cell lists in LDS (the ROI case).  Cells of (slightly more than) the tolerance, hashed into 8192
    // buckets; the points are counting-sorted by bucket INTO LDS (xyz + original index), so a bucket is a
    // contiguous run.  One task per (neighbouring cell offset, point in sorted order): the 64 lanes of a
    // wavefront are 64 consecutive sorted points, i.e. a handful of cells -- lanes of the same cell scan the
    // same run (identical LDS addresses: broadcasts, no bank conflicts, equal trip counts).  Same pairs, same
    // distance arithmetic, same partition as the other two searches; union-find indices stay the original
    // ones, so roots (smallest member index) and labels are unchanged.
    uint32_t* key = lds_parent + cap;          // bucket of point i
    uint32_t* cur = lds_parent + 2 * cap;      // kClusterGridBuckets words: counts -> run ends
    float3 lo = make_float3(3.0e38f, 3.0e38f, 3.0e38f);
    for (uint32_t i = tid; i < M; i += kFrameThreads) {
      const float4 q = P[i];
      lo.x = fminf(lo.x, q.x);
      lo.y = fminf(lo.y, q.y);
      lo.z = fminf(lo.z, q.z);
    }
    for (uint32_t k = tid; k < (uint32_t)kClusterGridBuckets; k += kFrameThreads) cur[k] = 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo.x = fminf(lo.x, __shfl_xor(lo.x, o, ILCC_WAVE));
      lo.y = fminf(lo.y, __shfl_xor(lo.y, o, ILCC_WAVE));
      lo.z = fminf(lo.z, __shfl_xor(lo.z, o, ILCC_WAVE));
    }
    float* scf = reinterpret_cast<float*>(sc);
    __syncthreads();
    if (lane_id() == 0) {
      scf[wave_id()] = lo.x;
      scf[16 + wave_id()] = lo.y;
      scf[32 + wave_id()] = lo.z;
    }
    __syncthreads();
    for (int w = 0; w < kFrameThreads / ILCC_WAVE; ++w) {
      lo.x = fminf(lo.x, scf[w]);
      lo.y = fminf(lo.y, scf[16 + w]);
      lo.z = fminf(lo.z, scf[32 + w]);
    }
    __syncthreads();
    const float inv_cell = 1.0f / ((float)c.p.cluster_tol * 1.001f);
    // counts per bucket
    for (uint32_t i = tid; i < M; i += kFrameThreads) {
      const float4 q = P[i];
      const int cx = (int)floorf((q.x - lo.x) * inv_cell), cy = (int)floorf((q.y - lo.y) * inv_cell),
                cz = (int)floorf((q.z - lo.z) * inv_cell);
      const uint32_t b = cell_hash(cx, cy, cz) & (uint32_t)(kClusterGridBuckets - 1);
      key[i] = b;
      atomicAdd(&cur[b], 1u);
    }
    __syncthreads();
    // exclusive scan of the 8192 counts (8 consecutive buckets per thread): cur[b] := start of run b
    {
      constexpr int kPer = kClusterGridBuckets / kFrameThreads;
      uint32_t v[kPer], sum = 0;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        v[k] = cur[tid * kPer + k];
        sum += v[k];
      }
      uint32_t incl = sum;   // inclusive scan over the wavefront
#pragma unroll
      for (int o = 1; o < ILCC_WAVE; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, ILCC_WAVE);
        if (lane_id() >= o) incl += t;
      }
      if (lane_id() == ILCC_WAVE - 1) sc[wave_id()] = incl;
      __syncthreads();
      uint32_t base = 0;
      for (int w = 0; w < wave_id(); ++w) base += sc[w];
      uint32_t run = base + incl - sum;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        cur[tid * kPer + k] = run;
        run += v[k];
      }
    }
    __syncthreads();
    // placement: afterwards cur[b] = END of run b (= start of run b + 1)
    for (uint32_t i = tid; i < M; i += kFrameThreads) {
      const float4 q = P[i];
      const uint32_t at = atomicAdd(&cur[key[i]], 1u);
      s_pts[at] = make_float4(q.x, q.y, q.z, __uint_as_float(i));
    }
    __syncthreads();
    K2_MARK(8);
    // Unions are the expensive part (two finds = chains of dependent LDS atomics, ~1-2 k cycles), and inside the
    // scan at most a lane or two need one at any step: done in place they would stall the other 60-odd lanes
    // every time (measured: 96 % of the kernel).  Instead a wavefront QUEUES the (i, j) pairs that pass the
    // cheap parent test and unites 64 of them at once, one per lane, whenever the queue fills.  The scan loop
    // runs a wave-uniform trip count (the longest run among the 64 lanes) so the queue length stays uniform.
    uint2* wq = reinterpret_cast<uint2*>(tile) + wave_id() * 128;   // 16 wavefronts x 128 entries = the 16 KiB tile
    const int lane = lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t qn = 0;
    uint32_t cidx = 0, sp = tid;
    while (sp >= M && cidx < 27u) {
      sp -= M;
      ++cidx;
    }
    // every lane of a wavefront makes the same number of trips through the task loop (inactive ones idle)
    const uint32_t n_tasks = (27u * M + kFrameThreads - 1) / kFrameThreads;
    for (uint32_t task = 0; task < n_tasks; ++task) {
      const bool live = cidx < 27u;
      float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
      uint32_t i = 0, at = 0, len = 0, qi = 0;
      if (live) {
        pi = s_pts[sp];
        i = __float_as_uint(pi.w);
        const int dz = (int)(cidx / 9u) - 1, dy = (int)((cidx / 3u) % 3u) - 1, dx = (int)(cidx % 3u) - 1;
        const int nx = (int)floorf((pi.x - lo.x) * inv_cell) + dx, ny = (int)floorf((pi.y - lo.y) * inv_cell) + dy,
                  nz = (int)floorf((pi.z - lo.z) * inv_cell) + dz;
        const uint32_t b = cell_hash(nx, ny, nz) & (uint32_t)(kClusterGridBuckets - 1);
        at = b ? cur[b - 1] : 0u;
        len = cur[b] - at;
        // i's parent as of now: equal parents mean "same set" for good, so a stale copy only costs a redundant unite
        qi = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      uint32_t wmax = len;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = __shfl_xor(wmax, o, ILCC_WAVE);
        wmax = t > wmax ? t : wmax;
      }
      for (uint32_t st = 0; st < wmax; ++st) {
        bool need = false;
        uint32_t j = 0;
        if (st < len) {
          const float4 q = s_pts[at + st];
          j = __float_as_uint(q.w);
          const float ex = q.x - pi.x, ey = q.y - pi.y, ez = q.z - pi.z;
          float d2 = ex * ex;
          d2 = d2 + ey * ey;
          d2 = d2 + ez * ez;
          if (j < i && d2 < tol2)   // each pair once
            need = qi != __hip_atomic_load(&parent[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const unsigned long long m = __ballot(need);
        if (m) {
          if (need) wq[qn + (uint32_t)__popcll(m & lt)] = make_uint2(i, j);
          qn += (uint32_t)__popcll(m);
          if (qn >= 64u) {   // unite the newest 64 pairs, one per lane
            qn -= 64u;
            const uint2 e = wq[qn + lane];
            uf_unite(parent, e.x, e.y);
            if (live) qi = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
      if (live) {
        sp += kFrameThreads;
        while (sp >= M && cidx < 27u) {
          sp -= M;
          ++cidx;
        }
      }
    }
    if ((uint32_t)lane < qn) {
      const uint2 e = wq[lane];
      uf_unite(parent, e.x, e.y);
    }
  } else {
  // ---- all pairs (j < i), tiles of 1024
    for (uint32_t ic = 0; ic < M; ic += kFrameThreads) {
      const uint32_t i = ic + tid;
      const bool vi = i < M;
      const float4 pi = vi ? P[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      for (uint32_t jc = 0; jc <= ic; jc += kFrameThreads) {
        __syncthreads();
        if (jc + tid < M) tile[tid] = P[jc + tid];
        __syncthreads();
        uint32_t lim = (M - jc < (uint32_t)kFrameThreads) ? M - jc : (uint32_t)kFrameThreads;
        if (jc == ic) lim = (tid < lim) ? tid : lim;   // only j < i inside the diagonal tile
        if (!vi) lim = 0;
        // wave-uniform upper bound so that LDS reads stay broadcast; lanes mask themselves out
        uint32_t wlim = lim;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t t = __shfl_xor(wlim, o, ILCC_WAVE);
          wlim = t > wlim ? t : wlim;
        }
        for (uint32_t jj = 0; jj < wlim; ++jj) {
          const float4 q = tile[jj];
          const float dx = q.x - pi.x, dy = q.y - pi.y, dz = q.z - pi.z;
          float d2 = dx * dx;
          d2 = d2 + dy * dy;
          d2 = d2 + dz * dz;
          if (jj < lim && d2 < tol2) {
            // most neighbours already share a parent after the first few hooks: skip the find loops
            const uint32_t qi = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t qj = __hip_atomic_load(&parent[jc + jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (qi != qj) uf_unite(parent, i, jc + jj);
          }
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  K2_MARK(2);
  cluster_finish<true>(c, f, parent, sc);
}

// Labels, component sizes, exact 1-NN of the click, the reference's choice rule, compaction: shared by the LDS path
// (parent = the workgroup's LDS parents) and the multi-workgroup path (parent = the frame's global parents).
template <bool LDS_PARENT>
__device__ void cluster_finish(const Ctx& c, uint32_t f, uint32_t* parent, uint32_t* sc) {
  ilcc_result* r = &c.res[f];
  const uint32_t M = (uint32_t)r->n_roi;
  const uint64_t beg = c.off[f];
  const float4* __restrict__ P = c.roi + beg;
  uint32_t* gparent = c.uf_parent + beg;
  uint32_t* count = c.uf_count + beg;
  const uint32_t tid = threadIdx.x;
#ifdef ILCC_K2_TIMING
  __shared__ unsigned long long tmark[12];
#endif
  // ---- flatten: label = root (smallest member index)
  for (uint32_t base = 0; base < M; base += kFrameThreads) {
    const uint32_t i = base + tid;
    uint32_t root = 0;
    if (i < M) root = uf_find(parent, i);
    __syncthreads();
    if (i < M) {
      parent[i] = root;
      if (LDS_PARENT) gparent[i] = root;   // labels kept in global for the fetch/debug path
    }
    __syncthreads();
  }

  K2_MARK(3);
  // ---- component sizes (wave-aggregated atomics on the root's counter)
  for (uint32_t base = 0; base < M; base += kFrameThreads) {
    const uint32_t i = base + tid;
    const bool v = i < M;
    const uint32_t lab = v ? parent[i] : 0xFFFFFFFFu;
    unsigned long long todo = __ballot(v);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t ll = __shfl(lab, leader, ILCC_WAVE);
      const unsigned long long same = __ballot(v && lab == ll);
      if (lane_id() == leader) atomicAdd(&count[ll], (uint32_t)__popcll(same));
      todo &= ~same;
    }
  }
  __syncthreads();

  K2_MARK(4);
  // ---- exact 1-NN of the click (float squared distance, ties -> lowest index)
  const float cx = c.clicks[3 * f], cy = c.clicks[3 * f + 1], cz = c.clicks[3 * f + 2];
  NnKey best{3.402823466e38f, 0xFFFFFFFFu};
  for (uint32_t i = tid; i < M; i += kFrameThreads) {
    const float4 q = P[i];
    const float dx = q.x - cx, dy = q.y - cy, dz = q.z - cz;
    float d2 = dx * dx;
    d2 = d2 + dy * dy;
    d2 = d2 + dz * dz;
    const NnKey k{d2, i};
    if (nn_less(k, best)) best = k;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    NnKey t;
    t.d2 = __shfl_down(best.d2, o, ILCC_WAVE);
    t.idx = __shfl_down(best.idx, o, ILCC_WAVE);
    if (nn_less(t, best)) best = t;
  }
  float* scf = reinterpret_cast<float*>(sc);
  if (lane_id() == 0) {
    scf[wave_id()] = best.d2;
    sc[16 + wave_id()] = best.idx;
  }
  __syncthreads();
  if (tid == 0) {
    NnKey b{scf[0], sc[16]};
    for (int w = 1; w < kFrameThreads / ILCC_WAVE; ++w) {
      const NnKey k{scf[w], sc[16 + w]};
      if (nn_less(k, b)) b = k;
    }
    sc[32] = b.idx;
  }
  __syncthreads();
  const uint32_t nn = sc[32];
  const uint32_t nn_label = parent[nn];

  K2_MARK(5);
  // ---- largest valid component (ties -> smallest root), i.e. sorted index 0
  const uint32_t cmin = (uint32_t)c.p.cluster_min, cmax = (uint32_t)c.p.cluster_max;
  uint32_t bsz = 0, broot = 0xFFFFFFFFu;
  for (uint32_t i = tid; i < M; i += kFrameThreads) {
    if (parent[i] != i) continue;
    const uint32_t sz = __hip_atomic_load(&count[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sz < cmin || sz > cmax) continue;
    if (sz > bsz || (sz == bsz && i < broot)) {
      bsz = sz;
      broot = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t s2 = __shfl_down(bsz, o, ILCC_WAVE);
    const uint32_t r2 = __shfl_down(broot, o, ILCC_WAVE);
    if (s2 > bsz || (s2 == bsz && r2 < broot)) {
      bsz = s2;
      broot = r2;
    }
  }
  __syncthreads();
  if (lane_id() == 0) {
    sc[wave_id()] = bsz;
    sc[16 + wave_id()] = broot;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t s0 = sc[0], r0 = sc[16];
    for (int w = 1; w < kFrameThreads / ILCC_WAVE; ++w) {
      const uint32_t s2 = sc[w], r2 = sc[16 + w];
      if (s2 > s0 || (s2 == s0 && r2 < r0)) {
        s0 = s2;
        r0 = r2;
      }
    }
    const uint32_t nsz = __hip_atomic_load(&count[nn_label], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t chosen = r0;                                   // plane_index = 0
    const bool found = nsz >= cmin && nsz <= cmax;          // find_board of get_chessboard_by_point (:91-102)
    if (found) chosen = nn_label;                           // cluster containing the click's NN
    r->found_board = found ? 1 : 0;
    sc[33] = chosen;
    sc[34] = (s0 == 0) ? 0u : 1u;
  }
  __syncthreads();
  const uint32_t chosen = sc[33];
  const bool any = sc[34] != 0;
  if (!any) {
    if (tid == 0) r->status = ILCC_NO_CLUSTER;
    return;
  }

  K2_MARK(6);
  // ---- stable compaction of the chosen component
  float4* __restrict__ dst = c.cluster + beg;
  uint32_t running = 0;
  for (uint32_t base = 0; base < M; base += kFrameThreads) {
    const uint32_t i = base + tid;
    const bool keep = (i < M) && parent[i] == chosen;
    uint32_t tot;
    const uint32_t rank = block_rank(keep, sc + 40, tot);
    if (keep) dst[running + rank] = P[i];
    running += tot;
  }
  if (tid == 0) r->n_cluster = (int32_t)running;
  K2_MARK(7);
#ifdef ILCC_K2_TIMING
  if (f == 0 && tid == 0)
    printf("K2 f0 M=%u cycles: flatten %llu count %llu nn %llu largest %llu compact %llu\n", M, tmark[3] - tmark[2],
           tmark[4] - tmark[3], tmark[5] - tmark[4], tmark[6] - tmark[5], tmark[7] - tmark[6]);
#endif
}

// ------------------------------------------------------------------ frames above the LDS capacity: several workgroups per frame
constexpr int kBigChunk = 256;        // points per work item
constexpr int kBigBlock = 1024;       // listed frames whose chunk prefix sums a workgroup keeps in LDS at a time

// calls fn(f, first point of the chunk) for this workgroup's share of the (listed frame, chunk) items; every thread
// of the workgroup makes the same calls
template <typename Fn>
__device__ __forceinline__ void for_each_big_chunk(const Ctx& c, uint32_t* s_pref, Fn fn) {
  const uint32_t nbig = __hip_atomic_load(c.big_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t base = 0; base < nbig; base += kBigBlock) {
    const uint32_t nb = min((uint32_t)kBigBlock, nbig - base);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t run = 0;
      for (uint32_t j = 0; j < nb; ++j) {
        s_pref[j] = run;
        run += ((uint32_t)c.res[c.big_list[base + j]].n_roi + kBigChunk - 1) / kBigChunk;
      }
      s_pref[nb] = run;
    }
    __syncthreads();
    const uint32_t total = s_pref[nb];
    for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) {
      uint32_t lo = 0, hi = nb;   // last j with s_pref[j] <= item
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_pref[mid] <= item) lo = mid; else hi = mid;
      }
      fn(c.big_list[base + lo], (item - s_pref[lo]) * kBigChunk);
    }
  }
}

__device__ __forceinline__ void big_cell(const Ctx& c, const float4& q, int& cx, int& cy, int& cz) {
  const float inv_cell = 1.0f / ((float)c.p.cluster_tol * 1.001f);   // cells of slightly more than the tolerance
  cx = (int)floorf(q.x * inv_cell);
  cy = (int)floorf(q.y * inv_cell);
  cz = (int)floorf(q.z * inv_cell);
}

// spatial hash of a frame: buckets chained through `next` (bucket order depends on the race of the insertions, the
// resulting partition does not)
__device__ __forceinline__ void big_insert_point(const Ctx& c, uint32_t f, uint32_t i) {
  const uint64_t beg = c.off[f];
  uint32_t* head = c.uf_hash_head + (uint64_t)f * kClusterHashSize;
  int cx, cy, cz;
  big_cell(c, c.roi[beg + i], cx, cy, cz);
  c.uf_hash_next[beg + i] = atomicExch(&head[cell_hash(cx, cy, cz)], i);
}

// point i tests the 27 cells around its own; pairs within the tolerance whose parents differ are united
__device__ __forceinline__ void big_search_point(const Ctx& c, uint32_t f, uint32_t i, float tol2) {
  const uint64_t beg = c.off[f];
  const float4* __restrict__ P = c.roi + beg;
  uint32_t* parent = c.uf_parent + beg;
  const uint32_t* head = c.uf_hash_head + (uint64_t)f * kClusterHashSize;
  const uint32_t* next = c.uf_hash_next + beg;
  const float4 pi = P[i];
  int cx, cy, cz;
  big_cell(c, pi, cx, cy, cz);
  // the point's own cell (pairs once: j < i) and its 13 forward neighbours (every pair of adjacent cells is met once, from
  // the cell that comes first in (dz, dy, dx) order); two cells sharing a bucket only add distance tests that fail or repeat
  for (int r = 13; r < 27; ++r) {
      {
        const int dz = r / 9 - 1, dy = (r / 3) % 3 - 1, dx = r % 3 - 1;
        uint32_t j = __hip_atomic_load(&head[cell_hash(cx + dx, cy + dy, cz + dz)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (j != 0xFFFFFFFFu) {
          if (r != 13 ? j != i : j < i) {
            const float4 q = P[j];
            const float ex = q.x - pi.x, ey = q.y - pi.y, ez = q.z - pi.z;
            float d2 = ex * ex;
            d2 = d2 + ey * ey;
            d2 = d2 + ez * ez;
            if (d2 < tol2) {
              const uint32_t qi = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const uint32_t qj = __hip_atomic_load(&parent[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (qi != qj) uf_unite(parent, i, j);
            }
          }
          j = next[j];
        }
      }
  }
}

__global__ __launch_bounds__(kBigChunk) void k2l_insert(Ctx c) {
  __shared__ uint32_t s_pref[kBigBlock + 1];
  for_each_big_chunk(c, s_pref, [&](uint32_t f, uint32_t first) {
    const uint32_t i = first + threadIdx.x;
    if (i < (uint32_t)c.res[f].n_roi) big_insert_point(c, f, i);
  });
}

__global__ __launch_bounds__(kBigChunk) void k2l_search(Ctx c) {
  __shared__ uint32_t s_pref[kBigBlock + 1];
  const float tol2 = (float)(c.p.cluster_tol * c.p.cluster_tol);
  for_each_big_chunk(c, s_pref, [&](uint32_t f, uint32_t first) {
    const uint32_t i = first + threadIdx.x;
    if (i < (uint32_t)c.res[f].n_roi) big_search_point(c, f, i, tol2);
  });
}

__global__ __launch_bounds__(kFrameThreads) void k2l_finish(Ctx c) {
  __shared__ uint32_t sc[128];
  const uint32_t nbig = __hip_atomic_load(c.big_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x >= nbig) return;
  const uint32_t f = c.big_list[blockIdx.x];
  cluster_finish<false>(c, f, c.uf_parent + c.off[f], sc);
}

__global__ __launch_bounds__(kFrameThreads) void k2_seeded_cluster(Ctx c) {
  extern __shared__ __align__(16) unsigned char smem[];
  const uint32_t cap = c.cluster_lds_points;
  float4* tile = reinterpret_cast<float4*>(smem);                          // 16 KiB
  uint32_t* sc = reinterpret_cast<uint32_t*>(smem + sizeof(float4) * kFrameThreads);  // 128 words
  uint32_t* lds_parent = sc + 128;                                          // cap parents, cap keys, 32 KiB of cell run ends
  float4* s_pts = reinterpret_cast<float4*>(lds_parent + 2 * cap + kClusterCells / 2);  // cap points (+ cap u16 occupied cells)
  const uint32_t f = blockIdx.x;
  if (c.res[f].status != ILCC_OK) return;
  const uint32_t M = (uint32_t)c.res[f].n_roi;
  if (M <= cap) {
    cluster_frame(c, f, lds_parent, tile, sc, s_pts, cap);
    return;
  }
  // above the LDS capacity: reset the frame's parents, component counters and hash table ...
  const uint64_t beg = c.off[f];
  uint32_t* gparent = c.uf_parent + beg;
  uint32_t* count = c.uf_count + beg;
  uint32_t* head = c.uf_hash_head + (uint64_t)f * kClusterHashSize;
  for (uint32_t i = threadIdx.x; i < M; i += kFrameThreads) {
    gparent[i] = i;
    count[i] = 0u;
  }
  for (uint32_t k = threadIdx.x; k < (uint32_t)kClusterHashSize; k += kFrameThreads) head[k] = 0xFFFFFFFFu;
  if (c.big_armed) {
    // ... and list the frame for the multi-workgroup kernels that follow on the stream
    if (threadIdx.x == 0) c.big_list[atomicAdd(c.big_count, 1u)] = f;
    return;
  }
  // The handle has not met such a frame yet and did not launch those kernels (on a stream of ROI-cropped VLP-16 batches
  // their three empty launches cost 2.6 % of the frame rate): this workgroup does the same work alone -- same hash, same
  // pairs, same partition, ~4x slower per batch of large frames -- and the handle arms the fast path for its later batches
  // (ilcc_reserve arms it up front).
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < M; i += kFrameThreads) big_insert_point(c, f, i);
  __syncthreads();
  const float tol2 = (float)(c.p.cluster_tol * c.p.cluster_tol);
  for (uint32_t i = threadIdx.x; i < M; i += kFrameThreads) big_search_point(c, f, i, tol2);
  __syncthreads();
  cluster_finish<false>(c, f, gparent, sc);
}

static_assert(kClusterGridBuckets * sizeof(uint32_t) == kClusterCells * sizeof(uint16_t), "the two LDS cell-list layouts share one 32 KiB region");
size_t cluster_lds_bytes(uint32_t cap) {
  return sizeof(float4) * kFrameThreads + 128 * sizeof(uint32_t) + 2 * sizeof(uint32_t) * (size_t)cap + sizeof(uint16_t) * kClusterCells +
         sizeof(float4) * (size_t)cap + sizeof(uint16_t) * (size_t)cap;
}

// up to 152.5 KiB of dynamic LDS at the largest capacity (> the 64 KiB default cap).  Called by ilcc_create for the
// handle's device: the attribute is kept per (function, device).
hipError_t set_kernel_attributes_k2() {
  return hipFuncSetAttribute((const void*)k2_seeded_cluster, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)cluster_lds_bytes(kClusterLdsPointsMax));
}

void launch_cluster(const Ctx& c, hipStream_t s) {
  hipLaunchKernelGGL(k2_seeded_cluster, dim3(c.n_frames), dim3(kFrameThreads), cluster_lds_bytes(c.cluster_lds_points), s, c);
  if (!c.big_armed) return;   // no frame above the LDS capacity seen by this handle so far: see k2_seeded_cluster
  const uint32_t grid = c.big_grid;
  hipLaunchKernelGGL(k2l_insert, dim3(grid), dim3(kBigChunk), 0, s, c);
  hipLaunchKernelGGL(k2l_search, dim3(grid), dim3(kBigChunk), 0, s, c);
  hipLaunchKernelGGL(k2l_finish, dim3(c.n_frames), dim3(kFrameThreads), 0, s, c);
}

}  // namespace ilcc
