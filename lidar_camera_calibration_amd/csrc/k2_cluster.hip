// K2 seeded_cluster -- replaces pcl::EuclideanClusterExtraction + nearestKSearch of
// LidarCornersEst::EuclideanCluster (/root/reference/ilcc2/src/LidarCornersEst.cpp:124-153).
//
// Single-linkage components of the radius graph (squared float distance dx*dx+dy*dy+dz*dz < (float)(tol*tol),
// FLANN's strict test).  Round 4: the components are built on CELLS, not on points.
//
//   The frame's ROI points are binned into a grid of cells of side s = 0.57 tol.  3 s^2 = 0.975 tol^2, so any two points
//   of one cell pass the distance test (with 2.5 % to spare, against ~1e-5 of float rounding in the binning): a cell is a
//   clique and needs no test at all.  Two points within tol sit at most 2 cells apart on every axis (tol / s = 1.754 < 2),
//   so the components of the points are the components of the graph whose nodes are the occupied cells and whose edges
//   join two cells (within +-2 per axis) that hold at least one pair of points within tol.  A VLP-16 ROI of ~1350 points
//   has ~300 occupied cells and the union-find runs on those: the point-level search of rounds 1-3 tested every point
//   against every candidate of 14 cells of side tol (~250 k distance tests and ~1 k dependent LDS union chains per frame,
//   a 1024-thread / 100 KB workgroup that could not get a CU beside K6); this one probes a bitmap for the occupied
//   neighbours (5 neighbours per 64-bit window read), skips pairs whose cells already share a root and stops a pair's
//   tests at the first hit.  One workgroup of 256 threads (1024 in batches of <= 64 frames), ~50 KB of LDS: bitmap of the
//   padded bounding grid + its rank directory (cell key -> dense cell id), five words per cell, the points sorted by cell
//   (frames above the LDS point capacity keep the sorted points in HBM/L2 -- BASELINE config 5's ~12 k ROI points).
//   Labels are the smallest member index of a component (atomicMin over its cells), sizes are sums of cell counts: the
//   partition, the 1-NN of the click, the reference's choice rule and the index-ordered output are what they were.
//
//   Frames the grid cannot hold (bounding grid above kFineBits cells: the un-cropped clouds of the online caller
//   get_chessboard_by_point, LidarCornersEst.cpp:72-115; or more occupied cells than the handle's capacity) take the
//   point-level spatial hash in global memory: their own workgroup resets the frame's parents and hash table and lists
//   the frame; k2l_insert / k2l_search (persistent grids over (frame, 256-point chunk) items) unite neighbours with
//   device-scope atomics; k2l_finish labels, sizes, picks and compacts.  A handle that has not met such a frame does not
//   launch those kernels (2.6 % of the frame rate when empty): the frame's workgroup then does the same work alone.
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
  const uint32_t kT = blockDim.x;   // 256 or 1024 threads
#ifdef ILCC_K2_TIMING
  __shared__ unsigned long long tmark[12];
#endif
  // ---- flatten: label = root (smallest member index)
  for (uint32_t base = 0; base < M; base += kT) {
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
  for (uint32_t base = 0; base < M; base += kT) {
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
  for (uint32_t i = tid; i < M; i += kT) {
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
    for (int w = 1; w < (int)(kT / ILCC_WAVE); ++w) {
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
  for (uint32_t i = tid; i < M; i += kT) {
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
    for (int w = 1; w < (int)(kT / ILCC_WAVE); ++w) {
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
  for (uint32_t base = 0; base < M; base += kT) {
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

__global__ __launch_bounds__(1024) void k2l_finish(Ctx c) {
  __shared__ uint32_t sc[128];
  const uint32_t nbig = __hip_atomic_load(c.big_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x >= nbig) return;
  const uint32_t f = c.big_list[blockIdx.x];
  cluster_finish<false>(c, f, c.uf_parent + c.off[f], sc);
}

// ------------------------------------------------------------------ components on cells (the ROI case)
constexpr int kFineBits = 96 * 1024;             // cells of the padded bounding grid the bitmap holds (12 KiB); a 2 x 3 x 4 m ROI box at
                                                 // tol 0.12 is <= 34 x 48 x 63 = 102 k cells padded, the bounding box of real ROI clouds ~50 k
constexpr int kFineWords = kFineBits / 32 + 2;   // + 2: the 5-bit neighbour windows read one word past their own
constexpr float kFineCellOverTol = 0.57f;        // s / tol: 3 s^2 = 0.9747 tol^2 < tol^2 (cell = clique), tol / s = 1.754 < 2 (neighbours within +-2 cells)
// the 13 forward (dy, dz) rows of the 5 x 5 x 5 neighbourhood, nearest first: row 0 is the cell's own row (dx = +1, +2), every
// other row a window of five cells dx = -2..2.  A pair of cells is met once, from the one with the smaller (z, y, x).
__device__ const signed char kRowDy[13] = {0, 1, 0, 1, -1, 2, 0, 2, -2, 1, -1, 2, -2};
__device__ const signed char kRowDz[13] = {0, 0, 1, 1, 1, 0, 2, 1, 1, 2, 2, 2, 2};

struct FineGrid {
  float3 lo;
  float inv;
  int nx, ny, nz;
};
__device__ __forceinline__ uint32_t fine_key(const FineGrid& g, const float4& q) {
  // + 2: two cells of padding on every side, so that key + (dx, dy, dz) never leaves the grid
  const int cx = (int)floorf((q.x - g.lo.x) * g.inv) + 2, cy = (int)floorf((q.y - g.lo.y) * g.inv) + 2,
            cz = (int)floorf((q.z - g.lo.z) * g.inv) + 2;
  return (uint32_t)(cx + g.nx * (cy + g.ny * cz));
}
// dense id of an occupied cell: set bits below its key (rank directory per 64-bit word)
__device__ __forceinline__ uint32_t fine_rank(const uint32_t* bm, const uint16_t* pre, uint32_t key) {
  const unsigned long long w = reinterpret_cast<const unsigned long long*>(bm)[key >> 6];
  return (uint32_t)pre[key >> 6] + (uint32_t)__popcll(w & ((1ull << (key & 63u)) - 1ull));
}

// exclusive scan over the workgroup of one value per thread (thread order); total via ref.  scratch: >= 17 words.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* scratch, uint32_t& total) {
  uint32_t incl = v;
#pragma unroll
  for (int o = 1; o < ILCC_WAVE; o <<= 1) {
    const uint32_t t = __shfl_up(incl, o, ILCC_WAVE);
    if (lane_id() >= o) incl += t;
  }
  const int nw = (int)(blockDim.x / ILCC_WAVE);
  __syncthreads();
  if (lane_id() == ILCC_WAVE - 1) scratch[wave_id()] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < nw; ++w) {
    const uint32_t cw = scratch[w];
    if (w < wave_id()) base += cw;
    tot += cw;
  }
  total = tot;
  return base + incl - v;
}

struct FineLds {
  uint32_t* sc;      // 160 words of scratch
  uint32_t* bm;      // kFineWords: occupancy bitmap of the padded bounding grid
  uint16_t* pre;     // kFineBits / 64: occupied cells below each 64-bit word
  uint32_t *ckey, *cstart, *ccnt, *cmin, *cpar;   // per occupied cell (capacity c.cluster_cells_cap; cstart one more)
  float *sx, *sy, *sz;   // the points sorted by cell (capacity c.cluster_lds_points)
};

// Returns false (uniformly) when the frame does not fit the grid / the cell capacity: the caller takes the point-level path.
template <bool PTS_LDS>
__device__ bool fine_cluster_frame(const Ctx& c, uint32_t f, const FineLds& L) {
  ilcc_result* r = &c.res[f];
  const uint32_t M = (uint32_t)r->n_roi;
  const uint64_t beg = c.off[f];
  const float4* __restrict__ P = c.roi + beg;
  float4* gpts = c.cluster + beg;   // PTS_LDS = false: the sorted points live here until the compaction overwrites it
  const uint32_t tid = threadIdx.x, kT = blockDim.x;
  const int nwv = (int)(kT / ILCC_WAVE);
  const float tol2 = (float)(c.p.cluster_tol * c.p.cluster_tol);
  uint32_t* sc = L.sc;
  float* scf = reinterpret_cast<float*>(sc);
  unsigned long long* stats = c.grid_iters + 3 * kIterSlots;   // [0] most occupied cells a frame needed, [1] frames the grid could not hold

  // ---- bounding box
  float3 lo = make_float3(3.0e38f, 3.0e38f, 3.0e38f), hi = make_float3(-3.0e38f, -3.0e38f, -3.0e38f);
  for (uint32_t i = tid; i < M; i += kT) {
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
  __syncthreads();
  if (lane_id() == 0) {
    scf[wave_id()] = lo.x; scf[16 + wave_id()] = lo.y; scf[32 + wave_id()] = lo.z;
    scf[64 + wave_id()] = hi.x; scf[80 + wave_id()] = hi.y; scf[96 + wave_id()] = hi.z;
  }
  __syncthreads();
  for (int w = 0; w < nwv; ++w) {
    lo.x = fminf(lo.x, scf[w]); lo.y = fminf(lo.y, scf[16 + w]); lo.z = fminf(lo.z, scf[32 + w]);
    hi.x = fmaxf(hi.x, scf[64 + w]); hi.y = fmaxf(hi.y, scf[80 + w]); hi.z = fmaxf(hi.z, scf[96 + w]);
  }
  __syncthreads();
  FineGrid g;
  g.lo = lo;
  g.inv = 1.0f / ((float)c.p.cluster_tol * kFineCellOverTol);
  const float ex = (hi.x - lo.x) * g.inv, ey = (hi.y - lo.y) * g.inv, ez = (hi.z - lo.z) * g.inv;
  if (!(ex < 8192.f && ey < 8192.f && ez < 8192.f)) {   // (also catches a NaN extent)
    if (tid == 0) atomicAdd(&stats[1], 1ull);
    return false;
  }
  g.nx = (int)floorf(ex) + 5;
  g.ny = (int)floorf(ey) + 5;
  g.nz = (int)floorf(ez) + 5;
  const unsigned long long cells = (unsigned long long)g.nx * (unsigned long long)g.ny * (unsigned long long)g.nz;
  if (cells > (unsigned long long)kFineBits) {
    if (tid == 0) atomicAdd(&stats[1], 1ull);
    return false;
  }
  const uint32_t n_w32 = ((uint32_t)cells + 31u) / 32u + 2u, n_w64 = ((uint32_t)cells + 63u) / 64u;

  // ---- occupancy bitmap, rank directory
  for (uint32_t k = tid; k < ((n_w32 + 1u) & ~1u); k += kT) L.bm[k] = 0u;
  __syncthreads();
  for (uint32_t i = tid; i < M; i += kT) {
    const uint32_t key = fine_key(g, P[i]);
    atomicOr(&L.bm[key >> 5], 1u << (key & 31u));
  }
  __syncthreads();
  uint32_t C = 0;
  {
    const unsigned long long* bm64 = reinterpret_cast<const unsigned long long*>(L.bm);
    const uint32_t per = (n_w64 + kT - 1u) / kT, w0 = tid * per, w1 = min(n_w64, w0 + per);
    uint32_t sum = 0;
    for (uint32_t w = w0; w < w1; ++w) sum += (uint32_t)__popcll(bm64[w]);
    uint32_t run = block_excl_scan(sum, sc, C);
    if (C <= c.cluster_cells_cap)
      for (uint32_t w = w0; w < w1; ++w) {
        L.pre[w] = (uint16_t)run;
        run += (uint32_t)__popcll(bm64[w]);
      }
  }
  if (tid == 0) atomicMax(&stats[0], (unsigned long long)C);
  if (C > c.cluster_cells_cap) return false;   // more occupied cells than the handle's LDS arrays hold: it grows them for the next batch
  __syncthreads();

  // ---- points per cell, smallest member index, key of every occupied cell
  for (uint32_t k = tid; k < C; k += kT) {
    L.ccnt[k] = 0u;
    L.cmin[k] = 0xFFFFFFFFu;
  }
  __syncthreads();
  for (uint32_t i = tid; i < M; i += kT) {
    const uint32_t key = fine_key(g, P[i]);
    const uint32_t cid = fine_rank(L.bm, L.pre, key);
    atomicAdd(&L.ccnt[cid], 1u);
    atomicMin(&L.cmin[cid], i);
    L.ckey[cid] = key;   // (every point of the cell writes the same word)
  }
  __syncthreads();
  {   // exclusive scan of the counts: cstart[k] = first sorted position of cell k
    const uint32_t per = (C + kT - 1u) / kT, k0 = tid * per, k1 = min(C, k0 + per);
    uint32_t sum = 0, tot;
    for (uint32_t k = k0; k < k1; ++k) sum += L.ccnt[k];
    uint32_t run = block_excl_scan(sum, sc, tot);
    for (uint32_t k = k0; k < k1; ++k) {
      L.cstart[k] = run;
      run += L.ccnt[k];
    }
    if (tid == 0) L.cstart[C] = M;
  }
  __syncthreads();
  // placement (the order inside a cell is the race of the atomics: nothing below depends on it); leaves ccnt all zero
  for (uint32_t i = tid; i < M; i += kT) {
    const float4 q = P[i];
    const uint32_t cid = fine_rank(L.bm, L.pre, fine_key(g, q));
    const uint32_t at = L.cstart[cid] + (atomicSub(&L.ccnt[cid], 1u) - 1u);
    if (PTS_LDS) {
      L.sx[at] = q.x;
      L.sy[at] = q.y;
      L.sz[at] = q.z;
    } else {
      gpts[at] = q;
    }
  }
  for (uint32_t k = tid; k < C; k += kT) L.cpar[k] = k;
  if (!PTS_LDS) __threadfence_block();
  __syncthreads();

  // ---- edges between cells.  (row, cell) items, rows outermost: by the time the far rows come up most cells of a
  // surface already share a root and a pair costs two finds.
  for (int row = 0; row < 13; ++row) {
    const int delta = (int)kRowDy[row] * g.nx + (int)kRowDz[row] * g.nx * g.ny;
    for (uint32_t A = tid; A < C; A += kT) {
      const uint32_t lo_bit = (uint32_t)((int)L.ckey[A] + delta - 2);
      const uint32_t w = lo_bit >> 5, sh = lo_bit & 31u;
      uint32_t bits = (uint32_t)((((unsigned long long)L.bm[w + 1] << 32) | (unsigned long long)L.bm[w]) >> sh) & 31u;
      if (row == 0) bits &= 24u;   // the cell's own row: only dx = +1, +2
      if (!bits) continue;
      const uint32_t a0 = L.cstart[A], a1 = L.cstart[A + 1];
      while (bits) {
        const uint32_t b = (uint32_t)__ffs((int)bits) - 1u;
        bits &= bits - 1u;
        const uint32_t B = fine_rank(L.bm, L.pre, lo_bit + b);
        const uint32_t ra = uf_find(L.cpar, A), rb = uf_find(L.cpar, B);
        if (ra == rb) continue;   // same component already (for good)
        const uint32_t b0 = L.cstart[B], b1 = L.cstart[B + 1];
        bool hit = false;
        for (uint32_t ia = a0; ia < a1 && !hit; ++ia) {
          float px, py, pz;
          if (PTS_LDS) { px = L.sx[ia]; py = L.sy[ia]; pz = L.sz[ia]; }
          else { const float4 q = gpts[ia]; px = q.x; py = q.y; pz = q.z; }
          for (uint32_t ib = b0; ib < b1; ++ib) {
            float qx, qy, qz;
            if (PTS_LDS) { qx = L.sx[ib]; qy = L.sy[ib]; qz = L.sz[ib]; }
            else { const float4 q = gpts[ib]; qx = q.x; qy = q.y; qz = q.z; }
            const float ex2 = qx - px, ey2 = qy - py, ez2 = qz - pz;
            float d2 = ex2 * ex2;
            d2 = d2 + ey2 * ey2;
            d2 = d2 + ez2 * ez2;
            if (d2 < tol2) {
              hit = true;
              break;
            }
          }
        }
        if (hit) uf_unite(L.cpar, ra, rb);
      }
    }
  }
  __syncthreads();

  // ---- components: root of every cell, size and smallest member index on the root (ccnt is zero since the placement;
  // ckey is free now and takes the component's smallest index)
  for (uint32_t base = 0; base < C; base += kT) {
    const uint32_t k = base + tid;
    uint32_t root = 0;
    if (k < C) root = uf_find(L.cpar, k);
    __syncthreads();
    if (k < C) {
      L.cpar[k] = root;
      L.ckey[k] = 0xFFFFFFFFu;
    }
    __syncthreads();
  }
  for (uint32_t k = tid; k < C; k += kT) {
    const uint32_t root = L.cpar[k];
    atomicAdd(&L.ccnt[root], L.cstart[k + 1] - L.cstart[k]);
    atomicMin(&L.ckey[root], L.cmin[k]);
  }
  __syncthreads();

  // ---- exact 1-NN of the click (float squared distance, ties -> lowest index)
  const float kx = c.clicks[3 * f], ky = c.clicks[3 * f + 1], kz = c.clicks[3 * f + 2];
  NnKey best{3.402823466e38f, 0xFFFFFFFFu};
  for (uint32_t i = tid; i < M; i += kT) {
    const float4 q = P[i];
    const float dx = q.x - kx, dy = q.y - ky, dz = q.z - kz;
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
  // ---- largest admissible component (ties -> smallest member index), i.e. sorted index 0
  const uint32_t cmin_sz = (uint32_t)c.p.cluster_min, cmax_sz = (uint32_t)c.p.cluster_max;
  uint32_t bsz = 0, bidx = 0xFFFFFFFFu, broot = 0xFFFFFFFFu;
  for (uint32_t k = tid; k < C; k += kT) {
    if (L.cpar[k] != k) continue;
    const uint32_t sz = L.ccnt[k], mi = L.ckey[k];
    if (sz < cmin_sz || sz > cmax_sz) continue;
    if (sz > bsz || (sz == bsz && mi < bidx)) {
      bsz = sz;
      bidx = mi;
      broot = k;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t s2 = __shfl_down(bsz, o, ILCC_WAVE), i2 = __shfl_down(bidx, o, ILCC_WAVE), r2 = __shfl_down(broot, o, ILCC_WAVE);
    if (s2 > bsz || (s2 == bsz && i2 < bidx)) {
      bsz = s2;
      bidx = i2;
      broot = r2;
    }
  }
  __syncthreads();
  if (lane_id() == 0) {
    scf[wave_id()] = best.d2;
    sc[16 + wave_id()] = best.idx;
    sc[32 + wave_id()] = bsz;
    sc[48 + wave_id()] = bidx;
    sc[64 + wave_id()] = broot;
  }
  __syncthreads();
  if (tid == 0) {
    NnKey nb{scf[0], sc[16]};
    uint32_t s0 = sc[32], i0 = sc[48], r0 = sc[64];
    for (int w = 1; w < nwv; ++w) {
      const NnKey k{scf[w], sc[16 + w]};
      if (nn_less(k, nb)) nb = k;
      const uint32_t s2 = sc[32 + w], i2 = sc[48 + w], r2 = sc[64 + w];
      if (s2 > s0 || (s2 == s0 && i2 < i0)) {
        s0 = s2;
        i0 = i2;
        r0 = r2;
      }
    }
    const uint32_t nn_root = L.cpar[fine_rank(L.bm, L.pre, fine_key(g, P[nb.idx]))];
    const uint32_t nsz = L.ccnt[nn_root];
    const bool found = nsz >= cmin_sz && nsz <= cmax_sz;   // find_board of get_chessboard_by_point (:91-102)
    r->found_board = found ? 1 : 0;
    sc[120] = found ? nn_root : r0;                         // cluster containing the click's NN, else plane_index = 0
    sc[121] = (s0 == 0) ? 0u : 1u;
  }
  __syncthreads();
  const uint32_t chosen = sc[120];
  if (sc[121] == 0u) {
    if (tid == 0) r->status = ILCC_NO_CLUSTER;
    return true;
  }

  // ---- stable compaction of the chosen component (index order)
  float4* __restrict__ dst = c.cluster + beg;
  uint32_t running = 0;
  for (uint32_t base = 0; base < M; base += kT) {
    const uint32_t i = base + tid;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    bool keep = false;
    if (i < M) {
      q = P[i];
      keep = L.cpar[fine_rank(L.bm, L.pre, fine_key(g, q))] == chosen;
    }
    uint32_t tot;
    const uint32_t rank = block_rank(keep, sc + 128, tot);
    if (keep) dst[running + rank] = q;
    running += tot;
  }
  if (tid == 0) r->n_cluster = (int32_t)running;
  return true;
}

size_t cluster_lds_bytes(uint32_t pts_cap, uint32_t cells_cap) {
  return 160 * sizeof(uint32_t) + sizeof(uint32_t) * kFineWords + sizeof(uint16_t) * (kFineBits / 64) +
         sizeof(uint32_t) * (5 * (size_t)cells_cap + 2) + 3 * sizeof(float) * (size_t)pts_cap;
}

__global__ __launch_bounds__(1024) void k2_seeded_cluster(Ctx c) {
  extern __shared__ __align__(16) unsigned char smem[];
  FineLds L;
  L.sc = reinterpret_cast<uint32_t*>(smem);                  // 160 words
  L.bm = L.sc + 160;                                        // 8-byte aligned: read as 64-bit words, too
  L.pre = reinterpret_cast<uint16_t*>(L.bm + kFineWords);
  const uint32_t cc = c.cluster_cells_cap;
  L.ckey = reinterpret_cast<uint32_t*>(L.pre + kFineBits / 64);
  L.cstart = L.ckey + cc;
  L.ccnt = L.cstart + cc + 2;
  L.cmin = L.ccnt + cc;
  L.cpar = L.cmin + cc;
  L.sx = reinterpret_cast<float*>(L.cpar + cc);
  L.sy = L.sx + c.cluster_lds_points;
  L.sz = L.sy + c.cluster_lds_points;
  const uint32_t f = blockIdx.x;
  if (c.res[f].status != ILCC_OK) return;
  const uint32_t M = (uint32_t)c.res[f].n_roi;
  const bool done = (M <= c.cluster_lds_points) ? fine_cluster_frame<true>(c, f, L) : fine_cluster_frame<false>(c, f, L);
  if (done) return;
  // the grid cannot hold this frame: point-level spatial hash in global memory.  Reset the frame's parents, component
  // counters and hash table ...
  __syncthreads();
  const uint32_t kT = blockDim.x;
  const uint64_t beg = c.off[f];
  uint32_t* gparent = c.uf_parent + beg;
  uint32_t* count = c.uf_count + beg;
  uint32_t* head = c.uf_hash_head + (uint64_t)f * kClusterHashSize;
  for (uint32_t i = threadIdx.x; i < M; i += kT) {
    gparent[i] = i;
    count[i] = 0u;
  }
  for (uint32_t k = threadIdx.x; k < (uint32_t)kClusterHashSize; k += kT) head[k] = 0xFFFFFFFFu;
  if (c.big_armed) {
    // ... and list the frame for the multi-workgroup kernels that follow on the stream
    if (threadIdx.x == 0) c.big_list[atomicAdd(c.big_count, 1u)] = f;
    return;
  }
  // The handle has not met such a frame yet and did not launch those kernels (on a stream of ROI-cropped VLP-16 batches
  // their three empty launches cost 2.6 % of the frame rate): this workgroup does the same work alone -- same hash, same
  // pairs, same partition, several times slower per batch of such frames -- and the handle arms the multi-workgroup path
  // for its later batches (ilcc_reserve arms it up front).
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < M; i += kT) big_insert_point(c, f, i);
  __syncthreads();
  const float tol2 = (float)(c.p.cluster_tol * c.p.cluster_tol);
  for (uint32_t i = threadIdx.x; i < M; i += kT) big_search_point(c, f, i, tol2);
  __syncthreads();
  cluster_finish<false>(c, f, gparent, L.sc);
}

// up to 160 KiB of dynamic LDS (> the 64 KiB default cap).  Called by ilcc_create for the handle's device: the attribute
// is kept per (function, device).
hipError_t set_kernel_attributes_k2() {
  return hipFuncSetAttribute((const void*)k2_seeded_cluster, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void launch_cluster(const Ctx& c, hipStream_t s) {
  const int threads = c.n_frames <= (uint32_t)kSmallBatchFrames ? 1024 : kFrameThreads;
  hipLaunchKernelGGL(k2_seeded_cluster, dim3(c.n_frames), dim3(threads), cluster_lds_bytes(c.cluster_lds_points, c.cluster_cells_cap), s, c);
  if (!c.big_armed) return;   // no frame the cell grid could not hold seen by this handle so far: see k2_seeded_cluster
  const uint32_t grid = c.big_grid;
  hipLaunchKernelGGL(k2l_insert, dim3(grid), dim3(kBigChunk), 0, s, c);
  hipLaunchKernelGGL(k2l_search, dim3(grid), dim3(kBigChunk), 0, s, c);
  hipLaunchKernelGGL(k2l_finish, dim3(c.n_frames), dim3(1024), 0, s, c);
}

}  // namespace ilcc
