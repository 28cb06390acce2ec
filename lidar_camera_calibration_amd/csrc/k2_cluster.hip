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
//   Frames the LDS grid cannot hold (bounding grid above kFineBits cells: the un-cropped clouds of the online caller
//   get_chessboard_by_point, LidarCornersEst.cpp:72-115; or more occupied cells than the handle's capacity) run the same
//   cell-level algorithm with the cells in a hash table in global memory (hashed_cluster_frame, round 5); only frames beyond
//   THAT path's limits (> 65 536 ROI points) take the point-level spatial hash of rounds 1-3, one workgroup.
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

// ------------------------------------------------------------------ point-level spatial hash (frames beyond the limits of the cell paths)
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

// ------------------------------------------------------------------ components on cells (the ROI case)
constexpr int kFineBitsOnline = 128 * 1024;      // ... in the first tier of the online caller: a +-1.25 m window at tol 0.10 is <= 49^3 = 118 k cells padded
constexpr int kFineBits = 96 * 1024;             // cells of the padded bounding grid the bitmap holds (12 KiB); a 2 x 3 x 4 m ROI box at
                                                 // tol 0.12 is <= 34 x 48 x 63 = 102 k cells padded, the bounding box of real ROI clouds ~50 k
__host__ __device__ inline uint32_t fine_words(uint32_t bits) { return bits / 32u + 2u; }   // + 2: the 5-bit neighbour windows read one word past their own
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
  uint32_t* bm;      // fine_words(cluster_bits): occupancy bitmap of the padded bounding grid
  uint16_t* pre;     // cluster_bits / 64: occupied cells below each 64-bit word
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
  unsigned long long* stats = c.grid_iters + 3 * kIterSlots;   // [0] most occupied cells a frame needed

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
  if (!(ex < 8192.f && ey < 8192.f && ez < 8192.f)) return false;   // (also catches a NaN extent)
  g.nx = (int)floorf(ex) + 5;
  g.ny = (int)floorf(ey) + 5;
  g.nz = (int)floorf(ez) + 5;
  const unsigned long long cells = (unsigned long long)g.nx * (unsigned long long)g.ny * (unsigned long long)g.nz;
  if (cells > (unsigned long long)c.cluster_bits) return false;
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
    // First tier of the online caller: the points are a WINDOW of +-w around the click, the answer must be the whole cloud's.
    // It is, when (1) the window's nearest point is the cloud's -- nothing outside the window is nearer than w; (2) that point's
    // component is admissible -- otherwise the reference falls back to the LARGEST component of the cloud, which a window cannot
    // know; (3) the component is complete -- checked below: no member within tol (+ 1 mm) of the window's faces, so no point
    // outside the window is within tol of it.  Anything else: the second tier clusters the whole cloud.
    if (c.online_tier == 1u) {
      const float w = c.online_window - 1e-3f;   // (the window's faces are float-rounded: a point outside is farther than this)
      sc[122] = (found && nb.d2 <= w * w) ? 1u : 0u;
    }
  }
  __syncthreads();
  const uint32_t chosen = sc[120];
  if (c.online_tier == 1u && sc[122] == 0u) {
    if (tid == 0) c.frame_flags[f] = 1u;
    return true;
  }
  if (sc[121] == 0u) {
    if (tid == 0) r->status = ILCC_NO_CLUSTER;
    return true;
  }

  // ---- stable compaction of the chosen component (index order)
  float4* __restrict__ dst = c.cluster + beg;
  uint32_t running = 0;
  const float safe = c.online_window - ((float)c.p.cluster_tol + 1e-3f);   // (first tier of the online caller: completeness)
  bool near_face = false;
  for (uint32_t base = 0; base < M; base += kT) {
    const uint32_t i = base + tid;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    bool keep = false;
    if (i < M) {
      q = P[i];
      keep = L.cpar[fine_rank(L.bm, L.pre, fine_key(g, q))] == chosen;
    }
    if (keep) near_face |= !(fabsf(q.x - kx) <= safe && fabsf(q.y - ky) <= safe && fabsf(q.z - kz) <= safe);
    uint32_t tot;
    const uint32_t rank = block_rank(keep, sc + 128, tot);
    if (keep) dst[running + rank] = q;
    running += tot;
  }
  if (c.online_tier == 1u) {
    uint32_t tot;
    (void)block_rank(near_face, sc + 128, tot);
    if (tot != 0u) {   // the component may continue outside the window
      if (tid == 0) c.frame_flags[f] = 1u;
      return true;
    }
  }
  if (tid == 0) r->n_cluster = (int32_t)running;
  return true;
}

// ------------------------------------------------------------------ components on cells, the cells in a HASH TABLE (round 5)
// Frames the LDS cell grid cannot hold -- the un-cropped clouds of the online caller (get_chessboard_by_point,
// LidarCornersEst.cpp:72-115: a bounding grid of 1 269 x 1 720 x 249 cells against a 96 k-bit bitmap), or more occupied cells than
// the LDS arrays -- used to fall to the POINT-level spatial hash (every point against every point of 14 cells of side tol, chained
// buckets, device-scope atomics from many workgroups: 3.75 ms per 128 un-cropped VLP-16 clouds).  They now run the cell-level
// algorithm of fine_cluster_frame -- cells of side 0.57 tol are cliques, edges join occupied cells within +-2 per axis that hold a
// pair of points within tol -- with the dense bitmap replaced by an open-addressing hash table in global memory keyed by BLOCKS
// of 4 x 4 x 4 cells: an entry holds the block's 64-bit occupancy mask and the dense id of its first cell, so the 124-cell
// neighbourhood of a cell is read with at most 7 table lookups (the 2 x 2 x 2 blocks its 5-cell windows overlap, minus its own).
// ONE workgroup per frame, and every atomic at WORKGROUP scope: on this part a device-scope atomic or load is executed beyond
// the XCD's L2 (the L2s of the eight XCDs are not coherent with each other inside a kernel) at several microseconds a piece --
// measured here: 8 400 cycles per table lookup with device-scope loads, and a chain of kernels over (frame, chunk) items with
// device-scope atomics took 2.7 ms for the same 128 clouds.  A frame's data is touched by its own workgroup only, whose
// read-modify-writes then execute in the L2 of its XCD.  Plain loads may hit stale lines of the CU's L1: every phase boundary
// invalidates it (acquire fence), and inside a phase staleness is benign by construction -- a table word goes from empty to a key
// once (a stale "empty" only costs a compare-and-swap that returns the key), union-find parents only ever move to smaller
// indices (a stale parent is still an ancestor) and a failed hook continues from the value the compare-and-swap returned.
// Scratch per frame (all dead at K2 time, rewritten by K3 / K4 / K5 / K5w / K7b afterwards): table keys = the frame's slice of
// uf_hash_head; base ids = board; masks = pca (low words) + optim (high words); per point / per cell words = uf_hash_next
// (cell of a point), uf_count (count, later component size), uf_parent (cell union-find), yz (smallest index | block-and-bit,
// later the root), walk_yz (sorted start).  Same cells, same pair arithmetic, same union rule, labels = smallest member index:
// the partition, the choice and the index-ordered output are those of the other paths (tests: the online caller, the wide sparse
// ROI and the large ROIs against the oracle's BFS).
constexpr uint32_t kHashEmpty = 0xFFFFFFFFu;
constexpr uint32_t kHashPointsMax = 65536u;            // frames above this keep the point-level path (table at most half full)
static_assert(2u * kHashPointsMax <= (uint32_t)kClusterHashSize, "hash table capacity");

constexpr uint32_t kHashMinFramePoints = 256;   // hashed-block path: 4 words per input point hold the smallest (1024-word) block table
struct HashFrame {
  float lox, loy, loz, inv;
  int32_t nbx, nby;       // blocks per axis (x, y); keys are bx + nbx * (by + nby * bz)
  uint32_t mask, shift;   // table size - 1, 32 - log2(size)
};
struct HashViews {
  const float4* P;
  float4* sorted;
  uint32_t *tkey, *tbase, *tlo, *thi;
  uint32_t *cell_of, *ccnt, *cpar, *cmin, *ckey, *cstart;
  uint32_t M;
};
__device__ __forceinline__ HashViews hash_views(const Ctx& c, uint32_t f) {
  const uint64_t beg = c.off[f];
  HashViews v;
  v.M = (uint32_t)c.res[f].n_roi;
  v.P = c.roi + beg;
  v.sorted = c.cluster + beg;
  v.tkey = c.uf_hash_head + (uint64_t)f * kClusterHashSize;
  v.tbase = reinterpret_cast<uint32_t*>(c.board + beg);
  v.tlo = reinterpret_cast<uint32_t*>(c.pca + beg);
  v.thi = reinterpret_cast<uint32_t*>(c.optim + beg);
  v.cell_of = c.uf_hash_next + beg;
  v.ccnt = c.uf_count + beg;
  v.cpar = c.uf_parent + beg;
  v.cmin = reinterpret_cast<uint32_t*>(c.yz + beg);
  v.ckey = v.cmin + v.M;
  v.cstart = reinterpret_cast<uint32_t*>(c.walk_yz + beg);   // M + 1 <= 2 M words
  return v;
}
__device__ __forceinline__ uint32_t hb_slot(const HashFrame& g, uint32_t key) { return (key * 0x9E3779B1u) >> g.shift; }
// block key and bit of a point's cell (cell coordinates + 2, as fine_key: neighbours never leave the padded grid)
__device__ __forceinline__ void hb_cell(const HashFrame& g, const float4& q, uint32_t& key, uint32_t& bit) {
  const int cx = (int)floorf((q.x - g.lox) * g.inv) + 2, cy = (int)floorf((q.y - g.loy) * g.inv) + 2,
            cz = (int)floorf((q.z - g.loz) * g.inv) + 2;
  key = (uint32_t)((cx >> 2) + g.nbx * ((cy >> 2) + g.nby * (cz >> 2)));
  bit = (uint32_t)((cx & 3) | ((cy & 3) << 2) | ((cz & 3) << 4));
}
__device__ __forceinline__ uint32_t hb_rank(uint32_t lo, uint32_t hi, uint32_t bit) {   // set bits below `bit`
  return bit < 32u ? (uint32_t)__popc(lo & ((1u << bit) - 1u)) : (uint32_t)__popc(lo) + (uint32_t)__popc(hi & ((1u << (bit - 32u)) - 1u));
}
// workgroup-scope read-modify-writes on global memory: executed in the XCD's L2
#define WG_ATOMIC(op, ptr, val) __hip_atomic_##op((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
__device__ __forceinline__ uint32_t wg_cas(uint32_t* p, uint32_t expect, uint32_t desired) {
  __hip_atomic_compare_exchange_strong(p, &expect, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return expect;   // the value found (== the `expect` passed in iff the exchange happened)
}
// phase boundary of a one-workgroup algorithm on global memory: everyone's stores and atomics are out, and nobody reads a stale L1 line
__device__ __forceinline__ void wg_phase() {
  __threadfence_block();
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (buffer_inv: invalidates the CU's L1)
}
// union-find on global parents owned by ONE workgroup: plain loads (a stale parent is still an ancestor: parents only move to
// smaller indices), hooks by workgroup-scope compare-and-swap; a failed hook continues from the parent the CAS returned
#ifdef ILCC_K2_TIMING
__device__ unsigned long long g_k2_hops, g_k2_finds, g_k2_pairs, g_k2_hits, g_k2_t[6];
#define K2_T(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; } while (0)
#define K2_COUNT(v, n) atomicAdd(&(v), (unsigned long long)(n))
#else
#define K2_COUNT(v, n) do {} while (0)
#endif
__device__ __forceinline__ uint32_t wg_find(uint32_t* parent, uint32_t x) {
  K2_COUNT(g_k2_finds, 1);
  for (;;) {
    K2_COUNT(g_k2_hops, 1);
    const uint32_t p = parent[x];
    if (p == x) return x;
    const uint32_t gp = parent[p];
    if (gp != p) parent[x] = gp;   // path halving (a non-root's word: hooks only ever touch roots)
    x = p;
  }
}
__device__ __forceinline__ void wg_unite(uint32_t* parent, uint32_t a, uint32_t b) {
  for (;;) {
    a = wg_find(parent, a);
    b = wg_find(parent, b);
    if (a == b) return;
    if (a < b) {
      const uint32_t t = a;
      a = b;
      b = t;
    }
    const uint32_t old = wg_cas(&parent[a], a, b);   // hook the larger root under the smaller one
    if (old == a) return;
    a = old;   // a was no root any more (our copy was stale): its real parent is smaller -- the loop terminates
  }
}

// Returns false (uniformly) when the frame is outside this path's limits: the caller takes the point-level path.
__device__ bool hashed_cluster_frame(const Ctx& c, uint32_t f, uint32_t* sc) {
  ilcc_result* r = &c.res[f];
  const HashViews v = hash_views(c, f);
  const uint32_t M = v.M;
  if (M == 0u || M > kHashPointsMax) return false;
  // the block arrays (tbase / tlo / thi: at least 1024 words each) live in the frame's OWN slices of board / pca / optim, four
  // words per INPUT point: a frame of fewer than 256 input points cannot hold them (it wrote into its neighbour's)
  if (c.off[f + 1] - c.off[f] < (uint64_t)kHashMinFramePoints) return false;
  const uint32_t tid = threadIdx.x, kT = blockDim.x;
  const int nwv = (int)(kT / ILCC_WAVE);
  const float tol2 = (float)(c.p.cluster_tol * c.p.cluster_tol);
  float* scf = reinterpret_cast<float*>(sc);
#ifdef ILCC_K2_TIMING
  __shared__ unsigned long long hmark[12];
#define HC_MARK(k) do { __syncthreads(); if (threadIdx.x == 0) hmark[k] = __builtin_readcyclecounter(); } while (0)
#else
#define HC_MARK(k) do {} while (0)
#endif
  HC_MARK(0);

  // ---- bounding box -> block grid (as in fine_cluster_frame)
  float3 lo = make_float3(3.0e38f, 3.0e38f, 3.0e38f), hi = make_float3(-3.0e38f, -3.0e38f, -3.0e38f);
  for (uint32_t i = tid; i < M; i += kT) {
    const float4 q = v.P[i];
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
  HashFrame g;
  g.lox = lo.x; g.loy = lo.y; g.loz = lo.z;
  g.inv = 1.0f / ((float)c.p.cluster_tol * kFineCellOverTol);
  const float ex = (hi.x - lo.x) * g.inv, ey = (hi.y - lo.y) * g.inv, ez = (hi.z - lo.z) * g.inv;
  if (!(ex < 8192.f && ey < 8192.f && ez < 8192.f)) return false;   // (also a NaN extent)
  {
    const int nx = (int)floorf(ex) + 5, ny = (int)floorf(ey) + 5, nz = (int)floorf(ez) + 5;
    g.nbx = (nx + 3) >> 2;
    g.nby = (ny + 3) >> 2;
    // 32-bit keys, 0xFFFFFFFF = empty
    if ((unsigned long long)g.nbx * (unsigned long long)g.nby * (unsigned long long)((nz + 3) >> 2) >= 0xFFFFFFFFull) return false;
  }
  uint32_t size = 1024u;
  while (size < 2u * M) size <<= 1;             // <= 2^17: blocks <= cells <= points, the table at most half full
  g.mask = size - 1u;
  g.shift = 32u - (uint32_t)__builtin_ctz(size);
  HC_MARK(1);

  // ---- reset
  for (uint32_t k = tid; k < size; k += kT) {
    v.tkey[k] = kHashEmpty;
    v.tlo[k] = 0u;
    v.thi[k] = 0u;
  }
  for (uint32_t i = tid; i < M; i += kT) {
    v.ccnt[i] = 0u;
    v.cmin[i] = 0xFFFFFFFFu;
    v.cpar[i] = i;
  }
  wg_phase();
  HC_MARK(2);
  // ---- every point: its block's entry (claimed with a compare-and-swap on the key), its cell's bit
  for (uint32_t i = tid; i < M; i += kT) {
    uint32_t key, bit;
    hb_cell(g, v.P[i], key, bit);
    uint32_t h = hb_slot(g, key);
    for (;;) {
      uint32_t e = v.tkey[h];
      if (e == kHashEmpty) e = wg_cas(&v.tkey[h], kHashEmpty, key);   // (a stale "empty" costs this CAS, which returns the key)
      if (e == kHashEmpty || e == key) break;
      h = (h + 1u) & g.mask;
    }
    WG_ATOMIC(fetch_or, bit < 32u ? &v.tlo[h] : &v.thi[h], 1u << (bit & 31u));
    v.cell_of[i] = (h << 6) | bit;
  }
  wg_phase();
  HC_MARK(3);
  // ---- first dense cell id of every block: exclusive scan of the masks' popcounts (whole tiles of 4 slots per thread)
  uint32_t C = 0;
  {
    const uint4* lo4 = reinterpret_cast<const uint4*>(v.tlo);
    const uint4* hi4 = reinterpret_cast<const uint4*>(v.thi);
    uint4* base4 = reinterpret_cast<uint4*>(v.tbase);
    for (uint32_t t0 = 0; t0 < size / 4u; t0 += kT) {
      const uint32_t q = t0 + tid;
      uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
      if (q < size / 4u) {
        const uint4 a = lo4[q], b = hi4[q];
        n0 = (uint32_t)(__popc(a.x) + __popc(b.x));
        n1 = (uint32_t)(__popc(a.y) + __popc(b.y));
        n2 = (uint32_t)(__popc(a.z) + __popc(b.z));
        n3 = (uint32_t)(__popc(a.w) + __popc(b.w));
      }
      uint32_t tot;
      const uint32_t run = C + block_excl_scan(n0 + n1 + n2 + n3, sc, tot);
      if (q < size / 4u) base4[q] = make_uint4(run, run + n0, run + n0 + n1, run + n0 + n1 + n2);
      C += tot;
    }
  }
  wg_phase();
  HC_MARK(4);
  // ---- every point: dense cell id (block's base + rank of the bit), cell count, smallest member index
  for (uint32_t i = tid; i < M; i += kT) {
    const uint32_t hb = v.cell_of[i], h = hb >> 6, bit = hb & 63u;
    const uint32_t cell = v.tbase[h] + hb_rank(v.tlo[h], v.thi[h], bit);
    v.cell_of[i] = cell;
    WG_ATOMIC(fetch_add, &v.ccnt[cell], 1u);
    WG_ATOMIC(fetch_min, &v.cmin[cell], i);
    v.ckey[cell] = hb;   // (every point of the cell writes the same word)
  }
  wg_phase();
  HC_MARK(5);
  // ---- first sorted position of every cell: exclusive scan of the counts (tiles of one cell per thread: coalesced)
  {
    uint32_t run0 = 0;
    for (uint32_t t0 = 0; t0 < C; t0 += kT) {
      const uint32_t k = t0 + tid;
      const uint32_t n = k < C ? v.ccnt[k] : 0u;
      uint32_t tot;
      const uint32_t run = run0 + block_excl_scan(n, sc, tot);
      if (k < C) v.cstart[k] = run;
      run0 += tot;
    }
    if (tid == 0) v.cstart[C] = M;
  }
  wg_phase();
  // ---- the points sorted by cell (leaves the counts all zero)
  for (uint32_t i = tid; i < M; i += kT) {
    const uint32_t cell = v.cell_of[i];
    v.sorted[v.cstart[cell] + (WG_ATOMIC(fetch_sub, &v.ccnt[cell], 1u) - 1u)] = v.P[i];
  }
  wg_phase();
  HC_MARK(6);

  // ---- edges: per cell, the occupied neighbour cells with a larger id; same root -> nothing; else the pair's points are tested
  // until the first hit (the reference's float arithmetic) and the cells united
  auto axis_mask = [](int a, int o) -> uint32_t {   // cells x of block (own + o) with |x + 4 o - a| <= 2, as a 4-bit mask
    const int l = max(a - 2 - 4 * o, 0), u = min(a + 2 - 4 * o, 3);
    return l > u ? 0u : ((1u << (u + 1)) - 1u) & ~((1u << l) - 1u);
  };
#ifdef ILCC_K2_TIMING
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#else
#define K2_T(k) do {} while (0)
#endif
  for (uint32_t A = tid; A < C; A += kT) {
    K2_T(5);
    const uint32_t hb = v.ckey[A], hA = hb >> 6, bitA = hb & 63u;
    const uint32_t keyA = v.tkey[hA];
    const int ax = (int)(bitA & 3u), ay = (int)((bitA >> 2) & 3u), az = (int)(bitA >> 4);
    const uint32_t a0 = v.cstart[A], a1 = v.cstart[A + 1];
    // the 5-cell window of an axis overlaps the cell's own block and ONE neighbour: the lower one when the cell sits in the
    // block's lower half, the upper one otherwise
    const int sx = ax < 2 ? -1 : 1, sy = ay < 2 ? -1 : 1, sz = az < 2 ? -1 : 1;
    // slots first, then ALL first probes in flight at once, then all masks: no chain of dependent loads per neighbour
    uint32_t hs[8], want[8], got[8];
    hs[0] = hA;
#pragma unroll
    for (int nb = 1; nb < 8; ++nb) {
      const int ox = (nb & 1) ? sx : 0, oy = (nb & 2) ? sy : 0, oz = (nb & 4) ? sz : 0;
      want[nb] = (uint32_t)((int)keyA + ox + g.nbx * (oy + g.nby * oz));
      hs[nb] = hb_slot(g, want[nb]);
    }
#pragma unroll
    for (int nb = 1; nb < 8; ++nb) got[nb] = v.tkey[hs[nb]];
#pragma unroll
    for (int nb = 1; nb < 8; ++nb) {
      uint32_t h = hs[nb], e = got[nb];
      while (e != want[nb] && e != kHashEmpty) {   // (a collision: rare at <= 50 % load)
        h = (h + 1u) & g.mask;
        e = v.tkey[h];
      }
      hs[nb] = e == kHashEmpty ? kHashEmpty : h;
    }
    K2_T(0);
    uint32_t mlo[8], mhi[8], base[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const uint32_t h = hs[nb] == kHashEmpty ? hA : hs[nb];   // (a harmless load for an absent block)
      mlo[nb] = v.tlo[h];
      mhi[nb] = v.thi[h];
      base[nb] = v.tbase[h];
    }
    uint32_t ra = wg_find(v.cpar, A);
    K2_T(1);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      if (hs[nb] == kHashEmpty) continue;
      const int ox = (nb & 1) ? sx : 0, oy = (nb & 2) ? sy : 0, oz = (nb & 4) ? sz : 0;
      const uint32_t mx = axis_mask(ax, ox), my = axis_mask(ay, oy), mz = axis_mask(az, oz);
      // window = { x + 4 y + 16 z : x in mx, y in my, z in mz }: products of disjoint bit patterns
      const uint32_t plane = mx * ((my & 1u) | ((my & 2u) << 3) | ((my & 4u) << 6) | ((my & 8u) << 9));   // 16 bits
      const unsigned long long win = (unsigned long long)plane *
                                     ((unsigned long long)(mz & 1u) | ((unsigned long long)(mz & 2u) << 15) | ((unsigned long long)(mz & 4u) << 30) | ((unsigned long long)(mz & 8u) << 45));
      unsigned long long occ = (((unsigned long long)mhi[nb] << 32) | mlo[nb]) & win;
      if (nb == 0) occ &= ~(1ull << bitA);
      while (occ) {
        const uint32_t b = (uint32_t)__builtin_ctzll(occ);
        occ &= occ - 1ull;
        const uint32_t B = base[nb] + hb_rank(mlo[nb], mhi[nb], b);
        if (B <= A) continue;   // every pair of cells once: from the smaller id
        K2_T(2);
        const uint32_t rb = wg_find(v.cpar, B);
        K2_T(3);
        if (ra == rb) continue;   // (a stale ra only costs a redundant test: wg_unite finds the roots itself)
        K2_COUNT(g_k2_pairs, 1);
        const uint32_t b0 = v.cstart[B], b1 = v.cstart[B + 1];
        // B's points four at a time (independent loads: one memory round trip per four), A's points inside (re-read for every
        // neighbour: L1 hits).  A serial walk over B's points for every point of A -- dense cells near the sensor hold dozens, and
        // two rings a cell apart never touch -- was most of this phase: one round trip per TEST, in the slowest lane of the wavefront
        bool hit = false;
        for (uint32_t ib = b0; ib < b1 && !hit; ib += 4u) {
          float4 q[4];
#pragma unroll
          for (uint32_t k = 0; k < 4u; ++k) q[k] = v.sorted[min(ib + k, b1 - 1u)];   // (the last point again past the end: harmless)
          for (uint32_t ia = a0; ia < a1 && !hit; ++ia) {
            const float4 pa = v.sorted[ia];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
              const float ex2 = q[k].x - pa.x, ey2 = q[k].y - pa.y, ez2 = q[k].z - pa.z;
              float d2 = ex2 * ex2;
              d2 = d2 + ey2 * ey2;
              d2 = d2 + ez2 * ez2;
              hit |= d2 < tol2;
            }
          }
        }
        K2_T(4);
        if (hit) {
          K2_COUNT(g_k2_hits, 1);
          wg_unite(v.cpar, ra, rb);
          ra = wg_find(v.cpar, A);
        }
        K2_T(3);
      }
    }
  }
#ifdef ILCC_K2_TIMING
  for (int k = 0; k < 6; ++k) atomicAdd(&g_k2_t[k], tacc[k]);
#endif
  wg_phase();
  HC_MARK(7);

  // ---- components: the root of every cell (kept in ckey: block-and-bit is no longer needed); size and smallest member index on the root
  for (uint32_t t0 = 0; t0 < C; t0 += kT) {
    const uint32_t A = t0 + tid;
    uint32_t root = 0;
    if (A < C) root = wg_find(v.cpar, A);
    wg_phase();   // (path halving rewrites parents: finish every find of the tile before the roots are stored)
    if (A < C) {
      v.ckey[A] = root;
      WG_ATOMIC(fetch_add, &v.ccnt[root], v.cstart[A + 1] - v.cstart[A]);   // the counts are zero since the placement
      if (root != A) WG_ATOMIC(fetch_min, &v.cmin[root], v.cmin[A]);        // (a non-root's word is final since the cells phase)
    }
  }
  wg_phase();
  HC_MARK(8);

  // ---- exact 1-NN of the click (float squared distance, ties -> lowest index)
  const float kx = c.clicks[3 * f], ky = c.clicks[3 * f + 1], kz = c.clicks[3 * f + 2];
  NnKey best{3.402823466e38f, 0xFFFFFFFFu};
  for (uint32_t i = tid; i < M; i += kT) {
    const float4 q = v.P[i];
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
    if (v.ckey[k] != k) continue;
    const uint32_t sz = v.ccnt[k], mi = v.cmin[k];
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
    const uint32_t nn_root = v.ckey[v.cell_of[nb.idx]];
    const uint32_t nsz = v.ccnt[nn_root];
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

  // ---- stable compaction of the chosen component (index order); the sorted points (same buffer) are dead from here on
  float4* __restrict__ dst = c.cluster + c.off[f];
  uint32_t running = 0;
  for (uint32_t base0 = 0; base0 < M; base0 += kT) {
    const uint32_t i = base0 + tid;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    bool keep = false;
    if (i < M) {
      q = v.P[i];
      keep = v.ckey[v.cell_of[i]] == chosen;
    }
    uint32_t tot;
    const uint32_t rank = block_rank(keep, sc + 128, tot);
    if (keep) dst[running + rank] = q;
    running += tot;
  }
  if (tid == 0) r->n_cluster = (int32_t)running;
  HC_MARK(9);
#ifdef ILCC_K2_TIMING
  if (tid == 0)
    printf("K2 counters (all frames so far): finds %llu hops %llu cell pairs tested %llu hits %llu; thread-cycles lookups %llu masks+findA %llu window %llu find/unite %llu tests %llu loop %llu\n", g_k2_finds, g_k2_hops, g_k2_pairs, g_k2_hits, g_k2_t[0], g_k2_t[1], g_k2_t[2], g_k2_t[3], g_k2_t[4], g_k2_t[5]);
  if (tid == 0)
    printf("K2 hashed M=%u C=%u table %u cycles: bbox %llu reset %llu insert %llu blocks %llu cells %llu sort %llu edges %llu comps %llu nn+compact %llu\n",
           M, C, size, hmark[1] - hmark[0], hmark[2] - hmark[1], hmark[3] - hmark[2], hmark[4] - hmark[3], hmark[5] - hmark[4], hmark[6] - hmark[5],
           hmark[7] - hmark[6], hmark[8] - hmark[7], hmark[9] - hmark[8]);
#endif
  return true;
}

// the point-level spatial hash of rounds 1-3 by ONE workgroup: the last resort for frames beyond the limits of the cell paths
__device__ void point_level_cluster_frame(const Ctx& c, uint32_t f, uint32_t* sc) {
  const uint32_t M = (uint32_t)c.res[f].n_roi, kT = blockDim.x;
  const uint64_t beg = c.off[f];
  uint32_t* gparent = c.uf_parent + beg;
  uint32_t* count = c.uf_count + beg;
  uint32_t* head = c.uf_hash_head + (uint64_t)f * kClusterHashSize;
  for (uint32_t i = threadIdx.x; i < M; i += kT) {
    gparent[i] = i;
    count[i] = 0u;
  }
  for (uint32_t k = threadIdx.x; k < (uint32_t)kClusterHashSize; k += kT) head[k] = 0xFFFFFFFFu;
  __threadfence_block();
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < M; i += kT) big_insert_point(c, f, i);
  __threadfence_block();
  __syncthreads();
  const float tol2 = (float)(c.p.cluster_tol * c.p.cluster_tol);
  for (uint32_t i = threadIdx.x; i < M; i += kT) big_search_point(c, f, i, tol2);
  __threadfence_block();
  __syncthreads();
  cluster_finish<false>(c, f, gparent, sc);
}

// ------------------------------------------------------------------ the same components, the whole chip on a few frames
// k2h_*: the second tier of the online caller (get_chessboard_by_point on un-cropped clouds, LidarCornersEst.cpp:72-115).  The
// first tier answers most frames from a window; the few it cannot vouch for -- listed in Ctx::list -- need the WHOLE cloud
// clustered, ~29 k points and ~20 k cells each.  One workgroup per frame (hashed_cluster_frame) is latency-bound there: every
// phase is a few dozen dependent memory round trips per thread, 2.1 ms per frame however few frames there are.  This chain of
// kernels spreads every phase of the same algorithm over (listed frame, 256-item chunk) work items on the whole chip; its
// atomics are device-scope (several workgroups share a frame) and therefore slow one by one -- 2.7 ms when all 128 clouds of a
// call are listed -- but with a tenth of the frames listed the chain is a tenth as long.
//   k2h_setup   per frame: bounding box -> block grid, table reset
//   k2h_insert  per point: its block's entry (claimed with a CAS on the key), its cell's bit
//   k2h_blocks  per frame: exclusive scan of the masks' popcounts -> first dense cell id of every block, number of cells
//   k2h_cells   per point: dense cell id (block's base + rank of the bit), cell count, smallest member index
//   k2h_starts  per frame: exclusive scan of the counts -> first sorted position of every cell
//   k2h_place   per point: the points sorted by cell (the frame's slice of the cluster buffer, until the compaction)
//   k2h_edges   per cell: occupied neighbour cells with a larger id; same root -> nothing; else the pair's points are tested
//               until the first hit (the reference's float arithmetic) and the cells united (hooks to the smaller id)
//   k2h_roots   per cell: its root; size and smallest member index accumulated on the root
//   k2h_finish  per frame: exact 1-NN of the click, the reference's choice rule, index-ordered compaction
// Same scratch as hashed_cluster_frame (hash_views).
constexpr int kListChunk = 256;       // items per work item
constexpr int kListBlock = 1024;      // listed frames whose chunk prefix sums a workgroup keeps in LDS at a time
struct ListedFrame {   // per frame (indexed by frame, valid for listed frames)
  HashFrame g;
  uint32_t n_cells;
  uint32_t valid;      // 0: outside the hashed path's limits: k2h_finish clusters the frame with the point-level search
  uint32_t pad[6];
};
static_assert(sizeof(ListedFrame) == 64, "ilcc_api.cpp allocates 64 bytes per frame");

// calls fn(f, first item of the chunk) for this workgroup's share of the (listed frame, chunk of kListChunk items) work items;
// count(f) = the frame's items (points, or cells).  Every thread of the workgroup makes the same calls.
template <typename Count, typename Fn>
__device__ __forceinline__ void for_each_listed_chunk(const Ctx& c, uint32_t* s_pref, Count count, Fn fn) {
  const uint32_t nl = __hip_atomic_load(c.list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t base = 0; base < nl; base += kListBlock) {
    const uint32_t nb = min((uint32_t)kListBlock, nl - base);
    __syncthreads();
    // chunks per listed frame, read in parallel, then an inclusive scan by the first wavefront: s_pref[j] = chunks of the frames before j
    for (uint32_t j = threadIdx.x; j < nb; j += blockDim.x) s_pref[j + 1] = (count(c.list[base + j]) + kListChunk - 1) / kListChunk;
    if (threadIdx.x == 0) s_pref[0] = 0u;
    __syncthreads();
    if (wave_id() == 0) {
      const uint32_t per = (nb + ILCC_WAVE - 1u) / ILCC_WAVE, k0 = min(nb + 1u, 1u + (uint32_t)lane_id() * per), k1 = min(nb + 1u, k0 + per);
      uint32_t sum = 0;
      for (uint32_t k = k0; k < k1; ++k) sum += s_pref[k];
      uint32_t incl = sum;
#pragma unroll
      for (int o = 1; o < ILCC_WAVE; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, ILCC_WAVE);
        if (lane_id() >= o) incl += t;
      }
      uint32_t run = incl - sum;
      for (uint32_t k = k0; k < k1; ++k) {
        run += s_pref[k];
        s_pref[k] = run;
      }
    }
    __syncthreads();
    const uint32_t total = s_pref[nb];
    for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) {
      uint32_t lo = 0, hi = nb;   // last j with s_pref[j] <= item
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_pref[mid] <= item) lo = mid; else hi = mid;
      }
      fn(c.list[base + lo], (item - s_pref[lo]) * kListChunk);
    }
  }
}

__global__ __launch_bounds__(1024) void k2h_setup(Ctx c, ListedFrame* frames) {
  __shared__ float scf[128];
  const uint32_t nl = __hip_atomic_load(c.list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x >= nl) return;
  const uint32_t f = c.list[blockIdx.x];
  const HashViews v = hash_views(c, f);
  const uint32_t M = v.M, tid = threadIdx.x, kT = blockDim.x;
  float3 lo = make_float3(3.0e38f, 3.0e38f, 3.0e38f), hi = make_float3(-3.0e38f, -3.0e38f, -3.0e38f);
  for (uint32_t i = tid; i < M; i += kT) {
    const float4 q = v.P[i];
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
  if (lane_id() == 0) {
    scf[wave_id()] = lo.x; scf[16 + wave_id()] = lo.y; scf[32 + wave_id()] = lo.z;
    scf[64 + wave_id()] = hi.x; scf[80 + wave_id()] = hi.y; scf[96 + wave_id()] = hi.z;
  }
  __syncthreads();
  for (int w = 0; w < (int)(kT / ILCC_WAVE); ++w) {
    lo.x = fminf(lo.x, scf[w]); lo.y = fminf(lo.y, scf[16 + w]); lo.z = fminf(lo.z, scf[32 + w]);
    hi.x = fmaxf(hi.x, scf[64 + w]); hi.y = fmaxf(hi.y, scf[80 + w]); hi.z = fmaxf(hi.z, scf[96 + w]);
  }
  ListedFrame L{};
  HashFrame& g = L.g;
  g.lox = lo.x; g.loy = lo.y; g.loz = lo.z;
  g.inv = 1.0f / ((float)c.p.cluster_tol * kFineCellOverTol);
  const float ex = (hi.x - lo.x) * g.inv, ey = (hi.y - lo.y) * g.inv, ez = (hi.z - lo.z) * g.inv;
  bool ok = M > 0u && M <= kHashPointsMax && ex < 8192.f && ey < 8192.f && ez < 8192.f &&   // (a NaN extent fails the comparisons)
            c.off[f + 1] - c.off[f] >= (uint64_t)kHashMinFramePoints;   // (the 1024-word block arrays fit the frame's slices)
  if (ok) {
    const int nx = (int)floorf(ex) + 5, ny = (int)floorf(ey) + 5, nz = (int)floorf(ez) + 5;
    g.nbx = (nx + 3) >> 2;
    g.nby = (ny + 3) >> 2;
    ok = (unsigned long long)g.nbx * (unsigned long long)g.nby * (unsigned long long)((nz + 3) >> 2) < 0xFFFFFFFFull;   // 32-bit keys, 0xFFFFFFFF = empty
  }
  if (!ok) {
    if (tid == 0) frames[f] = L;   // valid = 0, n_cells = 0
    return;
  }
  uint32_t size = 1024u;
  while (size < 2u * M) size <<= 1;
  g.mask = size - 1u;
  g.shift = 32u - (uint32_t)__builtin_ctz(size);
  L.valid = 1u;
  if (tid == 0) frames[f] = L;
  for (uint32_t k = tid; k < size; k += kT) {
    v.tkey[k] = kHashEmpty;
    v.tlo[k] = 0u;
    v.thi[k] = 0u;
  }
  for (uint32_t i = tid; i < M; i += kT) {
    v.ccnt[i] = 0u;
    v.cmin[i] = 0xFFFFFFFFu;
    v.cpar[i] = i;
  }
}

__global__ __launch_bounds__(kListChunk) void k2h_insert(Ctx c, const ListedFrame* frames) {
  __shared__ uint32_t s_pref[kListBlock + 1];
  for_each_listed_chunk(c, s_pref, [&](uint32_t f) { return frames[f].valid ? (uint32_t)c.res[f].n_roi : 0u; }, [&](uint32_t f, uint32_t first) {
    const HashViews v = hash_views(c, f);
    const uint32_t i = first + threadIdx.x;
    if (i >= v.M) return;
    const HashFrame g = frames[f].g;
    uint32_t key, bit;
    hb_cell(g, v.P[i], key, bit);
    uint32_t h = hb_slot(g, key);
    for (;;) {
      uint32_t e = __hip_atomic_load(&v.tkey[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e == kHashEmpty) e = atomicCAS(&v.tkey[h], kHashEmpty, key);
      if (e == kHashEmpty || e == key) break;
      h = (h + 1u) & g.mask;
    }
    atomicOr(bit < 32u ? &v.tlo[h] : &v.thi[h], 1u << (bit & 31u));
    v.cell_of[i] = (h << 6) | bit;
  });
}

__global__ __launch_bounds__(1024) void k2h_blocks(Ctx c, ListedFrame* frames) {
  __shared__ uint32_t sc[32];
  const uint32_t nl = __hip_atomic_load(c.list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x >= nl) return;
  const uint32_t f = c.list[blockIdx.x];
  if (!frames[f].valid) return;
  const HashViews v = hash_views(c, f);
  const uint32_t size = frames[f].g.mask + 1u;   // a power of two >= 1024: whole tiles of 4 slots per thread, 16-byte accesses
  const uint4* lo4 = reinterpret_cast<const uint4*>(v.tlo);
  const uint4* hi4 = reinterpret_cast<const uint4*>(v.thi);
  uint4* base4 = reinterpret_cast<uint4*>(v.tbase);
  uint32_t C = 0;
  for (uint32_t t0 = 0; t0 < size / 4u; t0 += blockDim.x) {
    const uint32_t q = t0 + threadIdx.x;
    uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    if (q < size / 4u) {
      const uint4 a = lo4[q], b = hi4[q];
      n0 = (uint32_t)(__popc(a.x) + __popc(b.x));
      n1 = (uint32_t)(__popc(a.y) + __popc(b.y));
      n2 = (uint32_t)(__popc(a.z) + __popc(b.z));
      n3 = (uint32_t)(__popc(a.w) + __popc(b.w));
    }
    uint32_t tot;
    const uint32_t run = C + block_excl_scan(n0 + n1 + n2 + n3, sc, tot);
    if (q < size / 4u) base4[q] = make_uint4(run, run + n0, run + n0 + n1, run + n0 + n1 + n2);
    C += tot;
  }
  if (threadIdx.x == 0) frames[f].n_cells = C;
}

__global__ __launch_bounds__(kListChunk) void k2h_cells(Ctx c, const ListedFrame* frames) {
  __shared__ uint32_t s_pref[kListBlock + 1];
  for_each_listed_chunk(c, s_pref, [&](uint32_t f) { return frames[f].valid ? (uint32_t)c.res[f].n_roi : 0u; }, [&](uint32_t f, uint32_t first) {
    const HashViews v = hash_views(c, f);
    const uint32_t i = first + threadIdx.x;
    if (i >= v.M) return;
    const uint32_t hb = v.cell_of[i], h = hb >> 6, bit = hb & 63u;
    const uint32_t cell = v.tbase[h] + hb_rank(v.tlo[h], v.thi[h], bit);
    v.cell_of[i] = cell;
    atomicAdd(&v.ccnt[cell], 1u);
    atomicMin(&v.cmin[cell], i);
    v.ckey[cell] = hb;   // (every point of the cell writes the same word)
  });
}

__global__ __launch_bounds__(1024) void k2h_starts(Ctx c, const ListedFrame* frames) {
  __shared__ uint32_t sc[32];
  const uint32_t nl = __hip_atomic_load(c.list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x >= nl) return;
  const uint32_t f = c.list[blockIdx.x];
  if (!frames[f].valid) return;
  const HashViews v = hash_views(c, f);
  const uint32_t C = frames[f].n_cells;
  uint32_t run0 = 0;
  for (uint32_t t0 = 0; t0 < C; t0 += blockDim.x) {
    const uint32_t k = t0 + threadIdx.x;
    const uint32_t n = k < C ? v.ccnt[k] : 0u;
    uint32_t tot;
    const uint32_t run = run0 + block_excl_scan(n, sc, tot);
    if (k < C) v.cstart[k] = run;
    run0 += tot;
  }
  if (threadIdx.x == 0) v.cstart[C] = v.M;
}

__global__ __launch_bounds__(kListChunk) void k2h_place(Ctx c, const ListedFrame* frames) {
  __shared__ uint32_t s_pref[kListBlock + 1];
  for_each_listed_chunk(c, s_pref, [&](uint32_t f) { return frames[f].valid ? (uint32_t)c.res[f].n_roi : 0u; }, [&](uint32_t f, uint32_t first) {
    const HashViews v = hash_views(c, f);
    const uint32_t i = first + threadIdx.x;
    if (i >= v.M) return;
    const uint32_t cell = v.cell_of[i];
    v.sorted[v.cstart[cell] + (atomicSub(&v.ccnt[cell], 1u) - 1u)] = v.P[i];   // leaves the counts all zero
  });
}

__global__ __launch_bounds__(kListChunk) void k2h_edges(Ctx c, const ListedFrame* frames) {
  __shared__ uint32_t s_pref[kListBlock + 1];
  const float tol2 = (float)(c.p.cluster_tol * c.p.cluster_tol);
  for_each_listed_chunk(c, s_pref, [&](uint32_t f) { return frames[f].n_cells; }, [&](uint32_t f, uint32_t first) {
    const uint32_t A = first + threadIdx.x;
    if (A >= frames[f].n_cells) return;
    const HashFrame g = frames[f].g;
    const HashViews v = hash_views(c, f);
    const uint32_t hb = v.ckey[A], hA = hb >> 6, bitA = hb & 63u;
    const uint32_t keyA = v.tkey[hA];
    const int ax = (int)(bitA & 3u), ay = (int)((bitA >> 2) & 3u), az = (int)(bitA >> 4);
    const uint32_t a0 = v.cstart[A], a1 = v.cstart[A + 1];
    const int sx = ax < 2 ? -1 : 1, sy = ay < 2 ? -1 : 1, sz = az < 2 ? -1 : 1;
    auto axis_mask = [](int a, int o) -> uint32_t {   // cells x of block (own + o) with |x + 4 o - a| <= 2, as a 4-bit mask
      const int l = max(a - 2 - 4 * o, 0), u = min(a + 2 - 4 * o, 3);
      return l > u ? 0u : ((1u << (u + 1)) - 1u) & ~((1u << l) - 1u);
    };
    uint32_t hs[8], want[8], got[8];
    hs[0] = hA;
#pragma unroll
    for (int nb = 1; nb < 8; ++nb) {
      const int ox = (nb & 1) ? sx : 0, oy = (nb & 2) ? sy : 0, oz = (nb & 4) ? sz : 0;
      want[nb] = (uint32_t)((int)keyA + ox + g.nbx * (oy + g.nby * oz));
      hs[nb] = hb_slot(g, want[nb]);
    }
#pragma unroll
    for (int nb = 1; nb < 8; ++nb) got[nb] = v.tkey[hs[nb]];
#pragma unroll
    for (int nb = 1; nb < 8; ++nb) {
      uint32_t h = hs[nb], e = got[nb];
      while (e != want[nb] && e != kHashEmpty) {
        h = (h + 1u) & g.mask;
        e = v.tkey[h];
      }
      hs[nb] = e == kHashEmpty ? kHashEmpty : h;
    }
    uint32_t mlo[8], mhi[8], base[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const uint32_t h = hs[nb] == kHashEmpty ? hA : hs[nb];
      mlo[nb] = v.tlo[h];
      mhi[nb] = v.thi[h];
      base[nb] = v.tbase[h];
    }
    uint32_t ra = uf_find(v.cpar, A);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      if (hs[nb] == kHashEmpty) continue;
      const int ox = (nb & 1) ? sx : 0, oy = (nb & 2) ? sy : 0, oz = (nb & 4) ? sz : 0;
      const uint32_t mx = axis_mask(ax, ox), my = axis_mask(ay, oy), mz = axis_mask(az, oz);
      const uint32_t plane = mx * ((my & 1u) | ((my & 2u) << 3) | ((my & 4u) << 6) | ((my & 8u) << 9));
      const unsigned long long win = (unsigned long long)plane *
                                     ((unsigned long long)(mz & 1u) | ((unsigned long long)(mz & 2u) << 15) | ((unsigned long long)(mz & 4u) << 30) | ((unsigned long long)(mz & 8u) << 45));
      unsigned long long occ = (((unsigned long long)mhi[nb] << 32) | mlo[nb]) & win;
      if (nb == 0) occ &= ~(1ull << bitA);
      while (occ) {
        const uint32_t b = (uint32_t)__builtin_ctzll(occ);
        occ &= occ - 1ull;
        const uint32_t B = base[nb] + hb_rank(mlo[nb], mhi[nb], b);
        if (B <= A) continue;   // every pair of cells once: from the smaller id
        const uint32_t rb = uf_find(v.cpar, B);
        if (ra == rb) continue;
        const uint32_t b0 = v.cstart[B], b1 = v.cstart[B + 1];
        bool hit = false;
        for (uint32_t ib = b0; ib < b1 && !hit; ib += 4u) {
          float4 q[4];
#pragma unroll
          for (uint32_t k = 0; k < 4u; ++k) q[k] = v.sorted[min(ib + k, b1 - 1u)];
          for (uint32_t ia = a0; ia < a1 && !hit; ++ia) {
            const float4 pa = v.sorted[ia];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
              const float ex2 = q[k].x - pa.x, ey2 = q[k].y - pa.y, ez2 = q[k].z - pa.z;
              float d2 = ex2 * ex2;
              d2 = d2 + ey2 * ey2;
              d2 = d2 + ez2 * ez2;
              hit |= d2 < tol2;
            }
          }
        }
        if (hit) {
          uf_unite(v.cpar, ra, rb);
          ra = uf_find(v.cpar, A);
        }
      }
    }
  });
}

__global__ __launch_bounds__(kListChunk) void k2h_roots(Ctx c, const ListedFrame* frames) {
  __shared__ uint32_t s_pref[kListBlock + 1];
  for_each_listed_chunk(c, s_pref, [&](uint32_t f) { return frames[f].n_cells; }, [&](uint32_t f, uint32_t first) {
    const uint32_t A = first + threadIdx.x;
    if (A >= frames[f].n_cells) return;
    const HashViews v = hash_views(c, f);
    const uint32_t root = uf_find(v.cpar, A);
    v.ckey[A] = root;   // (block-and-bit is no longer needed: the word now holds the cell's component)
    atomicAdd(&v.ccnt[root], v.cstart[A + 1] - v.cstart[A]);   // the counts are zero since the placement
    // smallest member index of the component, accumulated in place: a root's word only ever falls towards its component's minimum
    atomicMin(&v.cmin[root], __hip_atomic_load(&v.cmin[A], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  });
}

__global__ __launch_bounds__(1024) void k2h_finish(Ctx c, const ListedFrame* frames) {
  __shared__ uint32_t sc[160];
  const uint32_t nl = __hip_atomic_load(c.list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x >= nl) return;
  const uint32_t f = c.list[blockIdx.x];
  if (!frames[f].valid) {   // beyond the hashed path's limits (> 65 536 points, a bounding grid of > 2^32 blocks): point-level search
    point_level_cluster_frame(c, f, sc);
    return;
  }
  ilcc_result* r = &c.res[f];
  const HashViews v = hash_views(c, f);
  const uint32_t M = v.M, C = frames[f].n_cells, tid = threadIdx.x, kT = blockDim.x;
  const int nwv = (int)(kT / ILCC_WAVE);
  float* scf = reinterpret_cast<float*>(sc);
  const float kx = c.clicks[3 * f], ky = c.clicks[3 * f + 1], kz = c.clicks[3 * f + 2];
  NnKey best{3.402823466e38f, 0xFFFFFFFFu};
  for (uint32_t i = tid; i < M; i += kT) {
    const float4 q = v.P[i];
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
  const uint32_t cmin_sz = (uint32_t)c.p.cluster_min, cmax_sz = (uint32_t)c.p.cluster_max;
  uint32_t bsz = 0, bidx = 0xFFFFFFFFu, broot = 0xFFFFFFFFu;
  for (uint32_t k = tid; k < C; k += kT) {
    if (v.ckey[k] != k) continue;
    const uint32_t sz = v.ccnt[k], mi = v.cmin[k];
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
    const uint32_t nn_root = v.ckey[v.cell_of[nb.idx]];
    const uint32_t nsz = v.ccnt[nn_root];
    const bool found = nsz >= cmin_sz && nsz <= cmax_sz;   // find_board of get_chessboard_by_point (:91-102)
    r->found_board = found ? 1 : 0;
    sc[120] = found ? nn_root : r0;                         // cluster containing the click's NN, else plane_index = 0
    sc[121] = (s0 == 0) ? 0u : 1u;
  }
  __syncthreads();
  const uint32_t chosen = sc[120];
  if (sc[121] == 0u) {
    if (tid == 0) r->status = ILCC_NO_CLUSTER;
    return;
  }
  float4* __restrict__ dst = c.cluster + c.off[f];
  uint32_t running = 0;
  for (uint32_t base0 = 0; base0 < M; base0 += kT) {
    const uint32_t i = base0 + tid;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    bool keep = false;
    if (i < M) {
      q = v.P[i];
      keep = v.ckey[v.cell_of[i]] == chosen;
    }
    uint32_t tot;
    const uint32_t rank = block_rank(keep, sc + 128, tot);
    if (keep) dst[running + rank] = q;
    running += tot;
  }
  if (tid == 0) r->n_cluster = (int32_t)running;
}

// second tier of the online caller: the flagged frames, listed (one thread per frame: a few dozen frames at most matter)
__global__ __launch_bounds__(256) void k2h_list(Ctx c) {
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f < c.n_frames && c.frame_flags[f] != 0u && c.res[f].status == ILCC_OK) c.list[atomicAdd(c.list_count, 1u)] = f;
}

uint32_t cluster_bits_default() { return (uint32_t)kFineBits; }
uint32_t cluster_bits_online() { return (uint32_t)kFineBitsOnline; }
size_t cluster_lds_bytes(uint32_t pts_cap, uint32_t cells_cap, uint32_t bits) {
  return 160 * sizeof(uint32_t) + sizeof(uint32_t) * fine_words(bits) + sizeof(uint16_t) * (bits / 64) +
         sizeof(uint32_t) * (5 * (size_t)cells_cap + 2) + 3 * sizeof(float) * (size_t)pts_cap;
}

__global__ __launch_bounds__(1024) void k2_seeded_cluster(Ctx c) {
  extern __shared__ __align__(16) unsigned char smem[];
  FineLds L;
  L.sc = reinterpret_cast<uint32_t*>(smem);                  // 160 words
  L.bm = L.sc + 160;                                        // 8-byte aligned: read as 64-bit words, too
  L.pre = reinterpret_cast<uint16_t*>(L.bm + fine_words(c.cluster_bits));
  const uint32_t cc = c.cluster_cells_cap;
  L.ckey = reinterpret_cast<uint32_t*>(L.pre + c.cluster_bits / 64);
  L.cstart = L.ckey + cc;
  L.ccnt = L.cstart + cc + 2;
  L.cmin = L.ccnt + cc;
  L.cpar = L.cmin + cc;
  L.sx = reinterpret_cast<float*>(L.cpar + cc);
  L.sy = L.sx + c.cluster_lds_points;
  L.sz = L.sy + c.cluster_lds_points;
  const uint32_t f = blockIdx.x;
  if (c.res[f].status != ILCC_OK) return;
  // first tier of the online caller: frames K1 has already handed on (empty window) are the second tier's
  if (c.online_tier == 1u && c.frame_flags[f] != 0u) return;
  const uint32_t M = (uint32_t)c.res[f].n_roi;
  const bool done = (M <= c.cluster_lds_points) ? fine_cluster_frame<true>(c, f, L) : fine_cluster_frame<false>(c, f, L);
  if (done) return;
  if (c.online_tier == 1u) {   // the window's points do not fit the LDS grid (capacity): the second tier clusters the whole cloud
    if (threadIdx.x == 0) c.frame_flags[f] = 1u;
    return;
  }
  // the LDS cell grid cannot hold this frame: the same components on cells with the cells in a hash table (global memory) ...
  __syncthreads();
  if (hashed_cluster_frame(c, f, L.sc)) return;
  // ... and beyond that path's limits (> 65 536 ROI points in one frame, a bounding grid of > 2^32 blocks): point-level search
  __syncthreads();
  point_level_cluster_frame(c, f, L.sc);
}

// up to 160 KiB of dynamic LDS (> the 64 KiB default cap).  Called by ilcc_create for the handle's device: the attribute
// is kept per (function, device).
hipError_t set_kernel_attributes_k2() {
#ifdef ILCC_K2_TIMING
  return hipFuncSetAttribute((const void*)k2_seeded_cluster, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);   // (the timers' static LDS)
#else
  return hipFuncSetAttribute((const void*)k2_seeded_cluster, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
}

void launch_cluster(const Ctx& c, hipStream_t s) {
  if (c.online_tier == 2u) {   // the frames the online caller's first tier could not vouch for: the k2h_* chain over the listed frames
    ListedFrame* lf = static_cast<ListedFrame*>(c.list_frames);
    const uint32_t grid = c.list_grid;
    (void)hipMemsetAsync(c.list_count, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(k2h_list, dim3((c.n_frames + 255u) / 256u), dim3(256), 0, s, c);
    hipLaunchKernelGGL(k2h_setup, dim3(c.n_frames), dim3(1024), 0, s, c, lf);
    hipLaunchKernelGGL(k2h_insert, dim3(grid), dim3(kListChunk), 0, s, c, lf);
    hipLaunchKernelGGL(k2h_blocks, dim3(c.n_frames), dim3(1024), 0, s, c, lf);
    hipLaunchKernelGGL(k2h_cells, dim3(grid), dim3(kListChunk), 0, s, c, lf);
    hipLaunchKernelGGL(k2h_starts, dim3(c.n_frames), dim3(1024), 0, s, c, lf);
    hipLaunchKernelGGL(k2h_place, dim3(grid), dim3(kListChunk), 0, s, c, lf);
    hipLaunchKernelGGL(k2h_edges, dim3(grid), dim3(kListChunk), 0, s, c, lf);
    hipLaunchKernelGGL(k2h_roots, dim3(grid), dim3(kListChunk), 0, s, c, lf);
    hipLaunchKernelGGL(k2h_finish, dim3(c.n_frames), dim3(1024), 0, s, c, lf);
    return;
  }
  const int threads = (c.n_frames <= (uint32_t)kSmallBatchFrames || c.wide) ? 1024 : kFrameThreads;   // (small batches: latency, not CU footprint, matters)
  hipLaunchKernelGGL(k2_seeded_cluster, dim3(c.n_frames), dim3(threads), cluster_lds_bytes(c.cluster_lds_points, c.cluster_cells_cap, c.cluster_bits), s, c);
}

}  // namespace ilcc
