// K6 grid_cost -- the hot kernel.  Evaluates the ILCC intensity-grid objective of
// Optimization::get_theta_t (/root/reference/ilcc2/src/Optimization.cpp:94-160), i.e.
//     cost(theta,ty,tz,phase) = 1/2 * sum_k HuberLoss(0.1)( r_k^2 ),
// r_k = VirtualboardError::operator() (/root/reference/ilcc2/include/ilcc2/Optimization.h:31-107)
// for EVERY candidate of an exhaustive (theta, ty, tz) x colour-phase grid (the reference only
// walks this surface locally with Ceres from (0,0,0)).
//
// (Round 4: k6_group_prepass, further down, runs ONE such pre-pass for five consecutive thetas in front of the full pass; a
// full-pass workgroup starts from its group's rejected-tile mask, or exits at once when the group left no tile.)
// Mapping (gfx950): workgroup = (frame, theta index), 4 wavefronts (8 on frames staged above 2048 points).  The frame's
// labelled points are rotated by the workgroup's theta and staged ONCE into LDS.  A wavefront owns a tile of
// 4 x 4 (ty, tz) candidates; lane = candidate * 4 + slice: the four lanes of a quad evaluate the SAME
// candidate on four interleaved quarters of the point walk and keep private running sums (both
// colour phases).  The branch-and-bound test therefore needs only a quad reduction -- four DPP adds --
// and runs every 8 positions of the walk (2 points per lane): a tile stops the moment each of its candidates is provably
// beaten.  The walk starts with the border-class points closest to the board's outline (rim), then the other border-class
// points, then the interior-class ones -- the layout k5w_walk_order (end of this file) writes once per frame.  Before any
// tile is started, the full pass runs a BOX PRE-PASS: a lower bound for all 16 candidates of a tile from the out-of-board
// cost of 32 rim points at the tile's extreme translations (box_term); 93-99 % of the tiles are never started.  The first
// block of the walk lives in registers (DESIGN.md section 4 has the measurements behind every one of these choices).
// History of this mapping, measured on the 128-frame batch:
//  * lanes = points, 4 x 4 candidates in registers (round-1 first design): 14.5 VALU per evaluation
//    thanks to separable i/j terms, but every test needed a ~130-instruction transposed reduction over
//    32 accumulators and could only run at 1/16, 1/8, 1/4, 1/2 of the points: >= 400 instructions per
//    tile however hopeless the tile (2.8 M wave-instructions per frame, 0.89 ms per batch);
//  * one candidate per lane, 8 x 8 tiles: no reduction at all, but a surviving tile is a serial walk over
//    all M points by ONE wavefront (~50 us): the launch tails dominated (1.05 ms);
//  * this one: quad-sliced -- a quarter of the serial length, 4 x 4 pruning granularity.
// No MFMA: there is no dense contraction.  The cost volume is never written (in-kernel argmin)
// unless the diagnostic entry asks for it.
//
// Per point and candidate, with i = (y' + ty + W g/2)/g, j likewise (Optimization.h:45-46):
//   in board (0<i<W, 0<j<H):  r = dist(i, nearest integer) + dist(j, nearest integer) when the
//                             cell colour differs from the point's label, else 0   (:50-83)
//   out of board:             r = min(|i|,|i-W|) + min(|j|,|j-H|) if useOutofBoard    (:85-104)
//   1/2 rho(r^2) = q (r - q/2),  q = min(r, delta)                     (HuberLoss(0.1), :137)
// The cell is white iff topleftWhite xor ((floor i + floor j) odd)  (:53-61), so a mismatch
// under phase 0 is a match under phase 1: both phases come out of one pass.
// A lane's sums carry cost / 2 (the colour weight is 0 or 1/2; powers of two, exact): 29 VALU
// instructions per point and lane for a border-class point (out-of-board logic included; 26 inside the unrolled walk), 15
// for an interior-class point (it is in the board under every translation of the grid: accumulate_interior) --
// tools/k6_isa_count.sh counts them in the assembly of the probe kernels at the end of this file.
#include "ilcc_internal.h"
#include <type_traits>
#ifdef ILCC_K6_TIMING
#include <algorithm>
#include <vector>
#endif

namespace ilcc {

// -DILCC_K6_TIMING: per-wavefront cycle budget of the FULL pass (s_memtime around the phases of a workgroup's life), summed
// into k6_prof[] and read back through ilcc_debug_k6_profile (tools/dev_k6_timing.py).  Costs ~10 % of the kernel's time.
#ifdef ILCC_K6_TIMING
constexpr int kProfWaves = 1 << 17, kProfWords = 16;
__device__ unsigned long long k6_prof[kProfWaves * kProfWords];   // one record per wavefront: plain stores, no atomics
#define K6_NOW() __builtin_readcyclecounter()
#else
#define K6_NOW() 0ull
#endif
// Tuned constants (the measurements behind each value are in DESIGN.md section 4, "K6 tuning record")
#ifndef ILCC_SEED_SHIFT
#define ILCC_SEED_SHIFT 4   // round 3: an eighth (1/2: 323 k, 1/4: 331 k, 1/8: 335 k, 1/16: 329 k frames/s); round 6, with the bound anchored on all points anyway: 1/4: 1 247 k, 1/8: 1 282 k (locate 0.187 ms alone), 1/16 = the 128-point floor on VLP-16 frames: 1 288 k (0.166 ms), 1/32: the same; config 5 unchanged (90-91 k)
#endif
constexpr int kSeedShift = ILCC_SEED_SHIFT;     // subsampled (locate) launches walk M >> kSeedShift positions, at least Ctx::walk_limit
constexpr int kTile = 4;          // 4 x 4 candidates per wavefront; lane = ((a << 2) | b) << 2 | slice
constexpr int kSlices = 4;        // lanes (one quad) sharing a candidate, each on every 4th point of the walk
constexpr int kUnroll = 2;        // points per lane and block of the generic walk
constexpr int kStep = kSlices * kUnroll;
constexpr int kBoxShiftSmall = 3;   // box pre-pass: at least 1/8 of the frame's labelled points per tile (and at least Ctx::box_points) ...
constexpr int kBoxShiftLarge = 2;   // ... 1/4 in the 512-thread instance (frames of several thousand labelled points)
// the common pre-pass (k6_group_prepass) looks at M >> kGroupShift points of the rim-first walk: a quarter in the 256-thread instance, half in
// the 512-thread one (twice what each theta's own pre-pass looks at)
#ifndef ILCC_GROUP_SHIFT_SMALL
#define ILCC_GROUP_SHIFT_SMALL 2   // round 6 (the only filter in front of the walk now), k frames/s: 1 (half): 1 284-1 295, 2: 1 285-1 288, 3: 1 199-1 222
#endif
#ifndef ILCC_GROUP_SHIFT_LARGE
#define ILCC_GROUP_SHIFT_LARGE 1   // config 5: 1: 90.4-90.8, 2: 87.1-88.1
#endif
constexpr int kGroupShiftSmall = ILCC_GROUP_SHIFT_SMALL, kGroupShiftLarge = ILCC_GROUP_SHIFT_LARGE;
static_assert(kGroupShiftSmall >= 1 && kGroupShiftLarge >= 1,
              "k6_group_prepass stages M >> kGroupShift <= M / 2 points: launch_group_prepass sizes its LDS for that");
constexpr int kBoxFirstRound = 32;                 // box pre-pass: points of the first round (an eighth of the sample, at least this many) when many tiles are alive ...
constexpr int kBoxFirstRoundFrom = 8;              // ... = from this many tiles per wavefront on (8 lanes or fewer per tile)
// box pre-pass: tile ids per compaction round = the length of the list of live tiles in LDS, a multiple of the workgroup size (the
// 256-thread instance: 512 B -- with 1 792 staged points a workgroup then needs 22.9 KB and SEVEN fit a CU's 160 KB)
template <int THREADS>
constexpr int kBoxSegment = THREADS <= 256 ? 256 : 1024;
constexpr int kBoxTilesMax = 4096;                 // box pre-pass: tiles per workgroup its LDS bit mask holds
constexpr float kBoxSafety = 1.f - 0x1p-12f;
constexpr int kBoundRefresh = 256;                  // points between reloads of the frame's shared bound

// sum over the 4 lanes of a quad (every lane gets the total): two DPP adds
__device__ __forceinline__ float quad_sum(float v) {
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1 /*quad_perm [1,0,3,2]*/, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E /*quad_perm [2,3,0,1]*/, 0xf, 0xf, false));
  return v;
}

// sum over aligned groups of P lanes (P a power of two, 2 ... 64; wave-uniform): every lane of a group gets the same bits (each
// step adds two values that both partners hold: a + b == b + a)
__device__ __forceinline__ float lanes_sum(float v, uint32_t P) {
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1 /*quad_perm [1,0,3,2]*/, 0xf, 0xf, false));
  if (P > 2u) v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E /*quad_perm [2,3,0,1]*/, 0xf, 0xf, false));
  if (P > 4u) v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141 /*row_half_mirror*/, 0xf, 0xf, false));
  if (P > 8u) v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140 /*row_mirror*/, 0xf, 0xf, false));
  if (P > 16u) v += __shfl_xor(v, 16);
  if (P > 32u) v += __shfl_xor(v, 32);
  return v;
}

struct Best {
  float cost;
  uint32_t d2;
  uint32_t flat;
};
__device__ __forceinline__ bool better(float c, uint32_t d2, uint32_t flat, const Best& b) {
  return c < b.cost || (c == b.cost && (d2 < b.d2 || (d2 == b.d2 && flat < b.flat)));
}

struct PointTerms {   // uniform across the wavefront
  float pi, pj, hw;   // rotated coordinates / g, and 0.5 * (label == white)
};

// one point under this lane's translation: adds cost / 2 to (A0, A1)
template <bool OOB>
__device__ __forceinline__ void accumulate(const PointTerms& p, float ay, float az, float Wh, float Hh, float delta,
                                           float& A0, float& A1) {
  const float i = p.pi + ay, j = p.pj + az;
  const float fi = floorf(i), fj = floorf(j);
  const float ai = (i - fi) - 0.5f, aj = (j - fj) - 0.5f;       // dist to the nearest integer = 0.5 - |a|
  const float Rin = 1.f - (fabsf(ai) + fabsf(aj));              // dist_i + dist_j
  const float mf = __builtin_amdgcn_fractf(fmaf(0.5f, fi + fj, p.hw));   // 0.5 iff (floor i + floor j + white) odd
  const float nmf = 0.5f - mf;
  const float ui = fabsf(i - Wh) - Wh, uj = fabsf(j - Hh) - Hh; // < 0 inside; |.| = min(|i|, |i-W|)
  const bool oob = fmaxf(ui, uj) >= 0.f;                         // not (0 < i < W and 0 < j < H)
  auto sel = [&](float if_oob, float otherwise) -> float { return oob ? if_oob : otherwise; };
  float R, w0, w1;
  if (OOB) {
    R = sel(fabsf(ui) + fabsf(uj), Rin);
    w0 = sel(0.5f, mf);
    w1 = sel(0.5f, nmf);
  } else {
    R = sel(0.f, Rin);
    w0 = mf;
    w1 = nmf;
  }
  const float Q = fminf(R, delta);
  const float T = Q * fmaf(-0.5f, Q, R);    // q (r - q/2) = 1/2 rho(r^2)
  A0 = fmaf(T, w0, A0);                     // += cost / 2 under topleftWhite = false
  A1 = fmaf(T, w1, A1);
}

// The same term for a point that is IN the board under every translation of this workgroup's tables (see the staging
// below): the out-of-board half of accumulate<> -- 10 of its 27.5 instructions -- is dead for it.  Same operations on
// the in-board side, so the value is bit-identical to what accumulate<> computes for such a point.
__device__ __forceinline__ void accumulate_interior(const PointTerms& p, float ay, float az, float delta, float& A0, float& A1) {
  const float i = p.pi + ay, j = p.pj + az;
  const float fi = floorf(i), fj = floorf(j);
  const float ai = (i - fi) - 0.5f, aj = (j - fj) - 0.5f;
  const float R = 1.f - (fabsf(ai) + fabsf(aj));
  const float mf = __builtin_amdgcn_fractf(fmaf(0.5f, fi + fj, p.hw));
  const float nmf = 0.5f - mf;
  const float Q = fminf(R, delta);
  const float T = Q * fmaf(-0.5f, Q, R);
  A0 = fmaf(T, mf, A0);
  A1 = fmaf(T, nmf, A1);
}


// Box pre-pass, one point against one tile: adds to lb a lower bound of the point's term (cost / 2, either colour phase) for
// EVERY translation in [alo, ahi] x [zlo, zhi] -- see the pre-pass in grid_cost_body for the argument.
// (pi_lo, pi_hi), (pj_lo, pj_hi): the point's rotated coordinates -- one value each (lo == hi) in a workgroup's own pre-pass, the
// extremes over the thetas of a group in k6_group_prepass's common pre-pass (fl(p + a) is monotone in p as in a).
__device__ __forceinline__ void box_term(float pi_lo, float pi_hi, float pj_lo, float pj_hi, float alo, float ahi, float zlo, float zhi,
                                         float Wh, float Hh, float delta, float& lb, bool count = true) {
  const float i_lo = pi_lo + alo, i_hi = pi_hi + ahi, j_lo = pj_lo + zlo, j_hi = pj_hi + zhi;
  const float ui_lo = fabsf(i_lo - Wh) - Wh, ui_hi = fabsf(i_hi - Wh) - Wh;
  const float uj_lo = fabsf(j_lo - Hh) - Hh, uj_hi = fabsf(j_hi - Hh) - Hh;
  const bool out_all = fmaxf(fminf(ui_lo, ui_hi), fminf(uj_lo, uj_hi)) >= 0.f;   // out of the board everywhere in the box
  // |u| closest to zero over the box: the end value nearer to zero, 0 when the ends differ in sign
  const float R = fabsf(__builtin_amdgcn_fmed3f(ui_lo, ui_hi, 0.f)) + fabsf(__builtin_amdgcn_fmed3f(uj_lo, uj_hi, 0.f));
  const float Q = fminf(R, delta);
  const float T = Q * fmaf(-0.5f, Q, R);
  lb = (out_all && count) ? fmaf(T, 0.5f, lb) : lb;   // (count = false: a lane past the end of the sample, evaluated branch-free)
}

// the seed pass's best candidate of frame f and the seed workgroup (= seed theta index) that found it; wave-uniform.
// A handful of records: every lane reads them all (uniform addresses, scalar-cache loads) -- no cross-lane reduction.
__device__ __forceinline__ Best seed_argmin(const Ctx& c, uint32_t f, uint32_t& k2, uint32_t& ab) {
  const GridPartial* sp = c.seed_partial + (uint64_t)f * c.seed_blocks;
  Best sb{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu};
  k2 = 0;
  ab = 0;
  for (uint32_t q = 0; q < c.seed_blocks; ++q) {
    const GridPartial g = sp[q];
    if (better(g.cost, g.d2, g.flat, sb)) {
      sb = Best{g.cost, g.d2, g.flat};
      k2 = q;       // seed launch: workgroup q = seed theta q
      ab = g.pad;   // (a << 16) | b of the record's candidate in ITS launch's tables
    }
  }
  return sb;
}

// Box pre-pass of one workgroup over the tiles whose bit in s_dead is clear (grid_cost_body's own pre-pass behind the common one's
// mask; k6_group_prepass over all tiles).  Round 5: the live tiles are COMPACTED into a list and the workgroup's lanes dealt out
// over them -- P lanes per tile (a power of two, 2 ... 64), each on every P-th point of the sample -- so that all lanes work on
// tiles that still need work; and when many tiles are alive a first round on the sample's first kBoxFirstRound points weeds out
// the tiles far from the minimum (most of them) before the survivors get the whole sample.  Segments of kBoxSegment<THREADS> tile ids keep
// the list small.  A tile's bound is a sum in an order that depends on P: kBoxSafety covers that (it is a lower bound in real
// arithmetic whatever the order; see box_term).  Sets the tile's bit in s_dead when its bound exceeds lim_box; *s_alive = 1 when
// a tile survives the whole sample.  cnt: two counters used in turn.  interval(u): the u-th point's (i_lo, i_hi, j_lo, j_hi).
// Returns the (point, tile) evaluations this wavefront really did (wave-uniform).  Ends with every thread past its last barrier
// -- the caller synchronises before it reads s_dead / *s_alive.
template <int THREADS, class Interval>
__device__ __forceinline__ uint32_t box_prepass_rounds(int n_tiles, int ntb, int a_org, int b_org, int n_ty, int n_tz, const float* s_ay,
                                                       const float* s_az, uint32_t* s_dead, uint16_t* s_live, uint32_t* cnt, uint32_t* s_alive,
                                                       uint32_t n_pre, float lim_box, float Wh, float Hh, float delta2, Interval interval) {
  constexpr int kWaves = THREADS / ILCC_WAVE;
  const int lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(wave_id());
  uint32_t wave_evals = 0, turn = 0;
  for (int seg0 = 0; seg0 < n_tiles; seg0 += kBoxSegment<THREADS>) {
    for (int round = 0; round < 2; ++round) {
      uint32_t* n_live = &cnt[turn & 1u];   // (two counters in turn: the next compaction's reset cannot overtake this one's readers)
      ++turn;
      if (threadIdx.x == 0) *n_live = 0u;
      __syncthreads();   // (also: s_dead initialised / the previous round's bits set; the previous list no longer read)
      for (int q = seg0 + (int)threadIdx.x; q < min(seg0 + kBoxSegment<THREADS>, n_tiles); q += THREADS) {   // (the segment is a multiple of THREADS: whole wavefronts)
        const bool live = !((s_dead[q >> 5] >> (q & 31)) & 1u);
        const unsigned long long m = __ballot(live);
        uint32_t base = 0;
        if (lane == 0 && m != 0ull) base = atomicAdd(n_live, (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (live) s_live[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)q;
      }
      __syncthreads();
      const uint32_t L = *n_live;   // live tiles of this segment (any order: a tile's sum does not depend on its place in the list)
      if (L == 0u) break;
      // tiles per wavefront: the smallest power of two that deals all L out in one go, at most 32 (P >= 2 lanes per tile)
      uint32_t tpw = 1u;
      while (tpw < 32u && tpw * (uint32_t)kWaves < L) tpw <<= 1;
      const uint32_t P = 64u / tpw, lgP = (uint32_t)__builtin_ctz(P);
      const uint32_t chk = min(8u, max(2u, 32u / P));   // points per lane between two looks at "is every tile of this wavefront beaten already"
      // many tiles alive: a first round on a prefix of the sample; few: the whole sample at once
      const uint32_t n_first = max((uint32_t)kBoxFirstRound, n_pre >> 3);
      const bool first = round == 0 && tpw >= (uint32_t)kBoxFirstRoundFrom && n_pre > 2u * n_first;
      const uint32_t n_use = first ? n_first : n_pre;
      const uint32_t slice = (uint32_t)lane & (P - 1u);
      for (uint32_t j0 = (uint32_t)wid * tpw; j0 < L; j0 += (uint32_t)kWaves * tpw) {
        const uint32_t j = j0 + ((uint32_t)lane >> lgP);
        const bool todo = j < L;
        const int q = (int)s_live[todo ? j : 0u];
        const int qa = q / ntb, qb = q - qa * ntb;
        float alo = __builtin_inff(), ahi = -__builtin_inff(), zlo = __builtin_inff(), zhi = -__builtin_inff();
#pragma unroll
        for (int d = 0; d < kTile; ++d) {
          const float va = s_ay[min(a_org + qa * kTile + d, n_ty - 1)], vz = s_az[min(b_org + qb * kTile + d, n_tz - 1)];
          alo = fminf(alo, va);
          ahi = fmaxf(ahi, va);
          zlo = fminf(zlo, vz);
          zhi = fmaxf(zhi, vz);
        }
        // (the sum only grows: a wavefront whose tiles are all beaten already stops looking at further points)
        float lb = 0.f, both = 0.f;
        const uint32_t wave_tiles = (uint32_t)__popcll(__ballot(todo && slice == 0u));
        for (uint32_t u0 = 0; u0 < n_use; u0 += P * chk) {
          wave_evals += wave_tiles * min(P * chk, n_use - u0);
          auto block = [&](auto n) {   // unrolled and branch-free, the LDS reads of four points issued together
            constexpr int N = decltype(n)::value, SUB = N >= 4 ? 4 : N;
#pragma unroll
            for (int d0 = 0; d0 < N; d0 += SUB) {
              float4 v[SUB];
              bool ok[SUB];
#pragma unroll
              for (int e = 0; e < SUB; ++e) {
                const uint32_t u = u0 + (uint32_t)(d0 + e) * P + slice;
                ok[e] = u < n_use && todo;
                v[e] = interval(min(u, n_use - 1u));
              }
#pragma unroll
              for (int e = 0; e < SUB; ++e) box_term(v[e].x, v[e].y, v[e].z, v[e].w, alo, ahi, zlo, zhi, Wh, Hh, delta2, lb, ok[e]);
            }
          };
          if (chk == 8u)
            block(std::integral_constant<int, 8>{});
          else if (chk == 4u)
            block(std::integral_constant<int, 4>{});
          else
            block(std::integral_constant<int, 2>{});
          both = lanes_sum(lb, P);   // the same bits in all P lanes of a tile
          if (__ballot(todo && !(both * kBoxSafety > lim_box)) == 0ull) break;
        }
        if (slice == 0u && todo) {
          if (both * kBoxSafety > lim_box)
            atomicOr(&s_dead[q >> 5], 1u << (q & 31));   // (a tile's bit is only ever set by its own lanes)
          else if (!first)
            *s_alive = 1u;
        }
      }
      if (!first) break;   // (uniform: first depends on L only)
    }
  }
  return wave_evals;
}


// OVERFLOW (round 6; LDS_POINTS only): the frame holds more labelled points than the workgroup's LDS staging (Ctx::grid_lds_points:
// the handle stops growing it where a second workgroup would no longer fit the CU).  The first grid_lds_points walk positions are
// staged as always -- the box pre-pass's sample and the blocks nearly every tile dies on are among them --, a position past them is
// read from the walk layout in L2 and rotated by the lane that needs it: only the few tiles that walk (almost) the whole frame get
// there.  Same points, same order, same sums.  (Before: such a frame took the LDS-free body, which has no box pre-pass: a handful of
// frames with 6 682 labelled points made config 5's full pass 5.6 ms instead of 0.74 ms per 128 frames.)
template <bool OOB, bool VOLUME, bool LDS_POINTS, bool PRUNE, int THREADS, bool OVERFLOW = false>
__device__ __forceinline__ void grid_cost_body(const Ctx& c, float* volume, float2* s_ij, float* s_hw, Best* s_best,
                                               uint32_t* s_iters, uint32_t* s_cnt, float* s_ay, float* s_az, uint32_t* s_dead, uint16_t* s_live,
                                               uint32_t* s_next, const uint32_t kblk) {
  // kblk: this workgroup's index within the frame (= blockIdx.x)
  const uint32_t f = blockIdx.y;
  uint32_t k = kblk;   // theta index (refinement pass: set from the seed below)
  [[maybe_unused]] const unsigned long long t_entry = K6_NOW();
  [[maybe_unused]] unsigned long long t_rej = 0, t_surv = 0, n_rej = 0, n_surv = 0, n_done = 0, p_surv = 0;
  [[maybe_unused]] unsigned long long n_tests = 0, alive_sum = 0, alive_le4 = 0, alive_le2 = 0;   // bound tests after the first one
  const ilcc_result* r = &c.res[f];
  const int lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(wave_id());
  constexpr int kBoxShift = (THREADS == kGridThreadsLarge && kGridThreadsLarge != kGridThreads) ? kBoxShiftLarge : kBoxShiftSmall;
  GridPartial* out = &c.partial[(uint64_t)f * c.grid_blocks + kblk];
  // full pass behind k6_group_prepass: tiles the pre-pass common to this theta's group has already rejected (a bit mask in
  // global memory), or nothing at all to do when it rejected them all
  const uint32_t* s_dead0 = nullptr;
  bool group_dead = false;
  if constexpr (PRUNE && OOB && LDS_POINTS && !VOLUME) {
    if (c.grp_alive != nullptr) {
      const uint32_t tr = (uint32_t)f * c.grp_count + kblk / (uint32_t)kThetaGroup;
      const uint32_t st = c.grp_alive[tr];   // 0: every tile dead, 1: mask valid, 2: no common pre-pass ran for this group
      group_dead = st == 0u;
      if (st == 1u) s_dead0 = c.grp_mask + (uint64_t)tr * c.grp_words;
    }
  }
  if (r->status != ILCC_OK || group_dead) {
    if (threadIdx.x == 0) {
      out->cost = __builtin_inff();
      out->d2 = 0xFFFFFFFFu;
      out->flat = 0xFFFFFFFFu;
      out->pad = 0u;
    }
    return;
  }
  // walk_limit: the seed and refinement launches only look at a PREFIX of the walk -- the golden-ratio order makes it a
  // uniform sample of the board -- to find WHERE the minimum is; their sums are not costs of complete candidates and go to
  // a bound word of their own.  The anchor launch then evaluates the found neighbourhood on every point: that is the bound.
  const uint32_t Mfull = c.n_lab[f];
  const uint32_t M = c.walk_limit ? min(Mfull, max(c.walk_limit, Mfull >> kSeedShift)) : Mfull;   // an eighth of the points (measured: 1/2: 323 k, 1/4: 331 k, 1/8: 335 k, 1/16: 329 k frames/s), at least walk_limit
  const uint64_t beg = c.off[f];
  const float2* __restrict__ gyz = c.yz + beg;
  const uint8_t* __restrict__ glab = c.lab + beg;
  const int n_ty = c.p.n_ty, n_tz = c.p.n_tz;
  int nta = (n_ty + kTile - 1) / kTile, ntb = (n_tz + kTile - 1) / kTile;
  int a_org = 0, b_org = 0;   // first candidate of tile (0, 0)
  int t0 = 0;
  if (PRUNE && c.seed_partial != nullptr) {
    uint32_t k2u, ab;
    const Best sb = seed_argmin(c, f, k2u, ab);
    if (sb.flat != 0xFFFFFFFFu) {
      const int a2 = __builtin_amdgcn_readfirstlane((int)(ab >> 16)), b2 = __builtin_amdgcn_readfirstlane((int)(ab & 0xFFFFu));
      const int k2 = __builtin_amdgcn_readfirstlane((int)k2u);
      const int sa = min(a2 * c.seed_stride_t, n_ty - 1), sbb = min(b2 * c.seed_stride_t, n_tz - 1);
      if (c.refine_window) {
        // refinement pass: every candidate within +-radius theta steps and the 8 x 8 (ty, tz) window
        // around the seed argmin -> the frame's bound is (nearly always) the true minimum before the
        // full pass starts, which is what lets the full pass cut almost every tile after 8 points
        // (it scores every kRefineThetaStride-th theta of its range: refine_step_th; the anchor rounds -- k6_anchor, or the tail of k6_locate -- cover the ones in between)
        const int kc = c.seed_off_th + k2 * c.seed_stride_th + (int)kblk * c.refine_step_th - c.refine_radius_th;
        k = (uint32_t)__builtin_amdgcn_readfirstlane(min(max(kc, 0), c.p.n_th - 1));
        // refine_window = tiles per axis of the window: 2 (8 x 8 translations around the seed's argmin) or, on grids whose seed stride
        // is wider than that window (config 5: every 16th translation), 4 -- the seed's best can be half a stride from the minimum
        const int wt = max(2, c.refine_window);
        a_org = __builtin_amdgcn_readfirstlane(min(max(sa - (wt * kTile) / 2, 0), max(n_ty - wt * kTile, 0)));
        b_org = __builtin_amdgcn_readfirstlane(min(max(sbb - (wt * kTile) / 2, 0), max(n_tz - wt * kTile, 0)));
        nta = min(wt, nta);
        ntb = min(wt, ntb);
      } else {
        // full pass: start at the tile that holds the seed's best translation; the order never changes the result
        t0 = __builtin_amdgcn_readfirstlane((sa / kTile) * ntb + (sbb / kTile));
      }
    }
  }
  const float cth = c.cth[k], sth = c.sth[k];

  // Point order: k5w_walk_order (below) has written the frame's labelled points in WALK layout, once per frame instead of once
  // per workgroup: [interior class | rim | other border-class points], each part in golden-ratio order (slot s of the walk
  // holds point (s * S) mod M with S ~ 0.618 M coprime to M, so that every prefix of a part is a sample spread over the whole
  // board: the bound test cuts tiles sooner than ring order would).  INTERIOR: in the board under EVERY rotation and
  // translation of the tables -> accumulate_interior.  Staging is then a rotation of M points by this workgroup's theta.
  // A subsampled launch (walk_limit) stages a prefix of each part, in proportion.
  const uint32_t S = Mfull ? c.walk_stride[f] : 1u;   // (the walk through global memory of frames too large for LDS)
  uint32_t Mi = 0;
  uint32_t stage_lo = 0, stage_hi = 0;   // walk positions staged so far: [stage_lo, stage_hi)
  const float2* __restrict__ wyz = c.walk_yz + beg;
  const uint8_t* __restrict__ wlab = c.walk_lab + beg;
  const uint32_t Mi_all = LDS_POINTS ? c.walk_mi[f] : 0u, n_rim_all = LDS_POINTS ? c.walk_nrim[f] : 0u;
  uint32_t n_in = Mi_all, n_rm = n_rim_all;
  if (LDS_POINTS && M < Mfull) {
    n_in = (uint32_t)(((uint64_t)Mi_all * M) / Mfull);
    n_rm = (uint32_t)(((uint64_t)n_rim_all * M) / Mfull);
  }
  const uint32_t lds_n = OVERFLOW ? min(M, c.grid_lds_points) : M;   // walk positions [0, lds_n) live in LDS
  auto walk_source = [&](uint32_t sl) -> uint32_t {   // walk position -> index in the walk layout
    const uint32_t src = sl < n_in ? sl : sl < n_in + n_rm ? Mi_all + (sl - n_in) : Mi_all + n_rim_all + (sl - n_in - n_rm);
    return min(src, Mfull - 1u);
  };
  auto stage_point = [&](uint32_t sl) {
    const uint32_t src = walk_source(sl);
    const float2 v = wyz[src];
    // Rx(theta) on (0,y,z), already divided by g (Optimization.h:37-46)
    s_ij[sl] = make_float2(fmaf(-sth, v.y, cth * v.x), fmaf(cth, v.y, sth * v.x));
    s_hw[sl] = wlab[src] ? 0.5f : 0.f;
  };
  if (LDS_POINTS) {
    // (parts are prefixes: n_in <= Mi_all, n_rm <= n_rim_all, and the rest M - n_in - n_rm <= the other border points + 2:
    // floor() twice; the source index below is clamped for that)
    Mi = n_in;
    // The full pass stages the points its box pre-pass looks at first: a workgroup the pre-pass leaves no tile (half of
    // them: every theta more than a few steps from the minimum) never stages, or reads, the rest.
    const bool box_first = PRUNE && OOB && c.box_points != 0u && nta * ntb <= kBoxTilesMax && M > Mi;   // (= use_box below)
    stage_lo = box_first ? Mi : 0u;
    stage_hi = box_first ? Mi + min(max(c.box_points, Mfull >> kBoxShift), M - Mi) : lds_n;   // (OVERFLOW: the caller made sure the sample fits)
    for (uint32_t sl = stage_lo + threadIdx.x; sl < stage_hi; sl += THREADS) stage_point(sl);
  }
  // (ty, tz) tables in LDS: a cut-short tile lasts about as long as one L2 round trip, so its
  // prologue must not wait for global loads
  for (int i = threadIdx.x; i < c.p.n_ty; i += THREADS) s_ay[i] = c.ay[i];
  for (int i = threadIdx.x; i < c.p.n_tz; i += THREADS) s_az[i] = c.az[i];
  if (threadIdx.x == 0) *s_next = 0u;   // the workgroup's tile-chunk counter (see the tile loop)
  __syncthreads();
  [[maybe_unused]] const unsigned long long t_staged = K6_NOW();

  const float Wh = 0.5f * (float)c.p.board_w, Hh = 0.5f * (float)c.p.board_h;
  const float delta2 = (float)c.p.huber_delta;   // (the terms carry plain r and q: a lane's sums are cost / 2)
  const int n_tiles = nta * ntb;
  const uint32_t dk = (uint32_t)((int)k - c.c_th) * (uint32_t)((int)k - c.c_th);

  Best best{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu};
  uint32_t* bound = c.grid_bound + f;
  float shared_bound = __builtin_inff();   // what this wavefront last published
  uint32_t gb_bits = PRUNE ? __hip_atomic_load(bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7f800000u;
  uint32_t pts_done = 0, pts_in = 0;   // walk positions executed by this wavefront (all / interior class)

  // BOX PRE-PASS (full pass of the pipeline): a lower bound for all 16 candidates of a tile at once, at the price of ONE
  // evaluation per point.  A border-class point that is out of the board under EVERY translation of the tile's 4 x 4 box costs
  // each of them at least T(r_box), r_box = (smallest |u_i| over the box) + (smallest |u_j| over the box): u = |x - W/2| - W/2
  // is evaluated with accumulate<>'s own fp32 expressions at the two extreme translations of each axis, every operation in
  // it is monotone (floating-point rounding included) and a box is narrower than one square (the host checks), so the
  // extremes bound everything in between; T is non-decreasing in r, and both colour phases pay an out-of-board point.  Points
  // that are in the board somewhere in the box count 0.  lane = (tile, half of the points): 2 x box_points evaluations per
  // tile instead of the 16 x 8 of a first block -- and the rim points, which the walk puts first, are exactly the ones that
  // leave the board when the translation is wrong.  A tile whose bound already exceeds the frame's bound is never started.
  // kBoxSafety: the bound is a sum in another order than the candidates' own fp32 sums (<= 2^12 terms per lane).
  bool use_box = false;
  unsigned long long box_evals = 0;   // (point, tile) evaluations of this workgroup's pre-pass
  if constexpr (PRUNE && LDS_POINTS && OOB) {
    use_box = c.box_points != 0u && n_tiles <= kBoxTilesMax && M > Mi;
    if (use_box) {
      for (int w = threadIdx.x; w < (n_tiles + 31) / 32; w += THREADS) s_dead[w] = s_dead0 ? s_dead0[w] : 0u;
      if (threadIdx.x == 0) s_cnt[0] = 0u;   // "a tile of this workgroup is still alive" (s_cnt is free until the epilogue)
      const uint32_t n_pre = min(max(c.box_points, Mfull >> kBoxShift), M - Mi);   // the frame's bound grows with its point count: so must the sample that has to exceed it
      const float lim_box = 0.5f * (1.f + kTieEps) * __uint_as_float(gb_bits);
      // (tiles the group's common pre-pass has rejected are not looked at again)
      // Round 6: behind a VALID common mask (k6_group_prepass, state 1) the workgroup walks what the mask leaves, without a pre-pass of
      // its own -- the compaction rounds, three barriers and ~60 evaluations per live tile bought little once the common pre-pass
      // looked at twice the sample: full pass alone 0.323 -> 0.308 ms per 1024 frames (executed evaluations 251 -> 261 M), bench
      // 1 247 -> 1 274 k frames/s, config 5 87.3 -> 89.2 k; with groups of 7 thetas instead of 5 (the common pre-pass is then the only
      // filter and costs less per theta): 1 280 k / 92 k (groups of 3 / 4 / 5 / 7 / 9 / 11 / 15: 1 235 / 1 250 / 1 276 / 1 280 / 1 283 /
      // 1 271 / 1 177 k and 82.9 / 83.9 / 89.7 / 91.5 / 92.6 / 92.2 / 90.8 k).  Grids without a common pre-pass (state 2) keep their own.
#ifndef ILCC_K6_SKIP_OWN_PREPASS
#define ILCC_K6_SKIP_OWN_PREPASS 1
#endif
      uint32_t wave_evals = 0;
      if (ILCC_K6_SKIP_OWN_PREPASS && s_dead0 != nullptr) {
        if (threadIdx.x == 0) s_cnt[0] = 1u;   // (the group's state word said: some tile is alive)
      } else {
        wave_evals = box_prepass_rounds<THREADS>(n_tiles, ntb, a_org, b_org, n_ty, n_tz, s_ay, s_az, s_dead, s_live, s_cnt + 1, s_cnt, n_pre,
                                                 lim_box, Wh, Hh, delta2, [&](uint32_t u) {
                                                   const float2 v = s_ij[Mi + u];
                                                   return make_float4(v.x, v.x, v.y, v.y);
                                                 });
      }
      if (lane == 0) s_iters[wid] = wave_evals;   // (s_iters is free until the epilogue)
      __syncthreads();
      for (int w = 0; w < THREADS / ILCC_WAVE; ++w) box_evals += s_iters[w];
      const bool any_alive = s_cnt[0] != 0u;
      __syncthreads();
      if (!any_alive) {   // nothing to walk at this theta: the rest of the frame's points is never staged
        if (threadIdx.x == 0) {
          out->cost = __builtin_inff();
          out->d2 = 0xFFFFFFFFu;
          out->flat = 0xFFFFFFFFu;
          out->pad = 0u;
          atomicAdd(c.grid_iters + 2 * kIterSlots + (f & (kIterSlots - 1)), box_evals);
        }
        return;
      }
      for (uint32_t sl = threadIdx.x; sl < stage_lo; sl += THREADS) stage_point(sl);
      for (uint32_t sl = stage_hi + threadIdx.x; sl < lds_n; sl += THREADS) stage_point(sl);
      __syncthreads();
    }
  }
  float* vol = VOLUME ? volume + (uint64_t)f * (uint64_t)c.p.n_th * n_ty * n_tz * 2u : nullptr;
  const int my_s = lane & (kSlices - 1), my_c = lane >> 2;
  const int my_a = my_c >> 2, my_b = my_c & 3;

  // first block of each class in registers (see run_tile); needs a full block of both classes
  constexpr int kFirstIn = 0, kFirstBd = 2;   // interior- / border-class (rim-first) points per lane in the first, register-resident block
  const bool first_block = LDS_POINTS && Mi >= (uint32_t)(kFirstIn * kSlices) && M - Mi >= (uint32_t)(kFirstBd * kSlices);
  PointTerms first_in[kFirstIn > 0 ? kFirstIn : 1], first_bd[kFirstBd > 0 ? kFirstBd : 1];
  if (LDS_POINTS && first_block) {
#pragma unroll
    for (int u = 0; u < kFirstIn; ++u) {
      const uint32_t ai = (uint32_t)(u * kSlices + my_s);
      const float2 vi = s_ij[ai];
      first_in[u] = PointTerms{vi.x, vi.y, s_hw[ai]};
    }
#pragma unroll
    for (int u = 0; u < kFirstBd; ++u) {
      const uint32_t ab = Mi + (uint32_t)(u * kSlices + my_s);
      const float2 vb = s_ij[ab];
      first_bd[u] = PointTerms{vb.x, vb.y, s_hw[ab]};
    }
  }

  const bool whole_tiles = (n_ty % kTile) == 0 && (n_tz % kTile) == 0;   // (wave-uniform: a scalar branch per tile)
  // one 4 x 4 tile, quad-sliced (lane = candidate * 4 + slice), from walk positions (pin0, pbd0) on, sums starting at
  // (a0_init, a1_init)
  auto run_tile = [&](int tile_a, int tile_b, float a0_init, float a1_init, uint32_t pin0, uint32_t pbd0) {
    const int ia = a_org + tile_a * kTile + my_a, ib = b_org + tile_b * kTile + my_b;
    // grids whose axes are multiples of the tile (the default 40 x 40) have no partial tiles: no clamps, no owner test
    bool owner = true;
    unsigned long long owner_mask = ~0ull;   // an SGPR pair: the bound test is then ballot & mask, no VALU select
    float ay, az;
    if (whole_tiles) {
      ay = s_ay[ia];
      az = s_az[ib];
    } else {
      owner = ia < n_ty && ib < n_tz;
      owner_mask = __ballot(owner);
      ay = s_ay[min(ia, n_ty - 1)];
      az = s_az[min(ib, n_tz - 1)];
    }

    // Branch and bound (PRUNE): costs are sums of non-negative terms, so a candidate whose partial
    // sum already exceeds the best COMPLETE cost known for this frame cannot be the argmin.
    // Exact: only provably losing candidates are cut short.
    float A0 = a0_init, A1 = a1_init;
    bool pruned = false;
    // the shared bound is fetched ahead of its use (an L2 round trip is longer than a cut-short tile, and a
    // slightly stale bound only delays a cut): this tile starts with the word loaded during the previous
    // one and issues the load for the next refresh right away
    // sums are cost / 2.  The test keeps everything within kTieEps of the bound alive: fp32 sums cannot order
    // such candidates reliably, K7r re-orders them on exact fixed-point sums
    float lim2 = 0.5f * (1.f + kTieEps) * fminf(__uint_as_float(gb_bits), best.cost);
    if (PRUNE) gb_bits = __hip_atomic_load(bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // lane's points: walk positions my_s, my_s + 4, ...  (LDS_POINTS = false: point index (pos * S) mod M)
    // (M == 0: no point is ever fetched, every candidate costs 0; keep the modulo defined)
    uint32_t idx = Mfull ? (uint32_t)(((uint64_t)my_s * S) % Mfull) : 0u;
    const uint32_t idx_step = Mfull ? (uint32_t)(((uint64_t)kSlices * S) % Mfull) : 0u;
    uint32_t pos = 0;
    auto fetch = [&](uint32_t at) -> PointTerms {   // at = walk position of THIS lane's point
      if (LDS_POINTS) {
        if (OVERFLOW && at >= lds_n) {   // past the staged prefix: from the walk layout in L2, rotated here (stage_point's expressions)
          const uint32_t src = walk_source(at);
          const float2 v = wyz[src];
          return PointTerms{fmaf(-sth, v.y, cth * v.x), fmaf(cth, v.y, sth * v.x), wlab[src] ? 0.5f : 0.f};
        }
        const float2 v = s_ij[at];
        return PointTerms{v.x, v.y, s_hw[at]};
      } else {
        const float2 v = gyz[idx];
        const PointTerms t{fmaf(-sth, v.y, cth * v.x), fmaf(cth, v.y, sth * v.x), glab[idx] ? 0.5f : 0.f};
        idx += idx_step;
        if (idx >= Mfull) idx -= Mfull;
        return t;
      }
    };
    if constexpr (LDS_POINTS) {
      // Two interleaved walks: 8 interior points (cheap term), 8 border points (full term), test, ... --
      // interleaved so that every prefix still samples both the pattern (interior) and the outline (border) of the board
      uint32_t pin = pin0, pbd = pbd0;       // next walk position of each class
      uint32_t since_refresh = 0;
      auto beaten = [&]() -> bool {          // every candidate of the tile provably loses
        const float part = fminf(quad_sum(A0), quad_sum(A1));
#ifdef ILCC_K6_TIMING
        {
          const uint32_t alive = (uint32_t)__popcll(__ballot(!(part > lim2)) & owner_mask) >> 2;
          if (pin + (pbd - Mi) > (uint32_t)((kFirstIn + kFirstBd) * kSlices) && alive) {
            ++n_tests;
            alive_sum += alive;
            alive_le4 += alive <= 4u;
            alive_le2 += alive <= 2u;
          }
        }
#endif
        return (__ballot(!(part > lim2)) & owner_mask) == 0ull;
      };
      auto refresh = [&]() {
        since_refresh += kStep;
        if (since_refresh >= (uint32_t)kBoundRefresh) {
          since_refresh = 0;
          lim2 = 0.5f * (1.f + kTieEps) * fminf(__uint_as_float(gb_bits), best.cost);
          gb_bits = __hip_atomic_load(bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next refresh
        }
      };
      // The first block of each class is the SAME 16 walk positions for every tile of the workgroup, and 96 % of the tiles
      // die at the test that follows it: those points live in registers (first_in / first_bd, loaded once per wavefront), so
      // the rejection path of a tile touches LDS only for its (ty, tz) pair and never prefetches a block it will not use
      if (first_block) {
#pragma unroll
        for (int u = 0; u < kFirstIn; ++u) accumulate_interior(first_in[u], ay, az, delta2, A0, A1);
#pragma unroll
        for (int u = 0; u < kFirstBd; ++u) accumulate<OOB>(first_bd[u], ay, az, Wh, Hh, delta2, A0, A1);
        pin = kFirstIn * kSlices;
        pbd = Mi + kFirstBd * kSlices;
        if (PRUNE) {
          if (beaten())
            pruned = true;
          else
            refresh();
        }
      }
      // Survivors of the first test -- 72 % of the kernel's VALU instructions are spent here (tools/dev_k6_timing.py) -- walk
      // on in a loop made for the common case, "both classes still have a full block": one bound test per 8 + 8 positions, a
      // scalar trip count, and NO software prefetch: with 6-7 resident wavefronts per SIMD the LDS latency is covered by the
      // other wavefronts, and the registers of a second block in flight cost a wavefront of occupancy (measured with two
      // named register sets: 83 VGPRs, 267 k instead of 281 k frames/s).
      if (!(PRUNE && pruned)) {
        constexpr int kLoopIn = 0, kLoopBd = 3;   // interior- / border-class points per lane and trip of this loop
        constexpr uint32_t kStepIn = kLoopIn * kSlices, kStepBd = kLoopBd * kSlices;
        uint32_t both = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)min(kLoopIn ? (Mi - min(pin, Mi)) / (kLoopIn ? kStepIn : 1u) : 0x7FFFFFFFu, kLoopBd ? (M - pbd) / (kLoopBd ? kStepBd : 1u) : 0x7FFFFFFFu));
        for (; both; --both) {
          PointTerms bi[kLoopIn > 0 ? kLoopIn : 1], bb[kLoopBd > 0 ? kLoopBd : 1];
#pragma unroll
          for (int u = 0; u < kLoopIn; ++u) bi[u] = fetch(pin + u * kSlices + my_s);
#pragma unroll
          for (int u = 0; u < kLoopBd; ++u) bb[u] = fetch(pbd + u * kSlices + my_s);
#pragma unroll
          for (int u = 0; u < kLoopIn; ++u) accumulate_interior(bi[u], ay, az, delta2, A0, A1);
#pragma unroll
          for (int u = 0; u < kLoopBd; ++u) accumulate<OOB>(bb[u], ay, az, Wh, Hh, delta2, A0, A1);
          pin += kStepIn;
          pbd += kStepBd;
          since_refresh += kStepIn + kStepBd - kStep;   // (refresh() adds kStep)
          if (PRUNE) {
            if (beaten()) {
              pruned = true;
              break;
            }
            refresh();
          }
        }
      }
      // what is left when one class runs out of full blocks (a few dozen positions at the end of a complete walk)
      for (; !(PRUNE && pruned);) {
        const bool more_in = pin + kStep <= Mi, more_bd = pbd + kStep <= M;
        if (!more_in && !more_bd) break;
        if (more_in) {
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) accumulate_interior(fetch(pin + u * kSlices + my_s), ay, az, delta2, A0, A1);
          pin += kStep;
        }
        if (more_bd) {
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) accumulate<OOB>(fetch(pbd + u * kSlices + my_s), ay, az, Wh, Hh, delta2, A0, A1);
          pbd += kStep;
        }
        if (PRUNE) {
          if (beaten()) {
            pruned = true;
            break;
          }
          refresh();
        }
      }
      if (!(PRUNE && pruned)) {   // tails (< 8 points per class): one point per lane and trip, lanes past the end idle
        for (; pin < Mi; pin += kSlices) {
          const uint32_t at = pin + my_s;
          if (at < Mi) accumulate_interior(fetch(at), ay, az, delta2, A0, A1);
        }
        for (; pbd < M; pbd += kSlices) {
          const uint32_t at = pbd + my_s;
          if (at < M) accumulate<OOB>(fetch(at), ay, az, Wh, Hh, delta2, A0, A1);
        }
        pos = M;
        pts_in += Mi;
      } else {
        pos = pin + (pbd - Mi);
        pts_in += pin;
      }
    } else {
      PointTerms nxt[kUnroll];   // the next block's points are in flight while this block is evaluated
      if (kStep <= M) {
  #pragma unroll
        for (int u = 0; u < kUnroll; ++u) nxt[u] = fetch(u * kSlices + my_s);
      }
      for (; pos + kStep <= M; pos += kStep) {
        PointTerms pt[kUnroll];
  #pragma unroll
        for (int u = 0; u < kUnroll; ++u) pt[u] = nxt[u];
        if (pos + 2 * kStep <= M) {
  #pragma unroll
          for (int u = 0; u < kUnroll; ++u) nxt[u] = fetch(pos + kStep + u * kSlices + my_s);
        }
  #pragma unroll
        for (int u = 0; u < kUnroll; ++u) accumulate<OOB>(pt[u], ay, az, Wh, Hh, delta2, A0, A1);
        if (PRUNE) {
          const float part = fminf(quad_sum(A0), quad_sum(A1));
          if ((__ballot(!(part > lim2)) & owner_mask) == 0ull) {
            pruned = true;
            pos += kStep;
            break;
          }
          if (((pos + kStep) & (kBoundRefresh - 1)) == 0) {
            lim2 = 0.5f * (1.f + kTieEps) * fminf(__uint_as_float(gb_bits), best.cost);
            gb_bits = __hip_atomic_load(bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next refresh
          }
        }
      }
      if (!(PRUNE && pruned)) {
        if (!LDS_POINTS && Mfull) idx = (uint32_t)(((uint64_t)(pos + my_s) * S) % Mfull);
        for (; pos < M; pos += kSlices) {   // tail (< 12 points): one point per lane and trip, lanes past the end idle
          const uint32_t at = pos + my_s;
          if (at < M) {
            const PointTerms p1 = fetch(at);
            accumulate<OOB>(p1, ay, az, Wh, Hh, delta2, A0, A1);
          }
        }
        pos = M;
      }
    }
    pts_done += pos;
    if (PRUNE && pruned) return;

    const float t0s = quad_sum(A0), t1s = quad_sum(A1);   // the four slices of each candidate
    if (owner) {
      const uint32_t cell = ((uint32_t)k * (uint32_t)n_ty + (uint32_t)ia) * (uint32_t)n_tz + (uint32_t)ib;
      const uint32_t d2 = dk + (uint32_t)((ia - c.c_ty) * (ia - c.c_ty)) + (uint32_t)((ib - c.c_tz) * (ib - c.c_tz));
      const float c0 = 2.f * t0s, c1 = 2.f * t1s;
      if (c.tie_count != nullptr && my_s == 0 && fminf(c0, c1) <= 2.f * lim2) {   // full pass: near ties of the bound
        // completions are rare: afford a fresh look at the frame's bound so that little junk is listed while it is loose
        const float fresh = __uint_as_float(__hip_atomic_load(bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const float thr = (1.f + kTieEps) * fminf(fresh, fminf(best.cost, fminf(c0, c1)));
        // more than kTieCap entries: the count keeps growing, the consumer sees the overflow and falls back to the
        // (deterministic) fp32 argmin with ILCC_FLAG_TIE_OVERFLOW set -- which entries made it into the list never matters
        if (c0 <= thr) {
          const uint32_t at = atomicAdd(c.tie_count + f, 1u);
          if (at < (uint32_t)kTieCap) c.tie_list[(uint64_t)f * kTieCap + at] = GridPartial{c0, d2, 2u * cell, 0u};
        }
        if (c1 <= thr) {
          const uint32_t at = atomicAdd(c.tie_count + f, 1u);
          if (at < (uint32_t)kTieCap) c.tie_list[(uint64_t)f * kTieCap + at] = GridPartial{c1, d2, 2u * cell + 1u, 0u};
        }
      }
      if (better(c0, d2, 2u * cell, best)) best = Best{c0, d2, 2u * cell};
      if (better(c1, d2, 2u * cell + 1u, best)) best = Best{c1, d2, 2u * cell + 1u};
      if (VOLUME && my_s == 0) {
        vol[2u * cell] = c0;
        vol[2u * cell + 1u] = c1;
      }
    }
    if (PRUNE) {
      // share the wavefront's best complete cost with every workgroup of the frame
      float wb = best.cost;
#pragma unroll
      for (int o = ILCC_WAVE / 2; o > 0; o >>= 1) wb = fminf(wb, __shfl_xor(wb, o, ILCC_WAVE));
      // publish only what improves the frame's bound as this wavefront last saw it: atomics on a
      // word that thousands of wavefronts also load serialise in L2 (measured: 2.5x on the kernel)
      if (lane == 0 && wb < fminf(shared_bound, __uint_as_float(gb_bits))) {
        shared_bound = wb;
        atomicMin(bound, __float_as_uint(wb));   // costs are >= 0: uint order == float order
      }
    }
  };

  // (Measured and rejected: a PRE-PASS that evaluates the first 8 + 8 walk positions with lane = candidate for four tiles
  // at once -- one LDS broadcast per point, no quad reduction, one prologue and one bound test per four tiles -- and lets
  // only the surviving tiles into the quad-sliced loop.  An instruction-count model from simulated death times promised
  // -12 %; on the chip: 85 instead of 72 VGPRs (5 instead of 7 waves per SIMD) and sixteen serial evaluations per lane:
  // 0.611 instead of 0.564 ms per batch, 235 k instead of 250.6 k frames/s.)
  // This wavefront's tiles are wid, wid + 4, ... of the order rotated by t0 (the tile of the seed's best translation first).
  // 64 of them at a time, one per lane: the lanes look their tiles up in the pre-pass's bit mask, a ballot gives the ones
  // still alive, and only those are visited (a handful of a wavefront's 25: the loop over dead tiles was a third of the
  // kernel's scalar instructions).
  if constexpr (THREADS == kGridThreadsLarge && kGridThreadsLarge != kGridThreads) {
    // The 512-thread instance (frames of several thousand labelled points, ~1000 tiles per workgroup): which wavefront takes
    // which tile is decided at run time.  The tiles that survive the pre-pass AND walk far (the candidates around the minimum)
    // used to fall to whichever wavefront the static interleave gave them, and the workgroup waited at its last barrier for the
    // unlucky one (16 % of a wavefront's life on BASELINE config 5, tools/dev_k6_timing.py): 23.2 -> 24.0 k frames/s there.  (The
    // 256-thread instance keeps the static interleave below: with 100 tiles and 2.4 survivors per wavefront the claims cost
    // more than they balance -- 775 -> 740 k frames/s.)  A wavefront
    // claims CHUNKS of kChunk tile slots from a counter in LDS; slot n of chunk c is tile (c + n_chunks * n + t0) mod n_tiles,
    // so neighbouring tiles -- the expensive ones are neighbours -- sit in different chunks, and chunk 0 starts with the tile of
    // the seed's best translation.  The lanes look their slots up in the pre-pass's bit mask, a ballot gives the ones still
    // alive, and only those are visited.  The order never changes the result.
    constexpr int kChunk = 8;   // tile slots per claim
    static_assert(kChunk >= 1 && kChunk <= ILCC_WAVE, "a chunk is looked up by the lanes of one wavefront");
    const int n_chunks = __builtin_amdgcn_readfirstlane((n_tiles + kChunk - 1) / kChunk);
    const int ntb_s = __builtin_amdgcn_readfirstlane(ntb);
    for (;;) {
      int cl = 0;
      if (lane == 0) cl = (int)atomicAdd(s_next, 1u);
      const int ch = __builtin_amdgcn_readfirstlane(cl);
      if (ch >= n_chunks) break;
      bool alive = false;
      if (lane < kChunk) {
        const int slot = ch + n_chunks * lane;
        if (slot < n_tiles) {
          int q = slot + t0;
          if (q >= n_tiles) q -= n_tiles;
          alive = !use_box || !((s_dead[q >> 5] >> (q & 31)) & 1u);
        }
      }
      unsigned long long todo = __ballot(alive);
      while (todo) {
        const int bit = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(todo));
        todo &= todo - 1ull;
        int qs = ch + n_chunks * bit + t0;
        if (qs >= n_tiles) qs -= n_tiles;
        const int tile_a = __builtin_amdgcn_readfirstlane(qs / ntb_s), tile_b = __builtin_amdgcn_readfirstlane(qs - tile_a * ntb_s);
#ifdef ILCC_K6_TIMING
        const unsigned long long tt0 = K6_NOW();
        const uint32_t pd0 = pts_done;
        const float bc0 = best.cost;
        const uint32_t bf0 = best.flat;
#endif
        run_tile(tile_a, tile_b, 0.f, 0.f, 0u, Mi);
#ifdef ILCC_K6_TIMING
        {
          const unsigned long long dt = K6_NOW() - tt0;
          const uint32_t walked = pts_done - pd0;
          if (walked <= (uint32_t)(2 * kStep) && walked < M) {
            ++n_rej;
            t_rej += dt;
          } else {
            ++n_surv;
            t_surv += dt;
            p_surv += walked;
            if (walked >= M) ++n_done;
          }
          (void)bc0;
          (void)bf0;
        }
#endif
      }
    }
  } else {
    constexpr int kWaves = THREADS / ILCC_WAVE;
    const int per_wave = __builtin_amdgcn_readfirstlane(wid < n_tiles ? (n_tiles - wid + kWaves - 1) / kWaves : 0);
    const int ntb_s = __builtin_amdgcn_readfirstlane(ntb);
    for (int k0 = 0; k0 < per_wave; k0 += ILCC_WAVE) {
      bool alive = false;
      if (k0 + lane < per_wave) {
        int q = wid + kWaves * (k0 + lane) + t0;
        if (q >= n_tiles) q -= n_tiles;
        alive = !use_box || !((s_dead[q >> 5] >> (q & 31)) & 1u);
      }
      unsigned long long todo = __ballot(alive);
      while (todo) {
        const int bit = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(todo));
        todo &= todo - 1ull;
        int qs = wid + kWaves * (k0 + bit) + t0;
        if (qs >= n_tiles) qs -= n_tiles;
        const int tile_a = __builtin_amdgcn_readfirstlane(qs / ntb_s), tile_b = __builtin_amdgcn_readfirstlane(qs - tile_a * ntb_s);
#ifdef ILCC_K6_TIMING
        const unsigned long long tt0 = K6_NOW();
        const uint32_t pd0 = pts_done;
        const float bc0 = best.cost;
        const uint32_t bf0 = best.flat;
#endif
        run_tile(tile_a, tile_b, 0.f, 0.f, 0u, Mi);
#ifdef ILCC_K6_TIMING
        {
          const unsigned long long dt = K6_NOW() - tt0;
          const uint32_t walked = pts_done - pd0;
          if (walked <= (uint32_t)(2 * kStep) && walked < M) {
            ++n_rej;
            t_rej += dt;
          } else {
            ++n_surv;
            t_surv += dt;
            p_surv += walked;
            if (walked >= M) ++n_done;
          }
          (void)bc0;
          (void)bf0;
        }
#endif
      }
    }
  }
  [[maybe_unused]] const unsigned long long t_tiles = K6_NOW();

  // lanes hold different candidates: wavefront argmin, then across the 4 wavefronts
#pragma unroll
  for (int o = ILCC_WAVE / 2; o > 0; o >>= 1) {
    Best t;
    t.cost = __shfl_down(best.cost, o, ILCC_WAVE);
    t.d2 = __shfl_down(best.d2, o, ILCC_WAVE);
    t.flat = __shfl_down(best.flat, o, ILCC_WAVE);
    if (better(t.cost, t.d2, t.flat, best)) best = t;
  }
  if (lane == 0) {
    s_best[wid] = best;
    s_iters[wid] = pts_done;
    s_cnt[wid] = pts_in;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t it_sum = 0, in_sum = 0;
    for (int w = 0; w < THREADS / ILCC_WAVE; ++w) {
      it_sum += s_iters[w];
      in_sum += s_cnt[w];
    }
    atomicAdd(c.grid_iters + (f & (kIterSlots - 1)), (unsigned long long)it_sum);   // spread over 64 words
    atomicAdd(c.grid_iters + kIterSlots + (f & (kIterSlots - 1)), (unsigned long long)in_sum);
    if (use_box) atomicAdd(c.grid_iters + 2 * kIterSlots + (f & (kIterSlots - 1)), box_evals);
    Best b = s_best[0];
    for (int w = 1; w < THREADS / ILCC_WAVE; ++w)
      if (better(s_best[w].cost, s_best[w].d2, s_best[w].flat, b)) b = s_best[w];
    out->cost = b.cost;
    out->d2 = b.d2;
    out->flat = b.flat;
    // (a << 16) | b of the winner in this launch's tables: the next launch reads it instead of dividing the flat index
    uint32_t ab = 0;
    if (b.flat != 0xFFFFFFFFu) {
      const uint32_t cell = b.flat >> 1;
      ab = (((cell / (uint32_t)n_tz) % (uint32_t)n_ty) << 16) | (cell % (uint32_t)n_tz);
    }
    out->pad = ab;
  }
#ifdef ILCC_K6_TIMING
  if (lane == 0 && c.tie_count != nullptr) {   // the full pass only
    const unsigned long long t_end = K6_NOW();
    const uint32_t w = ((f * c.grid_blocks + kblk) * (THREADS / ILCC_WAVE) + (uint32_t)wid) & (kProfWaves - 1);
    unsigned long long* o = k6_prof + (size_t)w * kProfWords;
    o[0] = 1ull;
    o[1] = t_end - t_entry;
    o[2] = t_staged - t_entry;
    o[3] = n_rej;
    o[4] = t_rej;
    o[5] = n_surv;
    o[6] = t_surv;
    o[7] = p_surv;
    o[8] = n_done;
    o[9] = t_end - t_tiles;
    o[10] = (unsigned long long)M;
    o[11] = (unsigned long long)Mi;
    o[12] = n_tests;
    o[13] = alive_sum;
    o[14] = alive_le4;
    o[15] = alive_le2;
  }
#endif
}

// dynamic LDS: [grid_lds_points float2][grid_lds_points float][n_ty + n_tz floats]; frames with more
// labelled points than the staged capacity read (and rotate) them through L1/L2 instead.
// THREADS: 256 (4 wavefronts) is the measured optimum for VLP-16-sized frames; frames of several thousand labelled points
// (BASELINE config 5: 4 198) stage 50+ KB per workgroup, three workgroups fit a CU, and 512 threads then double the
// resident wavefronts per SIMD (3 -> 6).  Same sums either way: a candidate's points are added by its quad in walk order.
template <bool OOB, bool VOLUME, bool PRUNE, int THREADS>
__global__ __launch_bounds__(THREADS) void k6_grid_cost(Ctx c, float* volume) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ Best s_best[THREADS / ILCC_WAVE];
  __shared__ uint32_t s_iters[THREADS / ILCC_WAVE];
  __shared__ uint32_t s_cnt[THREADS / ILCC_WAVE];
  __shared__ uint32_t s_dead[kBoxTilesMax / 32];   // box pre-pass: one bit per tile of the workgroup
  __shared__ uint16_t s_live[kBoxSegment<THREADS>];         // box pre-pass: the tiles still alive, compacted
  __shared__ uint32_t s_next;                      // next chunk of tiles to hand to a wavefront
  float2* s_ij = reinterpret_cast<float2*>(smem);
  float* s_hw = reinterpret_cast<float*>(smem + sizeof(float2) * (size_t)c.grid_lds_points);
  float* s_ay = s_hw + c.grid_lds_points;   // n_ty floats
  float* s_az = s_ay + c.p.n_ty;            // n_tz floats
  const uint32_t Mall = c.n_lab[blockIdx.y];
  const uint32_t M = c.walk_limit ? min(Mall, max(c.walk_limit, Mall >> kSeedShift)) : Mall;
  (void)M;
  if (Mall <= c.grid_lds_points) {   // (k5w_walk_order has laid out every frame of at most kGridLdsPointsMax points)
    grid_cost_body<OOB, VOLUME, true, PRUNE, THREADS>(c, volume, s_ij, s_hw, s_best, s_iters, s_cnt, s_ay, s_az, s_dead, s_live, &s_next, blockIdx.x);
    return;
  }
  if constexpr (OOB && !VOLUME && PRUNE) {
    // the pipeline's full pass on a frame above the staging capacity: the staged prefix + the rest through L2, as long as the
    // interior class and the box pre-pass's sample (what every workgroup reads) are inside the prefix
    constexpr int kBoxShift = (THREADS == kGridThreadsLarge && kGridThreadsLarge != kGridThreads) ? kBoxShiftLarge : kBoxShiftSmall;
    if (c.walk_limit == 0u && c.box_points != 0u && Mall <= (uint32_t)kGridLdsPointsMax) {
      const uint32_t Mi = c.walk_mi[blockIdx.y];
      if (Mall > Mi && Mi + min(max(c.box_points, Mall >> kBoxShift), Mall - Mi) <= c.grid_lds_points) {
        grid_cost_body<OOB, VOLUME, true, PRUNE, THREADS, true>(c, volume, s_ij, s_hw, s_best, s_iters, s_cnt, s_ay, s_az, s_dead, s_live, &s_next,
                                                                blockIdx.x);
        return;
      }
    }
  }
  grid_cost_body<OOB, VOLUME, false, PRUNE, THREADS>(c, volume, s_ij, s_hw, s_best, s_iters, s_cnt, s_ay, s_az, s_dead, s_live, &s_next, blockIdx.x);
}

// k6_group_prepass: ONE box pre-pass for a GROUP of kThetaGroup consecutive thetas, in front of the full pass.  71 % of the (frame,
// theta) workgroups of the full pass die in their own box pre-pass -- staging, tables, three barriers, ~450 instructions per
// wavefront each: 30 % of the kernel -- and a theta step moves a point by less than a third of a tile's width.  Every pre-pass
// point is rotated by all thetas of the group (the term's own fp32 expressions) and the box bound takes the extremes: i_lo from the
// smallest rotated coordinate and the box's lowest translation, i_hi from the largest and the highest.  fl(p + a) is monotone
// in p as in a, so [i_lo, i_hi] contains the interval each theta's own pre-pass uses: the bound is a lower bound for all group x 16
// candidates by box_term's argument unchanged (a point whose images lie more than half a square apart on an axis is left
// out; the interval stays far narrower than a board).  Output per (frame, group): a state word -- 0: every tile rejected (the
// group's full-pass workgroups exit on their first instructions), 1: a bit mask of the rejected tiles follows (their own
// pre-pass starts from it and only looks at the rest), 2: no common pre-pass (conditions not met) -- and the mask.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k6_group_prepass(Ctx c, uint32_t* grp_alive, uint32_t* grp_mask) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ uint32_t s_iters[THREADS / ILCC_WAVE];
  __shared__ uint32_t s_dead3[kBoxTilesMax / 32];
  __shared__ uint16_t s_live[kBoxSegment<THREADS>];   // the tiles still alive, compacted (box_prepass_rounds)
  __shared__ uint32_t s_cnt2[2];
  __shared__ uint32_t s_any;
  const uint32_t f = blockIdx.y;
  const int k0 = kThetaGroup * (int)blockIdx.x, nk = min(kThetaGroup, c.p.n_th - k0);
  const uint32_t tr = f * c.grp_count + blockIdx.x;
  const uint32_t Mall = c.n_lab[f];
  // (frames above the full pass's staging capacity have a walk layout too -- k5w lays out every frame of at most kGridLdsPointsMax
  // points -- and their full pass reads this mask in its OVERFLOW form)
  const bool lds = Mall <= (uint32_t)kGridLdsPointsMax;
  const int n_ty = c.p.n_ty, n_tz = c.p.n_tz;
  const int nta = (n_ty + kTile - 1) / kTile, ntb = (n_tz + kTile - 1) / kTile, n_tiles = nta * ntb;
  const uint32_t Mi = lds ? c.walk_mi[f] : 0u;
  // (every condition is uniform over the workgroup)
  if (!(c.res[f].status == ILCC_OK && lds && nk > 1 && c.box_points != 0u && n_tiles <= kBoxTilesMax && Mall > Mi)) {
    if (threadIdx.x == 0) grp_alive[tr] = 2u;
    return;
  }
  const int lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(wave_id());
  // (the sample never exceeds what launch_group_prepass sized the LDS for: any prefix of the rim-first walk gives a valid bound)
  constexpr int kGroupShift = (THREADS == kGridThreadsLarge && kGridThreadsLarge != kGridThreads) ? kGroupShiftLarge : kGroupShiftSmall;
  const uint32_t n_pre = min(min(max(c.box_points, Mall >> kGroupShift), Mall - Mi),
                             max(c.grid_lds_points / 2u + 64u, c.box_points));
  float4* s_w4 = reinterpret_cast<float4*>(smem);   // n_pre x (pi_lo, pi_hi, pj_lo, pj_hi)
  float* s_ay = reinterpret_cast<float*>(s_w4 + n_pre);
  float* s_az = s_ay + n_ty;
  const float2* __restrict__ wyz = c.walk_yz + c.off[f];
  float cth[kThetaGroup], sth[kThetaGroup];
#pragma unroll
  for (int t = 0; t < kThetaGroup; ++t) {
    cth[t] = c.cth[k0 + min(t, nk - 1)];
    sth[t] = c.sth[k0 + min(t, nk - 1)];
  }
  for (uint32_t sl = threadIdx.x; sl < n_pre; sl += THREADS) {
    const float2 v = wyz[Mi + sl];   // the rim-first border-class part of the walk layout: what each theta's own pre-pass looks at
    float ilo = __builtin_inff(), ihi = -__builtin_inff(), jlo = __builtin_inff(), jhi = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < kThetaGroup; ++t) {
      const float pi = fmaf(-sth[t], v.y, cth[t] * v.x), pj = fmaf(cth[t], v.y, sth[t] * v.x);   // = the full pass's staging
      ilo = fminf(ilo, pi);
      ihi = fmaxf(ihi, pi);
      jlo = fminf(jlo, pj);
      jhi = fmaxf(jhi, pj);
    }
    // a point far from the rotation centre: leave it out (as a point at the board's centre, in the board under every
    // translation of the tables -- the host launches this kernel only then -- it contributes nothing to any bound)
    const bool wide = !(ihi - ilo <= 0.5f && jhi - jlo <= 0.5f);
    s_w4[sl] = wide ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(ilo, ihi, jlo, jhi);
  }
  for (int i = threadIdx.x; i < n_ty; i += THREADS) s_ay[i] = c.ay[i];
  for (int i = threadIdx.x; i < n_tz; i += THREADS) s_az[i] = c.az[i];
  for (int w = threadIdx.x; w < (n_tiles + 31) / 32; w += THREADS) s_dead3[w] = 0u;
  if (threadIdx.x == 0) s_any = 0u;
  __syncthreads();
  const float Wh = 0.5f * (float)c.p.board_w, Hh = 0.5f * (float)c.p.board_h, delta2 = (float)c.p.huber_delta;
  const float lim_box = 0.5f * (1.f + kTieEps) * __uint_as_float(__hip_atomic_load(c.grid_bound + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const uint32_t wave_evals = box_prepass_rounds<THREADS>(n_tiles, ntb, 0, 0, n_ty, n_tz, s_ay, s_az, s_dead3, s_live, s_cnt2, &s_any, n_pre, lim_box, Wh, Hh,
                                                          delta2, [&](uint32_t u) { return s_w4[u]; });
  if (lane == 0) s_iters[wid] = wave_evals;
  __syncthreads();
  const bool any_alive = s_any != 0u;
  if (any_alive)
    for (int w = threadIdx.x; w < (n_tiles + 31) / 32; w += THREADS) grp_mask[(uint64_t)tr * c.grp_words + w] = s_dead3[w];
  if (threadIdx.x == 0) {
    unsigned long long box_evals = 0;
    for (int w = 0; w < THREADS / ILCC_WAVE; ++w) box_evals += s_iters[w];
    atomicAdd(c.grid_iters + 2 * kIterSlots + (f & (kIterSlots - 1)), box_evals);
    grp_alive[tr] = any_alive ? 1u : 0u;
  }
}

void launch_group_prepass(const Ctx& c, hipStream_t s, uint32_t* grp_alive, uint32_t* grp_mask) {
  const dim3 grid(c.grp_count, c.n_frames);
  // LDS: the widened pre-pass points (16 B each: at most M >> 1 of the frame's M <= grid_lds_points labelled points, or box_points of
  // them when that is more -- the static_assert next to kGroupShiftSmall) and the (ty, tz) tables
  const size_t lds = sizeof(float4) * std::max<size_t>((size_t)c.grid_lds_points / 2 + 64, (size_t)c.box_points) + sizeof(float) * (size_t)(c.p.n_ty + c.p.n_tz);
  if (c.grid_lds_points > (uint32_t)kGridLargeFrom)
    hipLaunchKernelGGL((k6_group_prepass<kGridThreadsLarge>), grid, dim3(kGridThreadsLarge), lds, s, c, grp_alive, grp_mask);
  else
    hipLaunchKernelGGL((k6_group_prepass<kGridThreads>), grid, dim3(kGridThreads), lds, s, c, grp_alive, grp_mask);
}

// k6_locate (round 5): seed + refinement + anchor in ONE launch, one workgroup per frame.  The three launches it replaces only
// have to say WHERE the minimum is and publish the frame's bound; as 17 small workgroups per frame in three dependent launches
// they were 277 of the K6 stage's 733 us with a batch alone on the chip, at 16-28 % issue utilisation: each stages
// its points and tables again, reads the previous launch's argmin from global memory, and the anchor's workgroups run one
// wavefront of four.  Here a frame's 8 wavefronts
//   1. stage the SAMPLE of the walk (an eighth of the labelled points, a proportional prefix of each class of the walk layout)
//      once, rotated by the five seed thetas; 5 x (8 x 8 decimated translations) = 20 tiles, one per wavefront at a time,
//      pruned against a bound word in LDS; argmin in LDS;
//   2. rotate the sample by the (2 r + 1) thetas around the seed's and score their 8 x 8 windows of full-table translations
//      (36 tiles); argmin in LDS;
//   3. score the 4 x 4 tile around that argmin at theta - 1, theta, theta + 1 on EVERY labelled point (rotated on the fly from
//      the walk layout in L2), two wavefronts per theta on alternate blocks of the walk, and publish the best complete cost as
//      the frame's bound and its candidate as the full pass's starting tile.
// The same candidates on the same points as the three launches; the sums are grouped differently (the anchor's two half-walks
// are added at the end), which the full pass cannot see: the bound only has to be the cost of SOME complete candidate, and a
// candidate is cut, or listed as a near tie, with 2e-5 to spare -- 20 x the rounding of an fp32 sum of a thousand terms.
// Batches of fewer than kLocateMinFrames frames keep the three launches: there a frame's 17 workgroups are what fills the chip.
constexpr int kLocateThreads = 384;       // 6 wavefronts = 3 anchor thetas x 2 parts of the walk, all busy in the anchor.  Measured (k frames/s / locate alone): 256: 1056, 512: 1089, 1024: 1011-1033 when it was built; on the final build 384: 1266 / 0.184 ms, 512: 1268 / 0.214 ms, 768: 1251 / 0.201 ms
constexpr int kSeedPeek = 16;             // walk positions of every seed tile a wavefront looks at before it chooses which tile to evaluate first
constexpr int kAnchorThetas = 3;          // thetas around the refinement's argmin the anchor scores (1: locate 0.26 -> 0.21 ms alone, full pass 0.36 -> 0.40: a wash)
constexpr int kAnchorParts = (kLocateThreads / 64) / kAnchorThetas;  // wavefronts sharing a theta's walk
constexpr int kLocateWaves = kLocateThreads / ILCC_WAVE;
constexpr int kLocateThetasMax = 16;      // rotations of the sample kept in LDS at a time (seed: 5, refinement: 2 r + 1 = 9)
__global__ __launch_bounds__(kLocateThreads) void k6_locate(Ctx c, LocatePlan lp) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ Best s_best[kLocateWaves];
  __shared__ uint32_t s_ab[kLocateWaves];
  __shared__ uint32_t s_bound;              // float bits: best complete SAMPLE cost so far (seed and refinement share the sample)
  __shared__ uint32_t s_pick[4];            // argmin of a phase: flat, (a << 16) | b
  __shared__ uint32_t s_iters[kLocateWaves], s_iters_in[kLocateWaves];
  __shared__ float s_part[kAnchorThetas][kAnchorParts][16][2];     // anchor: (theta, part of the walk, candidate, phase) partial costs / 2
  const uint32_t f = blockIdx.x;
  GridPartial* out = lp.out + f;
  const int lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(wave_id());
  const uint32_t Mfull = c.n_lab[f];
  if (c.res[f].status != ILCC_OK) {
    if (threadIdx.x == 0) *out = GridPartial{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu, 0u};
    return;
  }
  const uint64_t beg = c.off[f];
  const bool walk_layout = Mfull <= (uint32_t)kGridLdsPointsMax;   // k5w_walk_order has laid the frame out; else: the points in golden-ratio order through L2
  const float2* __restrict__ wyz = walk_layout ? c.walk_yz + beg : c.yz + beg;
  const uint8_t* __restrict__ wlab = walk_layout ? c.walk_lab + beg : c.lab + beg;
  const uint32_t S = (!walk_layout && Mfull) ? c.walk_stride[f] : 1u;
  const uint32_t Mi_all = walk_layout ? c.walk_mi[f] : 0u, n_rim_all = walk_layout ? c.walk_nrim[f] : 0u;
  const uint32_t Ms = min(min(Mfull, max(lp.sample_min, Mfull >> kSeedShift)), lp.sample_cap);
  uint32_t n_in = Mi_all, n_rm = n_rim_all;
  if (Ms < Mfull) {
    n_in = (uint32_t)(((uint64_t)Mi_all * Ms) / Mfull);
    n_rm = (uint32_t)(((uint64_t)n_rim_all * Ms) / Mfull);
  }
  const int n_ty = c.p.n_ty, n_tz = c.p.n_tz, n_th = c.p.n_th;
  // dynamic LDS: the raw sample, its rotations, the (ty, tz) tables (full, then the seed's)
  float2* s_raw = reinterpret_cast<float2*>(smem);
  float* s_hw = reinterpret_cast<float*>(s_raw + lp.sample_cap);
  float2* s_rot = reinterpret_cast<float2*>(s_hw + lp.sample_cap);                  // [kLocateThetasMax][sample_cap]
  float* s_ay = reinterpret_cast<float*>(s_rot + (size_t)kLocateThetasMax * lp.sample_cap);
  float* s_az = s_ay + n_ty;
  float* s_ay2 = s_az + n_tz;
  float* s_az2 = s_ay2 + lp.n_ty2;
  for (uint32_t sl = threadIdx.x; sl < Ms; sl += kLocateThreads) {
    uint32_t src = sl < n_in ? sl : sl < n_in + n_rm ? Mi_all + (sl - n_in) : Mi_all + n_rim_all + (sl - n_in - n_rm);
    src = min(src, Mfull - 1u);
    if (!walk_layout) src = (uint32_t)(((uint64_t)sl * S) % Mfull);
    s_raw[sl] = wyz[src];
    s_hw[sl] = wlab[src] ? 0.5f : 0.f;
  }
  for (int i = threadIdx.x; i < n_ty; i += kLocateThreads) s_ay[i] = c.ay[i];
  for (int i = threadIdx.x; i < n_tz; i += kLocateThreads) s_az[i] = c.az[i];
  for (int i = threadIdx.x; i < lp.n_ty2; i += kLocateThreads) s_ay2[i] = lp.ay2[i];
  for (int i = threadIdx.x; i < lp.n_tz2; i += kLocateThreads) s_az2[i] = lp.az2[i];
  if (threadIdx.x == 0) s_bound = 0x7f800000u;
  const float Wh = 0.5f * (float)c.p.board_w, Hh = 0.5f * (float)c.p.board_h, delta2 = (float)c.p.huber_delta;
  const int my_s = lane & (kSlices - 1), my_c = lane >> 2, my_a = my_c >> 2, my_b = my_c & 3;
  uint32_t pts_done = 0, pts_in = 0;

  // the sample rotated by n_rot thetas (theta index of rotation r: th_of(r))
  auto rotate_sample = [&](int n_rot, auto th_of, const float* cth_tab, const float* sth_tab) {
    for (uint32_t item = threadIdx.x; item < (uint32_t)n_rot * Ms; item += kLocateThreads) {
      const uint32_t r = item / Ms, sl = item - r * Ms;
      const int k = th_of((int)r);
      const float cth = cth_tab[k], sth = sth_tab[k];
      const float2 v = s_raw[sl];
      s_rot[(size_t)r * lp.sample_cap + sl] = make_float2(fmaf(-sth, v.y, cth * v.x), fmaf(cth, v.y, sth * v.x));   // = k6_grid_cost's staging
    }
  };
  // one 4 x 4 tile on the rotated sample `rot`, quad-sliced like k6_grid_cost (border class first, a bound test every 8 walk
  // positions against the sample bound); false: every candidate provably loses
  auto sample_tile = [&](const float2* rot, float ay, float az, float& c0, float& c1) -> bool {
    float A0 = 0.f, A1 = 0.f;
    uint32_t since = 0;
    float lim2 = 0.5f * (1.f + kTieEps) * __uint_as_float(s_bound);
    uint32_t pos = 0;
    for (uint32_t at0 = n_in; at0 < Ms; at0 += kSlices) {   // (at0 is wave-uniform: every lane makes the same trips)
      const uint32_t at = at0 + (uint32_t)my_s;
      if (at < Ms) {
        const float2 v = rot[at];
        accumulate<true>(PointTerms{v.x, v.y, s_hw[at]}, ay, az, Wh, Hh, delta2, A0, A1);
      }
      pos += kSlices;
      since += kSlices;
      if (since >= 8u) {
        since = 0;
        const float part = fminf(quad_sum(A0), quad_sum(A1));
        if (__ballot(!(part > lim2)) == 0ull) {
          pts_done += pos;
          return false;
        }
        lim2 = 0.5f * (1.f + kTieEps) * __uint_as_float(s_bound);
      }
    }
    for (uint32_t at0 = 0; at0 < n_in; at0 += kSlices) {
      const uint32_t at = at0 + (uint32_t)my_s;
      if (at < n_in) {
        const float2 v = rot[at];
        accumulate_interior(PointTerms{v.x, v.y, s_hw[at]}, ay, az, delta2, A0, A1);
      }
      pos += kSlices;
      since += kSlices;
      if (since >= 8u) {
        since = 0;
        const float part = fminf(quad_sum(A0), quad_sum(A1));
        if (__ballot(!(part > lim2)) == 0ull) {
          pts_done += pos;
          pts_in += at0 + kSlices;
          return false;
        }
        lim2 = 0.5f * (1.f + kTieEps) * __uint_as_float(s_bound);
      }
    }
    pts_done += pos;
    pts_in += n_in;
    c0 = 2.f * quad_sum(A0);
    c1 = 2.f * quad_sum(A1);
    return true;
  };
  // a phase's argmin: every wavefront's best -> LDS -> s_pick (flat, ab)
  auto reduce_best = [&](Best best, uint32_t ab) {
#pragma unroll
    for (int o = ILCC_WAVE / 2; o > 0; o >>= 1) {
      Best t;
      t.cost = __shfl_down(best.cost, o, ILCC_WAVE);
      t.d2 = __shfl_down(best.d2, o, ILCC_WAVE);
      t.flat = __shfl_down(best.flat, o, ILCC_WAVE);
      const uint32_t tab = __shfl_down(ab, o, ILCC_WAVE);
      if (better(t.cost, t.d2, t.flat, best)) {
        best = t;
        ab = tab;
      }
    }
    if (lane == 0) {
      s_best[wid] = best;
      s_ab[wid] = ab;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      Best b = s_best[0];
      uint32_t bab = s_ab[0];
      for (int w = 1; w < kLocateWaves; ++w)
        if (better(s_best[w].cost, s_best[w].d2, s_best[w].flat, b)) {
          b = s_best[w];
          bab = s_ab[w];
        }
      s_pick[0] = b.flat;
      s_pick[1] = bab;
      s_pick[2] = __float_as_uint(b.cost);
      s_pick[3] = b.d2;
    }
    __syncthreads();
  };
  __syncthreads();

  // ---- 1. seed: n_th2 thetas x (n_ty2 x n_tz2 decimated translations) on the sample
  const int n_th2 = min(lp.n_th2, kLocateThetasMax);
  rotate_sample(n_th2, [&](int r) { return r; }, lp.cth2, lp.sth2);
  __syncthreads();
  {
    const int nta = (lp.n_ty2 + kTile - 1) / kTile, ntb = (lp.n_tz2 + kTile - 1) / kTile, per = nta * ntb, n_tiles = n_th2 * per;
    Best best{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t best_ab = 0;
    // The order of the tiles never changes the result (only provably losing candidates are cut, ties never), but it decides how
    // soon the sample bound is tight.  Every wavefront therefore first looks at the first kSeedPeek walk positions of each of
    // its tiles, evaluates the most promising one in full -- one of the eight is nearly always in the right basin -- and only then
    // walks the others, which now die at their first tests.
    int t_first = -1;
    {
      float peek_best = __builtin_inff();
      for (int t = wid; t < n_tiles; t += kLocateWaves) {
        const int k2 = t / per, q = t - k2 * per, ta = q / ntb, tb = q - ta * ntb;
        const int ia = ta * kTile + my_a, ib = tb * kTile + my_b;
        const float2* rot = s_rot + (size_t)k2 * lp.sample_cap;
        const float ay = s_ay2[min(ia, lp.n_ty2 - 1)], az = s_az2[min(ib, lp.n_tz2 - 1)];
        float A0 = 0.f, A1 = 0.f;
        for (uint32_t at0 = n_in; at0 < min(Ms, n_in + (uint32_t)kSeedPeek); at0 += kSlices) {
          const uint32_t at = at0 + (uint32_t)my_s;
          if (at < Ms) {
            const float2 v = rot[at];
            accumulate<true>(PointTerms{v.x, v.y, s_hw[at]}, ay, az, Wh, Hh, delta2, A0, A1);
          }
        }
        pts_done += min(Ms - n_in, (uint32_t)kSeedPeek);
        float pm = (ia < lp.n_ty2 && ib < lp.n_tz2) ? fminf(quad_sum(A0), quad_sum(A1)) : __builtin_inff();
#pragma unroll
        for (int o = ILCC_WAVE / 2; o > 0; o >>= 1) pm = fminf(pm, __shfl_xor(pm, o, ILCC_WAVE));
        if (pm < peek_best) {   // (wave-uniform)
          peek_best = pm;
          t_first = t;
        }
      }
    }
    const int n_mine = (n_tiles - wid + kLocateWaves - 1) / kLocateWaves;   // tiles of this wavefront
    for (int it = (t_first >= 0 ? -1 : 0); it < n_mine; ++it) {
      const int t = it < 0 ? t_first : wid + it * kLocateWaves;
      if (it >= 0 && t == t_first) continue;
      const int k2 = t / per, q = t - k2 * per, ta = q / ntb, tb = q - ta * ntb;
      const int ia = ta * kTile + my_a, ib = tb * kTile + my_b;
      const bool owner = ia < lp.n_ty2 && ib < lp.n_tz2;
      float c0, c1;
      if (!sample_tile(s_rot + (size_t)k2 * lp.sample_cap, s_ay2[min(ia, lp.n_ty2 - 1)], s_az2[min(ib, lp.n_tz2 - 1)], c0, c1)) continue;
      if (owner) {
        const uint32_t cell = ((uint32_t)k2 * (uint32_t)lp.n_ty2 + (uint32_t)ia) * (uint32_t)lp.n_tz2 + (uint32_t)ib;
        const uint32_t d2 = (uint32_t)((k2 - lp.c_th2) * (k2 - lp.c_th2) + (ia - lp.c_ty2) * (ia - lp.c_ty2) + (ib - lp.c_tz2) * (ib - lp.c_tz2));
        const uint32_t ab = ((uint32_t)ia << 16) | (uint32_t)ib;
        if (better(c0, d2, 2u * cell, best)) { best = Best{c0, d2, 2u * cell}; best_ab = ab; }
        if (better(c1, d2, 2u * cell + 1u, best)) { best = Best{c1, d2, 2u * cell + 1u}; best_ab = ab; }
      }
      float wb = owner ? fminf(c0, c1) : __builtin_inff();
#pragma unroll
      for (int o = ILCC_WAVE / 2; o > 0; o >>= 1) wb = fminf(wb, __shfl_xor(wb, o, ILCC_WAVE));
      if (lane == 0) atomicMin(&s_bound, __float_as_uint(wb));   // costs are >= 0: uint order == float order
    }
    reduce_best(best, best_ab);
  }
  if (s_pick[0] == 0xFFFFFFFFu) {   // (no candidate at all: an empty frame)
    if (threadIdx.x == 0) *out = GridPartial{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu, 0u};
    return;
  }
  // ---- 2. refinement: theta within +- refine_radius steps of the seed's, the 8 x 8 window of full-table translations around its argmin
  const int k2s = (int)((s_pick[0] >> 1) / (uint32_t)(lp.n_ty2 * lp.n_tz2));
  const int sa = min((int)(s_pick[1] >> 16) * lp.stride_t, n_ty - 1), sbb = min((int)(s_pick[1] & 0xFFFFu) * lp.stride_t, n_tz - 1);
  // every OTHER theta of the refinement's range: the anchor scores theta - 1, theta, theta + 1 around its argmin on every point, so
  // the thetas in between are still looked at -- measured: locate 0.265 -> 0.229 ms alone, the full pass's time and executed work
  // unchanged (the bound is as tight), bench 1085 -> 1112 k frames/s; every fourth theta, or every other translation: no further gain
  constexpr int kRefStride = kRefineThetaStride;
  const int n_ref = min(2 * (lp.refine_radius / kRefStride) + 1, kLocateThetasMax);
  const int k_ref0 = lp.off_th + k2s * lp.stride_th - (lp.refine_radius / kRefStride) * kRefStride;
  rotate_sample(n_ref, [&](int r) { return min(max(k_ref0 + kRefStride * r, 0), n_th - 1); }, c.cth, c.sth);
  __syncthreads();
  {
    const int a_org = min(max(sa - kTile, 0), max(n_ty - 2 * kTile, 0)), b_org = min(max(sbb - kTile, 0), max(n_tz - 2 * kTile, 0));
    const int nta = min(2, (n_ty + kTile - 1) / kTile), ntb = min(2, (n_tz + kTile - 1) / kTile), per = nta * ntb, n_tiles = n_ref * per;
    Best best{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t best_ab = 0;
    for (int t = wid; t < n_tiles; t += kLocateWaves) {
      const int r = t / per, q = t - r * per, ta = q / ntb, tb = q - ta * ntb;
      const int k = min(max(k_ref0 + kRefStride * r, 0), n_th - 1);
      const int ia = a_org + ta * kTile + my_a, ib = b_org + tb * kTile + my_b;
      const bool owner = ia < n_ty && ib < n_tz;
      float c0, c1;
      if (!sample_tile(s_rot + (size_t)r * lp.sample_cap, s_ay[min(ia, n_ty - 1)], s_az[min(ib, n_tz - 1)], c0, c1)) continue;
      if (owner) {
        const uint32_t cell = ((uint32_t)k * (uint32_t)n_ty + (uint32_t)ia) * (uint32_t)n_tz + (uint32_t)ib;
        const uint32_t d2 = (uint32_t)((k - c.c_th) * (k - c.c_th) + (ia - c.c_ty) * (ia - c.c_ty) + (ib - c.c_tz) * (ib - c.c_tz));
        const uint32_t ab = ((uint32_t)ia << 16) | (uint32_t)ib;
        if (better(c0, d2, 2u * cell, best)) { best = Best{c0, d2, 2u * cell}; best_ab = ab; }
        if (better(c1, d2, 2u * cell + 1u, best)) { best = Best{c1, d2, 2u * cell + 1u}; best_ab = ab; }
      }
      float wb = owner ? fminf(c0, c1) : __builtin_inff();
#pragma unroll
      for (int o = ILCC_WAVE / 2; o > 0; o >>= 1) wb = fminf(wb, __shfl_xor(wb, o, ILCC_WAVE));
      if (lane == 0) atomicMin(&s_bound, __float_as_uint(wb));
    }
    reduce_best(best, best_ab);
  }
  // ---- 3. anchor: the 4 x 4 tile around the refinement's argmin at theta - 1, theta, theta + 1 on EVERY labelled point
  {
    // (the refinement evaluates every candidate of the seed's argmin window again, so it always has a pick)
    const int kr = (int)((s_pick[0] >> 1) / (uint32_t)(n_ty * n_tz));
    const int ra = min((int)(s_pick[1] >> 16), n_ty - 1), rb = min((int)(s_pick[1] & 0xFFFFu), n_tz - 1);
    const int a_org = min(max(ra - 1, 0), max(n_ty - kTile, 0)), b_org = min(max(rb - 1, 0), max(n_tz - kTile, 0));
    const int ia = a_org + my_a, ib = b_org + my_b;
    const bool owner = ia < n_ty && ib < n_tz;
    if (wid < kAnchorThetas * kAnchorParts) {
      const int t = wid % kAnchorThetas, half = wid / kAnchorThetas;
      const int k = min(max(kr + t - kAnchorThetas / 2, 0), n_th - 1);
      const float cth = c.cth[k], sth = c.sth[k];
      const float ay = s_ay[min(ia, n_ty - 1)], az = s_az[min(ib, n_tz - 1)];
      float A0 = 0.f, A1 = 0.f;
      uint32_t idx = 0, pos = 0, pin = 0;
      // blocks of four walk positions go round the theta's wavefronts
      for (uint32_t at0 = (uint32_t)half * kSlices; at0 < Mfull; at0 += (uint32_t)kAnchorParts * kSlices) {
        const uint32_t at = at0 + (uint32_t)my_s;
        if (at < Mfull) {
          idx = walk_layout ? at : (uint32_t)(((uint64_t)at * S) % Mfull);
          const float2 v = wyz[idx];
          const PointTerms pt{fmaf(-sth, v.y, cth * v.x), fmaf(cth, v.y, sth * v.x), wlab[idx] ? 0.5f : 0.f};
          if (at < Mi_all)
            accumulate_interior(pt, ay, az, delta2, A0, A1);
          else
            accumulate<true>(pt, ay, az, Wh, Hh, delta2, A0, A1);
        }
        pos += kSlices;
        pin += at0 < Mi_all ? (uint32_t)kSlices : 0u;
      }
      pts_done += pos;
      pts_in += pin;
      const float t0s = quad_sum(A0), t1s = quad_sum(A1);
      if (my_s == 0) {
        s_part[t][half][my_c][0] = t0s;
        s_part[t][half][my_c][1] = t1s;
      }
    }
    __syncthreads();
    if (wid == 0) {
      Best best{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu};
      uint32_t best_ab = 0;
      if (lane < 16 * kAnchorThetas) {
        const int t = lane >> 4, cand = lane & 15, ca = cand >> 2, cb = cand & 3;
        const int k = min(max(kr + t - kAnchorThetas / 2, 0), n_th - 1), ja = a_org + ca, jb = b_org + cb;
        if (ja < n_ty && jb < n_tz) {
          float h0 = 0.f, h1 = 0.f;
#pragma unroll
          for (int q = 0; q < kAnchorParts; ++q) {
            h0 += s_part[t][q][cand][0];
            h1 += s_part[t][q][cand][1];
          }
          const float c0 = 2.f * h0, c1 = 2.f * h1;
          const uint32_t cell = ((uint32_t)k * (uint32_t)n_ty + (uint32_t)ja) * (uint32_t)n_tz + (uint32_t)jb;
          const uint32_t d2 = (uint32_t)((k - c.c_th) * (k - c.c_th) + (ja - c.c_ty) * (ja - c.c_ty) + (jb - c.c_tz) * (jb - c.c_tz));
          best = Best{c0, d2, 2u * cell};
          if (better(c1, d2, 2u * cell + 1u, best)) best = Best{c1, d2, 2u * cell + 1u};
          best_ab = ((uint32_t)ja << 16) | (uint32_t)jb;
        }
      }
#pragma unroll
      for (int o = ILCC_WAVE / 2; o > 0; o >>= 1) {
        Best tt;
        tt.cost = __shfl_down(best.cost, o, ILCC_WAVE);
        tt.d2 = __shfl_down(best.d2, o, ILCC_WAVE);
        tt.flat = __shfl_down(best.flat, o, ILCC_WAVE);
        const uint32_t tab = __shfl_down(best_ab, o, ILCC_WAVE);
        if (better(tt.cost, tt.d2, tt.flat, best)) {
          best = tt;
          best_ab = tab;
        }
      }
      if (lane == 0) {
        *out = GridPartial{best.cost, best.d2, best.flat, best_ab};
        if (best.flat != 0xFFFFFFFFu) atomicMin(c.grid_bound + f, __float_as_uint(best.cost));   // the frame's real bound: a complete candidate on every point
      }
    }
    (void)owner;
  }
  if (lane == 0) {
    s_iters[wid] = pts_done;
    s_iters_in[wid] = pts_in;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t it_sum = 0, in_sum = 0;
    for (int w = 0; w < kLocateWaves; ++w) {
      it_sum += s_iters[w];
      in_sum += s_iters_in[w];
    }
    atomicAdd(c.grid_iters + (f & (kIterSlots - 1)), (unsigned long long)it_sum);
    atomicAdd(c.grid_iters + kIterSlots + (f & (kIterSlots - 1)), (unsigned long long)in_sum);
  }
}

// k6_anchor (round 5): one ANCHOR round for batches that keep the separate locate launches (fewer than kLocateMinFrames frames:
// BASELINE config 5's 64-frame batches).  Workgroup = (theta - 1 | theta | theta + 1 around the previous launch's argmin, frame);
// its 8 wavefronts split the walk over ALL labelled points of ONE 4 x 4 tile of translations around that argmin (blocks of
// four walk positions go round the wavefronts, partial sums meet in LDS), where the anchor instance of k6_grid_cost walked the
// tile with one wavefront of eight: 0.16 ms per round on config 5, latency of a serial walk over 4 400 points.  At 0.03 ms a
// round can be repeated: every round re-centres the tile on the previous round's argmin -- a greedy descent on complete costs
// towards the grid minimum.  On a fine grid (config 5: steps of 1.6 mm and 0.25 degrees) the refinement's argmin, found on an
// eighth of the points, is several steps from it; a second round makes the bound 20 % tighter in executed work (225 -> 180 M
// point-candidate evaluations per 64 frames in the full pass, 224 -> 189 M in the pre-passes).
constexpr int kAnchorThreads = 512;
constexpr int kAnchorWaves = kAnchorThreads / ILCC_WAVE;
__global__ __launch_bounds__(kAnchorThreads) void k6_anchor(Ctx c) {
  __shared__ float s_part[kAnchorWaves][16][2];
  __shared__ uint32_t s_iters[kAnchorWaves], s_iters_in[kAnchorWaves];
  const uint32_t f = blockIdx.y, kblk = blockIdx.x;
  GridPartial* out = &c.partial[(uint64_t)f * c.grid_blocks + kblk];
  const int lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(wave_id());
  uint32_t k2u = 0, ab = 0;
  Best sb{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu};
  if (c.res[f].status == ILCC_OK) sb = seed_argmin(c, f, k2u, ab);
  if (sb.flat == 0xFFFFFFFFu) {
    if (threadIdx.x == 0) *out = GridPartial{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu, 0u};
    return;
  }
  const int n_ty = c.p.n_ty, n_tz = c.p.n_tz, n_th = c.p.n_th;
  const int kr = (int)((sb.flat >> 1) / (uint32_t)(n_ty * n_tz));   // (the records come from launches over the full tables)
  const int k = min(max(kr + (int)kblk - c.refine_radius_th, 0), n_th - 1);
  const int ra = min((int)(ab >> 16), n_ty - 1), rb = min((int)(ab & 0xFFFFu), n_tz - 1);
  const int a_org = min(max(ra - 1, 0), max(n_ty - kTile, 0)), b_org = min(max(rb - 1, 0), max(n_tz - kTile, 0));
  const int my_s = lane & (kSlices - 1), my_c = lane >> 2, my_a = my_c >> 2, my_b = my_c & 3;
  const int ia = a_org + my_a, ib = b_org + my_b;
  const uint32_t Mfull = c.n_lab[f];
  const uint64_t beg = c.off[f];
  const bool walk_layout = Mfull <= (uint32_t)kGridLdsPointsMax;
  const float2* __restrict__ wyz = walk_layout ? c.walk_yz + beg : c.yz + beg;
  const uint8_t* __restrict__ wlab = walk_layout ? c.walk_lab + beg : c.lab + beg;
  const uint32_t S = (!walk_layout && Mfull) ? c.walk_stride[f] : 1u;
  const uint32_t Mi_all = walk_layout ? c.walk_mi[f] : 0u;
  const float Wh = 0.5f * (float)c.p.board_w, Hh = 0.5f * (float)c.p.board_h, delta2 = (float)c.p.huber_delta;
  const float cth = c.cth[k], sth = c.sth[k];
  const float ay = c.ay[min(ia, n_ty - 1)], az = c.az[min(ib, n_tz - 1)];
  float A0 = 0.f, A1 = 0.f;
  uint32_t pos = 0, pin = 0;
  for (uint32_t at0 = (uint32_t)wid * kSlices; at0 < Mfull; at0 += (uint32_t)kAnchorWaves * kSlices) {
    const uint32_t at = at0 + (uint32_t)my_s;
    if (at < Mfull) {
      const uint32_t idx = walk_layout ? at : (uint32_t)(((uint64_t)at * S) % Mfull);
      const float2 v = wyz[idx];
      const PointTerms pt{fmaf(-sth, v.y, cth * v.x), fmaf(cth, v.y, sth * v.x), wlab[idx] ? 0.5f : 0.f};   // = k6_grid_cost's staging
      if (at < Mi_all)
        accumulate_interior(pt, ay, az, delta2, A0, A1);
      else
        accumulate<true>(pt, ay, az, Wh, Hh, delta2, A0, A1);
    }
    pos += kSlices;
    pin += at0 < Mi_all ? (uint32_t)kSlices : 0u;
  }
  const float t0s = quad_sum(A0), t1s = quad_sum(A1);
  if (my_s == 0) {
    s_part[wid][my_c][0] = t0s;
    s_part[wid][my_c][1] = t1s;
  }
  if (lane == 0) {
    s_iters[wid] = pos;
    s_iters_in[wid] = pin;
  }
  __syncthreads();
  if (wid == 0) {
    Best best{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t best_ab = 0;
    if (lane < 16) {
      const int ja = a_org + (lane >> 2), jb = b_org + (lane & 3);
      if (ja < n_ty && jb < n_tz) {
        float h0 = 0.f, h1 = 0.f;
#pragma unroll
        for (int q = 0; q < kAnchorWaves; ++q) {
          h0 += s_part[q][lane][0];
          h1 += s_part[q][lane][1];
        }
        const float c0 = 2.f * h0, c1 = 2.f * h1;
        const uint32_t cell = ((uint32_t)k * (uint32_t)n_ty + (uint32_t)ja) * (uint32_t)n_tz + (uint32_t)jb;
        const uint32_t d2 = (uint32_t)((k - c.c_th) * (k - c.c_th) + (ja - c.c_ty) * (ja - c.c_ty) + (jb - c.c_tz) * (jb - c.c_tz));
        best = Best{c0, d2, 2u * cell};
        if (better(c1, d2, 2u * cell + 1u, best)) best = Best{c1, d2, 2u * cell + 1u};
        best_ab = ((uint32_t)ja << 16) | (uint32_t)jb;
      }
    }
#pragma unroll
    for (int o = ILCC_WAVE / 2; o > 0; o >>= 1) {
      Best tt;
      tt.cost = __shfl_down(best.cost, o, ILCC_WAVE);
      tt.d2 = __shfl_down(best.d2, o, ILCC_WAVE);
      tt.flat = __shfl_down(best.flat, o, ILCC_WAVE);
      const uint32_t tab = __shfl_down(best_ab, o, ILCC_WAVE);
      if (better(tt.cost, tt.d2, tt.flat, best)) {
        best = tt;
        best_ab = tab;
      }
    }
    if (lane == 0) {
      *out = GridPartial{best.cost, best.d2, best.flat, best_ab};
      if (best.flat != 0xFFFFFFFFu) atomicMin(c.grid_bound + f, __float_as_uint(best.cost));   // a complete candidate on every point: a valid bound
      uint32_t it_sum = 0, in_sum = 0;
      for (int w = 0; w < kAnchorWaves; ++w) {
        it_sum += s_iters[w];
        in_sum += s_iters_in[w];
      }
      atomicAdd(c.grid_iters + (f & (kIterSlots - 1)), (unsigned long long)it_sum);
      atomicAdd(c.grid_iters + kIterSlots + (f & (kIterSlots - 1)), (unsigned long long)in_sum);
    }
  }
}

void launch_anchor(const Ctx& c, hipStream_t s) {
  hipLaunchKernelGGL(k6_anchor, dim3(c.grid_blocks, c.n_frames), dim3(kAnchorThreads), 0, s, c);
}

size_t locate_lds_bytes(uint32_t sample_cap, int n_ty, int n_tz, int n_ty2, int n_tz2) {
  return (sizeof(float2) + sizeof(float)) * (size_t)sample_cap + sizeof(float2) * (size_t)kLocateThetasMax * sample_cap +
         sizeof(float) * (size_t)(n_ty + n_tz + n_ty2 + n_tz2);
}

void launch_locate(const Ctx& c, hipStream_t s, const LocatePlan& lp) {
  const size_t lds = locate_lds_bytes(lp.sample_cap, c.p.n_ty, c.p.n_tz, lp.n_ty2, lp.n_tz2);
  hipLaunchKernelGGL(k6_locate, dim3(c.n_frames), dim3(kLocateThreads), lds, s, c, lp);
}

// K5w walk order: the frame's labelled points in the layout k6_grid_cost stages -- [interior | rim | other border], each part
// in golden-ratio walk order -- written ONCE per frame (round 3 until here: every one of a frame's ~80 K6 workgroups
// classified and partitioned the points again, 20 % of the full pass's VALU instructions).
//   interior: in the board under every rotation of the theta table and every translation of the (ty, tz) tables -- decided
//             by a bound, not by trying the 61 rotations (that loop was 10 M of the path's 386 M instructions per 512 frames
//             and found 7 % more points): |i| <= |y| max|cos| + |z| max|sin|, with a margin far above fp32 rounding; the
//             decimated tables of the seed launch are subsets of the full ones -> accumulate_interior is exact for these
//             points in every launch;
//   rim:      border-class and within kRimMilli thousandths of a square of the outline at the grid's centre candidate:
//             walked first, they are the points that leave the board when the translation is wrong (ordering only).
constexpr int kRimMilli = 300;   // rim = within 0.3 square of the outline
// 256 threads: in the pipeline the kernel runs beside another batch's full pass, and a 1024-thread workgroup needs 16 free wave
// slots on ONE CU while that pass keeps refilling them -- its span on the batch's stream was 0.65 ms against 0.04 ms alone, the
// largest item of a batch's front end (tools/dev_depth_timeline.py).  1024 -> 256 threads: bench 1105 -> 1157 k frames/s
constexpr int kWalkThreads = 256;
__global__ __launch_bounds__(kWalkThreads) void k5w_walk_order(Ctx c) {
  __shared__ uint8_t s_cls[kGridLdsPointsMax];
  __shared__ uint32_t s_in[kWalkThreads / ILCC_WAVE], s_rim[kWalkThreads / ILCC_WAVE], s_oth[kWalkThreads / ILCC_WAVE];
  const uint32_t f = blockIdx.x;
  if (c.res[f].status != ILCC_OK) return;
  const uint32_t M = c.n_lab[f];
  if (M == 0u || M > (uint32_t)kGridLdsPointsMax) {   // K6 walks such a frame through global memory in golden-ratio order
    if (threadIdx.x == 0) {
      c.walk_mi[f] = 0u;
      c.walk_nrim[f] = 0u;
    }
    return;
  }
  const int lane = lane_id();
  const int wid = wave_id();
  const uint64_t beg = c.off[f];
  const float2* __restrict__ gyz = c.yz + beg;
  const uint8_t* __restrict__ glab = c.lab + beg;
  const uint32_t S = c.walk_stride[f];
  const float Wh = 0.5f * (float)c.p.board_w, Hh = 0.5f * (float)c.p.board_h;
  const float ay_lo = c.ay[0], ay_hi = c.ay[c.p.n_ty - 1], az_lo = c.az[0], az_hi = c.az[c.p.n_tz - 1];
  const float ay_c = c.ay[c.c_ty], az_c = c.az[c.c_tz];
  const float rim_thr = -(float)kRimMilli * 1e-3f;
  const int n_th = c.p.n_th;
  float cmax = 0.f, smax = 0.f;   // (uniform: the tables are a few dozen values)
  for (int k = 0; k < n_th; ++k) {
    cmax = fmaxf(cmax, fabsf(c.cth[k]));
    smax = fmaxf(smax, fabsf(c.sth[k]));
  }
  const float room_i = fminf(fminf(ay_lo, ay_hi), 2.f * Wh - fmaxf(ay_lo, ay_hi)), room_j = fminf(fminf(az_lo, az_hi), 2.f * Hh - fmaxf(az_lo, az_hi));
  // class of every walk slot: 0 interior, 1 other border, 2 rim
  uint32_t cnt_in = 0, cnt_rim = 0;
  for (uint32_t sl = threadIdx.x; sl < M; sl += kWalkThreads) {
    const float2 v = gyz[(uint32_t)(((uint64_t)sl * S) % M)];
    // |i| <= |y| max|cos/g| + |z| max|sin/g| for every theta of the table (and likewise |j|): inside the room the translations
    // leave on both axes, with 1e-4 square to spare, the point is in the board for every candidate -- fp32 rounding of the
    // term's own expressions (a few 1e-7 on coordinates of a few squares) included
    const float bi = fmaf(fabsf(v.y), smax, fabsf(v.x) * cmax), bj = fmaf(fabsf(v.y), cmax, fabsf(v.x) * smax);
    const bool inside_always = bi + 1e-4f < room_i && bj + 1e-4f < room_j;
    int cl = 0;
    if (!inside_always) {
      const float cth = c.cth[c.c_th], sth = c.sth[c.c_th];
      const float pi = fmaf(-sth, v.y, cth * v.x), pj = fmaf(cth, v.y, sth * v.x);
      const float uc = fabsf((pi + ay_c) - Wh) - Wh, wc = fabsf((pj + az_c) - Hh) - Hh;
      cl = (fmaxf(uc, wc) > rim_thr) ? 2 : 1;
    }
    s_cls[sl] = (uint8_t)cl;
    cnt_in += cl == 0;
    cnt_rim += cl == 2;
  }
  cnt_in = wave_sum(cnt_in);
  cnt_rim = wave_sum(cnt_rim);
  if (lane == 0) {
    s_in[wid] = cnt_in;
    s_rim[wid] = cnt_rim;
  }
  __syncthreads();
  uint32_t Mi = 0, n_rim = 0;
  for (int w = 0; w < kWalkThreads / ILCC_WAVE; ++w) {
    Mi += s_in[w];
    n_rim += s_rim[w];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    c.walk_mi[f] = Mi;
    c.walk_nrim[f] = n_rim;
  }
  // stable partition, chunk by chunk (ballot ranks inside a wavefront, counts of the wavefronts in LDS): deterministic
  float2* __restrict__ wyz = c.walk_yz + beg;
  uint8_t* __restrict__ wlab = c.walk_lab + beg;
  uint32_t base_in = 0, base_rim = Mi, base_oth = Mi + n_rim;
  for (uint32_t c0 = 0; c0 < M; c0 += kWalkThreads) {
    const uint32_t sl = c0 + threadIdx.x;
    const int cl = sl < M ? (int)s_cls[sl] : -1;
    const unsigned long long m_in = __ballot(cl == 0), m_oth = __ballot(cl == 1), m_rim = __ballot(cl == 2);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) {
      s_in[wid] = (uint32_t)__popcll(m_in);
      s_rim[wid] = (uint32_t)__popcll(m_rim);
      s_oth[wid] = (uint32_t)__popcll(m_oth);
    }
    __syncthreads();
    uint32_t pre_in = 0, pre_rim = 0, pre_oth = 0, tot_in = 0, tot_rim = 0, tot_oth = 0;
    for (int w = 0; w < kWalkThreads / ILCC_WAVE; ++w) {
      const uint32_t a = s_in[w], r = s_rim[w], o = s_oth[w];
      if (w < wid) {
        pre_in += a;
        pre_rim += r;
        pre_oth += o;
      }
      tot_in += a;
      tot_rim += r;
      tot_oth += o;
    }
    if (cl >= 0) {
      const uint32_t at = cl == 0   ? base_in + pre_in + (uint32_t)__popcll(m_in & below)
                          : cl == 2 ? base_rim + pre_rim + (uint32_t)__popcll(m_rim & below)
                                    : base_oth + pre_oth + (uint32_t)__popcll(m_oth & below);
      const uint32_t i = (uint32_t)(((uint64_t)sl * S) % M);
      wyz[at] = gyz[i];
      wlab[at] = glab[i];
    }
    base_in += tot_in;
    base_rim += tot_rim;
    base_oth += tot_oth;
    __syncthreads();
  }
}

void launch_walk_order(const Ctx& c, hipStream_t s) {
  hipLaunchKernelGGL(k5w_walk_order, dim3(c.n_frames), dim3(kWalkThreads), 0, s, c);
}

// -DILCC_K6_ISA_PROBE (tools/k6_isa_count.sh): the two terms alone, N chained calls on N different points per kernel.  The
// VALU instructions of ONE evaluation = (instructions of the N = 3 kernel - instructions of the N = 1 kernel) / 2 -- counted
// by the script in the gfx950 assembly of THIS file with the library's own compiler flags, so bench.py's credit per executed
// evaluation is regenerated, not asserted.
#ifdef ILCC_K6_ISA_PROBE
template <int N, bool BORDER>
__global__ void k6_isa_probe(const float* __restrict__ pts, float ay, float az, float Wh, float Hh, float delta, float* out) {
  float A0 = 0.f, A1 = 0.f;
  const float lay = ay + (float)threadIdx.x, laz = az - (float)threadIdx.x;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const PointTerms p{pts[3 * k], pts[3 * k + 1], pts[3 * k + 2]};
    if (BORDER)
      accumulate<true>(p, lay, laz, Wh, Hh, delta, A0, A1);
    else
      accumulate_interior(p, lay, laz, delta, A0, A1);
  }
  out[2 * threadIdx.x] = A0;
  out[2 * threadIdx.x + 1] = A1;
}
template <int N>
__global__ void k6_isa_probe_box(const float* __restrict__ pts, float alo, float ahi, float zlo, float zhi, float Wh, float Hh, float delta,
                                 float* out) {
  float lb = 0.f;
  const float t = (float)threadIdx.x;
#pragma unroll
  for (int k = 0; k < N; ++k) box_term(pts[2 * k], pts[2 * k], pts[2 * k + 1], pts[2 * k + 1], alo + t, ahi + t, zlo - t, zhi - t, Wh, Hh, delta, lb);
  out[threadIdx.x] = lb;
}
template __global__ void k6_isa_probe_box<1>(const float*, float, float, float, float, float, float, float, float*);
template __global__ void k6_isa_probe_box<3>(const float*, float, float, float, float, float, float, float, float*);
template __global__ void k6_isa_probe<1, true>(const float*, float, float, float, float, float, float*);
template __global__ void k6_isa_probe<3, true>(const float*, float, float, float, float, float, float*);
template __global__ void k6_isa_probe<1, false>(const float*, float, float, float, float, float, float*);
template __global__ void k6_isa_probe<3, false>(const float*, float, float, float, float, float, float*);
#endif

#ifdef ILCC_K6_TIMING
extern "C" int ilcc_debug_k6_profile(unsigned long long* out16, int clear) {
  static std::vector<unsigned long long> host((size_t)kProfWaves * kProfWords);
  if (hipMemcpyFromSymbol(host.data(), HIP_SYMBOL(k6_prof), host.size() * sizeof(unsigned long long)) != hipSuccess) return -1;
  for (int k = 0; k < 16; ++k) out16[k] = 0;
  for (size_t w = 0; w < (size_t)kProfWaves; ++w)
    for (int k = 0; k < kProfWords; ++k) out16[k] += host[w * kProfWords + k];
  if (clear) {
    std::fill(host.begin(), host.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(k6_prof), host.data(), host.size() * sizeof(unsigned long long)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// executed-work unit of Ctx::grid_iters: one count = one point x one 16-candidate tile
uint32_t grid_cost_evals_per_count() { return kTile * kTile; }

// allow > 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU): the staged points plus the (ty, tz) tables, whose
// combined length params_ok bounds by kGridTableMax.  Called by ilcc_create for the handle's device.
hipError_t set_kernel_attributes_k6() {
  const void* fns[] = {(const void*)k6_grid_cost<true, true, false, kGridThreads>,  (const void*)k6_grid_cost<true, false, false, kGridThreads>,
                       (const void*)k6_grid_cost<false, true, false, kGridThreads>, (const void*)k6_grid_cost<false, false, false, kGridThreads>,
                       (const void*)k6_grid_cost<true, false, true, kGridThreads>,  (const void*)k6_grid_cost<false, false, true, kGridThreads>,
                       (const void*)k6_grid_cost<true, false, true, kGridThreadsLarge>,
                       (const void*)k6_group_prepass<kGridThreads>, (const void*)k6_group_prepass<kGridThreadsLarge>, (const void*)k6_locate};
  const int cap = (int)((sizeof(float2) + sizeof(float)) * (size_t)kGridLdsPointsMax + sizeof(float) * (size_t)kGridTableMax);
  for (const void* fn : fns) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

void launch_grid_cost(const Ctx& c, hipStream_t s, int32_t use_oob, float* cost_volume, bool prune) {
  const dim3 grid(c.grid_blocks, c.n_frames), block(kGridThreads);
  const size_t lds = (sizeof(float2) + sizeof(float)) * (size_t)c.grid_lds_points + sizeof(float) * (size_t)(c.p.n_ty + c.p.n_tz);
  // the diagnostic volume is always a complete evaluation (no pruning)
  if (cost_volume) {
    if (use_oob)
      hipLaunchKernelGGL((k6_grid_cost<true, true, false, kGridThreads>), grid, block, lds, s, c, cost_volume);
    else
      hipLaunchKernelGGL((k6_grid_cost<false, true, false, kGridThreads>), grid, block, lds, s, c, cost_volume);
  } else if (prune) {
    if (use_oob && c.grid_lds_points > (uint32_t)kGridLargeFrom)   // the pipeline's launches on large frames
      hipLaunchKernelGGL((k6_grid_cost<true, false, true, kGridThreadsLarge>), grid, dim3(kGridThreadsLarge), lds, s, c, cost_volume);
    else if (use_oob)
      hipLaunchKernelGGL((k6_grid_cost<true, false, true, kGridThreads>), grid, block, lds, s, c, cost_volume);
    else
      hipLaunchKernelGGL((k6_grid_cost<false, false, true, kGridThreads>), grid, block, lds, s, c, cost_volume);
  } else {
    if (use_oob)
      hipLaunchKernelGGL((k6_grid_cost<true, false, false, kGridThreads>), grid, block, lds, s, c, cost_volume);
    else
      hipLaunchKernelGGL((k6_grid_cost<false, false, false, kGridThreads>), grid, block, lds, s, c, cost_volume);
  }
}

}  // namespace ilcc
