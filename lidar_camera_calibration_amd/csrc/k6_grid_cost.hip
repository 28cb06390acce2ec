// K6 grid_cost -- the hot kernel.  Evaluates the ILCC intensity-grid objective of
// Optimization::get_theta_t (/root/reference/ilcc2/src/Optimization.cpp:94-160), i.e.
//     cost(theta,ty,tz,phase) = 1/2 * sum_k HuberLoss(0.1)( r_k^2 ),
// r_k = VirtualboardError::operator() (/root/reference/ilcc2/include/ilcc2/Optimization.h:31-107)
// for EVERY candidate of an exhaustive (theta, ty, tz) x colour-phase grid (the reference only
// walks this surface locally with Ceres from (0,0,0)).
//
// Mapping (gfx950): workgroup = (frame, theta index), 4 wavefronts.  The frame's labelled
// points (y, z in the plane frame + black/white label) are staged ONCE into LDS per workgroup.
// One wavefront owns a tile of kTileA x kTileB = 4 x 8 (ty, tz) candidates at the workgroup's theta:
// its 64 lanes stride over the points, each lane keeps 2 x 16 partial sums (both colour phases
// of the 16 candidates), and the sums are reduced across the wavefront with shuffles once per
// tile.  Sharing theta inside a tile means the rotation is done once per point, and the
// i-dependent terms (nearest-edge distance, cell parity, out-of-board distance) are computed
// once per ty and the j-dependent ones once per tz: ~10 VALU ops per (point, candidate) for
// both phases instead of ~30 for a candidate-at-a-time evaluation.  No MFMA: there is no
// dense contraction here.  The cost volume is never written (in-kernel argmin) unless the
// diagnostic entry asks for it.
//
// Per point and candidate, with i = (y' + ty + W g/2)/g, j likewise (Optimization.h:45-46):
//   in board (0<i<W, 0<j<H):  r = dist(i, nearest integer) + dist(j, nearest integer) when the
//                             cell colour differs from the point's label, else 0   (:50-83)
//   out of board:             r = min(|i|,|i-W|) + min(|j|,|j-H|) if useOutofBoard    (:85-104)
//   1/2 rho(r^2) = q (r - q/2),  q = min(r, delta)                     (HuberLoss(0.1), :137)
// The cell is white iff topleftWhite xor ((floor i + floor j) odd)  (:53-61), so a mismatch
// under phase 0 is a match under phase 1: both phases come out of one pass.
#include "ilcc_internal.h"

namespace ilcc {

typedef unsigned long long lanemask_t;
__device__ __forceinline__ lanemask_t ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool unballot(lanemask_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

#ifndef ILCC_K6_FIRST_CHECK_DIV
#define ILCC_K6_FIRST_CHECK_DIV 16
#endif
constexpr int kFirstCheckDiv = ILCC_K6_FIRST_CHECK_DIV;   // first bound check after n_iter / this many iterations
constexpr int kAcc = kTileA * kTileB * 2;   // partial sums per lane

// ---- transposed wavefront reduction of the 32 partial sums, in registers only ----------------
// kAcc (32 or 64) accumulators x 64 lanes -> every lane ends with one accumulator's total in acc[0].
// Every step halves the values per lane and doubles the lanes summed; partners are reached with
// v_permlane32_swap / v_permlane16_swap (gfx950) and DPP row rotations / quad permutes -- plain
// VALU instructions, no LDS round trips (a ds_bpermute chain costs several point-iterations of
// latency, which matters because the branch-and-bound checks reduce after every few iterations).
static_assert(kAcc == 32 || kAcc == 64, "4 x 4 or 4 x 8 candidate tiles");

// [aL+aH | bL+bH] over the two 32-lane halves
__device__ __forceinline__ float swap32_add(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// per 32-lane half: [a.row0+a.row1 | b.row0+b.row1] over its two 16-lane rows
__device__ __forceinline__ float swap16_add(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_xor4(float v) {   // lane i <- lane i^4 inside each 16-lane row
  // row_ror:n hands lane i the value of lane (i - n) mod 16: banks 1,3 take i-4, banks 0,2 take i+4
  const unsigned lo = __builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x124 /*row_ror:4*/, 0xf, 0xa, false);
  return __uint_as_float(__builtin_amdgcn_update_dpp(lo, __float_as_uint(v), 0x12C /*row_ror:12*/, 0xf, 0x5, false));
}
// v + v[lane ^ (1 << BIT)] in every lane
template <int BIT>
__device__ __forceinline__ float partner_add(float v) {
  if constexpr (BIT == 5) return swap32_add(v, v);
  else if constexpr (BIT == 4) return swap16_add(v, v);
  else if constexpr (BIT == 3) return v + dpp<0x128>(v);   // row_ror:8
  else if constexpr (BIT == 2) return v + dpp_xor4(v);
  else if constexpr (BIT == 1) return v + dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  else return v + dpp<0xB1>(v);                            // quad_perm [1,0,3,2]
}
// lanes with bit BIT clear end with lo + lo[partner], the others with hi + hi[partner]
template <int BIT>
__device__ __forceinline__ float halve(float lo, float hi, bool up) {
  if constexpr (BIT == 5) return swap32_add(lo, hi);
  else if constexpr (BIT == 4) return swap16_add(lo, hi);
  else {
    const float s0 = partner_add<BIT>(lo), s1 = partner_add<BIT>(hi);
    return up ? s1 : s0;
  }
}
// N values per lane, lane bits BIT..0 still to be summed over
template <int N, int BIT>
__device__ __forceinline__ void tsum(float (&acc)[kAcc], int lane) {
  if constexpr (N > 1) {
    const bool up = (lane & (1 << BIT)) != 0;
#pragma unroll
    for (int k = 0; k < N / 2; ++k) acc[k] = halve<BIT>(acc[k], acc[k + N / 2], up);
    if constexpr (BIT > 0) tsum<N / 2, BIT - 1>(acc, lane);
  } else {
    acc[0] = partner_add<BIT>(acc[0]);
    if constexpr (BIT > 0) tsum<1, BIT - 1>(acc, lane);
  }
}
// kAcc accumulators x 64 lanes -> acc[0] of lane l = total of accumulator l (kAcc = 64) or l >> 1 (kAcc = 32)
__device__ __forceinline__ void transposed_sum(float (&acc)[kAcc], int lane) { tsum<kAcc, 5>(acc, lane); }

struct Best {
  float cost;
  uint32_t d2;
  uint32_t flat;
};
__device__ __forceinline__ bool better(float c, uint32_t d2, uint32_t flat, const Best& b) {
  return c < b.cost || (c == b.cost && (d2 < b.d2 || (d2 == b.d2 && flat < b.flat)));
}

template <bool OOB, bool VOLUME, bool LDS_POINTS, bool PRUNE>
__device__ __forceinline__ void grid_cost_body(const Ctx& c, float* volume, float2* s_pts,
                                               uint8_t* s_lab, Best* s_best, uint32_t* s_iters, float* s_ay,
                                               float* s_az) {
  const uint32_t f = blockIdx.y;
  const uint32_t k = blockIdx.x;   // theta index
  const ilcc_result* r = &c.res[f];
  const int lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(wave_id());
  GridPartial* out = &c.partial[(uint64_t)f * c.grid_blocks + blockIdx.x];
  if (r->status != ILCC_OK) {
    if (threadIdx.x == 0) {
      out->cost = __builtin_inff();
      out->d2 = 0xFFFFFFFFu;
      out->flat = 0xFFFFFFFFu;
    }
    return;
  }
  const uint32_t M = c.n_lab[f];
  const uint64_t beg = c.off[f];
  const float2* __restrict__ gyz = c.yz + beg;
  const uint8_t* __restrict__ glab = c.lab + beg;
  const uint32_t Mpad = (M + ILCC_WAVE - 1) & ~(uint32_t)(ILCC_WAVE - 1);

  // Point order: trip `it` of a wavefront takes the points {lane * n_iter + it}, i.e. every trip is a
  // sample spread over the whole stream (ring order would hand a trip 64 neighbours on one ring,
  // which says little about a candidate; a spread sample lets the first bound check cut more tiles).
  const uint32_t n_iter_pts = Mpad / ILCC_WAVE;
  if (LDS_POINTS) {
    // stage once per workgroup, already in trip order: slot it*64 + lane <- point lane*n_iter + it
    for (uint32_t sl = threadIdx.x; sl < Mpad; sl += kGridThreads) {
      const uint32_t i = (sl & (ILCC_WAVE - 1)) * n_iter_pts + (sl >> 6);
      float2 v = make_float2(0.f, 0.f);
      uint8_t l = 0;
      if (i < M) {
        v = gyz[i];
        l = glab[i];
      }
      s_pts[sl] = v;
      s_lab[sl] = l;
    }
    __syncthreads();
  }

  // (ty, tz) tables in LDS: a cut-short tile lasts about as long as one L2 round trip, so its
  // prologue must not wait for global loads
  for (int i = threadIdx.x; i < c.p.n_ty; i += kGridThreads) s_ay[i] = c.ay[i];
  for (int i = threadIdx.x; i < c.p.n_tz; i += kGridThreads) s_az[i] = c.az[i];
  __syncthreads();
  const float cth = c.cth[k], sth = c.sth[k];
  const float Wh = 0.5f * (float)c.p.board_w, Hh = 0.5f * (float)c.p.board_h;
  const float delta = (float)c.p.huber_delta;
  const int n_ty = c.p.n_ty, n_tz = c.p.n_tz;
  const int nta = (n_ty + kTileA - 1) / kTileA, ntb = (n_tz + kTileB - 1) / kTileB;
  const int n_tiles = nta * ntb;
  const uint32_t dk = (uint32_t)((int)k - c.c_th) * (uint32_t)((int)k - c.c_th);

  Best best{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu};
  uint32_t* bound = c.grid_bound + f;
  float shared_bound = __builtin_inff();   // what this wavefront last published
  const uint32_t n_iter = Mpad / ILCC_WAVE;
  uint32_t iters_done = 0;
  float* vol = VOLUME ? volume + (uint64_t)f * (uint64_t)c.p.n_th * n_ty * n_tz * 2u : nullptr;

  // after the transposed reduction lane l owns accumulator l = (a*kTileB + b)*2 + phase
  const int my_l = (kAcc == ILCC_WAVE) ? lane : lane >> 1;   // the accumulator transposed_sum leaves in this lane
  const int my_ph = my_l & 1, my_b = (my_l >> 1) % kTileB, my_a = (my_l >> 1) / kTileB;

  // start at the tile that holds the seed pass's best translation so that the shared bound is
  // tight after the first round of tiles; the order never changes the result
  int t0 = 0;
  if (PRUNE && c.seed_partial != nullptr) {
    const GridPartial* sp = c.seed_partial + (uint64_t)f * c.seed_blocks;
    Best sb{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (uint32_t q = (uint32_t)lane; q < c.seed_blocks; q += ILCC_WAVE) {
      const GridPartial g = sp[q];
      if (better(g.cost, g.d2, g.flat, sb)) sb = Best{g.cost, g.d2, g.flat};
    }
#pragma unroll
    for (int o = ILCC_WAVE / 2; o > 0; o >>= 1) {
      Best tb;
      tb.cost = __shfl_xor(sb.cost, o, ILCC_WAVE);
      tb.d2 = __shfl_xor(sb.d2, o, ILCC_WAVE);
      tb.flat = __shfl_xor(sb.flat, o, ILCC_WAVE);
      if (better(tb.cost, tb.d2, tb.flat, sb)) sb = tb;
    }
    if (sb.flat != 0xFFFFFFFFu) {
      const uint32_t cell = sb.flat >> 1;
      const int b2 = (int)(cell % (uint32_t)c.seed_n_tz), a2 = (int)((cell / (uint32_t)c.seed_n_tz) % (uint32_t)c.seed_n_ty);
      const int sa = min(a2 * c.seed_stride_t, n_ty - 1), sbb = min(b2 * c.seed_stride_t, n_tz - 1);
      t0 = __builtin_amdgcn_readfirstlane((sa / kTileA) * ntb + (sbb / kTileB));
    }
  }

  // tiles wid, wid+4, ... of the rotated order, tracked as (row, column) so that the per-tile
  // prologue needs no integer division
  int t_first = wid + t0;
  if (t_first >= n_tiles) t_first -= n_tiles;
  int ta = t_first / ntb, tb = t_first - ta * ntb;
  for (int tt = wid; tt < n_tiles; tt += kGridThreads / ILCC_WAVE) {
    const int a0 = ta * kTileA, b0 = tb * kTileB;
    tb += kGridThreads / ILCC_WAVE;            // advance to this wavefront's next tile
    while (tb >= ntb) {
      tb -= ntb;
      ++ta;
    }
    if (ta >= nta) ta -= nta;
    float ayv[kTileA], azv[kTileB];
#pragma unroll
    for (int a = 0; a < kTileA; ++a) ayv[a] = s_ay[min(a0 + a, n_ty - 1)];
#pragma unroll
    for (int b = 0; b < kTileB; ++b) azv[b] = s_az[min(b0 + b, n_tz - 1)];
    float ayh[kTileA], azh[kTileB];
#pragma unroll
    for (int a = 0; a < kTileA; ++a) ayh[a] = 0.5f * ayv[a];
#pragma unroll
    for (int b = 0; b < kTileB; ++b) azh[b] = 0.5f * azv[b];

    float acc[kAcc];
#pragma unroll
    for (int k = 0; k < kAcc; ++k) acc[k] = 0.f;

    // Branch and bound (PRUNE): costs are sums of non-negative terms, so a candidate whose partial
    // sum already exceeds the best COMPLETE cost known for this frame cannot be the argmin.  After
    // 1/8, 1/4 and 1/2 of the points the partial sums are reduced; if every candidate of the tile is
    // beaten the wavefront moves on.  Exact: only provably losing candidates are cut short.
    float done_part = 0.f;           // this lane's candidate: reduced sum of the finished segments
    bool pruned = false;
    const int ia = a0 + my_a, ib = b0 + my_b;
    const bool owner = ia < n_ty && ib < n_tz;   // both lanes of a pair own the same candidate
    uint32_t next_check = PRUNE ? (n_iter >= kFirstCheckDiv ? n_iter / kFirstCheckDiv : 1) : 0xFFFFFFFFu;
    uint32_t it_no = 0;
    // the shared bound is fetched one segment ahead of its use: an L2 round trip is longer than a
    // cut-short tile, and a slightly stale bound only delays a cut
    uint32_t gb_bits = PRUNE ? __hip_atomic_load(bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7f800000u;

    for (uint32_t base = 0; base < Mpad; base += ILCC_WAVE) {
      const uint32_t idx = (uint32_t)lane * n_iter_pts + (base >> 6);   // point of this lane in this trip
      float2 p;
      uint32_t lab;
      if (LDS_POINTS) {
        p = s_pts[base + lane];
        lab = s_lab[base + lane];
      } else {
        p = idx < M ? gyz[idx] : make_float2(0.f, 0.f);
        lab = idx < M ? glab[idx] : 0;
      }
      const float dl = idx < M ? delta : 0.f;   // padded slots: q = min(r,0) = 0 -> no contribution
      // Lane predicates are kept as 64-bit wave masks in SGPR pairs (ballot), combined with SALU
      // ops per candidate and fed straight back to v_cndmask (inverse ballot): the VALU only
      // sees the float work.
      const lanemask_t white = ballot(lab != 0);
      // Rx(theta) on (0,y,z), already divided by g (Optimization.h:37-46)
      const float bi = fmaf(-sth, p.y, cth * p.x);
      const float bj = fmaf(cth, p.y, sth * p.x);
      const float bih = 0.5f * bi, bjh = 0.5f * bj;

      float di[kTileA], ui[kTileA];
      lanemask_t pa[kTileA], oa[kTileA];
#pragma unroll
      for (int a = 0; a < kTileA; ++a) {
        const float i = bi + ayv[a];
        di[a] = i - rintf(i);                       // |.| = distance to the nearest cell border
        // floor(i) odd  <=>  fract(i/2) >= 1/2 ; xor label -> colour mismatch contribution of i
        pa[a] = ballot(__builtin_amdgcn_fractf(bih + ayh[a]) >= 0.5f) ^ white;
        const float tt = i - Wh;
        oa[a] = ballot(!(fabsf(tt) < Wh));          // not (0 < i < W)
        ui[a] = fabsf(tt) - Wh;                     // |.| = min(|i|, |i-W|)
      }
      // tz candidates in groups of four: their masks live in SGPRs only while the group is combined
#pragma unroll
      for (int bg = 0; bg < kTileB; bg += 4) {
        float dj[4], uj[4];
        lanemask_t pb[4], ob[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float j = bj + azv[bg + b];
          dj[b] = j - rintf(j);
          pb[b] = ballot(__builtin_amdgcn_fractf(bjh + azh[bg + b]) >= 0.5f);
          const float tt = j - Hh;
          ob[b] = ballot(!(fabsf(tt) < Hh));
          uj[b] = fabsf(tt) - Hh;
        }
#pragma unroll
        for (int a = 0; a < kTileA; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const lanemask_t oob = oa[a] | ob[b];
            const lanemask_t mis0 = pa[a] ^ pb[b];   // colour mismatch under phase 0 (topleftWhite=false)
            float rin = fabsf(di[a]) + fabsf(dj[b]);
            float rr;
            if (OOB) {
              float rout = fabsf(ui[a]) + fabsf(uj[b]);
              // keep "select of two sums" (2 full-rate adds + 1 v_cndmask): LLVM would rewrite it into
              // a sum of two selects, and v_cndmask issues at about half the rate of v_add on gfx950
              asm volatile("" : "+v"(rin), "+v"(rout));
              rr = unballot(oob) ? rout : rin;
            } else {
              rr = unballot(oob) ? 0.f : rin;
            }
            const float q = fminf(rr, dl);
            const float h = q * fmaf(-0.5f, q, rr);
            float& x0 = acc[(a * kTileB + bg + b) * 2];
            float& x1 = acc[(a * kTileB + bg + b) * 2 + 1];
            if (OOB) {
              x0 += unballot(oob | mis0) ? h : 0.f;
              x1 += unballot(oob | ~mis0) ? h : 0.f;
            } else {
              x0 += unballot(mis0) ? h : 0.f;
              x1 += unballot(mis0) ? 0.f : h;
            }
          }
        if (kTileB > 4) __builtin_amdgcn_sched_barrier(0);   // keep the groups apart (SGPR pressure)
      }
      ++it_no;
      if (PRUNE && it_no == next_check && it_no < n_iter) {
        transposed_sum(acc, lane);
        done_part += acc[0];
#pragma unroll
        for (int k = 0; k < kAcc; ++k) acc[k] = 0.f;
        const float lim = fminf(__uint_as_float(gb_bits), best.cost);
        gb_bits = __hip_atomic_load(bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next check
        if (!__any(owner && !(done_part > lim))) {
          pruned = true;
          break;
        }
        next_check *= 2;
      }
    }
    iters_done += it_no;
    if (PRUNE && pruned) continue;

    transposed_sum(acc, lane);
    const float total = done_part + acc[0];
    if (owner) {
      const uint32_t cell = ((uint32_t)k * (uint32_t)n_ty + (uint32_t)ia) * (uint32_t)n_tz + (uint32_t)ib;
      const uint32_t d2 = dk + (uint32_t)((ia - c.c_ty) * (ia - c.c_ty)) + (uint32_t)((ib - c.c_tz) * (ib - c.c_tz));
      const uint32_t flat = 2u * cell + (uint32_t)my_ph;
      if (better(total, d2, flat, best)) best = Best{total, d2, flat};
      if (VOLUME) vol[flat] = total;
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
  }

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
    s_iters[wid] = iters_done;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t it_sum = 0;
    for (int w = 0; w < kGridThreads / ILCC_WAVE; ++w) it_sum += s_iters[w];
    atomicAdd(c.grid_iters + (f & (kIterSlots - 1)), (unsigned long long)it_sum);   // spread over 64 words
    Best b = s_best[0];
    for (int w = 1; w < kGridThreads / ILCC_WAVE; ++w)
      if (better(s_best[w].cost, s_best[w].d2, s_best[w].flat, b)) b = s_best[w];
    out->cost = b.cost;
    out->d2 = b.d2;
    out->flat = b.flat;
  }
}

// dynamic LDS: [grid_lds_points float2][grid_lds_points u8]; frames with more labelled points
// than the staged capacity read them through L1/L2 instead (same code, global pointers).
template <bool OOB, bool VOLUME, bool PRUNE>
__global__ __launch_bounds__(kGridThreads) void k6_grid_cost(Ctx c, float* volume) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ Best s_best[kGridThreads / ILCC_WAVE];
  __shared__ uint32_t s_iters[kGridThreads / ILCC_WAVE];
  float2* s_pts = reinterpret_cast<float2*>(smem);
  uint8_t* s_lab = smem + sizeof(float2) * (size_t)c.grid_lds_points;
  float* s_ay = reinterpret_cast<float*>(smem + (sizeof(float2) + 1) * (size_t)c.grid_lds_points);   // n_ty floats
  float* s_az = s_ay + c.p.n_ty;                                                                       // n_tz floats
  const uint32_t M = c.n_lab[blockIdx.y];
  if (M <= c.grid_lds_points)
    grid_cost_body<OOB, VOLUME, true, PRUNE>(c, volume, s_pts, s_lab, s_best, s_iters, s_ay, s_az);
  else
    grid_cost_body<OOB, VOLUME, false, PRUNE>(c, volume, s_pts, s_lab, s_best, s_iters, s_ay, s_az);
}

void launch_grid_cost(const Ctx& c, hipStream_t s, int32_t use_oob, float* cost_volume, bool prune) {
  const dim3 grid(c.grid_blocks, c.n_frames), block(kGridThreads);
  const size_t lds = (sizeof(float2) + 1) * (size_t)c.grid_lds_points + sizeof(float) * (size_t)(c.p.n_ty + c.p.n_tz);
  static bool attr_done = false;
  const void* fns[] = {(const void*)k6_grid_cost<true, true, false>,  (const void*)k6_grid_cost<true, false, false>,
                       (const void*)k6_grid_cost<false, true, false>, (const void*)k6_grid_cost<false, false, false>,
                       (const void*)k6_grid_cost<true, false, true>,  (const void*)k6_grid_cost<false, false, true>};
  if (!attr_done) {   // allow > 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU)
    const int cap = (int)((sizeof(float2) + 1) * (size_t)kGridLdsPointsMax + sizeof(float) * 2 * 1024);
    for (const void* fn : fns) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    attr_done = true;
  }
  // the diagnostic volume is always a complete evaluation (no pruning)
  if (cost_volume) {
    if (use_oob)
      hipLaunchKernelGGL((k6_grid_cost<true, true, false>), grid, block, lds, s, c, cost_volume);
    else
      hipLaunchKernelGGL((k6_grid_cost<false, true, false>), grid, block, lds, s, c, cost_volume);
  } else if (prune) {
    if (use_oob)
      hipLaunchKernelGGL((k6_grid_cost<true, false, true>), grid, block, lds, s, c, cost_volume);
    else
      hipLaunchKernelGGL((k6_grid_cost<false, false, true>), grid, block, lds, s, c, cost_volume);
  } else {
    if (use_oob)
      hipLaunchKernelGGL((k6_grid_cost<true, false, false>), grid, block, lds, s, c, cost_volume);
    else
      hipLaunchKernelGGL((k6_grid_cost<false, false, false>), grid, block, lds, s, c, cost_volume);
  }
}

}  // namespace ilcc
