// One-wavefront rectangular LAP (maximise), step-for-step the algorithm of
// scipy.optimize.linear_sum_assignment (Crouse 2016; restated in oracle/lap.c and pinned there against
// scipy), which the reference reaches through utils/hungarian.py:63.
//
// The reference does one D2H copy + scipy call + H2D copy per graph per Hungarian-stage iteration
// (utils/hungarian.py:34,51); here the matrix never leaves the GPU.  Parallelisation: the 64 lanes scan
// the unscanned-column list `remaining` in strides of 64; the sequential tie rule of the scalar scan
// ("strictly smaller wins; on equality an unassigned column wins, later positions overriding earlier")
// is reproduced exactly from per-lane (value, position, unassigned) triples:
//      gmin = min value;  if some minimal candidate is unassigned -> the LAST such position,
//                         else the FIRST minimal position.
// Duals and path costs are fp64 with scipy's evaluation order ((minVal + c) - u) - v, so near-ties
// resolve identically.  All scratch is wavefront-private LDS; lane 0 performs the scalar updates.
#pragma once
#include "common.h"

struct LapScratch {
  double* u;      // [nr]
  double* v;      // [nc]
  double* spc;    // [nc] shortest path costs
  int* path;      // [nc]
  int* row4col;   // [nc]
  int* col4row;   // [nr]
  int* remaining; // [nc]
  int* SR;        // [nr]
  int* SC;        // [nc]
};

__host__ __device__ inline size_t lap_scratch_bytes(int nr, int nc) {
  return (size_t)(nr + 2 * nc) * 8 + (size_t)(4 * nc + 2 * nr) * 4;
}

__device__ inline LapScratch lap_carve(void* base, int nr, int nc) {
  LapScratch s;
  double* d = (double*)base;
  s.u = d; s.v = d + nr; s.spc = s.v + nc;
  int* i = (int*)(s.spc + nc);
  s.path = i; s.row4col = i + nc; s.remaining = i + 2 * nc; s.SC = i + 3 * nc;
  s.col4row = i + 4 * nc; s.SR = s.col4row + nr;
  return s;
}

// wavefront-local ordering point for LDS traffic between lanes
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// val[i*si + j*sj]: value to MAXIMISE for row i, column j in the oriented frame nr <= nc.
// On return s.col4row[i] holds the column of every row (all rows are assigned).
__device__ __forceinline__ void lap_wave_solve(int nr, int nc, const float* val, int si, int sj, LapScratch s) {
  const int lane = threadIdx.x & 63;
  for (int i = lane; i < nr; i += 64) { s.u[i] = 0.0; s.col4row[i] = -1; }
  for (int j = lane; j < nc; j += 64) { s.v[j] = 0.0; s.row4col[j] = -1; s.path[j] = -1; }
  wave_sync();

  for (int cur = 0; cur < nr; ++cur) {
    double minVal = 0.0;
    int nrem = nc, i = cur, sink = -1;
    for (int it = lane; it < nc; it += 64) { s.remaining[it] = nc - it - 1; s.SC[it] = 0; s.spc[it] = INFINITY; }
    for (int r = lane; r < nr; r += 64) s.SR[r] = 0;
    wave_sync();

    while (sink == -1) {
      if (lane == 0) s.SR[i] = 1;
      const double ui = s.u[i];
      double lowest = INFINITY;
      int index = 0x7fffffff;
      bool un = false;
      for (int it = lane; it < nrem; it += 64) {
        const int j = s.remaining[it];
        const double r = minVal + (-(double)val[i * si + j * sj]) - ui - s.v[j];
        double sp = s.spc[j];
        if (r < sp) { s.path[j] = i; s.spc[j] = r; sp = r; }
        const bool free_col = (s.row4col[j] == -1);
        if (sp < lowest || (sp == lowest && free_col)) { lowest = sp; index = it; un = free_col; }
      }
      const double gmin = wave_min_f64(lowest);
      const bool cand = (lowest == gmin) && (index != 0x7fffffff);
      const bool any_un = __ballot(cand && un) != 0ull;
      int sel;
      if (any_un) sel = wave_max_i32((cand && un) ? index : -1);
      else sel = wave_min_i32(cand ? index : 0x7fffffff);
      minVal = gmin;
      // (finite inputs: gmin is finite, sel is valid)
      const int j = s.remaining[sel];
      const int owner = s.row4col[j];
      wave_sync();
      if (lane == 0) {
        s.SC[j] = 1;
        s.remaining[sel] = s.remaining[nrem - 1];
      }
      --nrem;
      if (owner == -1) sink = j; else i = owner;
      wave_sync();
    }

    // dual updates
    if (lane == 0) s.u[cur] += minVal;
    for (int r = lane; r < nr; r += 64)
      if (s.SR[r] && r != cur) s.u[r] += minVal - s.spc[s.col4row[r]];
    for (int j = lane; j < nc; j += 64)
      if (s.SC[j]) s.v[j] -= minVal - s.spc[j];
    wave_sync();

    // augment along the stored path (scalar walk, at most nr steps; every lane follows, lane 0 writes)
    int j = sink;
    for (;;) {
      const int r = s.path[j];
      const int t = s.col4row[r];
      wave_sync();
      if (lane == 0) { s.row4col[j] = r; s.col4row[r] = j; }
      wave_sync();
      j = t;
      if (r == cur) break;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Register-resident variant for nc <= 64 (every graph of <= 64 nodes against the 32-slot universe).
// Lane j owns column j (dual v_j, shortest-path cost, predecessor, owner row, position in scipy's `remaining`
// list); lane i also owns row i (dual u_i, assigned column).  No LDS traffic besides the cost reads; the scan's
// arg-min runs on DPP row operations instead of ds_bpermute shuffles.  Same steps, same tie rules, same fp64
// evaluation order as lap_wave_solve above (and as oracle/lap.c / scipy).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xf, false); }

#define TTDG_DPP_REDUCE(OP)                                            \
  OP(0xB1, 0xf)  /* quad_perm [1,0,3,2] */                             \
  OP(0x4E, 0xf)  /* quad_perm [2,3,0,1] */                             \
  OP(0x141, 0xf) /* row_half_mirror     */                             \
  OP(0x140, 0xf) /* row_mirror          */                             \
  OP(0x142, 0xa) /* row_bcast15 -> rows 1,3 */                         \
  OP(0x143, 0xc) /* row_bcast31 -> rows 2,3 */

__device__ __forceinline__ double wave_min_f64_dpp(double v) {
#define OP(C, R)                                                                           \
  {                                                                                        \
    const int lo = dpp_mov<C, R>(__double2loint(v)), hi = dpp_mov<C, R>(__double2hiint(v)); \
    v = fmin(v, __hiloint2double(hi, lo));                                                 \
  }
  TTDG_DPP_REDUCE(OP)
#undef OP
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
// 32-lane form (columns 0..31 only, the upper half-wave holds +inf): the row_bcast31 stage is skipped, result in lane 31
__device__ __forceinline__ double wave_min_f64_dpp32(double v) {
#define OP(C, R)                                                                           \
  {                                                                                        \
    const int lo = dpp_mov<C, R>(__double2loint(v)), hi = dpp_mov<C, R>(__double2hiint(v)); \
    v = fmin(v, __hiloint2double(hi, lo));                                                 \
  }
  OP(0xB1, 0xf) OP(0x4E, 0xf) OP(0x141, 0xf) OP(0x140, 0xf) OP(0x142, 0xa)
#undef OP
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 31), hi = __builtin_amdgcn_readlane(__double2hiint(v), 31);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int wave_min_i32_dpp(int v) {
#define OP(C, R) v = min(v, dpp_mov<C, R>(v));
  TTDG_DPP_REDUCE(OP)
#undef OP
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i32_dpp(int v) {
#define OP(C, R) v = max(v, dpp_mov<C, R>(v));
  TTDG_DPP_REDUCE(OP)
#undef OP
  return __builtin_amdgcn_readlane(v, 63);
}
// fp32 wavefront reductions on DPP (result broadcast through lane 63); sum uses 0 as the fill for masked rows
__device__ __forceinline__ float wave_max_f32_dpp(float v) {
#define OP(C, R) v = fmaxf(v, __int_as_float(dpp_mov<C, R>(__float_as_int(v))));
  TTDG_DPP_REDUCE(OP)
#undef OP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum_f32_dpp(float v) {
#define OP(C, R) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), C, R, 0xf, false));
  TTDG_DPP_REDUCE(OP)
#undef OP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_sum_i32_dpp(int v) {
#define OP(C, R) v += __builtin_amdgcn_update_dpp(0, v, C, R, 0xf, false);
  TTDG_DPP_REDUCE(OP)
#undef OP
  return __builtin_amdgcn_readlane(v, 63);
}
// value held by the lane 32 positions away (v_permlane32_swap)
__device__ __forceinline__ float other_half(float v) {
  const int x = __float_as_int(v);
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return __int_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

// ---- butterflies of the block-layout Sinkhorn kernels (gagm.hip, sinkhorn.hip): lane = (bi = lane >> 3, bj = lane & 7);
// sums / maxima over bj (three low lane bits: DPP quad_perm x 2, row_half_mirror) and over bi (three high bits: DPP row_ror:8,
// v_permlane16_swap, v_permlane32_swap); every lane of the group ends up with the group's result
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  const int x = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false));   // every lane has a source: `old` is dead
}
__device__ __forceinline__ float lane_xor16(float v) {
  const int x = __float_as_int(v);
  const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  return __int_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float bj_sum(float v) { v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); return v; }
__device__ __forceinline__ float bj_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v));
  return v;
}
__device__ __forceinline__ float bi_sum(float v) { v += dpp_f<0x128>(v); v += lane_xor16(v); v += other_half(v); return v; }
__device__ __forceinline__ float bi_max(float v) {
  v = fmaxf(v, dpp_f<0x128>(v)); v = fmaxf(v, lane_xor16(v)); v = fmaxf(v, other_half(v));
  return v;
}

// Exact fp64 minimum over the wavefront, hand-scheduled: per stage one `s_nop 1` (VALU-write -> DPP-read hazard),
// two v_mov_b32_dpp (lo/hi words; rows excluded by row_mask keep their own value) and one v_min_f64.  hipcc's own
// lowering of the same reduction spends six instructions per stage (two plain moves and a canonicalising v_max_f64
// on top).  `wide` = more than 32 participating lanes (otherwise the row_bcast31 stage is skipped, result in lane 31).
#define TTDG_F64_MIN_STAGE(CTRL)                                                              \
  {                                                                                           \
    int lo = __double2loint(v), hi = __double2hiint(v);                                       \
    int tlo = lo, thi = hi;                                                                   \
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 " CTRL "\n\tv_mov_b32_dpp %1, %3 " CTRL    \
                 : "+v"(tlo), "+v"(thi) : "v"(lo), "v"(hi));                                  \
    const double o = __hiloint2double(thi, tlo);                                              \
    asm volatile("v_min_f64 %0, %1, %2" : "=v"(v) : "v"(v), "v"(o));                          \
  }
__device__ __forceinline__ double wave_min_f64_fast(double v, bool wide) {
  TTDG_F64_MIN_STAGE("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
  TTDG_F64_MIN_STAGE("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
  TTDG_F64_MIN_STAGE("row_half_mirror row_mask:0xf bank_mask:0xf")
  TTDG_F64_MIN_STAGE("row_mirror row_mask:0xf bank_mask:0xf")
  TTDG_F64_MIN_STAGE("row_bcast:15 row_mask:0xa bank_mask:0xf")
  if (wide) {
    TTDG_F64_MIN_STAGE("row_bcast:31 row_mask:0xc bank_mask:0xf")
  }
  const int src = wide ? 63 : 31;
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// returns this lane's assigned column (valid for lanes < nr)
// kMinImpl: 0 = compiler-lowered fp64 DPP min (default: measured 6 % faster on MI355X, tools/bench_lap.py),
//           1 = hand-scheduled inline-asm stages (kept for A/B runs).
//           (Tried in round 2 and dropped: the minimum over order-preserving 64-bit integer keys as two rounds of
//           single-instruction v_min_u32_dpp stages - 10 DPP instructions instead of ~30 - was 7-9 % SLOWER (41.8 vs 38.9 us
//           per 30x32 LAP): the second round depends on a readlane of the first, which lengthens the per-step chain more
//           than the shorter stages save.  The chain, not the instruction count, bounds this solver.)
// kRegCost: the lane's column of the cost matrix (nr <= 32 rows) is preloaded into 32 registers and read back with a
//           wavefront-uniform dynamic index (s_set_gpr_idx / v_movrel): the cost read leaves the per-step dependency chain
//           (an LDS or L2 round trip per step otherwise)
template <int kMinImpl = 0, bool kRegCost = false, bool kNarrow = false>
__device__ __forceinline__ int lap_wave_solve_reg_impl(int nr, int nc, const float* val, int si, int sj) {
  const int lane = threadIdx.x & 63;
  double u = 0.0, v = 0.0, spc = INFINITY;
  int col4row = -1, row4col = -1, path = -1, pos = 0;
  const bool is_col = lane < nc;
  typedef float f32x32 __attribute__((ext_vector_type(32)));
  f32x32 cost;
  if (kRegCost) {
#pragma unroll
    for (int r = 0; r < 32; ++r) cost[r] = (is_col && r < nr) ? val[r * si + lane * sj] : 0.f;
  }
  for (int cur = 0; cur < nr; ++cur) {
    double minVal = 0.0;
    int nrem = nc, i = cur, sink = -1;
    bool active = is_col, SC = false, SR = false;
    pos = nc - 1 - lane;
    spc = INFINITY;
    while (sink == -1) {
      if (lane == i) SR = true;
      const double ui = readlane_f64(u, i);
      const float ci = kRegCost ? cost[i] : 0.f;       // i is wavefront-uniform
      if (active) {
        const double r = minVal + (-(double)(kRegCost ? ci : val[i * si + lane * sj])) - ui - v;
        if (r < spc) { path = i; spc = r; }
      }
      const double gmin = (kMinImpl == 1) ? wave_min_f64_fast(active ? spc : INFINITY, nc > 32)
                          : kNarrow      ? wave_min_f64_dpp32(active ? spc : INFINITY)
                                         : wave_min_f64_dpp(active ? spc : INFINITY);
      const bool is_min = active && spc == gmin;
      const unsigned long long minmask = __ballot(is_min);
      int jsel;
      if (__builtin_popcountll(minmask) == 1) {
        jsel = __builtin_ctzll(minmask);               // unique minimum: no tie rule needed (the common case)
      } else {
        const bool un = is_min && row4col == -1;
        int selpos;
        if (__ballot(un) != 0ull) selpos = wave_max_i32_dpp(un ? pos : -1);
        else selpos = wave_min_i32_dpp(is_min ? pos : 0x7fffffff);
        jsel = __builtin_ctzll(__ballot(active && pos == selpos));
      }
      const int selpos = __builtin_amdgcn_readlane(pos, jsel);
      minVal = gmin;
      const int owner = __builtin_amdgcn_readlane(row4col, jsel);
      if (lane == jsel) { SC = true; active = false; }
      else if (active && pos == nrem - 1) pos = selpos;
      --nrem;
      if (owner == -1) sink = jsel; else i = owner;
    }
    // dual updates (u of the rows on the alternating tree, v of the scanned columns)
    const int c4r = (col4row >= 0) ? col4row : 0;
    const double spc_of_my_col = __shfl(spc, c4r, 64);
    if (lane == cur) u += minVal;
    else if (SR) u += minVal - spc_of_my_col;
    if (SC) v -= minVal - spc;
    // augment
    int j = sink;
    for (;;) {
      const int r = __builtin_amdgcn_readlane(path, j);
      const int t = __builtin_amdgcn_readlane(col4row, r);
      if (lane == j) row4col = r;
      if (lane == r) col4row = j;
      j = t;
      if (r == cur) break;
    }
  }
  return col4row;
}

// nc <= 32 (every graph of <= 32 nodes: 32 universe columns) runs the five-stage half-wave minimum
template <int kMinImpl = 0, bool kRegCost = false>
__device__ __forceinline__ int lap_wave_solve_reg(int nr, int nc, const float* val, int si, int sj) {
  if (kMinImpl == 0 && nc <= 32) return lap_wave_solve_reg_impl<0, kRegCost, true>(nr, nc, val, si, sj);
  return lap_wave_solve_reg_impl<kMinImpl, kRegCost, false>(nr, nc, val, si, sj);
}

// ---------------------------------------------------------------------------------------------------
// Register-resident variant for 64 < nc <= 64*CW columns (graphs of up to 256 nodes against the 32-slot universe,
// solved transposed: nr = 32 rows, nc = n columns).  Lane l owns the CW columns l, l+64, ...; everything else as in
// lap_wave_solve_reg: same steps, same tie rules on scipy's `remaining` positions, same fp64 evaluation order.
template <int CW>
__device__ __forceinline__ int lap_wave_solve_regw(int nr, int nc, const float* val, int si, int sj) {
  const int lane = threadIdx.x & 63;
  double u = 0.0;
  int col4row = -1;
  double v[CW], spc[CW];
  int row4col[CW], path[CW], pos[CW];
  bool is_col[CW];
#pragma unroll
  for (int w = 0; w < CW; ++w) { v[w] = 0.0; spc[w] = INFINITY; row4col[w] = -1; path[w] = -1; pos[w] = 0; is_col[w] = lane + 64 * w < nc; }
  for (int cur = 0; cur < nr; ++cur) {
    double minVal = 0.0;
    int nrem = nc, i = cur, sink = -1;
    bool active[CW], SC[CW], SR = false;
#pragma unroll
    for (int w = 0; w < CW; ++w) { active[w] = is_col[w]; SC[w] = false; pos[w] = nc - 1 - (lane + 64 * w); spc[w] = INFINITY; }
    while (sink == -1) {
      if (lane == i) SR = true;
      const double ui = readlane_f64(u, i);
      double lmin = INFINITY;
#pragma unroll
      for (int w = 0; w < CW; ++w)
        if (active[w]) {
          const double r = minVal + (-(double)val[i * si + (lane + 64 * w) * sj]) - ui - v[w];
          if (r < spc[w]) { path[w] = i; spc[w] = r; }
          lmin = fmin(lmin, spc[w]);
        }
      const double gmin = wave_min_f64_dpp(lmin);
      bool is_min[CW];
      unsigned long long mm[CW];
      int cnt = 0;
#pragma unroll
      for (int w = 0; w < CW; ++w) { is_min[w] = active[w] && spc[w] == gmin; mm[w] = __ballot(is_min[w]); cnt += __builtin_popcountll(mm[w]); }
      int wsel = 0, lsel = 0;
      if (cnt == 1) {                                   // unique minimum: no tie rule needed (the common case)
#pragma unroll
        for (int w = 0; w < CW; ++w) if (mm[w]) { wsel = w; lsel = __builtin_ctzll(mm[w]); }
      } else {
        bool anyun = false;
        int lmaxp = -1, lminp = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < CW; ++w) {
          const bool un = is_min[w] && row4col[w] == -1;
          anyun |= __ballot(un) != 0ull;
          if (un) lmaxp = max(lmaxp, pos[w]);
          if (is_min[w]) lminp = min(lminp, pos[w]);
        }
        const int selpos = anyun ? wave_max_i32_dpp(lmaxp) : wave_min_i32_dpp(lminp);
#pragma unroll
        for (int w = 0; w < CW; ++w) {
          const unsigned long long m = __ballot(active[w] && pos[w] == selpos);
          if (m) { wsel = w; lsel = __builtin_ctzll(m); }
        }
      }
      int selpos2 = 0, owner = -1;
#pragma unroll
      for (int w = 0; w < CW; ++w)
        if (w == wsel) { selpos2 = __builtin_amdgcn_readlane(pos[w], lsel); owner = __builtin_amdgcn_readlane(row4col[w], lsel); }
      minVal = gmin;
#pragma unroll
      for (int w = 0; w < CW; ++w) {
        if (w == wsel && lane == lsel) { SC[w] = true; active[w] = false; }
        else if (active[w] && pos[w] == nrem - 1) pos[w] = selpos2;
      }
      --nrem;
      if (owner == -1) sink = lsel + 64 * wsel; else i = owner;
    }
    // dual updates (u of the rows on the alternating tree, v of the scanned columns)
    const int c4r = (col4row >= 0) ? col4row : 0;
    double spc_of_my_col = 0.0;
#pragma unroll
    for (int w = 0; w < CW; ++w) {
      const double x = __shfl(spc[w], c4r & 63, 64);
      if ((c4r >> 6) == w) spc_of_my_col = x;
    }
    if (lane == cur) u += minVal;
    else if (SR) u += minVal - spc_of_my_col;
#pragma unroll
    for (int w = 0; w < CW; ++w) if (SC[w]) v[w] -= minVal - spc[w];
    // augment
    int j = sink;
    for (;;) {
      const int wj = j >> 6, lj = j & 63;
      int r = 0;
#pragma unroll
      for (int w = 0; w < CW; ++w) if (w == wj) r = __builtin_amdgcn_readlane(path[w], lj);
      const int t = __builtin_amdgcn_readlane(col4row, r);
#pragma unroll
      for (int w = 0; w < CW; ++w) if (w == wj && lane == lj) row4col[w] = r;
      if (lane == r) col4row = j;
      j = t;
      if (r == cur) break;
    }
  }
  return col4row;
}

// ---------------------------------------------------------------------------------------------------
// [r5] INTEGER statement of lap_wave_solve_regw for cost blocks of a narrow range - the blocks scipy's tie rules decide.
// What follows a collapsed Sinkhorn stage (U ~ 1 / n) is a V whose entries agree to a few thousand ulp: no uniqueness certificate
// exists (lap_certified.h), and the step-by-step fp64 solver spends ~2400 cycles per Dijkstra step on it (528 steps: 1.2 M cycles per
// graph, a quarter of the BASELINE cfg-3 solve) in one dependent chain of two-register DPP moves and v_min_f64.
// When every entry of the block is a normal float32 and the binary exponents span <= 6, every entry is an integer multiple of
// q = 2^(emin - 23 - 127) below 2^30 q, and so is everything scipy's solver computes from them in float64 (sums and differences of a
// few hundred such integers, far below 2^53): its arithmetic is EXACT, its comparisons are integer comparisons.  Two more facts:
//   * adding a constant K to every cost changes no decision of the solver: by induction over the rows, u_i of every assigned row and
//     every shortest-path value / minVal of a search carry the offset K, v_j does not, and each comparison has K on both sides;
//     so the costs may be shifted to c' = c - min c >= 0, R = max c';
//   * v_j = 0 on unassigned columns (only scanned columns change, and a scanned unassigned column is the sink), so the shortest
//     augmenting path from the new row is at most its direct edge to a free column: minVal <= R throughout; v_j moves by at most
//     minVal per augmentation: |v_j| <= 32 R.
// With R < 2^17 (checked by the caller) the minimum of a scan is below 2^18 and every candidate below 2^24: int32 arithmetic
// without overflow, and the scan's arg-min - value, then scipy's tie rule (some minimal column unassigned -> the LAST such position
// of `remaining`, else the FIRST minimal position) - is ONE integer wavefront minimum over the key
//     (min(spc, 2^19 - 1) << 11) | (unassigned ? 1023 - pos : 1024 + pos)          (pos < 1024; clamped values are never minimal).
// Same steps, same decisions as lap_wave_solve_regw / oracle/lap.c / scipy, by construction and by test: oracle/lap_int.py restates the
// admission, the shift and the one-key arg-min in numpy and tests/test_oracle_lap.py holds that statement to scipy on the CPU; the GPU
// parity tests compare the projection with scipy.optimize.linear_sum_assignment on the device's own V and with
// cfg.variant = TTDG_GAGM_NO_INT_LAP.
// cst: shifted integer costs, cst[col * ldc + row] (LDS), nr = 32 rows, nc <= 64 * CW columns; returns col4row of row `lane`.
#define LAP_INT_RANGE_BITS 17
template <int CW>
__device__ __forceinline__ int lap_wave_solve_int(int nc, const int* cst, int ldc) {
  // Branch-free inner loop: a column's state is five integers - v, spc, path, row4col and pos, with pos = -1 once the column has been
  // scanned (scipy's SC set; its spc / path are frozen from then on because its candidate is replaced by BIG).
  constexpr int CAP = (1 << 19) - 1, BIG = 1 << 30;
  const int lane = threadIdx.x & 63;
  int u = 0, col4row = -1;
  int v[CW], spc[CW], row4col[CW], path[CW], pos[CW], cofs[CW], pos0[CW];
#pragma unroll
  for (int w = 0; w < CW; ++w) {
    const int j = lane + 64 * w;
    v[w] = 0; spc[w] = BIG; row4col[w] = -1; path[w] = -1;
    pos0[w] = j < nc ? nc - 1 - j : -1;                // position in scipy's `remaining` at the start of every search
    pos[w] = -1;
    cofs[w] = min(j, nc - 1) * ldc;
  }
  for (int cur = 0; cur < 32; ++cur) {
    int minVal = 0, nrem = nc, i = cur, sink = -1;
    int SRm = 0;                                       // rows on the alternating tree, as a bit mask (wavefront-uniform)
    int csgn[CW], cbase[CW], xun[CW];                  // tie code of the column = cbase + csgn * pos (row4col changes only between the rows)
#pragma unroll
    for (int w = 0; w < CW; ++w) {
      pos[w] = pos0[w]; spc[w] = BIG;
      const bool un = row4col[w] < 0;
      csgn[w] = un ? -1 : 1; cbase[w] = un ? 1023 : 1024;
      xun[w] = un ? -1 - (lane + 64 * w) : row4col[w];   // what the search needs to know about the column it selects: owner row, or (< 0) the sink
    }
    while (sink == -1) {
      SRm |= 1 << i;
      const int base = minVal - __builtin_amdgcn_readlane(u, i);
      int c[CW];
#pragma unroll
      for (int w = 0; w < CW; ++w) c[w] = cst[cofs[w] + i];
      int kmin = 0x7fffffff;
#pragma unroll
      for (int w = 0; w < CW; ++w) {
        const int r = pos[w] >= 0 ? base + (c[w] - v[w]) : BIG;
        path[w] = r < spc[w] ? i : path[w];
        spc[w] = min(spc[w], r);
        const int key = (min(spc[w], CAP) << 11) | (cbase[w] + csgn[w] * pos[w]);
        kmin = min(kmin, pos[w] >= 0 ? key : 0x7fffffff);
      }
      const int gkey = wave_min_i32_dpp(kmin);
      minVal = gkey >> 11;
      const int code = gkey & 2047;
      const int selpos = code < 1024 ? 1023 - code : code - 1024;
      int xv = 0;
      unsigned long long any = 0ull;
      --nrem;
#pragma unroll
      for (int w = 0; w < CW; ++w) {
        const bool sel = pos[w] == selpos;             // (selpos >= 0: scanned columns, at -1, never match)
        any |= __ballot(sel);
        xv = sel ? xun[w] : xv;
        pos[w] = sel ? -1 : (pos[w] == nrem ? selpos : pos[w]);     // the last position moves into the hole
      }
      const int x = __builtin_amdgcn_readlane(xv, __builtin_ctzll(any));
      if (x < 0) sink = -1 - x; else i = x;
      if (nrem <= 0 && sink == -1) sink = 0;           // (cannot happen: a free column always exists; never spin)
    }
    // dual updates (u of the rows on the alternating tree, v of the scanned columns)
    const int c4r = (col4row >= 0) ? col4row : 0;
    int spc_of_my_col = 0;
#pragma unroll
    for (int w = 0; w < CW; ++w) {
      const int x = __shfl(spc[w], c4r & 63, 64);
      if ((c4r >> 6) == w) spc_of_my_col = x;
    }
    if (lane == cur) u += minVal;
    else if (lane < 32 && ((SRm >> lane) & 1)) u += minVal - spc_of_my_col;
#pragma unroll
    for (int w = 0; w < CW; ++w) if (pos[w] < 0 && pos0[w] >= 0) v[w] -= minVal - spc[w];
    // augment
    int j = sink;
    for (int guard = 0; guard < 34; ++guard) {
      const int wj = j >> 6, lj = j & 63;
      int r = 0;
#pragma unroll
      for (int w = 0; w < CW; ++w) if (w == wj) r = __builtin_amdgcn_readlane(path[w], lj);
      const int t = __builtin_amdgcn_readlane(col4row, r);
#pragma unroll
      for (int w = 0; w < CW; ++w) if (w == wj && lane == lj) row4col[w] = r;
      if (lane == r) col4row = j;
      j = t;
      if (r == cur) break;
    }
  }
  return col4row;
}
