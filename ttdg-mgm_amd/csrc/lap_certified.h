// Whole-workgroup rectangular LAP (maximise; 32 rows = universe slots, nc = nodes of one graph as columns) for the Hungarian stage
// of the multi-workgroup GA-MGM solver, with a CERTIFICATE that makes it exchangeable with the scipy-order solver of lap_device.h.
//
// The reference projects with scipy.optimize.linear_sum_assignment (utils/hungarian.py:34-63): a successive-shortest-path
// solver that starts from nothing on every call - 32 augmentations x a Dijkstra scan over the node columns, one data-dependent
// step at a time (lap_wave_solve_regw: ~90 us for 32 x 256 on one wavefront; 16 of the 24 iterations of BASELINE cfg-3 are spent
// there).  Consecutive Hungarian-stage iterations solve nearly the same problem, so here:
//   1. the column duals v of the previous iteration (kept per graph in the workspace) are the starting point; every wavefront
//      takes rows: u_i = min_j (c_ij - v_j) and the arg-min column; rows that claim a column alone keep it.  Columns that carry
//      a negative dual but were claimed by nobody return to v = 0 and the rows are re-priced (complementary slackness of the
//      rectangular problem: an unmatched column has v = 0), until nothing changes;
//   2. wavefront 0 augments the rows that lost their claim: the same register-resident Dijkstra as lap_wave_solve_regw, fp64
//      duals, from the partial matching (typically 0-3 rows instead of 32);
//   3. the WHOLE workgroup certifies the result from (m, v) alone.  With u_i := c_{i,m(i)} - v_{m(i)} and S = max|c| + max|v|:
//        (i)   m is injective, every v_j <= 0, v_j = 0 on unmatched columns;
//        (ii)  every reduced cost rc_ij = c_ij - u_i - v_j >= -1e-12 S  (dual feasibility up to fp64 rounding);
//        (iii) call an entry TIGHT when rc_ij < 1e-9 S, and build the directed graph on the 32 rows plus one node F:
//              i -> i' when (i, m(i')) is tight, i -> F when i has a tight entry in an unmatched column, F -> i when
//              v_{m(i)} > -1e-9 S.  The graph must be ACYCLIC (33 nodes: bit masks, sinks peeled by ballot).
//      Any other assignment m' costs  cost(m) + sum_{m'} rc + sum_{cols(m) \ cols(m')} |v_j|.  If m' uses a non-tight entry the
//      first sum is >= 1e-9 S - 31e-12 S > 0; if it drops a column with |v_j| >= 1e-9 S likewise; otherwise m' differs from m
//      along tight alternating cycles, or tight paths from an unmatched column to a dropped column with |v| < 1e-9 S - both are
//      directed cycles of the graph.  So an acyclic graph proves that m is the UNIQUE optimum with a gap of ~1e-9 S, seven
//      orders of magnitude above the fp64 rounding of scipy's own duals - and the unique optimum is what scipy returns, whatever
//      its scan order and tie rules.  (The plain test "every non-matched reduced cost > 0" is not usable: the duals a shortest-
//      path solver ends with make every edge of its search trees tight.)  The certificate depends on (m, v) only, not on how they
//      were found: a mistake in steps 1-2 can cost time, never the answer.
//   4. no certificate (exact ties - e.g. duplicated nodes - or a non-finite entry): the caller runs the scipy-order solver and
//      counts the fallback (info[13] of ttdg_gagm_solve; certified solves in info[12]).
#pragma once
#include "lap_device.h"

#define LAP_CERT_MAX_TIGHT 256
struct LapCertScratch {
  double* v;      // [nc] column duals (<= 0)
  double* u;      // [32] row duals
  int* row4col;   // [nc]
  int* col4row;   // [32]
  int* arg;       // [32]
  int* flag;      // [0] a dual was reset this round, [1] certificate holds, [2] max |c| (float bits), [3] max |v| (float bits)
  unsigned* adj;  // [33][2]: tight-entry graph of the certificate (bit i' of word 0: row i', bit 0 of word 1: the node F)
};

__host__ __device__ inline size_t lap_cert_scratch_bytes(int nc) { return (size_t)(nc + 32) * 8 + (size_t)(nc + 32 + 32 + 4 + 68) * 4; }

__device__ inline LapCertScratch lap_cert_carve(void* base, int nc) {   // base 8-byte aligned
  LapCertScratch s;
  double* d = (double*)base;
  s.v = d; s.u = d + nc;
  int* i = (int*)(s.u + 32);
  s.row4col = i; s.col4row = i + nc; s.arg = s.col4row + 32; s.flag = s.arg + 32; s.adj = (unsigned*)(s.flag + 4);
  return s;
}

// wavefront 0: augment every unmatched row from the partial matching (register-resident columns: lane l owns l, l + 64, ...)
template <int CW>
__device__ __forceinline__ void lap_cert_augment(int nc, const float* vl, const LapCertScratch& s, int* stat) {
  const int lane = threadIdx.x & 63;
  double u = lane < 32 ? s.u[lane] : 0.0;
  int col4row = lane < 32 ? s.col4row[lane] : 0;
  double v[CW], spc[CW];
  int row4col[CW], path[CW];
  bool is_col[CW];
#pragma unroll
  for (int w = 0; w < CW; ++w) {
    const int j = lane + 64 * w;
    is_col[w] = j < nc;
    v[w] = is_col[w] ? s.v[j] : 0.0;
    row4col[w] = is_col[w] ? s.row4col[j] : -1;
    path[w] = -1;
    spc[w] = INFINITY;
  }
  unsigned long long fr = __ballot(lane < 32 && col4row == -1);
  int nsteps = 0;
  if (stat && lane == 0) atomicAdd(&stat[3], __builtin_popcountll(fr));
  while (fr) {
    const int cur = __builtin_ctzll(fr);
    fr &= fr - 1;
    double minVal = 0.0;
    int i = cur, sink = -1;
    bool active[CW], SC[CW], SR = false;
#pragma unroll
    for (int w = 0; w < CW; ++w) { active[w] = is_col[w]; SC[w] = false; spc[w] = INFINITY; }
    while (sink == -1) {
      ++nsteps;
      if (lane == i) SR = true;
      const double ui = readlane_f64(u, i);
      double lmin = INFINITY;
#pragma unroll
      for (int w = 0; w < CW; ++w)
        if (active[w]) {
          const double r = minVal + ((-(double)vl[(lane + 64 * w) * 33 + i] - ui) - v[w]);
          if (r < spc[w]) { path[w] = i; spc[w] = r; }
          lmin = fmin(lmin, spc[w]);
        }
      const double gmin = wave_min_f64_dpp(lmin);
      if (!(gmin < INFINITY)) { sink = -2; break; }       // non-finite costs: leave the row unmatched, the certificate refuses
      int wsel = -1, lsel = 0;
#pragma unroll
      for (int w = 0; w < CW; ++w) {
        const unsigned long long mm = __ballot(active[w] && spc[w] == gmin);
        if (mm != 0ull && wsel < 0) { wsel = w; lsel = __builtin_ctzll(mm); }
      }
      int owner = -1;
#pragma unroll
      for (int w = 0; w < CW; ++w)
        if (w == wsel) owner = __builtin_amdgcn_readlane(row4col[w], lsel);
      minVal = gmin;
#pragma unroll
      for (int w = 0; w < CW; ++w)
        if (w == wsel && lane == lsel) { SC[w] = true; active[w] = false; }
      if (owner == -1) sink = lsel + 64 * wsel; else i = owner;
    }
    if (sink < 0) break;
    // dual updates (rows on the alternating tree, scanned columns), then the augmentation along the stored path
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
  if (lane < 32) s.col4row[lane] = col4row;
#pragma unroll
  for (int w = 0; w < CW; ++w) if (is_col[w]) s.v[lane + 64 * w] = v[w];
  if (stat && lane == 0) atomicAdd(&stat[4], nsteps);
}

// vl: the graph's V tile in LDS, vl[node * 33 + slot]; warm: the previous iteration's column duals (global, nc doubles) or null.
// Returns true (workgroup-uniform) when the certificate holds: s.col4row[slot] is then the node of every universe slot and s.v the
// duals to keep for the next iteration.  PT = threads of the workgroup (every thread must call).
template <int PT, int CW>
__device__ __forceinline__ bool lap_certified_solve(int nc, const float* vl, const LapCertScratch& s, const double* warm, int* stat = nullptr) {
  constexpr int PW = PT / 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int j = tid; j < nc; j += PT) {
    const double x = warm ? warm[j] : 0.0;
    s.v[j] = (x <= 0.0) ? x : 0.0;         // (also turns a NaN of a stale buffer into 0)
  }
  if (tid == 0) s.flag[1] = 1;
  __syncthreads();
  // ---- 1. prices and claims
  for (int round = 0; round < 34; ++round) {
    for (int j = tid; j < nc; j += PT) s.row4col[j] = -1;
    if (tid == 0) s.flag[0] = 0;
    for (int i = wave; i < 32; i += PW) {
      double best = INFINITY;
      int bj = 0;
      for (int j = lane; j < nc; j += 64) {
        const double x = -(double)vl[j * 33 + i] - s.v[j];
        if (x < best) { best = x; bj = j; }
      }
      const double g = wave_min_f64_dpp(best);
      const unsigned long long m = __ballot(best == g);
      const int src = m ? __builtin_ctzll(m) : 0;
      const int jj = __builtin_amdgcn_readlane(bj, src);
      if (lane == 0) { s.u[i] = g; s.arg[i] = jj; }
    }
    __syncthreads();
    if (wave == 0) {
      const int a = lane < 32 ? s.arg[lane] : -1 - lane;
      bool lose = false;
      for (int k = 0; k < 31; ++k) {
        const int ak = __builtin_amdgcn_readlane(a, k);
        if (lane > k && a == ak) lose = true;
      }
      if (lane < 32) {
        s.col4row[lane] = lose ? -1 : a;
        if (!lose) s.row4col[a] = lane;
      }
    }
    __syncthreads();
    bool ch = false;
    for (int j = tid; j < nc; j += PT)
      if (s.v[j] < 0.0 && s.row4col[j] == -1) { s.v[j] = 0.0; ch = true; }
    if (ch) s.flag[0] = 1;
    __syncthreads();
    const int changed = s.flag[0];
    __syncthreads();
    if (stat && tid == 0) atomicAdd(&stat[2], 1);
    if (!changed) break;
  }
  // ---- 2. augment the rows that lost their claim.  More than a quarter of the rows without a column means the prices carried
  // nothing over (the first Hungarian iteration behind a collapsed Sinkhorn stage: every slot wants the same nodes, and the block
  // is ties all over): the shortest-path work would equal the scipy-order solver's and the certificate would refuse anyway
  {
    int losers = 0;
    if (tid < 32) losers = s.col4row[tid] == -1;
    const int nl = __builtin_popcountll(__ballot(losers != 0));       // (rows live in wavefront 0)
    if (tid == 0) s.flag[0] = nl;
    __syncthreads();
    const int nfree = s.flag[0];
    __syncthreads();
    if (nfree > 8) return false;
  }
  if (wave == 0) lap_cert_augment<CW>(nc, vl, s, stat);
  __syncthreads();
  // ---- 3. certificate on (m, v)
  for (int j = tid; j < nc; j += PT) s.row4col[j] = -1;
  if (tid < 66) s.adj[tid] = 0u;
  if (tid == 0) { s.flag[0] = 0; s.flag[2] = 0; s.flag[3] = 0; }
  __syncthreads();
  {
    float mc = 0.f, mv = 0.f;
    for (int j = tid; j < nc; j += PT) {
      mv = fmaxf(mv, (float)fabs(s.v[j]) * 1.000001f);
#pragma unroll 8
      for (int i = 0; i < 32; ++i) mc = fmaxf(mc, fabsf(vl[j * 33 + i]));
    }
    mc = wave_max_f32_dpp(mc); mv = wave_max_f32_dpp(mv);
    if (lane == 0) { atomicMax(&s.flag[2], __float_as_int(mc)); atomicMax(&s.flag[3], __float_as_int(mv)); }     // non-negative floats order as ints
  }
  if (tid < 32) {
    const int j = s.col4row[tid];
    bool good = j >= 0 && j < nc;
    if (good) {
      if (atomicExch(&s.row4col[j], tid) != -1) good = false;          // two rows on one column
      s.u[tid] = -(double)vl[j * 33 + tid] - s.v[j];
    } else {
      s.u[tid] = 0.0;
    }
    if (!good) s.flag[1] = 0;
  }
  __syncthreads();
  const double S = (double)__int_as_float(s.flag[2]) + (double)__int_as_float(s.flag[3]);
  const double t_lo = 1e-12 * S, t_hi = 1e-9 * S;
  bool ok = S > 0.0 && S < 1e300;                  // (an all-zero block is all ties; a non-finite entry certifies nothing)
  // nc <= PT on every caller: a thread owns at most ONE column.  [r5] Every thread walks the 32 entries of its column on its own
  // (reduced costs in float64, dual feasibility, a 32-bit mask of its tight entries: no cross-lane traffic), the workgroup counts
  // the tight entries, and only then the few that exist go to the 33 adjacency words as LDS atomics.  A block with more than
  // LAP_CERT_MAX_TIGHT tight entries is refused outright (ties all over: the graph would be cyclic, and one atomic per tight entry
  // of such a block once cost a million cycles) - refusing is always safe, the caller runs the scipy-order solver.  Round 4 reduced
  // every ROW's tight entries over the wavefront first (32 x a six-stage DPP OR + ballot + readlane per wavefront): 20 k of the
  // 35 k cycles of a certified LAP at 32 x 256 (in-kernel clocks, round 5).
  {
    const int j = tid;
    const bool have = j < nc;
    const double vj = have ? s.v[j] : 0.0;
    const int rj = have ? s.row4col[j] : -1;
    unsigned tmask = 0u;
    if (have) {
      if (!(vj <= 0.0)) ok = false;
      if (rj < 0 && vj != 0.0) ok = false;
      const float* col = vl + j * 33;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) {
        const double rc = (-(double)col[i] - s.u[i]) - vj;
        if (i != rj) {
          if (!(rc >= -t_lo)) ok = false;
          else if (rc < t_hi) tmask |= 1u << i;
        }
      }
    }
    const int wcnt = wave_sum_i32_dpp(__popc(tmask));
    if (lane == 0 && wcnt) atomicAdd(&s.flag[0], wcnt);      // (flag[0] is free again: zeroed below before the count)
    __syncthreads();
    if (s.flag[0] > LAP_CERT_MAX_TIGHT) ok = false;
    else if (have) {
      if (rj >= 0 && vj > -t_hi) atomicOr(&s.adj[2 * 32], 1u << rj);                      // F -> rj (at most 32 of these)
      while (tmask) {
        const int i = __builtin_ctz(tmask);
        tmask &= tmask - 1;
        if (rj >= 0) atomicOr(&s.adj[2 * i], 1u << rj);                                   // i -> rj
        else atomicOr(&s.adj[2 * i + 1], 1u);                                             // i -> F
      }
    }
  }
  if (!ok) s.flag[1] = 0;
  __syncthreads();
  if (wave == 0) {      // acyclic <=> peeling sinks empties the graph
    unsigned long long my = lane < 33 ? ((unsigned long long)s.adj[2 * lane] | ((unsigned long long)(s.adj[2 * lane + 1] & 1u) << 32)) : 0ull;
    unsigned long long alive = (1ull << 33) - 1ull;
    for (int round = 0; round < 34; ++round) {
      const bool sink = lane < 33 && ((alive >> lane) & 1ull) && (my & alive) == 0ull;
      const unsigned long long m = __ballot(sink);
      if (m == 0ull) break;
      alive &= ~m;
    }
    if (lane == 0 && alive != 0ull) s.flag[1] = 0;
  }
  __syncthreads();
  return s.flag[1] != 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Scipy-order LAP on the WHOLE workgroup (32 rows = universe slots, nc <= PT node columns, one column per thread): step for step
// the algorithm of lap_wave_solve_regw - the same shortest-path scan, the same fp64 evaluation order ((minVal + c) - u_i) - v_j,
// the same tie rule on scipy's `remaining` positions (some minimal column unassigned -> the LAST such position, else the FIRST
// minimal position; the last position moves into the hole) - with the arg-min of a step taken over all wavefronts in ONE LDS
// meeting: every wavefront reduces its own columns (DPP minimum; among ITS minima the largest position of an unassigned one and
// the smallest position, with the columns and the owner row behind them) and publishes that record; after the barrier every
// thread combines the records of the wavefronts that hold the global minimum.  Records are double-buffered, so one barrier per
// step is all.  One wavefront owning four columns per lane spends ~2400 cycles per step in one dependent fp64 chain (measured:
// 1.25 M cycles for the 32 x 256 block of all-tied costs that follows a collapsed Sinkhorn stage - a quarter of the cfg-3 solve).
struct LapBlockScratch {
  double* u;        // [32]
  double* spc;      // [nc]   (published for the dual update)
  double* wmin;     // [2][16] per-wavefront minimum, two slots
  int* path;        // [nc]
  int* row4col;     // [nc]
  int* col4row;     // [32]
  int* SR;          // [32]
  int* rec;         // [2][16][8]: largest position of an unassigned minimum (-1: none), its column; smallest position of a minimum, its column, its owner row
};
__host__ __device__ inline size_t lap_block_scratch_bytes(int nc) { return (size_t)(32 + nc + 32) * 8 + (size_t)(2 * nc + 32 + 32 + 256) * 4; }
__device__ inline LapBlockScratch lap_block_carve(void* base, int nc) {     // base 8-byte aligned
  LapBlockScratch s;
  double* d = (double*)base;
  s.u = d; s.spc = d + 32; s.wmin = s.spc + nc;
  int* i = (int*)(s.wmin + 32);
  s.path = i; s.row4col = i + nc; s.col4row = s.row4col + nc; s.SR = s.col4row + 32; s.rec = s.SR + 32;
  return s;
}

// every thread of the workgroup calls; on return s.col4row[slot] = node of every universe slot
template <int PT>
__device__ __forceinline__ void lap_block_solve_exact(int nc, const float* vl, const LapBlockScratch& s) {
  constexpr int NW = PT / 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = tid;
  const bool is_col = j < nc;
  double v = 0.0, spc = INFINITY;
  int row4col = -1, path = -1;
  if (tid < 32) { s.u[tid] = 0.0; s.col4row[tid] = -1; }
  __syncthreads();
  int slot = 0;
  for (int cur = 0; cur < 32; ++cur) {
    double minVal = 0.0;
    int nrem = nc, i = cur, sink = -1;
    bool active = is_col, SC = false;
    int pos = nc - 1 - j;
    spc = INFINITY;
    if (tid < 32) s.SR[tid] = 0;
    __syncthreads();
    while (sink == -1) {
      if (tid == 0) s.SR[i] = 1;
      const double ui = s.u[i];
      if (active) {
        const double r = minVal + (-(double)vl[j * 33 + i]) - ui - v;
        if (r < spc) { path = i; spc = r; }
      }
      const double wm = wave_min_f64_dpp(active ? spc : INFINITY);
      {
        const bool lmin = active && spc == wm;
        const bool lun = lmin && row4col == -1;
        const unsigned long long mm = __ballot(lmin), mu = __ballot(lun);
        int mx = -1, jmx = 0, mn = 0x7fffffff, jmn = 0, omn = -1;
        if (mm != 0ull) {                               // wavefront-uniform
          if ((mm & (mm - 1)) == 0ull) {                // one minimum in this wavefront: no reductions
            const int l = __builtin_ctzll(mm);
            mn = __builtin_amdgcn_readlane(pos, l); jmn = 64 * wave + l; omn = __builtin_amdgcn_readlane(row4col, l);
            if (mu != 0ull) { mx = mn; jmx = jmn; }
          } else {
            mn = wave_min_i32_dpp(lmin ? pos : 0x7fffffff);
            const int l = __builtin_ctzll(__ballot(lmin && pos == mn));
            jmn = 64 * wave + l; omn = __builtin_amdgcn_readlane(row4col, l);
            if (mu != 0ull) {
              mx = wave_max_i32_dpp(lun ? pos : -1);
              jmx = 64 * wave + __builtin_ctzll(__ballot(lun && pos == mx));
            }
          }
        }
        if (lane == 0) {
          s.wmin[slot * 16 + wave] = wm;
          int* r = s.rec + (slot * 16 + wave) * 8;
          r[0] = mx; r[1] = jmx; r[2] = mn; r[3] = jmn; r[4] = omn;
        }
      }
      __syncthreads();
      double gmin = s.wmin[slot * 16];
#pragma unroll
      for (int w = 1; w < NW; ++w) gmin = fmin(gmin, s.wmin[slot * 16 + w]);
      int mxu = -1, jmxu = 0, mnp = 0x7fffffff, jmnp = 0, omnp = -1;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if (s.wmin[slot * 16 + w] == gmin) {
          const int* r = s.rec + (slot * 16 + w) * 8;
          if (r[0] > mxu) { mxu = r[0]; jmxu = r[1]; }
          if (r[2] < mnp) { mnp = r[2]; jmnp = r[3]; omnp = r[4]; }
        }
      slot ^= 1;
      const int selpos = mxu >= 0 ? mxu : mnp;          // (finite costs: at least one minimum exists)
      const int jsel = mxu >= 0 ? jmxu : jmnp, owner = mxu >= 0 ? -1 : omnp;
      if (j == jsel) { SC = true; active = false; }
      else if (active && pos == nrem - 1) pos = selpos;
      --nrem;
      minVal = gmin;
      if (owner == -1) sink = jsel; else i = owner;
      if (nrem < 0) { sink = jsel; break; }             // (cannot happen with finite costs; never spin)
    }
    // dual updates (rows on the alternating tree, scanned columns), then the augmentation along the stored path
    if (is_col) { s.spc[j] = spc; s.path[j] = path; s.row4col[j] = row4col; }
    __syncthreads();
    if (tid < 32) {
      if (tid == cur) s.u[tid] += minVal;
      else if (s.SR[tid]) s.u[tid] += minVal - s.spc[s.col4row[tid]];
    }
    if (SC) v -= minVal - spc;
    __syncthreads();
    if (tid == 0) {
      int jj = sink;
      for (int guard = 0; guard < 34; ++guard) {
        const int r = s.path[jj];
        const int t = s.col4row[r];
        s.row4col[jj] = r;
        s.col4row[r] = jj;
        jj = t;
        if (r == cur) break;
      }
    }
    __syncthreads();
    if (is_col) row4col = s.row4col[j];
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// [r5] Admission to the integer scipy-order solver (lap_device.h: lap_wave_solve_int) and the conversion of the block, by the whole
// workgroup: every entry of the graph's V tile (vl[node * 33 + slot], n nodes x 32 slots) must be a normal float32, the binary
// exponents must span <= 6, and the integer range R = (max - min) / 2^(emin - 150) must stay below 2^LAP_INT_RANGE_BITS.  Then the
// tile is overwritten IN PLACE with the shifted integer costs of the minimisation scipy solves (cost = -V: c' = (max V - V) / q >= 0)
// and true is returned (workgroup-uniform); otherwise the tile is untouched.  scr: 8 ints of LDS.  Every thread must call.
template <int PT>
__device__ __forceinline__ bool lap_int_admit(int n, float* vl, int* scr) {
  const int tid = threadIdx.x, ne = n * 32;
  if (tid == 0) { scr[0] = 1; scr[1] = 255; scr[2] = 0; scr[3] = 0x7fffffff; scr[4] = -0x7fffffff; }
  __syncthreads();
  {
    int lo = 255, hi = 0;
    bool ok = true;
    for (int e = tid; e < ne; e += PT) {
      const int ex = (__float_as_int(vl[(e >> 5) * 33 + (e & 31)]) >> 23) & 0xff;
      ok &= ex != 0 && ex != 255;
      lo = min(lo, ex); hi = max(hi, ex);
    }
    lo = wave_min_i32_dpp(lo); hi = wave_max_i32_dpp(hi);
    if ((tid & 63) == 0) { atomicMin(&scr[1], lo); atomicMax(&scr[2], hi); }
    if (!ok) scr[0] = 0;
  }
  __syncthreads();
  const int emin = scr[1];
  const bool narrow = scr[0] != 0 && scr[2] - emin <= 6;
  __syncthreads();                       // (a declining workgroup reuses the scratch at once)
  if (!narrow) return false;
  auto as_int = [emin](float x) {
    const int b = __float_as_int(x);
    const int m = ((b & 0x7fffff) | 0x800000) << (((b >> 23) & 0xff) - emin);       // < 2^30
    return b < 0 ? -m : m;
  };
  {
    int lo = 0x7fffffff, hi = -0x7fffffff;
    for (int e = tid; e < ne; e += PT) {
      const int iv = as_int(vl[(e >> 5) * 33 + (e & 31)]);
      lo = min(lo, iv); hi = max(hi, iv);
    }
    lo = wave_min_i32_dpp(lo); hi = wave_max_i32_dpp(hi);
    if ((tid & 63) == 0) { atomicMin(&scr[3], lo); atomicMax(&scr[4], hi); }
  }
  __syncthreads();
  const int vmax = scr[4];
  const bool small = (long long)vmax - (long long)scr[3] < (1ll << LAP_INT_RANGE_BITS);
  __syncthreads();
  if (!small) return false;
  for (int e = tid; e < ne; e += PT) {
    float* p = vl + (e >> 5) * 33 + (e & 31);
    *p = __int_as_float(vmax - as_int(*p));
  }
  __syncthreads();
  return true;
}
