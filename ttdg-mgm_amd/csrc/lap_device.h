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
