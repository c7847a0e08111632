// A4 + A5 fused — the pair stage of MGM3_unsup.forward (reference multi_graph_matching.py:504-525):
//     for every ordered pair of graphs a >= b:   M_ab = Affinity(x_a, x_b)  (utils/affinity.py:44-57)
//                                                Wds[a,b] = Sinkhorn(M_ab, tau = 0.05, 20 sweeps, dummy rows)   (:518-522)
//                                                Wds[b,a] = Wds[a,b]^T                                          (:523-525)
// Round 2 ran this as two launches with an HBM round trip in between: affinity_fwd wrote 16 K-split partial planes (2 MB for a
// 55 KB result, PMC traffic 10.5x the algorithmic bytes) which sinkhorn_pairs_fwd re-read, and the Sinkhorn kernel walked its
// lines one at a time through LDS (two passes and two dependent shuffle reductions per line, a barrier per sweep): 10 + 104 us
// for the ten 20..40-node pairs of a TTA step.  Here ONE workgroup per pair keeps the block on the CU from the P / Q rows to the
// doubly-stochastic result:
//   phase 1  M_ij = sum_k w2_k relu(P_ik + Q_jk) + b2 over the whole hidden dimension (same 1.5-instruction inner loop as
//            affinity.hip: packed add + fma with the |.| modifier, the linear half as two row sums), 4x4 register tiles,
//            16-row sub-tiles beyond the block's edge skipped;  (M - b2) goes to `aff` once (the backward's input, 1x bytes),
//            L = (M) * log2(e) / tau goes to LDS in the oriented frame (rows = the smaller graph);
//   phase 2  the block moves into REGISTERS: lane = column, wavefront w owns rows w, w+4, ... (<= 16 per lane).  A row sweep is
//            <= 16 independent DPP wavefront reductions (no LDS, no barrier); a column sweep is <= 16 in-lane terms, the four
//            wavefront partials meet in LDS once (ONE barrier per column sweep, double-buffered) and every wavefront finishes
//            all columns redundantly.  Potential form (f_p = lse_q(L - g), g_q = lse_p(L - f)), previous potential as the
//            stabiliser after the first pair with an exact two-pass fallback when a sum leaves [2^-80, 2^80] - the arithmetic
//            of sinkhorn_pairs_fwd_reg_kernel (sinkhorn.hip), which the 256-node path has been running since round 1.
//            The per-sweep potentials the backward needs are logged in LDS and written out once.
// The backward (ttdg_pair_stage_bwd) is the same register layout run in reverse: dY in registers, L rebuilt from `aff`,
// potentials from the log; row sums = DPP reductions, column sums meet in LDS; it emits dM, which affinity.hip's backward takes.
// Graphs of more than 64 nodes take the round-2 kernels (ttdg_affinity_pairwise_fwd + ttdg_sinkhorn_pairs_fwd).
#include "sinkhorn_device.h"

#define PS_T 64          /* nodes per graph on this path */
#define PS_BK 64         /* hidden units per slab */
#define PS_LDK (PS_BK + 4)
#define PS_LDM (PS_T + 1)
#define PS_RW 16         /* rows per lane: 4 wavefronts x 16 */
#define PS_WAVES 4
#define PS_BIG 1.2e24f
#define PS_SMALL 8.3e-25f
typedef float ps_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float ps_fma_abs(float w, float x, float acc) {
  asm("v_fma_f32 %0, %1, |%2|, %0" : "+v"(acc) : "v"(w), "v"(x));
  return acc;
}
__device__ __forceinline__ bool ps_sane(float s) { return s > PS_SMALL && s < PS_BIG; }
// LDS-only barrier: the potential log and the result stores of a sweep must not be waited for (a __syncthreads() drains vmcnt)
__device__ __forceinline__ void ps_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void ps_pair_of(int idx, int& a, int& b) {
  a = 0;
  while ((a + 1) * (a + 2) / 2 <= idx) ++a;   // pairs ordered (0,0),(1,0),(1,1),(2,0)... as ttdg_sinkhorn_pairs_fwd
  b = idx - a * (a + 1) / 2;
}

// One column sweep.  kExact: two-pass (column maxima through LDS first); otherwise the previous potential g is the stabiliser
// and the return value says whether some column sum left the sane range - every wavefront sees the same sums, so all of them
// take the exact path together.
template <bool kExact, int NR>
__device__ __forceinline__ bool ps_col_sweep(const float (&L)[PS_RW], const float (&f)[PS_RW], float g, float td0, int mult, int r, int c,
                                             int wave, int lane, float (&s_part)[2][PS_WAVES * PS_T], int& buf, float& gn) {
  // rows beyond r (and lanes beyond c) carry L = -inf and f = 0: their terms are exp2(-inf) = 0 and max(-inf), so the loops
  // below need no guards - NR independent transcendentals and a tree sum instead of a dependent add / select chain
  float sh;
  if (kExact) {
    float t[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) t[i] = L[i] - f[i];
#pragma unroll
    for (int st = 1; st < NR; st <<= 1)
#pragma unroll
      for (int i = 0; i + st < NR; i += 2 * st) t[i] = fmaxf(t[i], t[i + st]);
    s_part[buf][wave * PS_T + lane] = t[0];
    ps_barrier();
    sh = td0;
#pragma unroll
    for (int w = 0; w < PS_WAVES; ++w) sh = fmaxf(sh, s_part[buf][w * PS_T + lane]);
    if (lane >= c) sh = 0.f;
    buf ^= 1;
  } else {
    sh = g;
  }
  float e[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) e[i] = fast_exp2(L[i] - f[i] - sh);
#pragma unroll
  for (int st = 1; st < NR; st <<= 1)
#pragma unroll
    for (int i = 0; i + st < NR; i += 2 * st) e[i] += e[i + st];
  s_part[buf][wave * PS_T + lane] = e[0];
  ps_barrier();
  float s = (s_part[buf][0 * PS_T + lane] + s_part[buf][1 * PS_T + lane]) + (s_part[buf][2 * PS_T + lane] + s_part[buf][3 * PS_T + lane]);
  buf ^= 1;
  if (mult > 0) s += (float)mult * fast_exp2(td0 - sh);
  const bool live = lane < c;
  gn = live ? sh + fast_log2(s) : 0.f;
  return !kExact && __ballot(live && !ps_sane(s)) != 0ull;
}

template <int NR>
__device__ __forceinline__ float ps_col_step(const float (&L)[PS_RW], const float (&f)[PS_RW], float g, float td0, int mult, int r, int c, int wave,
                                             int lane, float (&s_part)[2][PS_WAVES * PS_T], int& buf, bool first) {
  float gn;
  bool exact = first;
  if (!exact) exact = ps_col_sweep<false, NR>(L, f, g, td0, mult, r, c, wave, lane, s_part, buf, gn);
  if (exact) ps_col_sweep<true, NR>(L, f, g, td0, mult, r, c, wave, lane, s_part, buf, gn);
  return gn;
}

// One 32-deep slab of the affinity block: TA x TB live 16 x 16 sub-tiles per thread tile (compile-time: no guards, the
// ds_read_b128 of a k-step are all in flight before the first fma).
template <int TA, int TB>
struct PsOperands {
  float4 p[TA][2], q[TB][2], w[2];
};

template <int TA, int TB>
__device__ __forceinline__ void ps_slab_load(PsOperands<TA, TB>& o, const float (&Ps)[PS_T][PS_LDK], const float (&Qs)[PS_T][PS_LDK],
                                             const float (&Ws)[PS_BK], int tx, int ty, int kk) {
#pragma unroll
  for (int x = 0; x < TA; ++x) {
    o.p[x][0] = *reinterpret_cast<const float4*>(&Ps[ty + 16 * x][kk]);
    o.p[x][1] = *reinterpret_cast<const float4*>(&Ps[ty + 16 * x][kk + 4]);
  }
#pragma unroll
  for (int y = 0; y < TB; ++y) {
    o.q[y][0] = *reinterpret_cast<const float4*>(&Qs[tx + 16 * y][kk]);
    o.q[y][1] = *reinterpret_cast<const float4*>(&Qs[tx + 16 * y][kk + 4]);
  }
  o.w[0] = *reinterpret_cast<const float4*>(&Ws[kk]);
  o.w[1] = *reinterpret_cast<const float4*>(&Ws[kk + 4]);
}

template <int TA, int TB>
__device__ __forceinline__ void ps_slab_fma(const PsOperands<TA, TB>& o, float (&acc)[4][4]) {
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int x = 0; x < TA; ++x)
#pragma unroll
      for (int y = 0; y < TB; ++y) {
        const ps_f32x2 x0 = (ps_f32x2){o.p[x][h].x, o.p[x][h].y} + (ps_f32x2){o.q[y][h].x, o.q[y][h].y};
        const ps_f32x2 x1 = (ps_f32x2){o.p[x][h].z, o.p[x][h].w} + (ps_f32x2){o.q[y][h].z, o.q[y][h].w};
        acc[x][y] = ps_fma_abs(o.w[h].x, x0.x, acc[x][y]);
        acc[x][y] = ps_fma_abs(o.w[h].y, x0.y, acc[x][y]);
        acc[x][y] = ps_fma_abs(o.w[h].z, x1.x, acc[x][y]);
        acc[x][y] = ps_fma_abs(o.w[h].w, x1.y, acc[x][y]);
      }
}

// One 64-deep slab of the affinity block: TA x TB live 16 x 16 sub-tiles per thread tile (compile-time: no guards).  With one
// wavefront per SIMD nothing hides an LDS round trip, so the reads are software-pipelined by hand: the operands of the NEXT
// eight hidden units are requested (two register sets, ping-pong) before the 12 (TA TB) VALU instructions of the current
// eight are issued; __builtin_amdgcn_sched_barrier keeps the compiler from sinking the requests behind the arithmetic.
template <int TA, int TB>
__device__ __forceinline__ void ps_affinity_slab(const float (&Ps)[PS_T][PS_LDK], const float (&Qs)[PS_T][PS_LDK], const float (&Ws)[PS_BK], int tx,
                                                 int ty, float (&acc)[4][4]) {
  PsOperands<TA, TB> a, b;
  ps_slab_load<TA, TB>(a, Ps, Qs, Ws, tx, ty, 0);
#pragma unroll
  for (int kk = 0; kk < PS_BK; kk += 16) {
    ps_slab_load<TA, TB>(b, Ps, Qs, Ws, tx, ty, kk + 8);
    __builtin_amdgcn_sched_barrier(0);
    ps_slab_fma<TA, TB>(a, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (kk + 16 < PS_BK) ps_slab_load<TA, TB>(a, Ps, Qs, Ws, tx, ty, kk + 16);
    __builtin_amdgcn_sched_barrier(0);
    ps_slab_fma<TA, TB>(b, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Row sweep over the rows a wavefront owns, NR of them at most (compile-time): f_p = lse_q(L_pq - g_q).  The common case - the
// previous potential stabilises the exponentials - is branch-free: NR independent DPP reductions the compiler interleaves;
// only when some sum leaves the sane range (or in the first pair) is the exact max-subtracted form run, for all rows.
template <int NR>
__device__ __forceinline__ void ps_row_sweep(const float (&L)[PS_RW], float (&f)[PS_RW], float g, int r, int wave, bool first) {
  float s[NR];
  bool bad = first;
  if (!first) {
#pragma unroll
    for (int i = 0; i < NR; ++i) s[i] = wave_sum_f32_dpp(fast_exp2(L[i] - g - f[i]));
#pragma unroll
    for (int i = 0; i < NR; ++i) bad |= (wave + PS_WAVES * i < r) && !ps_sane(s[i]);
  }
  if (!bad) {
#pragma unroll
    for (int i = 0; i < NR; ++i) f[i] = (wave + PS_WAVES * i < r) ? f[i] + fast_log2(s[i]) : 0.f;
    return;
  }
  float m[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) m[i] = wave_max_f32_dpp(L[i] - g);
#pragma unroll
  for (int i = 0; i < NR; ++i) s[i] = wave_sum_f32_dpp(fast_exp2(L[i] - g - m[i]));
#pragma unroll
  for (int i = 0; i < NR; ++i) f[i] = (wave + PS_WAVES * i < r) ? m[i] + fast_log2(s[i]) : 0.f;
}

__global__ __launch_bounds__(256) void pair_stage_fwd_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                             const float* __restrict__ w2, const float* __restrict__ b2, int H,
                                                             ttdg_graphs_t gr, float tau, int iters, float* __restrict__ aff,
                                                             float* __restrict__ Wds, float* __restrict__ pot, int cmax) {
  __shared__ __attribute__((aligned(16))) float Ps[2][PS_T][PS_LDK];
  __shared__ __attribute__((aligned(16))) float Qs[2][PS_T][PS_LDK];
  __shared__ __attribute__((aligned(16))) float Ws[2][PS_BK];
  __shared__ float Arow[PS_T], Brow[PS_T];
  __shared__ float mat[PS_T * PS_LDM];
  __shared__ float s_part[2][PS_WAVES * PS_T];
  __shared__ float plog[SK_MAXK * (PS_T + 1)];
  int a, b;
  ps_pair_of(blockIdx.x, a, b);
  const int M = gr.off[gr.G];
  const int i0 = gr.off[a], j0 = gr.off[b];
  const int na = gr.off[a + 1] - i0, nb = gr.off[b + 1] - j0;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const bool flip = nb < na;                 // rows <= cols (reference :518-522): rows = the smaller graph
  const int r = flip ? nb : na, c = flip ? na : nb, mult = c - r;
  const int ta = (na + 15) >> 4, tb = (nb + 15) >> 4;     // 16-row sub-tiles that hold anything

  // ---- phase 1: the affinity block ----
  // 64-deep slabs, double buffered in LDS AND in registers: slab s + 1 is requested from L2 before slab s is consumed and
  // stored into the other LDS buffer after it - one barrier per slab, the memory round trip hidden behind ~900 VALU
  // instructions; the loop body is instantiated per (ta, tb): no guards inside.
  float acc[4][4], tot[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = tot[x][y] = 0.f;
  float pa[4] = {0.f, 0.f, 0.f, 0.f}, qa[4] = {0.f, 0.f, 0.f, 0.f};
  const int lrow = tid >> 4, lk = (tid & 15) * 4;         // staging map: 16 lanes x float4 cover one 64-wide row, 16 rows per pass
  float4 pv[4], qv[4], wv;
  auto request = [&](int k0) {
    wv = *reinterpret_cast<const float4*>(w2 + k0 + lk);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int row = lrow + 16 * h;
      pv[h] = row < na ? *reinterpret_cast<const float4*>(P + (size_t)(i0 + row) * H + k0 + lk) : make_float4(0.f, 0.f, 0.f, 0.f);
      qv[h] = row < nb ? *reinterpret_cast<const float4*>(Q + (size_t)(j0 + row) * H + k0 + lk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto deposit = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int row = lrow + 16 * h;
      *reinterpret_cast<float4*>(&Ps[buf][row][lk]) = pv[h];
      *reinterpret_cast<float4*>(&Qs[buf][row][lk]) = qv[h];
      pa[h] = fmaf(wv.x, pv[h].x, fmaf(wv.y, pv[h].y, fmaf(wv.z, pv[h].z, fmaf(wv.w, pv[h].w, pa[h]))));
      qa[h] = fmaf(wv.x, qv[h].x, fmaf(wv.y, qv[h].y, fmaf(wv.z, qv[h].z, fmaf(wv.w, qv[h].w, qa[h]))));
    }
    if (lrow == 0) *reinterpret_cast<float4*>(&Ws[buf][lk]) = make_float4(0.5f * wv.x, 0.5f * wv.y, 0.5f * wv.z, 0.5f * wv.w);
  };
  request(0);
  deposit(0);
  __syncthreads();
  const int shape = (ta - 1) * 4 + (tb - 1);
  const int nslab = H / PS_BK;
  for (int sl = 0; sl < nslab; ++sl) {
    const int cur = sl & 1;
    if (sl + 1 < nslab) request((sl + 1) * PS_BK);
    switch (shape) {
#define PS_CASE(A, B) case (A - 1) * 4 + (B - 1): ps_affinity_slab<A, B>(Ps[cur], Qs[cur], Ws[cur], tx, ty, acc); break;
      PS_CASE(1, 1) PS_CASE(1, 2) PS_CASE(1, 3) PS_CASE(1, 4) PS_CASE(2, 1) PS_CASE(2, 2) PS_CASE(2, 3) PS_CASE(2, 4)
      PS_CASE(3, 1) PS_CASE(3, 2) PS_CASE(3, 3) PS_CASE(3, 4) PS_CASE(4, 1) PS_CASE(4, 2) PS_CASE(4, 3) PS_CASE(4, 4)
#undef PS_CASE
    }
    if (sl + 1 < nslab) deposit(cur ^ 1);
    if ((sl & 1) == 1) {
      // [r6] every 128 hidden units the running sums move to a second register set and the chains restart: a 512-long fp32 chain per entry lost
      // ~3 x what the fp32 reference loses on the same inputs (|Wds - fp64| 1.9e-6 against 4.6e-7, dM 2e-5 against 6e-6: tools/probe_pair_stage_sizes.py)
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) { tot[x][y] += acc[x][y]; acc[x][y] = 0.f; }
    }
    __syncthreads();
  }
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] += tot[x][y];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { pa[h] += __shfl_xor(pa[h], o, 64); qa[h] += __shfl_xor(qa[h], o, 64); }
    if ((tid & 15) == 0) { Arow[lrow + 16 * h] = 0.5f * pa[h]; Brow[lrow + 16 * h] = 0.5f * qa[h]; }
  }
  __syncthreads();
  const float bias = b2 ? *b2 : 0.f, scale = TTDG_LOG2E / tau;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const int i = ty + 16 * x;
    if (i >= na) continue;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int j = tx + 16 * y;
      if (j >= nb) continue;
      const float m = acc[x][y] + (Arow[i] + Brow[j]);
      if (aff) aff[(size_t)(i0 + i) * M + j0 + j] = m;                  // without b2, as one plane of the K-split form
      mat[(flip ? j : i) * PS_LDM + (flip ? i : j)] = (m + bias) * scale;
    }
  }
  __syncthreads();

  // ---- phase 2: 20 sweeps in registers ----
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int potld = PS_T + 1;
  float L[PS_RW], f[PS_RW];
#pragma unroll
  for (int i = 0; i < PS_RW; ++i) {
    const int p = wave + PS_WAVES * i;
    L[i] = (p < r && lane < c) ? mat[p * PS_LDM + lane] : -INFINITY;
    f[i] = 0.f;
  }
  float g = 0.f, tdn = SK_DUMMY;
  int buf = 0;
  for (int it = 0; it < iters; ++it) {
    if ((it & 1) == 0) {
      const int nrw = (r - wave + PS_WAVES - 1) / PS_WAVES;      // rows this wavefront owns
      if (nrw <= 4) ps_row_sweep<4>(L, f, g, r, wave, it < 2);
      else if (nrw <= 8) ps_row_sweep<8>(L, f, g, r, wave, it < 2);
      else if (nrw <= 12) ps_row_sweep<12>(L, f, g, r, wave, it < 2);
      else ps_row_sweep<16>(L, f, g, r, wave, it < 2);
#pragma unroll
      for (int i = 0; i < PS_RW; ++i) {
        const int p = wave + PS_WAVES * i;
        if (p < r && lane == 0) plog[it * potld + p] = f[i];
      }
      if (mult > 0) {          // the dummy row (one virtual row of multiplicity c - r), always in the exact form
        const float td = lane < c ? -g : -INFINITY;
        const float dm = wave_max_f32_dpp(td);
        const float ds = wave_sum_f32_dpp(fast_exp2(td - dm));
        // [r6] the dummy row's normalised value -(dm + log2 ds) is kept as it is for the column sweep: rounds 3-5 formed fd = SK_DUMMY + dm + log2 ds
        // and read it back as SK_DUMMY - fd - a round trip through |fd| ~ 144 that put ulp(144) = 1.5e-5 (relative) on the dummy mass of every
        // column sum in every sweep (ragged pairs lost 4 x what the fp32 reference loses, equal-size pairs 2 x: tools/probe_pair_stage_sizes.py)
        tdn = -(dm + fast_log2(ds));
        if (tid == 0) plog[it * potld + r] = -tdn;       // logged without the fill: the backward forms exp2(-x - g) with no round trip either
      }
    } else {
      const float td0 = (mult > 0) ? tdn : -INFINITY;
      // NR is chosen from r alone (not from this wavefront's row count): all four wavefronts take the same instantiation and
      // meet at the same barriers
      const int nrm = (r + PS_WAVES - 1) / PS_WAVES;
      if (nrm <= 4) g = ps_col_step<4>(L, f, g, td0, mult, r, c, wave, lane, s_part, buf, it < 2);
      else if (nrm <= 8) g = ps_col_step<8>(L, f, g, td0, mult, r, c, wave, lane, s_part, buf, it < 2);
      else if (nrm <= 12) g = ps_col_step<12>(L, f, g, td0, mult, r, c, wave, lane, s_part, buf, it < 2);
      else g = ps_col_step<16>(L, f, g, td0, mult, r, c, wave, lane, s_part, buf, it < 2);
      if (wave == 0 && lane < c) plog[it * potld + lane] = g;
    }
  }
  // ---- result: Wds[a,b] and its mirror; the potential log ----
  float* wab = Wds + (size_t)i0 * M + j0;      // element (i in a, j in b)
  float* wba = Wds + (size_t)j0 * M + i0;      // element (j in b, i in a)
#pragma unroll
  for (int i = 0; i < PS_RW; ++i) {
    const int p = wave + PS_WAVES * i;
    if (p < r && lane < c) {
      const float v = fast_exp2(L[i] - f[i] - g);
      const int ia = flip ? lane : p, jb = flip ? p : lane;
      wab[(size_t)ia * M + jb] = v;
      if (a != b) wba[(size_t)jb * M + ia] = v;
    }
  }
  if (pot) {
    __syncthreads();
    float* pt = pot + (size_t)blockIdx.x * iters * (cmax + 1);
    for (int e = tid; e < iters * (cmax + 1); e += 256) {
      const int it = e / (cmax + 1), x = e - it * (cmax + 1);
      const int n = (it & 1) ? c : r + (mult > 0 ? 1 : 0);
      if (x < n) pt[e] = plog[it * potld + x];
    }
  }
}

extern "C" int ttdg_pair_stage_fwd(const float* P, const float* Q, const float* w2, const float* b2, int H, ttdg_graphs_t gr,
                                   float tau, int iters, float* aff, float* Wds, float* pot, ttdg_stream_t stream) {
  TTDG_REQUIRE(P && Q && w2 && Wds && tau > 0.f, "pair_stage_fwd: bad arguments");
  TTDG_REQUIRE(iters >= 0 && iters <= SK_MAXK, "pair_stage_fwd: iters out of range");
  TTDG_REQUIRE(H > 0 && H % PS_BK == 0, "pair_stage_fwd: H must be a multiple of 64");
  if (int e = ttdg_validate_graphs(gr)) return e;
  int cmax = 0;
  for (int g = 0; g < gr.G; ++g) cmax = gr.off[g + 1] - gr.off[g] > cmax ? gr.off[g + 1] - gr.off[g] : cmax;
  TTDG_REQUIRE(cmax <= PS_T, "pair_stage_fwd: graphs of more than 64 nodes take ttdg_affinity_pairwise_fwd + ttdg_sinkhorn_pairs_fwd");
  hipLaunchKernelGGL(pair_stage_fwd_kernel, dim3(gr.G * (gr.G + 1) / 2), dim3(256), 0, (hipStream_t)stream, P, Q, w2, b2, H, gr, tau,
                     iters, aff, Wds, pot, cmax);
  return ttdg_launch_status("pair_stage_fwd");
}

// rows sweep of the backward, NR owned rows at most: S_p = sum_q dY_pq (DPP), dY_pq -= exp2(L_pq - f_p - g_q) S_p
template <int NR>
__device__ __forceinline__ void ps_bwd_rows(const float (&L)[PS_RW], float (&dY)[PS_RW], const float* frow, float gq, int r, int wave) {
  float S[NR], fp[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int p = wave + PS_WAVES * i;
    fp[i] = p < r ? frow[p] : 0.f;
    S[i] = wave_sum_f32_dpp(dY[i]);
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) dY[i] = (wave + PS_WAVES * i < r) ? dY[i] - fast_exp2(L[i] - fp[i] - gq) * S[i] : 0.f;
}

// ---- backward of the Sinkhorn half, same register layout ------------------------------------------------------------------
//   dY = dOut * out on the real rows, 0 on the dummy row; for k = K-1 .. 0 (potentials as of just after sweep k):
//     k even: S_p = sum_q dY_pq  (dummy row: sum_q dd_q);  dY_pq -= exp2(L_pq - f_p - g_q) S_p;  dd_q -= exp2(DUMMY - f_d - g_q) S_d
//     k odd : S_q = sum_p dY_pq + mult dd_q;               dY_pq -= exp2(L_pq - f_p - g_q) S_q;  dd_q -= exp2(DUMMY - f_d - g_q) S_q
//   dM = dY / tau   (statement and derivation: sk_backward in sinkhorn.hip)
__global__ __launch_bounds__(256) void pair_stage_bwd_kernel(const float* __restrict__ aff, const float* __restrict__ b2,
                                                             const float* __restrict__ pot, const float* __restrict__ dWds,
                                                             ttdg_graphs_t gr, float tau, int iters, float* __restrict__ dM, int cmax) {
  __shared__ float s_part[2][PS_WAVES * PS_T];
  __shared__ float plog[SK_MAXK * (PS_T + 1)];
  // pairs with a > b only: (1,0),(2,0),(2,1),...
  int a = 1, idx = blockIdx.x;
  while (idx >= a) { idx -= a; ++a; }
  const int b = idx;
  const int pair_fwd = a * (a + 1) / 2 + b;
  const int M = gr.off[gr.G];
  const int i0 = gr.off[a], j0 = gr.off[b];
  const int na = gr.off[a + 1] - i0, nb = gr.off[b + 1] - j0;
  const bool flip = nb < na;
  const int r = flip ? nb : na, c = flip ? na : nb, mult = c - r;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int potld = PS_T + 1;
  const float* pt = pot + (size_t)pair_fwd * iters * (cmax + 1);
  for (int e = tid; e < iters * (cmax + 1); e += 256) {
    const int it = e / (cmax + 1), x = e - it * (cmax + 1);
    if (x <= PS_T) plog[it * potld + x] = pt[e];
  }
  const float bias = b2 ? *b2 : 0.f, scale = TTDG_LOG2E / tau;
  // oriented element (p, q): a-node ia = flip ? q : p, b-node jb = flip ? p : q
  float L[PS_RW], dY[PS_RW];
#pragma unroll
  for (int i = 0; i < PS_RW; ++i) {
    const int p = wave + PS_WAVES * i;
    const bool ok = p < r && lane < c;
    const int ia = flip ? lane : p, jb = flip ? p : lane;
    L[i] = ok ? (aff[(size_t)(i0 + ia) * M + j0 + jb] + bias) * scale : -INFINITY;
    dY[i] = ok ? dWds[(size_t)(j0 + jb) * M + i0 + ia] : 0.f;      // the loss reads Wds[b-rows, a-cols] (:615-631)
  }
  __syncthreads();
  const int klast_row = (iters - 1) & ~1, klast_col = ((iters - 1) & 1) ? iters - 1 : iters - 2;
  const bool dummy = mult > 0;
  {
    const float gq = (klast_col >= 1 && lane < c) ? plog[klast_col * potld + lane] : 0.f;
#pragma unroll
    for (int i = 0; i < PS_RW; ++i) {
      const int p = wave + PS_WAVES * i;
      if (p < r) {
        const float fp = iters >= 1 ? plog[klast_row * potld + p] : 0.f;
        dY[i] *= fast_exp2(L[i] - fp - gq);          // lanes >= c: exp2(-inf) = 0 times 0
      }
    }
  }
  float dd = 0.f;     // dY of the dummy row at column `lane` (replicated in every wavefront)
  int buf = 0;
  const int nrw = (r - wave + PS_WAVES - 1) / PS_WAVES;
  for (int k = iters - 1; k >= 0; --k) {
    const bool rows = (k & 1) == 0;
    const int kf = rows ? k : k - 1, kg = rows ? k - 1 : k;      // logs holding f / g as of just after sweep k
    const float gq = (kg >= 0 && lane < c) ? plog[kg * potld + lane] : 0.f;
    const float fdum = (dummy && kf >= 0) ? plog[kf * potld + r] : 0.f;
    const float ed = (dummy && lane < c) ? fast_exp2(-fdum - gq) : 0.f;          // (fdum: logged without the fill; kf >= 0 for every k >= 0)
    if (rows) {
      // branch-free over the rows a wavefront owns (rows beyond r carry dY = 0, L = -inf: their update is 0 * S)
      if (nrw <= 4) ps_bwd_rows<4>(L, dY, plog + kf * potld, gq, r, wave);
      else if (nrw <= 8) ps_bwd_rows<8>(L, dY, plog + kf * potld, gq, r, wave);
      else if (nrw <= 12) ps_bwd_rows<12>(L, dY, plog + kf * potld, gq, r, wave);
      else ps_bwd_rows<16>(L, dY, plog + kf * potld, gq, r, wave);
      if (dummy) {
        const float S = wave_sum_f32_dpp(dd);
        dd -= ed * S;
      }
    } else {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < PS_RW; ++i)
        if (wave + PS_WAVES * i < r) acc += dY[i];
      s_part[buf][wave * PS_T + lane] = acc;
      ps_barrier();
      float S = 0.f;
#pragma unroll
      for (int w = 0; w < PS_WAVES; ++w) S += s_part[buf][w * PS_T + lane];
      buf ^= 1;
      S += (float)mult * dd;
#pragma unroll
      for (int i = 0; i < PS_RW; ++i) {
        const int p = wave + PS_WAVES * i;
        const float fp = (p < r && kf >= 0) ? plog[kf * potld + p] : 0.f;
        dY[i] = p < r ? dY[i] - fast_exp2(L[i] - fp - gq) * S : 0.f;
      }
      if (dummy) dd -= ed * S;
    }
  }
  const float inv_tau = 1.f / tau;
#pragma unroll
  for (int i = 0; i < PS_RW; ++i) {
    const int p = wave + PS_WAVES * i;
    if (p < r && lane < c) {
      const int ia = flip ? lane : p, jb = flip ? p : lane;
      dM[(size_t)(i0 + ia) * M + j0 + jb] = dY[i] * inv_tau;
    }
  }
}

extern "C" int ttdg_pair_stage_bwd(const float* aff, const float* b2, const float* pot, const float* dWds, ttdg_graphs_t gr, float tau,
                                   int iters, float* dM, ttdg_stream_t stream) {
  TTDG_REQUIRE(aff && pot && dWds && dM && tau > 0.f, "pair_stage_bwd: bad arguments");
  TTDG_REQUIRE(iters >= 0 && iters <= SK_MAXK, "pair_stage_bwd: iters out of range");
  if (int e = ttdg_validate_graphs(gr)) return e;
  int cmax = 0;
  for (int g = 0; g < gr.G; ++g) cmax = gr.off[g + 1] - gr.off[g] > cmax ? gr.off[g + 1] - gr.off[g] : cmax;
  TTDG_REQUIRE(cmax <= PS_T, "pair_stage_bwd: graphs of more than 64 nodes take ttdg_sinkhorn_pairs_bwd");
  const int npairs = gr.G * (gr.G - 1) / 2;
  if (npairs == 0) return 0;
  hipLaunchKernelGGL(pair_stage_bwd_kernel, dim3(npairs), dim3(256), 0, (hipStream_t)stream, aff, b2, pot, dWds, gr, tau, iters, dM, cmax);
  return ttdg_launch_status("pair_stage_bwd");
}
