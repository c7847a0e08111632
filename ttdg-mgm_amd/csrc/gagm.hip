// A6 + A7 — graduated-assignment multi-graph matching, the whole solve in ONE persistent workgroup.
// Reference: GA_GM.forward / gagm, multi_graph_matching.py:223-244, 300-389 (num_clusters == 1, so
// cluster_M == 1 and hung_iter is True), projector Sinkhorn (utils/sinkhorn.py -> pygmtools [3P]) then
// Hungarian (utils/hungarian.py -> scipy [3P]).
//
// What the reference does per iteration: UU^T (M x M), chain_matmul(A, UU^T, A, U), W U, then a 20-sweep
// Sinkhorn (20+ kernel launches) or G host round-trips to scipy, then two norm() host syncs; up to
// 5 x 200 Sinkhorn-stage + 200 Hungarian-stage iterations (data dependent).  Here nothing leaves the
// GPU and nothing is launched inside the loop:
//   B = A U (block diagonal)      S = U^T B (32x32)      V = (2q B S + W U) / G      [A (UU^T) A U == B S]
//   projection: one wavefront per graph, either a register-resident log-Sinkhorn (the n x 32 block is
//   held twice, column-wise and row-wise, so both sweeps are lane-local; dual potentials are exchanged
//   through wavefront-private LDS) or the on-device LAP of lap_device.h;
//   convergence: ||U - lastU|| < tol or U == lastU2 (exact 2-cycle), reduced in LDS.
// State (U, lastU, lastU2/B, V) lives in LDS when 4*M*32 floats fit, else in the L2-resident workspace.
#include "lap_device.h"

#define NU TTDG_UNIV
#define GA_HIST 256   /* states remembered for the cycle shortcut (>= max_iter to cover a whole stage) */
#define NEG_BIG (-INFINITY)


// ---- register-resident Sinkhorn projection of one graph block, one wavefront --------------------------
// V block: Vb[node*32 + univ], n nodes.  Oriented problem: r = min(n,32) rows, c = max(n,32) columns,
// (c - r) dummy rows.  CW = ceil(c / 64) columns per lane.
template <int CW>
__device__ __forceinline__ void sk_wave_project(const float* Vb, int n, float scale, int iters, float* Ub, float* fbuf, float* gbuf,
                                                bool square_tr = false) {
  const int lane = threadIdx.x & 63;
  const bool tr = n > NU || square_tr;   // rows = universe, cols = nodes (square_tr: a 32-node block kept transposed, see gagm_kernel)
  const int r = tr ? NU : n, c = tr ? n : NU, mult = c - r;
  const float D = -100.0f * TTDG_LOG2E;
  // column layout: lane owns columns q = lane + 64*w; Lc[w][p]
  float Lc[CW][NU];
#pragma unroll
  for (int w = 0; w < CW; ++w) {
    const int q = lane + 64 * w;
#pragma unroll
    for (int p = 0; p < NU; ++p) {
      float v = NEG_BIG;
      if (q < c && p < r) v = (tr ? Vb[q * NU + p] : Vb[p * NU + q]) * scale;
      Lc[w][p] = v;
    }
  }
  // row layout: lane (p = lane & 31, h = lane >> 5) owns row p, columns q in [h*32*CW, (h+1)*32*CW)
  const int rp = lane & 31, rh = lane >> 5;
  const int qbase = rh * 32 * CW;
  float Lr[32 * CW];
#pragma unroll
  for (int k = 0; k < 32 * CW; ++k) {
    const int q = qbase + k;
    float v = NEG_BIG;
    if (q < c && rp < r) v = (tr ? Vb[q * NU + rp] : Vb[rp * NU + q]) * scale;
    Lr[k] = v;
  }
  // potentials: g lives in the column lanes (registers) and in gbuf; f in fbuf[0..31], dummy in fbuf[32]
  float gq[CW];
#pragma unroll
  for (int w = 0; w < CW; ++w) { gq[w] = 0.f; if (lane + 64 * w < 64 * CW) gbuf[lane + 64 * w] = 0.f; }
  if (lane < 33) fbuf[lane] = 0.f;
  wave_sync();

  for (int it = 0; it < iters; ++it) {
    if ((it & 1) == 0) {
      // rows: f_p = lse_q(L_pq - g_q), lane-local over the lane's half row, then combine the two halves
      float m0 = NEG_BIG, m1 = NEG_BIG, m2 = NEG_BIG, m3 = NEG_BIG;
#pragma unroll
      for (int k = 0; k < 32 * CW; k += 4) {
        const float4 g4 = *reinterpret_cast<const float4*>(gbuf + qbase + k);
        m0 = fmaxf(m0, Lr[k] - g4.x); m1 = fmaxf(m1, Lr[k + 1] - g4.y); m2 = fmaxf(m2, Lr[k + 2] - g4.z); m3 = fmaxf(m3, Lr[k + 3] - g4.w);
      }
      float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
      m = fmaxf(m, other_half(m));
      const float ms = (m == NEG_BIG) ? 0.f : m;   // rows >= r hold no finite entry
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int k = 0; k < 32 * CW; k += 4) {
        const float4 g4 = *reinterpret_cast<const float4*>(gbuf + qbase + k);
        s0 += fast_exp2(Lr[k] - g4.x - ms); s1 += fast_exp2(Lr[k + 1] - g4.y - ms);
        s2 += fast_exp2(Lr[k + 2] - g4.z - ms); s3 += fast_exp2(Lr[k + 3] - g4.w - ms);
      }
      float s = (s0 + s1) + (s2 + s3);
      s += other_half(s);
      if (lane < 32) fbuf[lane] = (lane < r) ? ms + fast_log2(s) : 0.f;
      if (mult > 0) {
        // dummy row: fd = D + lse_q(-g_q) over the c real columns
        float dm = NEG_BIG;
#pragma unroll
        for (int w = 0; w < CW; ++w) if (lane + 64 * w < c) dm = fmaxf(dm, -gq[w]);
        dm = wave_max_f32_dpp(dm);
        float ds = 0.f;
#pragma unroll
        for (int w = 0; w < CW; ++w) if (lane + 64 * w < c) ds += fast_exp2(-gq[w] - dm);
        ds = wave_sum_f32_dpp(ds);
        if (lane == 0) fbuf[32] = D + dm + fast_log2(ds);
      }
    } else {
      // cols: g_q = lse over real rows p < r and `mult` dummy rows, lane-local
      const float td = (mult > 0) ? D - fbuf[32] : NEG_BIG;
#pragma unroll
      for (int w = 0; w < CW; ++w) {
        float m0 = td, m1 = NEG_BIG, m2 = NEG_BIG, m3 = NEG_BIG;
#pragma unroll
        for (int p = 0; p < NU; p += 4) {
          const float4 f4 = *reinterpret_cast<const float4*>(fbuf + p);
          m0 = fmaxf(m0, Lc[w][p] - f4.x); m1 = fmaxf(m1, Lc[w][p + 1] - f4.y);
          m2 = fmaxf(m2, Lc[w][p + 2] - f4.z); m3 = fmaxf(m3, Lc[w][p + 3] - f4.w);
        }
        const float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        const float ms = (m == NEG_BIG) ? 0.f : m;
        float s0 = (mult > 0) ? (float)mult * fast_exp2(td - ms) : 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int p = 0; p < NU; p += 4) {
          const float4 f4 = *reinterpret_cast<const float4*>(fbuf + p);
          s0 += fast_exp2(Lc[w][p] - f4.x - ms); s1 += fast_exp2(Lc[w][p + 1] - f4.y - ms);
          s2 += fast_exp2(Lc[w][p + 2] - f4.z - ms); s3 += fast_exp2(Lc[w][p + 3] - f4.w - ms);
        }
        const float s = (s0 + s1) + (s2 + s3);
        const int q = lane + 64 * w;
        gq[w] = (q < c) ? ms + fast_log2(s) : 0.f;
        gbuf[q] = gq[w];
      }
    }
    wave_sync();
  }
  // U = exp(L - f - g) in the row layout (contiguous stores when rows are nodes)
  const float fp = fbuf[rp];
#pragma unroll
  for (int k = 0; k < 32 * CW; ++k) {
    const int q = qbase + k;
    if (q < c && rp < r) {
      const float v = fast_exp2(Lr[k] - fp - gbuf[q]);
      if (tr) Ub[q * NU + rp] = v; else Ub[rp * NU + q] = v;
    }
  }
}

// ---- narrow projection: n <= 32 nodes (the common case: the sampler yields ~20-35 nodes per image) -------------------
// 32 x 32 problem after dummy rows; every lane owns 16 entries in BOTH layouts (row = lane & 31, half = lane >> 5), the
// dummy row lives in the free row slot n (its column-layout copy carries +log2(multiplicity)), so both sweeps are the
// same 16-element loop with no special cases.  A single wavefront issues ~1 instruction per 4-5 cycles, so the loop is
// written for instruction count: packed fp32 subtracts/adds, and NO max pass - after any sweep y = L - f - g <= 0, so
// the previous potential is a valid stabiliser (exp2 arguments <= 0, sums in (0, 32]); the exact two-pass form is used
// for the first sweep and whenever a sum underflows (< 2^-60).
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sk_line_exact(const f32x2 (&t)[8], bool used) {
  float m = NEG_BIG;
#pragma unroll
  for (int k = 0; k < 8; ++k) m = fmaxf(m, fmaxf(t[k].x, t[k].y));
  m = fmaxf(m, other_half(m));
  const float ms = used ? m : 0.f;
  f32x2 acc = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 8; ++k) { const f32x2 a = t[k] - ms; acc += (f32x2){fast_exp2(a.x), fast_exp2(a.y)}; }
  float s = acc.x + acc.y;
  s += other_half(s);
  return used ? ms + fast_log2(s) : 0.f;
}

__device__ __forceinline__ void sk_wave_project_narrow(const float* Vb, int n, float scale, int iters, float* Ub, float* fbuf,
                                                       float* gbuf) {
  const int lane = threadIdx.x & 63, lo = lane & 31, hi = lane >> 5;
  const int mult = NU - n;
  const float D = -100.0f * TTDG_LOG2E;
  const int nrow = n + (mult > 0 ? 1 : 0);
  const bool row_used = lo < nrow;
  f32x2 Lr[8], Lc[8];
  const float dcol = (mult > 0) ? D + fast_log2((float)mult) : NEG_BIG;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int q = hi * 16 + k;                                  // row layout: row lo, column q
    const float vr = (lo < n) ? Vb[lo * NU + q] * scale : ((lo == n && mult > 0) ? D : NEG_BIG);
    const int p = hi * 16 + k;                                  // column layout: column lo, row p
    const float vc = (p < n) ? Vb[p * NU + lo] * scale : (p == n ? dcol : NEG_BIG);
    if (k & 1) { Lr[k >> 1].y = vr; Lc[k >> 1].y = vc; } else { Lr[k >> 1].x = vr; Lc[k >> 1].x = vc; }
  }
  if (lane < 32) { fbuf[lane] = 0.f; gbuf[lane] = 0.f; }
  float fown = 0.f, gown = 0.f;
  wave_sync();
  for (int it = 0; it < iters; ++it) {
    const bool rows = (it & 1) == 0;
    const float* pot = (rows ? gbuf : fbuf) + hi * 16;          // the OTHER side's potentials for my 16 entries
    f32x2 t[8];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const float4 p4 = *reinterpret_cast<const float4*>(pot + 2 * k);
      t[k] = (rows ? Lr[k] : Lc[k]) - (f32x2){p4.x, p4.y};
      t[k + 1] = (rows ? Lr[k + 1] : Lc[k + 1]) - (f32x2){p4.z, p4.w};
    }
    const bool used = rows ? row_used : true;
    const float own = rows ? fown : gown;
    float fresh;
    bool exact = (it == 0);
    if (!exact) {
      f32x2 acc = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 8; ++k) { const f32x2 a = t[k] - own; acc += (f32x2){fast_exp2(a.x), fast_exp2(a.y)}; }
      float sm = acc.x + acc.y;
      sm += other_half(sm);
      fresh = used ? own + fast_log2(sm) : 0.f;
      exact = __ballot(used && !(sm >= 8.6736174e-19f && sm <= 1.0e6f)) != 0ull;      // 2^-60: underflow guard
    }
    if (exact) fresh = sk_line_exact(t, used);
    if (rows) { fown = fresh; if (hi == 0) fbuf[lo] = fresh; }
    else      { gown = fresh; if (hi == 0) gbuf[lo] = fresh; }
    wave_sync();
  }
  // U[p][lo] = exp(L - f_p - g_lo) from the column layout (conflict-free stores); the dummy slot is not stored
  {
    const float* fp = fbuf + hi * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int p = hi * 16 + k;
      const float l = (k & 1) ? Lc[k >> 1].y : Lc[k >> 1].x;
      if (p < n) Ub[p * NU + lo] = fast_exp2(l - fp[k] - gown);
    }
  }
}

// ---- block-layout projection (graphs of up to 64 nodes): ONE copy of the matrix, no LDS, one exp per entry per sweep PAIR ------
// Lane (bi = lane >> 3, bj = lane & 7) owns a 4 x CB block: rows 4 bi .. 4 bi + 3, CB columns.  Row sums are butterflies over
// the three low lane bits (DPP quad_perm x 2, row_half_mirror), column sums over the three high bits (DPP row_ror:8,
// v_permlane16_swap, v_permlane32_swap): every lane ends up with the potentials of its own rows and columns, so nothing is
// exchanged through LDS and each sweep is one short dependency chain.  A sweep pair evaluates e = exp2(L - f - g) once: the
// row sweep yields s_a = sum_b e and f_a += log2 s_a, and the column sweep needs exp2(L - f_new - g) = e / s_a - a multiply
// by v_rcp_f32 instead of a second transcendental - so c_b = sum_a w_a e / s_a, g_b += log2 c_b.  The potentials are the state
// (every pair starts from L, f, g: nothing drifts).  As in the other register projectors the previous potentials stabilise the
// exponentials; the first pair, and any pair in which a sum leaves [2^-60, 1e6], runs in the exact max-subtracted form.
//   kTr = false (n <= 32): rows = nodes, the dummy row sits in row slot n with weight (32 - n) in the column sums; CB = 4
//   kTr = true  (32 <= n <= 64): rows = universe slots, columns q = bj + 8 b (b < ceil(n / 8)), the (n - 32) identical dummy
//                                rows are one extra, wavefront-replicated row; CB = ceil(n / 8)
template <bool kTr, int CB>
__device__ __forceinline__ void sk_wave_project_blk(const float* Vb, int n, float scale, int iters, float* Ub) {
  const int lane = threadIdx.x & 63, bi = lane >> 3, bj = lane & 7;
  const float D = -100.0f * TTDG_LOG2E;
  const int c = kTr ? n : NU;                       // columns of the oriented problem
  const int mult = kTr ? n - NU : NU - n;           // identical dummy rows
  constexpr int cbu = CB;                            // kTr: the caller instantiates CB = ceil(n / 8)
  float L[4][CB], f[4], g[CB], w[4];
  bool rused[4], cused[CB];
#pragma unroll
  for (int b = 0; b < CB; ++b) { cused[b] = kTr ? (bj + 8 * b < c) : true; g[b] = 0.f; }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int p = 4 * bi + a;
    f[a] = 0.f;
    if (kTr) { rused[a] = true; w[a] = 1.f; }
    else { rused[a] = p < n || (p == n && mult > 0); w[a] = p < n ? 1.f : (rused[a] ? (float)mult : 0.f); }
  }
  if (kTr) {
#pragma unroll
    for (int b = 0; b < CB; ++b) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < cbu && cused[b]) v = *reinterpret_cast<const float4*>(Vb + (bj + 8 * b) * NU + 4 * bi);
      const bool u = b < cbu && cused[b];
      L[0][b] = u ? v.x * scale : NEG_BIG; L[1][b] = u ? v.y * scale : NEG_BIG;
      L[2][b] = u ? v.z * scale : NEG_BIG; L[3][b] = u ? v.w * scale : NEG_BIG;
    }
  } else {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int p = 4 * bi + a;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < n) v = *reinterpret_cast<const float4*>(Vb + p * NU + 4 * bj);
      const float fill = rused[a] ? D : NEG_BIG;
      L[a][0] = p < n ? v.x * scale : fill; L[a][1] = p < n ? v.y * scale : fill;
      L[a][2] = p < n ? v.z * scale : fill; L[a][3] = p < n ? v.w * scale : fill;
    }
  }
  float fd = 0.f;                                    // kTr: potential of the dummy rows
  const bool dummy = kTr && mult > 0;
  for (int it = 0; it < iters; it += 2) {
    const bool cols_too = it + 1 < iters;
    float fn[4], gn[CB], fdn = fd;
    bool exact = it == 0;
    if (!exact) {
      float e[4][CB], s[4], ed[CB], sd = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < CB; ++b)
          if (b < cbu) { e[a][b] = fast_exp2((L[a][b] - f[a]) - g[b]); acc += e[a][b]; }
        s[a] = bj_sum(acc);
      }
      // every sum that is used must stay inside [2^-60, 1e6]: tracked as one running minimum and maximum (a NaN can only come
      // from 0 * inf, i.e. after a zero or infinite sum that the two bounds have already caught)
      float lo = 1.f, hi = 1.f;
      if (dummy) {
#pragma unroll
        for (int b = 0; b < CB; ++b)
          if (b < cbu) { ed[b] = cused[b] ? fast_exp2((D - fd) - g[b]) : 0.f; sd += ed[b]; }
        sd = bj_sum(sd);
        lo = sd; hi = sd;
        fdn = fd + fast_log2(sd);
      }
      float r[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float su = rused[a] ? s[a] : 1.f;
        lo = fminf(lo, su); hi = fmaxf(hi, su);
        fn[a] = rused[a] ? f[a] + fast_log2(s[a]) : 0.f;
        r[a] = rused[a] ? w[a] * __builtin_amdgcn_rcpf(s[a]) : 0.f;
      }
      if (cols_too) {
        const float rd = dummy ? (float)mult * __builtin_amdgcn_rcpf(sd) : 0.f;
#pragma unroll
        for (int b = 0; b < CB; ++b)
          if (b < cbu) {
            float acc = (e[0][b] * r[0] + e[1][b] * r[1]) + (e[2][b] * r[2] + e[3][b] * r[3]);
            acc = bi_sum(acc);
            if (dummy) acc += ed[b] * rd;
            const float cu = cused[b] ? acc : 1.f;
            lo = fminf(lo, cu); hi = fmaxf(hi, cu);
            gn[b] = cused[b] ? g[b] + fast_log2(acc) : 0.f;
          }
      }
      exact = __ballot(!(lo >= 8.6736174e-19f && hi <= 1.0e6f)) != 0ull;
    }
    if (exact) {
      // rows, max-subtracted
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float m = NEG_BIG;
#pragma unroll
        for (int b = 0; b < CB; ++b) if (b < cbu) m = fmaxf(m, L[a][b] - g[b]);
        m = bj_max(m);
        const float ms = (m == NEG_BIG) ? 0.f : m;
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < CB; ++b) if (b < cbu) acc += fast_exp2((L[a][b] - g[b]) - ms);
        acc = bj_sum(acc);
        fn[a] = rused[a] ? ms + fast_log2(acc) : 0.f;
      }
      if (dummy) {
        float m = NEG_BIG;
#pragma unroll
        for (int b = 0; b < CB; ++b) if (b < cbu && cused[b]) m = fmaxf(m, -g[b]);
        m = bj_max(m);
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < CB; ++b) if (b < cbu && cused[b]) acc += fast_exp2(-g[b] - m);
        acc = bj_sum(acc);
        fdn = D + m + fast_log2(acc);
      }
      if (cols_too) {
        const float td = dummy ? D - fdn : NEG_BIG;
#pragma unroll
        for (int b = 0; b < CB; ++b)
          if (b < cbu) {
            float m = fmaxf(fmaxf(L[0][b] - fn[0], L[1][b] - fn[1]), fmaxf(L[2][b] - fn[2], L[3][b] - fn[3]));
            m = fmaxf(bi_max(m), td);
            const float ms = (m == NEG_BIG) ? 0.f : m;
            float acc = (w[0] * fast_exp2((L[0][b] - fn[0]) - ms) + w[1] * fast_exp2((L[1][b] - fn[1]) - ms)) +
                        (w[2] * fast_exp2((L[2][b] - fn[2]) - ms) + w[3] * fast_exp2((L[3][b] - fn[3]) - ms));
            acc = bi_sum(acc);
            if (dummy) acc += (float)mult * fast_exp2(td - ms);
            gn[b] = cused[b] ? ms + fast_log2(acc) : 0.f;
          }
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) f[a] = fn[a];
    fd = fdn;
    if (cols_too) {
#pragma unroll
      for (int b = 0; b < CB; ++b) if (b < cbu) g[b] = gn[b];
    }
  }
  // U = exp(L - f - g)
  if (kTr) {
#pragma unroll
    for (int b = 0; b < CB; ++b)
      if (b < cbu && cused[b]) {
        float4 o;
        o.x = fast_exp2((L[0][b] - f[0]) - g[b]); o.y = fast_exp2((L[1][b] - f[1]) - g[b]);
        o.z = fast_exp2((L[2][b] - f[2]) - g[b]); o.w = fast_exp2((L[3][b] - f[3]) - g[b]);
        *reinterpret_cast<float4*>(Ub + (bj + 8 * b) * NU + 4 * bi) = o;
      }
  } else {
#pragma unroll
    for (int a = 0; a < 4; ++a)
      if (4 * bi + a < n) {
        float4 o;
        o.x = fast_exp2((L[a][0] - f[a]) - g[0]); o.y = fast_exp2((L[a][1] - f[a]) - g[1]);
        o.z = fast_exp2((L[a][2] - f[a]) - g[2]); o.w = fast_exp2((L[a][3] - f[a]) - g[3]);
        *reinterpret_cast<float4*>(Ub + (4 * bi + a) * NU + 4 * bj) = o;
      }
  }
}

// one instantiation per number of column groups in use: no guards inside the sweeps
__device__ __forceinline__ void sk_project_blk(const float* Vb, int n, bool rows_are_nodes, float scale, int iters, float* Ub) {
  if (rows_are_nodes) { sk_wave_project_blk<false, 4>(Vb, n, scale, iters, Ub); return; }
  switch ((n + 7) >> 3) {
    case 4: sk_wave_project_blk<true, 4>(Vb, n, scale, iters, Ub); break;      // the 32-node block that stays transposed
    case 5: sk_wave_project_blk<true, 5>(Vb, n, scale, iters, Ub); break;
    case 6: sk_wave_project_blk<true, 6>(Vb, n, scale, iters, Ub); break;
    case 7: sk_wave_project_blk<true, 7>(Vb, n, scale, iters, Ub); break;
    default: sk_wave_project_blk<true, 8>(Vb, n, scale, iters, Ub); break;
  }
}

// Convergence terms of one graph block (cnt = 32 n values), by the wavefront that has just projected it: adds the block's share
// of ||U - lastU||^2 and ||U - lastU2||^2, then lastU2 <- lastU.  lastU2 is in global memory: a batch of float4 loads is issued
// before anything depends on it (one L2 round trip per batch).
__device__ __forceinline__ void conv_block(const float* Unew, const float* Uold, float* Uprev2, float* U1snap, int cnt, float& d1,
                                           float& d2) {
  const int lane = threadIdx.x & 63;
  const int n4 = cnt >> 2;
  const float4* N4 = reinterpret_cast<const float4*>(Unew);
  const float4* O4 = reinterpret_cast<const float4*>(Uold);
  float4* P4 = reinterpret_cast<float4*>(Uprev2);
  float4* S4 = reinterpret_cast<float4*>(U1snap);
  for (int e0 = 0; e0 < n4; e0 += 64 * 4) {
    float4 p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + q * 64 + lane;
      p[q] = P4[min(e, n4 - 1)];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + q * 64 + lane;
      if (e < n4) {
        const float4 un = N4[e], uc = O4[e];
        if (U1snap) S4[e] = un;
        float a, b;
        a = un.x - uc.x; b = un.x - p[q].x; d1 = fmaf(a, a, d1); d2 = fmaf(b, b, d2);
        a = un.y - uc.y; b = un.y - p[q].y; d1 = fmaf(a, a, d1); d2 = fmaf(b, b, d2);
        a = un.z - uc.z; b = un.z - p[q].z; d1 = fmaf(a, a, d1); d2 = fmaf(b, b, d2);
        a = un.w - uc.w; b = un.w - p[q].w; d1 = fmaf(a, a, d1); d2 = fmaf(b, b, d2);
        P4[e] = uc;
      }
    }
  }
}

template <int GA_WAVES>
__device__ __forceinline__ float block_sum2(float a, float b, float* red, float& outb) {
  a = wave_sum_f32_dpp(a);
  b = wave_sum_f32_dpp(b);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) { red[wave] = a; red[GA_WAVES + wave] = b; }
  __syncthreads();
  float sa = 0.f, sb = 0.f;
#pragma unroll
  for (int w = 0; w < GA_WAVES; ++w) { sa += red[w]; sb += red[GA_WAVES + w]; }
  outb = sb;
  return sa;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// workspace layout (floats): [V0: M*32][U1: M*32][W^T: M*Mp][state: 4*32*(Mp+1)][history: GA_HIST*M bytes]
// leading dimension of W^T / B^T: M rounded up to the 16-row MFMA tile (round 1's 32-row tiles needed 32: a 129-node
// multi-graph then paid for 160 columns of W^T and fell out of LDS)
__host__ __device__ inline int ga_mp(int M) { return (M + 15) & ~15; }
__host__ __device__ inline size_t ga_ws_hist_off(int M) {
  const int Mp = ga_mp(M);
  return (size_t)2 * M * NU + (size_t)M * Mp + (size_t)4 * NU * (Mp + 1);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- MFMA tiles of the three products (v_mfma_f32_16x16x4_f32, exact fp32) ------------------------------------------------
// Operand mapping: A[i][k] in lane (i = lane & 15, k = lane >> 4), B[k][j] likewise, D[4 * (lane >> 4) + r][lane & 15].
// All three walk K in blocks of 16 (four MFMA k-steps) with the operands of the NEXT block requested before the MFMAs of the
// current one are issued (two register sets, ping-pong): left to itself the compiler schedules load - wait - MFMA pair by pair
// (an LDS round trip per k-step: 170 cycles for 64 cycles of matrix work); __builtin_amdgcn_sched_barrier pins the order.
// Addresses past the end of K are clamped and the operand zeroed by select: no branches, one wait per block.
struct MmBlk { float a[4], b0[4], b1[4]; };

__device__ __forceinline__ void mm_issue(const MmBlk& x, f32x4& acc0, f32x4& acc1) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.a[q], x.b0[q], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.a[q], x.b1[q], acc1, 0, 0, 0);
  }
}

// K loop shared by the tiles: load(blk, k0) requests one block of raw operands (clamped addresses, nothing depends on the
// values yet); fix(blk, k0) zeroes / scales the A operand of the k-steps past the end right before the block's MFMAs, so the
// wait for a block's loads sits AFTER the previous block's matrix work.  One zero operand is enough: B stays raw.
template <class Load, class Fix>
__device__ __forceinline__ void mm_k_loop(int K, Load load, Fix fix, f32x4& acc0, f32x4& acc1) {
  MmBlk p, q;
  load(p, 0);
  for (int k0 = 0;;) {
    if (k0 + 16 < K) load(q, k0 + 16);
    __builtin_amdgcn_sched_barrier(0);
    fix(p, k0);
    mm_issue(p, acc0, acc1);
    __builtin_amdgcn_sched_barrier(0);
    k0 += 16;
    if (k0 >= K) break;
    if (k0 + 16 < K) load(p, k0 + 16);
    __builtin_amdgcn_sched_barrier(0);
    fix(q, k0);
    mm_issue(q, acc0, acc1);
    __builtin_amdgcn_sched_barrier(0);
    k0 += 16;
    if (k0 >= K) break;
  }
}

// V rows [i0, i0 + 16): (2q * B S + W U) / G; two 16-column accumulators share the A operand.  16-row tiles: a multi-graph of
// ~120 nodes is 8 tiles = one per wavefront (round 1's 32-row tiles kept half of the workgroup idle).
//   W U : A operand = W^T stored [k][i] (leading dimension Mp, zero padded): 16 consecutive words per k; B operand = U[k][u]
//   B S : A operand = B^T stored [v][i] (leading dimension Mp + 1); B operand = S[v][u], summed here from its KS K-split partials
template <int KS>
__device__ __forceinline__ void v_row_tile16(int i0, int M, int Mp, const float* WT, const float* Ucur, const float* BT,
                                             const float* S, float qw2, float invG, float* V, float* V0snap) {
  const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  mm_k_loop(M, [&](MmBlk& x, int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kc = min(k0 + 4 * q + kq, M - 1);
      x.a[q] = WT[(size_t)kc * Mp + i0 + li];
      x.b0[q] = Ucur[kc * NU + li];
      x.b1[q] = Ucur[kc * NU + 16 + li];
    }
  }, [&](MmBlk& x, int k0) {
    if (k0 + 16 > M) {
#pragma unroll
      for (int q = 0; q < 4; ++q) x.a[q] = (k0 + 4 * q + kq < M) ? x.a[q] : 0.f;
    }
  }, acc0, acc1);
  const int ldb = Mp + 1;
  const bool rok = i0 + li < M;
  mm_k_loop(NU, [&](MmBlk& x, int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + 4 * q + kq;
      x.a[q] = BT[k * ldb + i0 + li];                // i0 + li < Mp: inside the buffer; rows >= M are not B
      x.b0[q] = S[k * NU + li];
      x.b1[q] = S[k * NU + 16 + li];
      if (KS == 2) { x.b0[q] += S[NU * NU + k * NU + li]; x.b1[q] += S[NU * NU + k * NU + 16 + li]; }
    }
  }, [&](MmBlk& x, int) {
#pragma unroll
    for (int q = 0; q < 4; ++q) x.a[q] = rok ? x.a[q] * qw2 : 0.f;
  }, acc0, acc1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = i0 + 4 * kq + r;
    if (row < M) {
      const float v0 = acc0[r] * invG, v1 = acc1[r] * invG;
      V[row * NU + li] = v0;
      V[row * NU + 16 + li] = v1;
      if (V0snap) { V0snap[row * NU + li] = v0; V0snap[row * NU + 16 + li] = v1; }
    }
  }
}

// One 16 x 16 tile (ti, tj) of S = U^T B over the node rows [rbeg, rend): A operand = U[r][u] (16 consecutive words per r),
// B operand = B[r][v] read from the transposed store X[v][r] (odd leading dimension: conflict-free).  Even and odd k-steps
// go to two accumulators (two independent MFMA chains), added at the end.
__device__ __forceinline__ void s_tile16(int ti, int tj, int rbeg, int rend, int ldb, const float* Ucur, const float* X, float* Sout) {
  const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  struct Blk { float a[4], b[4]; };
  const int K = rend - rbeg;
  auto load = [&](Blk& x, int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rc = min(rbeg + k0 + 4 * q + kq, rend - 1);
      x.a[q] = Ucur[rc * NU + ti * 16 + li];
      x.b[q] = X[(tj * 16 + li) * ldb + rc];
    }
  };
  auto issue = [&](Blk& x, int k0) {
    if (k0 + 16 > K) {
#pragma unroll
      for (int q = 0; q < 4; ++q) x.a[q] = (k0 + 4 * q + kq < K) ? x.a[q] : 0.f;
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.a[0], x.b[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.a[1], x.b[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.a[2], x.b[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.a[3], x.b[3], acc1, 0, 0, 0);
  };
  Blk p, q;
  load(p, 0);
  for (int k0 = 0;;) {
    if (k0 + 16 < K) load(q, k0 + 16);
    __builtin_amdgcn_sched_barrier(0);
    issue(p, k0);
    __builtin_amdgcn_sched_barrier(0);
    k0 += 16;
    if (k0 >= K) break;
    if (k0 + 16 < K) load(p, k0 + 16);
    __builtin_amdgcn_sched_barrier(0);
    issue(q, k0);
    __builtin_amdgcn_sched_barrier(0);
    k0 += 16;
    if (k0 >= K) break;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) Sout[(ti * 16 + 4 * kq + r) * NU + tj * 16 + li] = acc0[r] + acc1[r];
}

// B_g = A_g U_g, rows [i0, i0 + 16) of graph g (n nodes starting at node o), stored transposed X[u][node] (leading
// dimension ldb).  kAT: the A block is the TRANSPOSED copy staged in LDS (At[k][i]: 16 consecutive words per k),
// otherwise the caller's row-major block in global memory.
template <bool kAT>
__device__ __forceinline__ void b_row_tile16(int o, int n, int i0, const float* Ag, const float* Ucur, float* X, int ldb) {
  const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int i = i0 + li, ic = min(i, n - 1);
  const bool iok = i < n;
  mm_k_loop(n, [&](MmBlk& x, int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kc = min(k0 + 4 * q + kq, n - 1);
      x.a[q] = kAT ? Ag[kc * n + ic] : Ag[ic * n + kc];
      x.b0[q] = Ucur[(o + kc) * NU + li];
      x.b1[q] = Ucur[(o + kc) * NU + 16 + li];
    }
  }, [&](MmBlk& x, int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) x.a[q] = (iok && k0 + 4 * q + kq < n) ? x.a[q] : 0.f;
  }, acc0, acc1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = i0 + 4 * kq + r;
    if (row < n) {
      X[li * ldb + o + row] = acc0[r];
      X[(16 + li) * ldb + o + row] = acc1[r];
    }
  }
}

// 512 threads (8 wavefronts, 2 per SIMD): the register-resident Sinkhorn block needs the 256-VGPR budget.
// CWMAX 1: graphs up to 64 nodes; CWMAX 2: up to 128 nodes.
// kLds: solver state (U, lastU, lastU2/B^T, V) in LDS; kWLds / kALds: W^T / the packed A blocks in LDS as well
// (both are constants of the solve: staged once, read every iteration).
template <bool kLds, bool kWLds, bool kALds, int GA_THREADS, int CWMAX>
__global__ __launch_bounds__(GA_THREADS) void gagm_kernel(const float* __restrict__ Apack, const float* __restrict__ W,
                                                          const float* __restrict__ U0, ttdg_graphs_t gr,
                                                          ttdg_gagm_cfg_t cfg, float* __restrict__ Uout,
                                                          int32_t* __restrict__ info, float* __restrict__ ws, int cmaxp, int flags) {
  extern __shared__ __attribute__((aligned(16))) float ga_smem[];
  constexpr int GA_WAVES = GA_THREADS / 64;
  const int M = gr.off[gr.G], G = gr.G;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int MU = M * NU;
  const int Mp = ga_mp(M);
  const int SB = NU * (Mp + 1);              // one state buffer: M x 32 row-major, or 32 x (Mp+1) for B^T
  int asz = 0, nmax = 0, nmin = 1 << 30;
  for (int g = 0; g < G; ++g) { const int n = gr.off[g + 1] - gr.off[g]; asz += n * n; nmax = max(nmax, n); nmin = min(nmin, n); }
  // Sinkhorn orientation of a 32-node block: in a batch of unequal sizes whose largest
  // graph exceeds the universe the whole padded batch is transposed and only blocks with n_g < 32 are transposed back,
  // so an exactly-32-node block keeps rows = universe; with equal sizes (or no graph above 32) rows = nodes
  const bool sq_tr = (nmax > NU) && (nmin != nmax);
  // workspace (global): [V0 snapshot MU][first projected U, MU][W^T: M x Mp][state 4*SB when !kLds]
  float* V0snap = ws;
  float* U1snap = ws + MU;
  float* WTg = ws + 2 * MU;
  float* lds = ga_smem;
  float* base = kLds ? lds : WTg + (size_t)M * Mp;
  if (kLds) lds += 3 * SB;
  // lastU2 is only touched by the convergence check: it lives in the (L2-resident) workspace, read and rewritten by the
  // wavefront that has just projected the graph (conv_block) - round 1 kept it in 16 registers per thread, which the register
  // allocator spilled to scratch around the projection and reloaded one dependent access at a time (11k cycles per iteration)
  float* Ucur = base;
  float* X = base + SB;
  float* V = base + 2 * SB;
  float* Uprev_g = WTg + (size_t)M * Mp + (size_t)3 * SB;
  const float* WT = WTg;
  const float* Ap = Apack;
  if (kWLds) { WT = lds; lds += M * Mp; }
  if (kALds) { Ap = lds; lds += (asz + 3) & ~3; }
  float* Spart = lds;            // 2 x 1024: the K-split partial tiles of S (the V tiles add them)
  float* red = Spart + 2 * NU * NU;  // 64
  float* wex = red + 64;         // GA_WAVES * (40 + cmaxp)
  const int wex_stride = 40 + cmaxp;
  unsigned char* lapb = (unsigned char*)(wex + GA_WAVES * wex_stride);   // LDS LAP scratch: CWMAX == 2 only
  const size_t lap_stride = (CWMAX == 2) ? ((lap_scratch_bytes(NU, cmaxp) + 15) & ~(size_t)15) : 0;
  unsigned char* s_gid = lapb + GA_WAVES * lap_stride;                    // node -> graph, M bytes
  unsigned long long* hmatch = (unsigned long long*)(s_gid + ((M + 15) & ~15));   // scratch words of the cycle shortcut
  unsigned long long* hhash = hmatch + 4;                                         // GA_HIST state hashes
  unsigned char* hist = (unsigned char*)(ws + ga_ws_hist_off(M));              // GA_HIST x M state codes, in the L2-resident workspace

  __shared__ int s_off[TTDG_MAX_GRAPHS + 4];     // 2 x 68 ints: the static LDS total stays a multiple of 16 B (dynamic base alignment)
  __shared__ int s_aoff[TTDG_MAX_GRAPHS + 4];   // start of graph g's block in Apack
  if (tid <= G) s_off[tid] = gr.off[tid];
  if (tid == 0) {
    int a = 0;
    for (int g = 0; g < G; ++g) { s_aoff[g] = a; const int n = gr.off[g + 1] - gr.off[g]; a += n * n; }
  }
  for (int r = tid; r < M; r += GA_THREADS) s_gid[r] = (unsigned char)graph_of(gr, r);
  for (int e = tid; e < MU; e += GA_THREADS) { Ucur[e] = U0[e]; Uprev_g[e] = 0.f; }   // lastU = zeros (:305)
  {  // W^T[k][i] = W[i][k], zero padded to Mp columns; A blocks
    float* wt = kWLds ? const_cast<float*>(WT) : WTg;
    for (int e = tid; e < M * Mp; e += GA_THREADS) {
      const int k = e / Mp, i = e - k * Mp;
      wt[e] = (i < M) ? W[(size_t)i * M + k] : 0.f;
    }
    if (kALds) {   // every block TRANSPOSED (At[k][i]): the MFMA A operand of B = A U reads 16 consecutive words per k
      float* ap = const_cast<float*>(Ap);
      int a0 = 0;
      for (int g = 0; g < G; ++g) {
        const int n = gr.off[g + 1] - gr.off[g];
        for (int e = tid; e < n * n; e += GA_THREADS) {
          const int i = e / n, k = e - i * n;
          ap[a0 + k * n + i] = Apack[a0 + e];
        }
        a0 += n * n;
      }
    }
  }
  __syncthreads();

  float tau = cfg.tau0;
  bool hungarian = cfg.start_hungarian != 0;
  int stage = 0, total = 0;
  const float qw2 = cfg.quad_weight * 2.f, invG = 1.f / (float)G;
  bool first = true;
  const int ldb = Mp + 1;
  // phase timers (cfg.profile): kept in LDS, not in registers that would be live across the whole solve
  __shared__ long long s_tph[6];
  if (tid < 6) s_tph[tid] = 0;
#define GA_PHASE(k)                                                    \
  if (cfg.profile && tid == 0) {                                       \
    const long long now = (long long)__builtin_readcyclecounter();     \
    s_tph[k] += now - s_tph[5];                                        \
    s_tph[5] = now;                                                    \
  }
  __syncthreads();
  if (cfg.profile && tid == 0) s_tph[5] = (long long)__builtin_readcyclecounter();

  for (;;) {   // stages (:311)
    int i = 0;
    for (; i < cfg.max_iter; ++i) {   // :312
      // ---- B = A U, block diagonal, stored transposed: X[u][row]; 16-row MFMA tiles dealt round-robin to the wavefronts ----
      {
        int t = 0;
        for (int g = 0; g < G; ++g) {
          const int o = s_off[g], n = s_off[g + 1] - o;
          for (int i0 = 0; i0 < n; i0 += 16, ++t)
            if ((t % GA_WAVES) == wave) b_row_tile16<kALds>(o, n, i0, Ap + s_aoff[g], Ucur, X, ldb);
        }
      }
      __syncthreads();
      GA_PHASE(0)
      // ---- S = U^T B : four 16 x 16 tiles x KS halves of K = M, one per wavefront; the V tiles add the KS partials ----
      {
        constexpr int KS = GA_WAVES / 4;
        const int kh = ((M + KS - 1) / KS + 3) & ~3;
        const int half = wave >> 2, t = wave & 3;
        if (half * kh < M) s_tile16(t >> 1, t & 1, half * kh, min(M, (half + 1) * kh), ldb, Ucur, X, Spart + half * NU * NU);
        else for (int e = lane; e < 16 * 16; e += 64) Spart[half * NU * NU + ((t >> 1) * 16 + (e >> 4)) * NU + (t & 1) * 16 + (e & 15)] = 0.f;
      }
      __syncthreads();
      GA_PHASE(1)
      // ---- V = (2q B S + W U) / G : one wavefront per 16-row tile, MFMA ----
      for (int t = wave; t * 16 < M; t += GA_WAVES)
        v_row_tile16<GA_WAVES / 4>(t * 16, M, Mp, WT, Ucur, X, Spart, qw2, invG, V, first ? V0snap : nullptr);
      __syncthreads();
      GA_PHASE(2)
      // ---- projection (X <- projected U) and the graph's convergence terms, one wavefront per graph ----
      float d1 = 0.f, d2 = 0.f;
      for (int g = wave; g < G; g += GA_WAVES) {
        const int o = s_off[g], n = s_off[g + 1] - o;
        float* Ub = X + o * NU;
        const float* Vb = V + o * NU;
        if (!hungarian) {
          float* fbuf = wex + wave * wex_stride;
          float* gbuf = fbuf + 40;
          const float scale = TTDG_LOG2E / tau;
          const bool rows_are_nodes = n < NU || (n == NU && !sq_tr);
          if (!(flags & 1) && n <= 64) sk_project_blk(Vb, n, rows_are_nodes, scale, cfg.sk_iter, Ub);
          else if (rows_are_nodes) sk_wave_project_narrow(Vb, n, scale, cfg.sk_iter, Ub, fbuf, gbuf);
          else if (CWMAX == 1 || n <= 64) sk_wave_project<1>(Vb, n, scale, cfg.sk_iter, Ub, fbuf, gbuf, n == NU);
          else sk_wave_project<CWMAX>(Vb, n, scale, cfg.sk_iter, Ub, fbuf, gbuf);
        } else {
          const bool tr = n > NU;
          const int nr = tr ? NU : n, nc = tr ? n : NU;
          for (int e = lane; e < n * NU; e += 64) Ub[e] = 0.f;
          if (CWMAX == 1 || nc <= 64) {
            const int b = lap_wave_solve_reg<0, true>(nr, nc, Vb, tr ? 1 : NU, tr ? NU : 1);
            wave_sync();
            if (lane < nr) { if (tr) Ub[b * NU + lane] = 1.f; else Ub[lane * NU + b] = 1.f; }
          } else {
            LapScratch sc = lap_carve(lapb + wave * lap_stride, nr, nc);
            lap_wave_solve(nr, nc, Vb, tr ? 1 : NU, tr ? NU : 1, sc);
            wave_sync();
            for (int a = lane; a < nr; a += 64) {
              const int b = sc.col4row[a];
              if (tr) Ub[b * NU + a] = 1.f; else Ub[a * NU + b] = 1.f;
            }
          }
        }
        if (G == 2 && g == 0)   // :358-359
          for (int e = lane; e < n * NU; e += 64) Ub[e] = ((e >> 5) == (e & 31)) ? 1.f : 0.f;
        wave_sync();
        conv_block(Ub, Ucur + o * NU, Uprev_g + o * NU, first ? U1snap + o * NU : nullptr, n * NU, d1, d2);
      }
      GA_PHASE(3)
      // ---- convergence (:361): the per-wavefront terms are added up; both barriers of the reduction also publish X ----
      float s2;
      const float s1 = block_sum2<GA_WAVES>(d1, d2, red, s2);
      GA_PHASE(4)
      // lastU <- U, U <- new: the two LDS buffers swap (lastU2 was updated element-wise above)
      float* t = Ucur; Ucur = X; X = t;
      first = false;
      ++total;
      if (sqrtf(s1) < cfg.tol || s2 == 0.f) break;
      // ---- exact cycle shortcut (Hungarian stage) ----------------------------------------------------------
      // The Hungarian-stage map U -> LAP(V(U)) is a deterministic map on a finite set, and the reference only
      // detects periods 1 and 2 (:361): on longer cycles it burns all `max_iter` iterations and returns whatever
      // state iteration max_iter-1 lands on.  We remember the last HIST states (one byte per node: its universe
      // slot); when the new state equals the one p >= 3 iterations back, every remaining iteration is known:
      // S(i + k) = S(i - p + k mod p).  Jump straight to the final state - bit-identical to running them all.
      if (hungarian && !cfg.no_cycle_skip && i < GA_HIST) {
        // state code + 64-bit hash (xor of per-row mixes); the full state goes to the workspace, the hash stays in LDS
        if (tid == 0) hmatch[0] = 0ull;
        __syncthreads();
        unsigned long long hx = 0;
        for (int r = tid; r < M; r += GA_THREADS) {
          int code = 255;
#pragma unroll
          for (int u = 0; u < NU; ++u) if (Ucur[r * NU + u] != 0.f) code = u;
          hist[(size_t)i * M + r] = (unsigned char)code;
          unsigned long long z = (unsigned long long)(r * 256 + code) + 0x9E3779B97F4A7C15ull;     // splitmix64
          z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
          z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
          hx ^= z ^ (z >> 31);
        }
        if (hx) atomicXor(&hmatch[0], hx);
        __syncthreads();
        const unsigned long long hcur = hmatch[0];
        __syncthreads();
        if (tid == 0) { hhash[i] = hcur; hmatch[1] = 0ull; }
        __syncthreads();
        // most recent earlier iteration with the same hash (then verified exactly)
        if (tid < i && hhash[tid] == hcur) atomicMax(&hmatch[1], (unsigned long long)(tid + 1));
        __syncthreads();
        int prev = (int)hmatch[1] - 1;
        if (prev >= 0) {                                // exact check of the candidate (hash collisions must not jump)
          __syncthreads();
          if (tid == 0) hmatch[2] = 1ull;
          __syncthreads();
          for (int r = tid; r < M; r += GA_THREADS)
            if (hist[(size_t)prev * M + r] != hist[(size_t)i * M + r]) hmatch[2] = 0ull;
          __syncthreads();
          if (hmatch[2] == 0ull) prev = -1;
        }
        const int p = (prev >= 0) ? i - prev : 0;
        if (p >= 3) {                                   // periods 1 and 2 are the reference's own exits
          const int R = cfg.max_iter - 1 - i;           // iterations the reference would still run
          const int src = i - p + (R % p);              // iteration whose state equals S(max_iter - 1)
          __syncthreads();
          for (int e = tid; e < MU; e += GA_THREADS) Ucur[e] = (hist[(size_t)src * M + (e >> 5)] == (e & 31)) ? 1.f : 0.f;
          if (tid == 0) { info[14] = p; info[15] = i; }
          total += R;
          i = cfg.max_iter;
          __syncthreads();
          break;
        }
      }
    }
    const int its = (i < cfg.max_iter) ? i + 1 : cfg.max_iter;
    if (tid == 0 && stage < 6) info[stage] = its;
    ++stage;
    if (hungarian) break;            // :374-376
    if (cfg.max_stages > 0 && stage >= cfg.max_stages) break;
    if (tau > cfg.min_tau) tau *= cfg.gamma;   // :377-379
    else hungarian = true;           // :382-383
    __syncthreads();
  }
  __syncthreads();
  for (int e = tid; e < MU; e += GA_THREADS) Uout[e] = Ucur[e];
  if (tid == 0) {
    info[6] = total; info[7] = stage; info[8] = 0;
    for (int k = 0; k < 5; ++k) info[9 + k] = (int32_t)(s_tph[k] >> 6);   // cycle counter ticks / 64 per phase
  }
#undef GA_PHASE
}

static size_t ga_fixed_lds_bytes(int cmaxp, int GA_WAVES, int cwmax, int M) {
  // static LDS (s_off, s_aoff, timers) ~ 0.6 KB + 2 partial S tiles, reduction scratch, per-wave potentials, optional LDS-LAP
  // scratch, node->graph bytes
  const size_t lap = cwmax == 2 ? GA_WAVES * ((lap_scratch_bytes(NU, cmaxp) + 15) & ~(size_t)15) : 0;
  return (size_t)1024 + (size_t)(2 * NU * NU + 64 + GA_WAVES * (40 + cmaxp)) * sizeof(float) + lap + ((M + 15) & ~15) + 32 + GA_HIST * 8;
}


// gagm_large.hip
#define GAGM_LARGE_FROM_DEFAULT 320   // measured (tools/bench_gagm_scale.py): 140 vs 89 us per iteration at 451 nodes, 392 vs 95 at 892; 69 vs 69 at 221
size_t ttdg_gagm_large_ws_bound(int M);
int ttdg_gagm_large_solve(const float* Apack, const float* W, const float* U0, ttdg_graphs_t gr, ttdg_gagm_cfg_t cfg,
                          float* U, int32_t* info, void* ws, hipStream_t st);

// total node count above which graphs that WOULD fit the single-workgroup kernel still take the multi-workgroup solver:
// the single workgroup streams the M x M matrix W once per iteration, which stops paying once W has left LDS and the
// per-iteration W U product outgrows one CU (Mode S: the gathered multi-graph of 8 ranks is ~1000 nodes)
// [r4] the A/B selectors travel in cfg.variant (include/ttdg_mgm.h): the library keeps no process-global switches for the solver

extern "C" size_t ttdg_gagm_workspace_bytes(int M) {
  const size_t small = ga_ws_hist_off(M) * sizeof(float) + (size_t)GA_HIST * M + 64;
  const size_t large = ttdg_gagm_large_ws_bound(M);
  return small > large ? small : large;
}

extern "C" int ttdg_gagm_solve(const float* Apack, const float* W, const float* U0, ttdg_graphs_t gr,
                               ttdg_gagm_cfg_t cfg, float* U, int32_t* info, void* ws, ttdg_stream_t stream) {
  TTDG_REQUIRE(Apack && W && U0 && U && info && ws, "gagm: null pointer");
  if (int e = ttdg_validate_graphs(gr)) return e;
  TTDG_REQUIRE(cfg.max_iter >= 1 && cfg.sk_iter >= 0 && cfg.tau0 > 0.f && cfg.gamma > 0.f && cfg.gamma < 1.f,
               "gagm: bad configuration");
  int cmax = NU, asz = 0;
  for (int g = 0; g < gr.G; ++g) {
    const int n = gr.off[g + 1] - gr.off[g];
    cmax = n > cmax ? n : cmax;
    asz += n * n;
  }
  const bool fits_single = cmax <= 128 && gr.off[gr.G] <= 4096;
  const bool large = !fits_single || (cfg.variant & TTDG_GAGM_FORCE_LARGE) ||
                     (gr.off[gr.G] >= GAGM_LARGE_FROM_DEFAULT && !(cfg.variant & TTDG_GAGM_FORCE_SINGLE));
  if (large)   // beyond one CU: the multi-workgroup solver (gagm_large.hip), same schedule and outputs
    return ttdg_gagm_large_solve(Apack, W, U0, gr, cfg, U, info, ws, (hipStream_t)stream);
  TTDG_LIMIT(gr.off[gr.G] <= 4096, "gagm: more than 4096 nodes in total");
  const int cmaxp = (cmax + 63) & ~63;
  const int M = gr.off[gr.G], Mp = ga_mp(M);
  // 512 threads (8 wavefronts).  With the 256-VGPR cap the kernel spills ~100 VGPRs, all at phase boundaries (none inside
  // the Sinkhorn / LAP loops); the spill-free 256-thread build (one wavefront per SIMD, 440 VGPRs) was measured SLOWER on
  // the bench (30.2 vs 27.1 us per iteration, 67.9 vs 72.0 images/s): B = A U and the convergence sweep want the eight
  // wavefronts.  It stays selectable for A/B runs (cfg.variant & TTDG_GAGM_256_THREADS).
  const int threads = (cmax <= 64 && (cfg.variant & TTDG_GAGM_256_THREADS)) ? 256 : 512;
  const int waves = threads / 64;
  const int cw = cmax <= 64 ? 1 : 2;
  const size_t fixed = ga_fixed_lds_bytes(cmaxp, waves, cw, M);
  const size_t state = (size_t)3 * NU * (Mp + 1) * sizeof(float);
  const size_t wb = (size_t)M * Mp * sizeof(float), ab = (size_t)((asz + 3) & ~3) * sizeof(float);
  const size_t cap = 158 * 1024;
  // what fits decides what is staged: state, then W^T (largest per-iteration reader), then the A blocks
  const int mode = (fixed + state + wb + ab <= cap) ? 3 : (fixed + state + wb <= cap) ? 2 : (fixed + state <= cap) ? 1 : 0;
  const size_t bytes = fixed + (mode >= 1 ? state : 0) + (mode >= 2 ? wb : 0) + (mode >= 3 ? ab : 0);
  hipStream_t st = (hipStream_t)stream;
#define GA_LAUNCH_T(L, WL, AL, C, T)                                                                                 \
  do {                                                                                                               \
    TTDG_ALLOW_LDS((gagm_kernel<L, WL, AL, T, C>), bytes);                                                           \
    hipLaunchKernelGGL((gagm_kernel<L, WL, AL, T, C>), dim3(1), dim3(T), bytes, st, Apack, W, U0, gr, cfg, U, info, (float*)ws, cmaxp, cfg.variant & TTDG_GAGM_LDS_PROJECTORS); \
  } while (0)
#define GA_LAUNCH(L, WL, AL, C)                                          \
  do {                                                                   \
    if (C == 1 && threads == 256) GA_LAUNCH_T(L, WL, AL, 1, 256);        \
    else GA_LAUNCH_T(L, WL, AL, C, 512);                                 \
  } while (0)
#define GA_MODES(C)                                             \
  do {                                                          \
    if (mode == 3) GA_LAUNCH(true, true, true, C);              \
    else if (mode == 2) GA_LAUNCH(true, true, false, C);        \
    else if (mode == 1) GA_LAUNCH(true, false, false, C);       \
    else GA_LAUNCH(false, false, false, C);                     \
  } while (0)
  if (cw == 1) GA_MODES(1); else GA_MODES(2);
#undef GA_MODES
#undef GA_LAUNCH
#undef GA_LAUNCH_T
  return ttdg_launch_status("gagm");
}


// ---- micro-benchmark hook: `reps` back-to-back projections of G graphs of n nodes, one wavefront per graph, from an
// LDS-resident V exactly as inside gagm_kernel (tools/bench_gagm_parts.py).  mode 0 = Sinkhorn, 1 = LAP.
__global__ __launch_bounds__(512) void debug_project_kernel(const float* __restrict__ Vg, int n, int G, float scale, int iters,
                                                            int reps, int mode, float* __restrict__ Ug, long long* __restrict__ ticks) {
  extern __shared__ __attribute__((aligned(16))) float dbg_smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* V = dbg_smem;                       // G*n*32
  float* U = V + G * n * NU;
  float* wex = U + G * n * NU;               // per wave 40 + 64
  for (int e = threadIdx.x; e < G * n * NU; e += blockDim.x) V[e] = Vg[e];
  __syncthreads();
  const long long t0 = (long long)__builtin_readcyclecounter();
  if (wave < G) {
    const float* Vb = V + wave * n * NU;
    float* Ub = U + wave * n * NU;
    float* fbuf = wex + wave * 104;
    for (int r = 0; r < reps; ++r) {
      if (mode == 0) sk_project_blk(Vb, n, n <= NU, scale, iters, Ub);
      else if (mode == 2) { if (n <= NU) sk_wave_project_narrow(Vb, n, scale, iters, Ub, fbuf, fbuf + 40); else sk_wave_project<1>(Vb, n, scale, iters, Ub, fbuf, fbuf + 40); }
      else {
        const bool tr = n > NU;
        const int nr = tr ? NU : n, nc = tr ? n : NU;
        for (int e = lane; e < n * NU; e += 64) Ub[e] = 0.f;
        const int b = lap_wave_solve_reg<0, true>(nr, nc, Vb, tr ? 1 : NU, tr ? NU : 1);
        wave_sync();
        if (lane < nr) { if (tr) Ub[b * NU + lane] = 1.f; else Ub[lane * NU + b] = 1.f; }
      }
      wave_sync();
    }
  }
  __syncthreads();
  const long long t1 = (long long)__builtin_readcyclecounter();
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
  for (int e = threadIdx.x; e < G * n * NU; e += blockDim.x) Ug[e] = U[e];
}

extern "C" int ttdg_debug_project(const float* V, int n, int G, float tau, int iters, int reps, int mode, float* U,
                                  long long* ticks, ttdg_stream_t stream) {
  TTDG_REQUIRE(V && U && ticks && n >= 1 && n <= 64 && G >= 1 && G <= 8, "debug_project: bad arguments");
  const size_t bytes = ((size_t)2 * G * n * NU + 8 * 104) * sizeof(float);
  TTDG_ALLOW_LDS(debug_project_kernel, bytes);
  hipLaunchKernelGGL(debug_project_kernel, dim3(1), dim3(512), bytes, (hipStream_t)stream, V, n, G, TTDG_LOG2E / tau, iters, reps,
                     mode, U, ticks);
  return ttdg_launch_status("debug_project");
}
