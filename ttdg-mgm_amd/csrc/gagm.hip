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
__host__ __device__ inline size_t ga_ws_hist_off(int M) {
  const int Mp = (M + 31) & ~31;
  return (size_t)2 * M * NU + (size_t)M * Mp + (size_t)4 * NU * (Mp + 1);
}

// V rows [i0, i0+32): (2q * B S + W U) / G on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//   W U : A operand = W^T stored [k][i] (leading dimension Mp, zero padded) -> the 32 lanes of a half-wave read 32
//         consecutive words; B operand = U[k][u].
//   B S : A operand = B^T stored [v][i] (leading dimension Mp + 1); B operand = S[v][u].
__device__ __forceinline__ void v_row_tile(int i0, int M, int Mp, const float* WT, const float* Ucur, const float* BT,
                                           const float* S, float qw2, float invG, float* V, float* V0snap) {
  const int lane = threadIdx.x & 63, li = lane & 31, kh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // operands are fetched eight k-pairs ahead of the MFMAs that consume them (LDS / L2 latency off the chain)
  int k0 = 0;
  for (; k0 + 16 <= M; k0 += 16) {
    float a[8], b[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = k0 + 2 * q + kh;
      a[q] = WT[(size_t)k * Mp + i0 + li];
      b[q] = Ucur[k * NU + li];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
  }
  for (; k0 < M; k0 += 2) {
    const int k = k0 + kh;
    const bool ok = k < M;
    const float a = ok ? WT[(size_t)k * Mp + i0 + li] : 0.f;
    const float b = ok ? Ucur[k * NU + li] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  const int ldb = Mp + 1;
  {
    float a[16], b[16];
    const bool rok = i0 + li < M;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = 2 * q + kh;
      a[q] = rok ? BT[k * ldb + i0 + li] * qw2 : 0.f;
      b[q] = S[k * NU + li];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    if (row < M) {
      const float val = acc[r] * invG;
      V[row * NU + li] = val;
      if (V0snap) V0snap[row * NU + li] = val;
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
                                                          int32_t* __restrict__ info, float* __restrict__ ws, int cmaxp) {
  extern __shared__ __attribute__((aligned(16))) float ga_smem[];
  constexpr int GA_WAVES = GA_THREADS / 64;
  const int M = gr.off[gr.G], G = gr.G;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int MU = M * NU;
  const int Mp = (M + 31) & ~31;
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
  // lastU2 is only touched by the convergence check, always by the same thread for the same element: it lives in
  // registers (kLds: M <= 256 -> at most 16 values per thread) or in the workspace, never in LDS
  float* Ucur = base;
  float* X = base + SB;
  float* V = base + 2 * SB;
  float* Uprev_g = base + 3 * SB;             // used only when !kLds
  constexpr int UPK = 8192 / GA_THREADS;     // lastU2 values per thread: covers M * 32 <= 8192
  float uprev[UPK];
#pragma unroll
  for (int k = 0; k < UPK; ++k) uprev[k] = 0.f;
  const float* WT = WTg;
  const float* Ap = Apack;
  if (kWLds) { WT = lds; lds += M * Mp; }
  if (kALds) { Ap = lds; lds += (asz + 3) & ~3; }
  float* S = lds;                // 1024
  float* Spart = S + NU * NU;    // 4 x 1024 partial tiles of S
  float* red = Spart + 4 * NU * NU;  // 64
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
  for (int e = tid; e < MU; e += GA_THREADS) { Ucur[e] = U0[e]; if (!kLds) Uprev_g[e] = 0.f; }   // lastU = zeros (:305)
  {  // W^T[k][i] = W[i][k], zero padded to Mp columns; A blocks
    float* wt = kWLds ? const_cast<float*>(WT) : WTg;
    for (int e = tid; e < M * Mp; e += GA_THREADS) {
      const int k = e / Mp, i = e - k * Mp;
      wt[e] = (i < M) ? W[(size_t)i * M + k] : 0.f;
    }
    if (kALds) {
      float* ap = const_cast<float*>(Ap);
      for (int e = tid; e < asz; e += GA_THREADS) ap[e] = Apack[e];
    }
  }
  __syncthreads();

  float tau = cfg.tau0;
  bool hungarian = cfg.start_hungarian != 0;
  int stage = 0, total = 0;
  const float qw2 = cfg.quad_weight * 2.f, invG = 1.f / (float)G;
  bool first = true;
  const int ldb = Mp + 1;
  long long tick = 0, tph[5] = {0, 0, 0, 0, 0};
#define GA_PHASE(k)                                                    \
  if (cfg.profile && tid == 0) {                                       \
    const long long now = (long long)__builtin_readcyclecounter();     \
    tph[k] += now - tick;                                              \
    tick = now;                                                        \
  }
  if (cfg.profile && tid == 0) tick = (long long)__builtin_readcyclecounter();

  for (;;) {   // stages (:311)
    int i = 0;
    for (; i < cfg.max_iter; ++i) {   // :312
      // ---- B = A U, block diagonal, stored transposed: X[u][row] ----
      for (int e = tid; e < M * 8; e += GA_THREADS) {
        const int row = e >> 3, u = (e & 7) * 4;
        const int g = s_gid[row];
        const int o = s_off[g], n = s_off[g + 1] - o;
        const float* arow = Ap + s_aoff[g] + (size_t)(row - o) * n;
        const float* ub = Ucur + o * NU + u;
        float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
        int j = 0;
        // eight columns per round, all sixteen LDS reads issued before the first FMA (the two-at-a-time loop below pays
        // one LDS round trip per pair); same two accumulators, same order: bit-identical sums
        for (; j + 8 <= n; j += 8) {
          float a[8];
          float4 x[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) { a[q] = arow[j + q]; x[q] = *reinterpret_cast<const float4*>(ub + (j + q) * NU); }
#pragma unroll
          for (int q = 0; q < 8; q += 2) {
            c0.x = fmaf(a[q], x[q].x, c0.x); c0.y = fmaf(a[q], x[q].y, c0.y); c0.z = fmaf(a[q], x[q].z, c0.z); c0.w = fmaf(a[q], x[q].w, c0.w);
            c1.x = fmaf(a[q + 1], x[q + 1].x, c1.x); c1.y = fmaf(a[q + 1], x[q + 1].y, c1.y);
            c1.z = fmaf(a[q + 1], x[q + 1].z, c1.z); c1.w = fmaf(a[q + 1], x[q + 1].w, c1.w);
          }
        }
        for (; j + 2 <= n; j += 2) {
          const float a0 = arow[j], a1 = arow[j + 1];
          const float4 x0 = *reinterpret_cast<const float4*>(ub + j * NU), x1 = *reinterpret_cast<const float4*>(ub + (j + 1) * NU);
          c0.x = fmaf(a0, x0.x, c0.x); c0.y = fmaf(a0, x0.y, c0.y); c0.z = fmaf(a0, x0.z, c0.z); c0.w = fmaf(a0, x0.w, c0.w);
          c1.x = fmaf(a1, x1.x, c1.x); c1.y = fmaf(a1, x1.y, c1.y); c1.z = fmaf(a1, x1.z, c1.z); c1.w = fmaf(a1, x1.w, c1.w);
        }
        if (j < n) {
          const float a0 = arow[j];
          const float4 x0 = *reinterpret_cast<const float4*>(ub + j * NU);
          c0.x = fmaf(a0, x0.x, c0.x); c0.y = fmaf(a0, x0.y, c0.y); c0.z = fmaf(a0, x0.z, c0.z); c0.w = fmaf(a0, x0.w, c0.w);
        }
        X[(u + 0) * ldb + row] = c0.x + c1.x;
        X[(u + 1) * ldb + row] = c0.y + c1.y;
        X[(u + 2) * ldb + row] = c0.z + c1.z;
        X[(u + 3) * ldb + row] = c0.w + c1.w;
      }
      __syncthreads();
      GA_PHASE(0)
      // ---- S = U^T B : 32x32 output, K = M split over the first four wavefronts (MFMA), partials summed in LDS ----
      if (wave < 4) {
        const int li = lane & 31, kh = lane >> 5;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int chunk = ((M + 7) / 8) * 2;              // rows per wavefront, even
        const int rbeg = wave * chunk, rend = min(M, rbeg + chunk);
        for (int r0 = rbeg; r0 < rend; r0 += 2) {
          const int r = r0 + kh;
          const bool ok = r < rend;
          const float a = ok ? Ucur[r * NU + li] : 0.f;      // (u = li, k = r)
          const float b = ok ? X[li * ldb + r] : 0.f;        // (k = r, v = li)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Spart[wave * NU * NU + ((r & 3) + 8 * (r >> 2) + 4 * kh) * NU + li] = acc[r];
      }
      __syncthreads();
      for (int e = tid; e < NU * NU; e += GA_THREADS)
        S[e] = (Spart[e] + Spart[NU * NU + e]) + (Spart[2 * NU * NU + e] + Spart[3 * NU * NU + e]);
      __syncthreads();
      GA_PHASE(1)
      // ---- V = (2q B S + W U) / G : one wavefront per 32-row tile, MFMA ----
      for (int t = wave; t * 32 < M; t += GA_WAVES)
        v_row_tile(t * 32, M, Mp, WT, Ucur, X, S, qw2, invG, V, first ? V0snap : nullptr);
      __syncthreads();
      GA_PHASE(2)
      // ---- projection (X <- projected U), one wavefront per graph ----
      for (int g = wave; g < G; g += GA_WAVES) {
        const int o = s_off[g], n = s_off[g + 1] - o;
        float* Ub = X + o * NU;
        const float* Vb = V + o * NU;
        if (!hungarian) {
          float* fbuf = wex + wave * wex_stride;
          float* gbuf = fbuf + 40;
          const float scale = TTDG_LOG2E / tau;
          if (n < NU || (n == NU && !sq_tr)) sk_wave_project_narrow(Vb, n, scale, cfg.sk_iter, Ub, fbuf, gbuf);
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
      }
      __syncthreads();
      GA_PHASE(3)
      if (G == 2) {   // :358-359
        const int n0 = s_off[1];
        for (int e = tid; e < n0 * NU; e += GA_THREADS) X[e] = ((e >> 5) == (e & 31)) ? 1.f : 0.f;
        __syncthreads();
      }
      // ---- convergence (:361) ----
      float d1 = 0.f, d2 = 0.f;
      if (kLds) {
#pragma unroll
        for (int k = 0; k < UPK; ++k) {
          const int e = tid + k * GA_THREADS;
          if (e < MU) {
            const float un = X[e], uc = Ucur[e];
            if (first) U1snap[e] = un;
            const float a = un - uc, b = un - uprev[k];
            d1 = fmaf(a, a, d1);
            d2 = fmaf(b, b, d2);
            uprev[k] = uc;                       // lastU2 <- lastU
          }
        }
      } else {
        for (int e = tid; e < MU; e += GA_THREADS) {
          const float un = X[e], uc = Ucur[e];
          if (first) U1snap[e] = un;
          const float a = un - uc, b = un - Uprev_g[e];
          d1 = fmaf(a, a, d1);
          d2 = fmaf(b, b, d2);
          Uprev_g[e] = uc;
        }
      }
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
    for (int k = 0; k < 5; ++k) info[9 + k] = (int32_t)(tph[k] >> 6);   // cycle counter ticks / 64 per phase
  }
#undef GA_PHASE
}

static size_t ga_fixed_lds_bytes(int cmaxp, int GA_WAVES, int cwmax, int M) {
  // static LDS (s_off, s_aoff) ~ 0.6 KB + S, 4 partial S tiles, reduction scratch, per-wave potentials, optional LDS-LAP
  // scratch, node->graph bytes
  const size_t lap = cwmax == 2 ? GA_WAVES * ((lap_scratch_bytes(NU, cmaxp) + 15) & ~(size_t)15) : 0;
  return (size_t)1024 + (size_t)(5 * NU * NU + 64 + GA_WAVES * (40 + cmaxp)) * sizeof(float) + lap + ((M + 15) & ~15) + 32 + GA_HIST * 8;
}

static inline int ga_mp(int M) { return (M + 31) & ~31; }

// gagm_large.hip
#define GAGM_LARGE_FROM_DEFAULT 320   // measured (tools/bench_gagm_scale.py): 140 vs 89 us per iteration at 451 nodes, 392 vs 95 at 892; 69 vs 69 at 221
size_t ttdg_gagm_large_ws_bound(int M);
int ttdg_gagm_large_solve(const float* Apack, const float* W, const float* U0, ttdg_graphs_t gr, ttdg_gagm_cfg_t cfg,
                          float* U, int32_t* info, void* ws, hipStream_t st);

// total node count above which graphs that WOULD fit the single-workgroup kernel still take the multi-workgroup solver:
// the single workgroup streams the M x M matrix W once per iteration, which stops paying once W has left LDS and the
// per-iteration W U product outgrows one CU (Mode S: the gathered multi-graph of 8 ranks is ~1000 nodes)
static int g_gagm_large_from = GAGM_LARGE_FROM_DEFAULT;
static int g_gagm_threads = 0;   // 0 / 512 = the default 512 threads; 256 = the spill-free one-wavefront-per-SIMD build (A/B runs)
extern "C" int ttdg_debug_set_gagm_threads(int threads) { g_gagm_threads = (threads == 256 || threads == 512) ? threads : 0; return 0; }
extern "C" int ttdg_debug_set_gagm_large_from(int total_nodes) { g_gagm_large_from = total_nodes > 0 ? total_nodes : GAGM_LARGE_FROM_DEFAULT; return 0; }

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
  if (cmax > 128 || gr.off[gr.G] >= g_gagm_large_from)   // beyond one CU: the multi-workgroup solver (gagm_large.hip), same schedule and outputs
    return ttdg_gagm_large_solve(Apack, W, U0, gr, cfg, U, info, ws, (hipStream_t)stream);
  TTDG_LIMIT(gr.off[gr.G] <= 4096, "gagm: more than 4096 nodes in total");
  const int cmaxp = (cmax + 63) & ~63;
  const int M = gr.off[gr.G], Mp = ga_mp(M);
  // 512 threads (8 wavefronts).  With the 256-VGPR cap the kernel spills ~100 VGPRs, all at phase boundaries (none inside
  // the Sinkhorn / LAP loops); the spill-free 256-thread build (one wavefront per SIMD, 440 VGPRs) was measured SLOWER on
  // the bench (30.2 vs 27.1 us per iteration, 67.9 vs 72.0 images/s): B = A U and the convergence sweep want the eight
  // wavefronts.  It stays selectable for A/B runs (ttdg_debug_set_gagm_threads).
  const int threads = (cmax <= 64 && g_gagm_threads == 256) ? 256 : 512;
  const int waves = threads / 64;
  const int cw = cmax <= 64 ? 1 : 2;
  const size_t fixed = ga_fixed_lds_bytes(cmaxp, waves, cw, M);
  const size_t state = (size_t)3 * NU * (Mp + 1) * sizeof(float);
  const size_t wb = (size_t)M * Mp * sizeof(float), ab = (size_t)((asz + 3) & ~3) * sizeof(float);
  const size_t cap = 158 * 1024;
  // what fits decides what is staged: state, then W^T (largest per-iteration reader), then the A blocks
  const bool regs_ok = (size_t)M * NU <= (size_t)8192;   // lastU2 in registers: 8192 / threads values per thread
  const int mode = !regs_ok ? 0 : (fixed + state + wb + ab <= cap) ? 3 : (fixed + state + wb <= cap) ? 2 : (fixed + state <= cap) ? 1 : 0;
  const size_t bytes = fixed + (mode >= 1 ? state : 0) + (mode >= 2 ? wb : 0) + (mode >= 3 ? ab : 0);
  hipStream_t st = (hipStream_t)stream;
#define GA_LAUNCH_T(L, WL, AL, C, T)                                                                                 \
  do {                                                                                                               \
    TTDG_ALLOW_LDS((gagm_kernel<L, WL, AL, T, C>), bytes);                                                           \
    hipLaunchKernelGGL((gagm_kernel<L, WL, AL, T, C>), dim3(1), dim3(T), bytes, st, Apack, W, U0, gr, cfg, U, info, (float*)ws, cmaxp); \
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
      if (mode == 0) { if (n <= NU) sk_wave_project_narrow(Vb, n, scale, iters, Ub, fbuf, fbuf + 40); else sk_wave_project<1>(Vb, n, scale, iters, Ub, fbuf, fbuf + 40); }
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
