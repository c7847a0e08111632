// A5 — log-space Sinkhorn (reference utils/sinkhorn.py:85-87 -> pygmtools.sinkhorn [3P, absent];
// specification: SURVEY.md Appendix B; oracle: oracle/sinkhorn_spec.py).
//
// Formulation.  The reference rewrites the whole matrix at every sweep (log_s -= logsumexp).  We keep the
// scaled input L = s/tau constant and carry the dual potentials instead:
//     row sweep:  f_p = lse_q(L_pq - g_q)      col sweep:  g_q = lse_p(L_pq - f_p)      y = L - f (+) g
// which is the same map in exact arithmetic, reads the matrix once per sweep, writes only r+c numbers,
// and lets the backward pass rebuild every intermediate y^(k) from K small vectors instead of K matrices.
// Dummy rows (dummy_row=True): the (c - r) appended rows are identical (all -100 after tau scaling) and
// stay identical under both sweeps, so they are ONE virtual row with multiplicity c - r.
// Everything runs in base 2 (L2 = L*log2 e) on v_exp_f32 / v_log_f32.
//
// One workgroup per matrix; a line (row or column) is reduced by a sub-group of 16/32/64 lanes with
// xor-shuffles; the matrix lives in LDS when it fits (<= 36k floats), otherwise it is re-read from L2.
#include "sinkhorn_device.h"

// ---- pair stage (multi_graph_matching.py:504-525) --------------------------------------------------
// (Round 2 tried the solver's one-wavefront block-layout projector here: equally accurate - max |Wds - oracle| 1.4e-6 vs 1.9e-6
// on planted case p4 - but not faster for the ten 20..40-node pairs of a TTA step (116 vs 102 us under rocprofv3: neither kernel
// is bound by its sweeps), and its different rounding alone moved p4's solve off the reference's permutation in the
// tau = 0.00625 stage; removed.)
__device__ __forceinline__ void pair_of(int idx, int G, int& a, int& b) {
  a = 0;
  while ((a + 1) * (a + 2) / 2 <= idx) ++a;  // pairs ordered (0,0),(1,0),(1,1),(2,0)...
  b = idx - a * (a + 1) / 2;
}

template <bool kLds>
__global__ void sinkhorn_pairs_fwd_kernel(const float* __restrict__ part, int ksplit, const float* __restrict__ b2,
                                          ttdg_graphs_t gr, float tau, int iters, float* __restrict__ Wds,
                                          float* __restrict__ pot, int cmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int a, b;
  pair_of(blockIdx.x, gr.G, a, b);
  const int M = gr.off[gr.G];
  const int na = gr.off[a + 1] - gr.off[a], nb = gr.off[b + 1] - gr.off[b];
  const float* blk = part + (size_t)gr.off[a] * M + gr.off[b];  // element (i in a, j in b) at blk[i*M + j]
  float* wab = Wds + (size_t)gr.off[a] * M + gr.off[b];
  float* wba = Wds + (size_t)gr.off[b] * M + gr.off[a];
  SkProb pb;
  pb.src = blk;
  pb.splane = (int64_t)M * M;
  pb.nplanes = ksplit;
  pb.bias = b2 ? *b2 : 0.f;
  pb.scale = TTDG_LOG2E / tau;
  if (nb >= na) {  // rows = a-nodes (reference: end_y - start_y >= end_x - start_x -> no transpose)
    pb.r = na; pb.c = nb; pb.sp = M; pb.sq = 1;
    pb.out = wab; pb.op = M; pb.oq = 1;
    pb.mir = (a != b) ? wba : nullptr; pb.mp = 1; pb.mq = M;
  } else {         // rows = b-nodes (transposed before, transposed back after)
    pb.r = nb; pb.c = na; pb.sp = 1; pb.sq = M;
    pb.out = wab; pb.op = 1; pb.oq = M;
    pb.mir = wba; pb.mp = M; pb.mq = 1;
  }
  pb.mult = pb.c - pb.r;
  pb.potld = cmax + 1;
  pb.pot = pot ? pot + (size_t)blockIdx.x * iters * pb.potld : nullptr;
  sk_forward<kLds>(pb, smem, iters);
}


// ---- register-resident pair stage for 128 < c <= 256 (BASELINE cfg-3: 256-node graphs) ---------------------------------
// A 256 x 256 fp32 matrix (256 KB) does not fit the 160 KB LDS, but it fits the register file of ONE workgroup:
// 1024 threads x 64 VGPRs.  Wavefront w owns rows [16w, 16w+16), lane l owns columns 4l..4l+3, so
//   row sweep : 4 in-lane terms + one DPP wavefront reduction per row, no LDS, no barrier;
//   col sweep : 16 in-lane terms per column, the 16 wavefront partials meet in LDS (one barrier), and every wavefront
//               finishes all 256 columns redundantly (no second barrier, identical values everywhere).
// After the first row+col pair y = L - f - g <= 0, so the previous potential is a valid stabiliser: sweeps >= 2 are
// single-pass (no max); a line whose sum leaves [2^-80, 2^80] falls back to the exact two-pass form (rows: per row,
// wavefront-uniform; columns: ballot over the wavefront, the same decision in every wavefront).
#define SKR_THREADS 1024
#define SKR_WAVES 16
#define SKR_C 256
#define SKR_RW (SKR_C / SKR_WAVES)   /* rows per wavefront */
#define SKR_BIG 1.2e24f
#define SKR_SMALL 8.3e-25f

__device__ __forceinline__ float skr_sgpr(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ bool skr_sane(float s) { return s > SKR_SMALL && s < SKR_BIG; }

// element (p, q) through a wavefront-uniform row pointer + a 32-bit lane offset (scalar base + vector offset addressing)
__device__ __forceinline__ float skr_load(const SkProb& pb, const float* rowp, int qo) {
  float v = pb.bias;
  for (int k = 0; k < pb.nplanes; ++k) v += rowp[(size_t)k * pb.splane + qo];
  return v * pb.scale;
}

__device__ __forceinline__ void sk_pair_problem(const float* part, int ksplit, const float* b2, const ttdg_graphs_t& gr, float tau,
                                                int a, int b, SkProb& pb) {
  const int M = gr.off[gr.G];
  const int na = gr.off[a + 1] - gr.off[a], nb = gr.off[b + 1] - gr.off[b];
  pb.src = part + (size_t)gr.off[a] * M + gr.off[b];
  pb.splane = (int64_t)M * M;
  pb.nplanes = ksplit;
  pb.bias = b2 ? *b2 : 0.f;
  pb.scale = TTDG_LOG2E / tau;
  if (nb >= na) { pb.r = na; pb.c = nb; pb.sp = M; pb.sq = 1; }
  else          { pb.r = nb; pb.c = na; pb.sp = 1; pb.sq = M; }
  pb.mult = pb.c - pb.r;
}


// One column sweep of the register-resident kernel.  kExact: two-pass (column maximum through LDS first); otherwise the
// previous potential g is the stabiliser and the return value says whether some column sum left the sane range (the
// same ballot in every wavefront, so all of them take the exact path together).
template <bool kExact>
__device__ __forceinline__ bool skr_col_sweep(const float (&L)[SKR_RW][4], const float (&f)[SKR_RW], const float (&g)[4], float td0,
                                              int mult, int p0, int q0, int r, int c, int wave,
                                              float (&s_part)[2][SKR_WAVES * SKR_C], int& buf, float (&gn)[4]) {
  float sh[4];
  if (kExact) {
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int i = 0; i < SKR_RW; ++i)
      if (p0 + i < r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mx[j] = fmaxf(mx[j], L[i][j] - f[i]);
      }
    *reinterpret_cast<float4*>(&s_part[buf][wave * SKR_C + q0]) = make_float4(mx[0], mx[1], mx[2], mx[3]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[j] = td0;
#pragma unroll
    for (int w = 0; w < SKR_WAVES; ++w) {
      const float4 v = *reinterpret_cast<const float4*>(&s_part[buf][w * SKR_C + q0]);
      sh[0] = fmaxf(sh[0], v.x); sh[1] = fmaxf(sh[1], v.y); sh[2] = fmaxf(sh[2], v.z); sh[3] = fmaxf(sh[3], v.w);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (q0 + j >= c) sh[j] = 0.f;
    buf ^= 1;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[j] = g[j];
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i)
    if (p0 + i < r) {
      const float fi = f[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += fast_exp2(L[i][j] - fi - sh[j]);
    }
  *reinterpret_cast<float4*>(&s_part[buf][wave * SKR_C + q0]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < SKR_WAVES; ++w) {
    const float4 v = *reinterpret_cast<const float4*>(&s_part[buf][w * SKR_C + q0]);
    s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
  }
  buf ^= 1;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (mult > 0) s[j] += (float)mult * fast_exp2(td0 - sh[j]);
    const bool live = q0 + j < c;
    gn[j] = live ? sh[j] + fast_log2(s[j]) : 0.f;
    bad |= live && !skr_sane(s[j]);
  }
  return !kExact && __ballot(bad) != 0ull;
}

__global__ __launch_bounds__(SKR_THREADS) void sinkhorn_pairs_fwd_reg_kernel(const float* __restrict__ part, int ksplit,
                                                                             const float* __restrict__ b2, ttdg_graphs_t gr, float tau,
                                                                             int iters, float* __restrict__ Wds,
                                                                             float* __restrict__ pot, int cmax) {
  __shared__ __attribute__((aligned(16))) float s_part[2][SKR_WAVES * SKR_C];
  int a, b;
  pair_of(blockIdx.x, gr.G, a, b);
  const int M = gr.off[gr.G];
  SkProb pb;
  sk_pair_problem(part, ksplit, b2, gr, tau, a, b, pb);
  float* wab = Wds + (size_t)gr.off[a] * M + gr.off[b];
  float* wba = Wds + (size_t)gr.off[b] * M + gr.off[a];
  const bool flip = pb.sq != 1;
  pb.out = wab; pb.op = flip ? 1 : M; pb.oq = flip ? M : 1;
  pb.mir = (flip || a != b) ? wba : nullptr; pb.mp = flip ? M : 1; pb.mq = flip ? 1 : M;
  const int potld = cmax + 1;
  float* pt = pot ? pot + (size_t)blockIdx.x * iters * potld : nullptr;
  const int r = pb.r, c = pb.c, mult = pb.mult;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave: provably uniform -> scalar row pointers
  const int p0 = wave * SKR_RW, q0 = lane * 4;

  // branch-free load: clamped (always valid) addresses, every plane's 128 loads in flight together
  const int sq = (int)pb.sq;
  unsigned qo[4];   // 32-bit BYTE offsets: scalar row base + vector offset addressing (blocks span < 4 GB)
#pragma unroll
  for (int j = 0; j < 4; ++j) qo[j] = (unsigned)(min(q0 + j, c - 1) * sq) * 4u;
  float L[SKR_RW][4];   // single plane only (the host routes K-split inputs to the LDS kernel): no loop for LICM to hoist 128 addresses out of
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i) {   // 128 unconditional loads in flight (clamped addresses); the empty asm keeps them from being sunk into branches
    const char* rowp = reinterpret_cast<const char*>(pb.src + (int64_t)min(p0 + i, r - 1) * pb.sp);
#pragma unroll
    for (int j = 0; j < 4; ++j) L[i][j] = *reinterpret_cast<const float*>(rowp + qo[j]);
  }
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(L[i][j]));
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) L[i][j] = (p0 + i < r && q0 + j < c) ? (L[i][j] + pb.bias) * pb.scale : -INFINITY;
  float f[SKR_RW], g[4] = {0.f, 0.f, 0.f, 0.f}, fd = 0.f;
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i) f[i] = 0.f;
  int buf = 0;

  for (int it = 0; it < iters; ++it) {
    if ((it & 1) == 0) {
      // ---- rows: f_p = lse_q(L_pq - g_q) ----
      float flog = 0.f;
#pragma unroll
      for (int i = 0; i < SKR_RW; ++i) {
        if (p0 + i < r) {
          const float t0 = L[i][0] - g[0], t1 = L[i][1] - g[1], t2 = L[i][2] - g[2], t3 = L[i][3] - g[3];
          float sh = f[i], s = 0.f;
          bool exact = it < 2;
          if (!exact) {
            s = wave_sum_f32_dpp((fast_exp2(t0 - sh) + fast_exp2(t1 - sh)) + (fast_exp2(t2 - sh) + fast_exp2(t3 - sh)));
            exact = !skr_sane(s);
          }
          if (exact) {
            sh = wave_max_f32_dpp(fmaxf(fmaxf(t0, t1), fmaxf(t2, t3)));
            s = wave_sum_f32_dpp((fast_exp2(t0 - sh) + fast_exp2(t1 - sh)) + (fast_exp2(t2 - sh) + fast_exp2(t3 - sh)));
          }
          f[i] = skr_sgpr(sh + fast_log2(s));
        }
        flog = (lane == i) ? f[i] : flog;
      }
      if (mult > 0) {   // the dummy row, always in the exact form (4 terms per lane)
        float dm = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (q0 + j < c) dm = fmaxf(dm, -g[j]);
        dm = wave_max_f32_dpp(dm);
        float ds = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (q0 + j < c) ds += fast_exp2(-g[j] - dm);
        ds = wave_sum_f32_dpp(ds);
        fd = skr_sgpr(SK_DUMMY + dm + fast_log2(ds));
      }
      if (pt) {
        if (lane < SKR_RW && p0 + lane < r) pt[it * potld + p0 + lane] = flog;
        if (mult > 0 && tid == 0) pt[it * potld + r] = fd;
      }
    } else {
      // ---- cols: g_q = lse over the r real rows and `mult` copies of the dummy row ----
      // (two straight-line instances instead of a retry loop: the loop form doubles the register pressure)
      const float td0 = (mult > 0) ? SK_DUMMY - fd : -INFINITY;
      float gn[4];
      bool exact = it < 2;
      if (!exact) exact = skr_col_sweep<false>(L, f, g, td0, mult, p0, q0, r, c, wave, s_part, buf, gn);
      if (exact) skr_col_sweep<true>(L, f, g, td0, mult, p0, q0, r, c, wave, s_part, buf, gn);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = gn[j];
      if (pt && wave == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (q0 + j < c) pt[it * potld + q0 + j] = g[j];
      }
    }
  }
  unsigned oo[4], mo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { oo[j] = (unsigned)((q0 + j) * (int)pb.oq) * 4u; mo[j] = (unsigned)((q0 + j) * (int)pb.mq) * 4u; }
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i) {
    const int p = p0 + i;
    if (p < r) {
      char* orow = reinterpret_cast<char*>(pb.out + (int64_t)p * pb.op);
      char* mrow = reinterpret_cast<char*>(pb.mir + (int64_t)p * pb.mp);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (q0 + j < c) {
          const float v = fast_exp2(L[i][j] - f[i] - g[j]);
          *reinterpret_cast<float*>(orow + oo[j]) = v;
          if (pb.mir) *reinterpret_cast<float*>(mrow + mo[j]) = v;
        }
      }
    }
  }
}

static inline size_t sk_lds_bytes(int rmax, int cmax, bool mat, int nmat) {
  size_t n = (size_t)(cmax + 1) + cmax + 4;
  if (mat) n += (size_t)nmat * rmax * (cmax | 1);
  return n * sizeof(float);
}
#define SK_LDS_CAP (150 * 1024)

static inline int max_graph(const ttdg_graphs_t& gr) {
  int m = 0;
  for (int g = 0; g < gr.G; ++g) m = gr.off[g + 1] - gr.off[g] > m ? gr.off[g + 1] - gr.off[g] : m;
  return m;
}

extern "C" int ttdg_sinkhorn_pairs_fwd(const float* part, int ksplit, const float* b2, ttdg_graphs_t gr, float tau,
                                       int iters, float* Wds, float* pot, ttdg_stream_t stream) {
  TTDG_REQUIRE(part && Wds && ksplit >= 1 && tau > 0.f, "sinkhorn_pairs_fwd: bad arguments");
  TTDG_REQUIRE(iters >= 0 && iters <= SK_MAXK, "sinkhorn_pairs_fwd: iters out of range");
  if (int e = ttdg_validate_graphs(gr)) return e;
  const int cmax = max_graph(gr);
  const int npairs = gr.G * (gr.G + 1) / 2;
  const bool lds = sk_lds_bytes(cmax, cmax, true, 1) <= SK_LDS_CAP;
  const size_t bytes = sk_lds_bytes(cmax, cmax, lds, 1);
  const int threads = cmax <= 64 ? 256 : 1024;
  hipStream_t st = (hipStream_t)stream;
  if (cmax > 128 && cmax <= SKR_C && ksplit == 1) {
    hipLaunchKernelGGL(sinkhorn_pairs_fwd_reg_kernel, dim3(npairs), dim3(SKR_THREADS), 0, st, part, ksplit, b2, gr, tau, iters,
                       Wds, pot, cmax);
    return ttdg_launch_status("sinkhorn_pairs_fwd_reg");
  }
  if (lds) {
    TTDG_ALLOW_LDS((sinkhorn_pairs_fwd_kernel<true>), bytes);
    hipLaunchKernelGGL((sinkhorn_pairs_fwd_kernel<true>), dim3(npairs), dim3(threads), bytes, st, part, ksplit, b2, gr, tau,
                       iters, Wds, pot, cmax);
  } else {
    hipLaunchKernelGGL((sinkhorn_pairs_fwd_kernel<false>), dim3(npairs), dim3(threads), bytes, st, part, ksplit, b2, gr,
                       tau, iters, Wds, pot, cmax);
  }
  return ttdg_launch_status("sinkhorn_pairs_fwd");
}

// ---- backward of the pair stage ------------------------------------------------------------------
// dY = dOut * out on the real rows, 0 on the dummy row; for k = K-1 .. 0:
//   line sums  S = sum over the line of dY   (dummy row counted `mult` times in column lines)
//   dY -= exp(y^(k)) * S,   y^(k) = L - f^(k) - g^(k) rebuilt from the logged potentials.
// dL = dY after sweep 0;  dM = dL / tau.
// Backward of one oriented Sinkhorn problem (shared by the pair stage and the stand-alone batched operator).
// pt: logged potentials of the forward (iters x potld); dout(p,q) / dm(p,q) addressed with the given strides.
// smem carve: [f: c+1][g: c][ls: c+1][dd: c][L: r*ldm (kLds)][dY: r*ldm (kLds)]
template <bool kLds>
__device__ void sk_backward(const SkProb& pb, const float* __restrict__ pt, int potld, const float* __restrict__ dout, int64_t dop,
                            int64_t doq, float* __restrict__ dm, int64_t dmp, int64_t dmq, float* smem, int iters, float tau) {
  const int r = pb.r, c = pb.c, mult = pb.mult;
  const int ldm = c | 1;
  float* f = smem;            // r+1
  float* g = f + c + 1;       // c
  float* ls = g + c;          // line sums, c+1
  float* dd = ls + c + 1;     // dY of the dummy row, c
  float* mat = dd + c;        // L   (kLds)
  float* dy = kLds ? mat + r * ldm : nullptr;  // dY  (kLds) else lives in the dM block
  const int tid = threadIdx.x, nthr = blockDim.x;
  auto DY = [&](int p, int q) -> float& { return kLds ? dy[p * ldm + q] : dm[p * dmp + q * dmq]; };
  auto LV = [&](int p, int q) -> float { return kLds ? mat[p * ldm + q] : sk_load(pb, p, q); };

  // final potentials: last row sweep / last col sweep
  const int klast_row = (iters - 1) & ~1, klast_col = ((iters - 1) & 1) ? iters - 1 : iters - 2;
  for (int p = tid; p <= r; p += nthr) f[p] = (iters >= 1 && (p < r || mult > 0)) ? pt[klast_row * potld + p] : 0.f;
  for (int q = tid; q < c; q += nthr) { g[q] = (klast_col >= 1) ? pt[klast_col * potld + q] : 0.f; dd[q] = 0.f; }
  if (kLds)
    for (int e = tid; e < r * c; e += nthr) { const int p = e / c, q = e - p * c; mat[p * ldm + q] = sk_load(pb, p, q); }
  __syncthreads();
  for (int e = tid; e < r * c; e += nthr) {
    const int p = e / c, q = e - p * c;
    const float out = fast_exp2(LV(p, q) - f[p] - g[q]);
    DY(p, q) = dout[p * dop + q * doq] * out;
  }
  __syncthreads();

  for (int k = iters - 1; k >= 0; --k) {
    const int sg = sk_group((k & 1) ? r : c);
    const int sl = tid & (sg - 1), sgi = tid / sg, nsg = nthr / sg;
    // potentials as of just after sweep k
    if ((k & 1) == 0) { for (int p = tid; p <= r; p += nthr) if (p < r || mult > 0) f[p] = pt[k * potld + p]; }
    else              { for (int q = tid; q < c; q += nthr) g[q] = pt[k * potld + q]; }
    if ((k & 1) == 0) { for (int q = tid; q < c; q += nthr) g[q] = (k >= 1) ? pt[(k - 1) * potld + q] : 0.f; }
    else              { for (int p = tid; p <= r; p += nthr) if (p < r || mult > 0) f[p] = pt[(k - 1) * potld + p]; }
    __syncthreads();
    if ((k & 1) == 0) {
      const int nlines = r + (mult > 0 ? 1 : 0);
      for (int p = sgi; p < nlines; p += nsg) {
        float s = 0.f;
        if (p == r) { for (int q = sl; q < c; q += sg) s += dd[q]; }
        else        { for (int q = sl; q < c; q += sg) s += DY(p, q); }
        s = sub_sum(s, sg);
        if (sl == 0) ls[p] = s;
      }
      __syncthreads();
      for (int e = tid; e < r * c; e += nthr) {
        const int p = e / c, q = e - p * c;
        DY(p, q) -= fast_exp2(LV(p, q) - f[p] - g[q]) * ls[p];
      }
      if (mult > 0)
        for (int q = tid; q < c; q += nthr) dd[q] -= fast_exp2(SK_DUMMY - f[r] - g[q]) * ls[r];
    } else {
      for (int q = sgi; q < c; q += nsg) {
        float s = 0.f;
        for (int p = sl; p < r; p += sg) s += DY(p, q);
        s = sub_sum(s, sg);
        if (sl == 0) ls[q] = s + (float)mult * dd[q];
      }
      __syncthreads();
      for (int e = tid; e < r * c; e += nthr) {
        const int p = e / c, q = e - p * c;
        DY(p, q) -= fast_exp2(LV(p, q) - f[p] - g[q]) * ls[q];
      }
      if (mult > 0)
        for (int q = tid; q < c; q += nthr) dd[q] -= fast_exp2(SK_DUMMY - f[r] - g[q]) * ls[q];
    }
    __syncthreads();
  }
  const float inv_tau = 1.f / tau;
  for (int e = tid; e < r * c; e += nthr) {
    const int p = e / c, q = e - p * c;
    dm[p * dmp + q * dmq] = DY(p, q) * inv_tau;
  }
}

template <bool kLds>
__global__ void sinkhorn_pairs_bwd_kernel(const float* __restrict__ part, int ksplit, const float* __restrict__ b2,
                                          const float* __restrict__ pot, const float* __restrict__ dWds,
                                          ttdg_graphs_t gr, float tau, int iters, float* __restrict__ dM, int cmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // pairs with a > b only: index over (1,0),(2,0),(2,1),...
  int a = 1, idx = blockIdx.x;
  while (idx >= a) { idx -= a; ++a; }
  const int b = idx;
  const int pair_fwd = a * (a + 1) / 2 + b;
  const int M = gr.off[gr.G];
  const int na = gr.off[a + 1] - gr.off[a], nb = gr.off[b + 1] - gr.off[b];
  const float* blk = part + (size_t)gr.off[a] * M + gr.off[b];
  const float* dout = dWds + (size_t)gr.off[b] * M + gr.off[a];  // loss reads Wds[b-rows, a-cols] = out^T
  float* dm = dM + (size_t)gr.off[a] * M + gr.off[b];
  SkProb pb;
  pb.src = blk; pb.splane = (int64_t)M * M; pb.nplanes = ksplit;
  pb.bias = b2 ? *b2 : 0.f; pb.scale = TTDG_LOG2E / tau;
  int64_t dop, doq, dmp, dmq;  // strides of dOut / dM in oriented (p,q) coordinates
  if (nb >= na) { pb.r = na; pb.c = nb; pb.sp = M; pb.sq = 1; dop = 1; doq = M; dmp = M; dmq = 1; }
  else          { pb.r = nb; pb.c = na; pb.sp = 1; pb.sq = M; dop = M; doq = 1; dmp = 1; dmq = M; }
  pb.mult = pb.c - pb.r;
  const int potld = cmax + 1;
  sk_backward<kLds>(pb, pot + (size_t)pair_fwd * iters * potld, potld, dout, dop, doq, dm, dmp, dmq, smem, iters, tau);
}

// register-resident backward for 128 < c <= 256: dY lives in registers (16 x 4 per lane, same ownership as the forward
// kernel), L is streamed from L2 every sweep in groups of 4 rows (the next group's 16 loads are in flight under the
// current group's exps), potentials come from the forward log; row sums are DPP wavefront reductions, column sums meet
// in LDS with one barrier per column sweep (double-buffered, every wavefront finishes all columns).
#define SKR_GROUP 4
__device__ __forceinline__ void skr_load_group(const SkProb& pb, int p0, int grp, int r, const unsigned (&qo)[4], float (&dst)[SKR_GROUP][4]) {
#pragma unroll
  for (int a = 0; a < SKR_GROUP; ++a) {
    const char* rowp = reinterpret_cast<const char*>(pb.src + (int64_t)min(p0 + grp * SKR_GROUP + a, r - 1) * pb.sp);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[a][j] = *reinterpret_cast<const float*>(rowp + qo[j]);
  }
}

__global__ __launch_bounds__(SKR_THREADS) void sinkhorn_pairs_bwd_reg_kernel(const float* __restrict__ part, int ksplit,
                                                                             const float* __restrict__ b2, const float* __restrict__ pot,
                                                                             const float* __restrict__ dWds, ttdg_graphs_t gr, float tau,
                                                                             int iters, float* __restrict__ dM, int cmax) {
  __shared__ __attribute__((aligned(16))) float s_part[2][SKR_WAVES * SKR_C];
  int a = 1, idx = blockIdx.x;
  while (idx >= a) { idx -= a; ++a; }
  const int b = idx;
  const int pair_fwd = a * (a + 1) / 2 + b;
  const int M = gr.off[gr.G];
  SkProb pb;
  sk_pair_problem(part, ksplit, b2, gr, tau, a, b, pb);
  const bool flip = pb.sq != 1;
  const float* dout = dWds + (size_t)gr.off[b] * M + gr.off[a];
  float* dm = dM + (size_t)gr.off[a] * M + gr.off[b];
  const int64_t dop = flip ? M : 1, dmp = flip ? 1 : M;
  const int doq = flip ? 1 : M, dmq = flip ? M : 1;
  const int r = pb.r, c = pb.c, mult = pb.mult, potld = cmax + 1;
  const float* pt = pot + (size_t)pair_fwd * iters * potld;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave: provably uniform -> scalar row pointers
  const int p0 = wave * SKR_RW, q0 = lane * 4;
  const float bias = pb.bias, scale = pb.scale;
  unsigned qo[4], qd[4], qm[4];   // 32-bit byte offsets of the lane's columns in L / dOut / dM
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = min(q0 + j, c - 1);
    qo[j] = (unsigned)(q * (int)pb.sq) * 4u; qd[j] = (unsigned)(q * doq) * 4u; qm[j] = (unsigned)(q * dmq) * 4u;
  }

  float f[SKR_RW], g[4], fd, dY[SKR_RW][4], dd[4] = {0.f, 0.f, 0.f, 0.f};
  // potentials of sweep index kf (rows) / kg (cols); a negative index means "still zero"
#define SKR_LOAD_POT(kf, kg)                                                                             \
  {                                                                                                      \
    const float fv = ((kf) >= 0 && lane < SKR_RW && p0 + lane < r) ? pt[(kf) * potld + p0 + lane] : 0.f;     \
    _Pragma("unroll") for (int i = 0; i < SKR_RW; ++i) f[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fv), i)); \
    fd = ((kf) >= 0 && mult > 0) ? pt[(kf) * potld + r] : 0.f;                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) g[j] = ((kg) >= 0 && q0 + j < c) ? pt[(kg) * potld + q0 + j] : 0.f; \
  }
  {
    const int klast_row = (iters - 1) & ~1, klast_col = ((iters - 1) & 1) ? iters - 1 : iters - 2;
    SKR_LOAD_POT(klast_row, (klast_col >= 1 ? klast_col : -1))
  }
  float Lb[2][SKR_GROUP][4];
  // dY = dOut * out on the real rows
  skr_load_group(pb, p0, 0, r, qo, Lb[0]);
#pragma unroll
  for (int grp = 0; grp < SKR_RW / SKR_GROUP; ++grp) {
    if (grp + 1 < SKR_RW / SKR_GROUP) skr_load_group(pb, p0, grp + 1, r, qo, Lb[(grp + 1) & 1]);
#pragma unroll
    for (int a2 = 0; a2 < SKR_GROUP; ++a2) {
      const int i = grp * SKR_GROUP + a2, p = p0 + i;
      const char* drow = reinterpret_cast<const char*>(dout + (int64_t)min(p, r - 1) * dop);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float o = fast_exp2((Lb[grp & 1][a2][j] + bias) * scale - f[i] - g[j]);
        dY[i][j] = (p < r && q0 + j < c) ? *reinterpret_cast<const float*>(drow + qd[j]) * o : 0.f;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  int buf = 0;
  for (int k = iters - 1; k >= 0; --k) {
    if ((k & 1) == 0) { SKR_LOAD_POT(k, (k >= 1 ? k - 1 : -1)) } else { SKR_LOAD_POT(k - 1, k) }
    float de[4];   // exp(y) of the dummy row in the lane's columns
#pragma unroll
    for (int j = 0; j < 4; ++j) de[j] = (mult > 0 && q0 + j < c) ? fast_exp2(SK_DUMMY - fd - g[j]) : 0.f;
    float ls[4] = {0.f, 0.f, 0.f, 0.f};
    const bool rows = (k & 1) == 0;
    if (!rows) {
      float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < SKR_RW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[j] += dY[i][j];
      *reinterpret_cast<float4*>(&s_part[buf][wave * SKR_C + q0]) = make_float4(cs[0], cs[1], cs[2], cs[3]);
      __syncthreads();
#pragma unroll
      for (int w = 0; w < SKR_WAVES; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(&s_part[buf][w * SKR_C + q0]);
        ls[0] += v.x; ls[1] += v.y; ls[2] += v.z; ls[3] += v.w;
      }
      buf ^= 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) ls[j] += (float)mult * dd[j];
    }
    skr_load_group(pb, p0, 0, r, qo, Lb[0]);
#pragma unroll
    for (int grp = 0; grp < SKR_RW / SKR_GROUP; ++grp) {
      if (grp + 1 < SKR_RW / SKR_GROUP) skr_load_group(pb, p0, grp + 1, r, qo, Lb[(grp + 1) & 1]);
#pragma unroll
      for (int a2 = 0; a2 < SKR_GROUP; ++a2) {
        const int i = grp * SKR_GROUP + a2;
        if (p0 + i < r) {
          float lrow = 0.f;
          if (rows) lrow = wave_sum_f32_dpp((dY[i][0] + dY[i][1]) + (dY[i][2] + dY[i][3]));
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float e = fast_exp2((Lb[grp & 1][a2][j] + bias) * scale - f[i] - g[j]);
            if (q0 + j < c) dY[i][j] -= e * (rows ? lrow : ls[j]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (rows) {
      if (mult > 0) {
        const float lsd = wave_sum_f32_dpp((dd[0] + dd[1]) + (dd[2] + dd[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) dd[j] -= de[j] * lsd;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) dd[j] -= de[j] * ls[j];
    }
  }
#undef SKR_LOAD_POT
  const float inv_tau = 1.f / tau;
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i) {
    const int p = p0 + i;
    if (p < r) {
      char* mrow = reinterpret_cast<char*>(dm + (int64_t)p * dmp);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (q0 + j < c) *reinterpret_cast<float*>(mrow + qm[j]) = dY[i][j] * inv_tau;
    }
  }
}

extern "C" int ttdg_sinkhorn_pairs_bwd(const float* part, int ksplit, const float* b2, const float* pot,
                                       const float* dWds, ttdg_graphs_t gr, float tau, int iters, float* dM,
                                       ttdg_stream_t stream) {
  TTDG_REQUIRE(part && pot && dWds && dM && ksplit >= 1 && tau > 0.f, "sinkhorn_pairs_bwd: bad arguments");
  TTDG_REQUIRE(iters >= 1 && iters <= SK_MAXK, "sinkhorn_pairs_bwd: iters out of range");
  if (int e = ttdg_validate_graphs(gr)) return e;
  if (gr.G < 2) return 0;
  const int cmax = max_graph(gr);
  const int npairs = gr.G * (gr.G - 1) / 2;
  const size_t base = (size_t)(4 * (cmax + 1) + 4) * sizeof(float);
  const size_t need = base + (size_t)2 * cmax * (cmax | 1) * sizeof(float);
  const bool lds = need <= SK_LDS_CAP;
  const size_t bytes = lds ? need : base;
  const int threads = cmax <= 64 ? 256 : 1024;
  hipStream_t st = (hipStream_t)stream;
  if (cmax > 128 && cmax <= SKR_C && ksplit == 1) {
    hipLaunchKernelGGL(sinkhorn_pairs_bwd_reg_kernel, dim3(npairs), dim3(SKR_THREADS), 0, st, part, ksplit, b2, pot, dWds, gr, tau,
                       iters, dM, cmax);
    return ttdg_launch_status("sinkhorn_pairs_bwd_reg");
  }
  if (lds) {
    TTDG_ALLOW_LDS((sinkhorn_pairs_bwd_kernel<true>), bytes);
    hipLaunchKernelGGL((sinkhorn_pairs_bwd_kernel<true>), dim3(npairs), dim3(threads), bytes, st, part, ksplit, b2, pot, dWds,
                       gr, tau, iters, dM, cmax);
  } else {
    hipLaunchKernelGGL((sinkhorn_pairs_bwd_kernel<false>), dim3(npairs), dim3(threads), bytes, st, part, ksplit, b2, pot,
                       dWds, gr, tau, iters, dM, cmax);
  }
  return ttdg_launch_status("sinkhorn_pairs_bwd");
}

// ---- stand-alone batched operator (GModule.utils.sinkhorn.Sinkhorn.forward) ------------------------
template <bool kLds>
__global__ void sinkhorn_batched_kernel(const float* __restrict__ s, int64_t sb, int64_t sr, int64_t sc, int R, int C,
                                        const int32_t* __restrict__ n1, const int32_t* __restrict__ n2, int dummy,
                                        float tau, int iters, float* __restrict__ out, float* __restrict__ pot) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bi = blockIdx.x;
  // Appendix B steps 1-3: the valid block is nr x nc inside the (R, C) frame; orient it rows <= cols.
  // Whole-batch transposition (C < R) and per-matrix transposition (nr > nc) both reduce to "the shorter
  // valid side becomes the row axis"; on ties the frame orientation decides (rows stay rows).
  int nr = n1 ? n1[bi] : R, nc = n2 ? n2[bi] : C;
  float* o = out + (size_t)bi * R * C;
  for (int e = threadIdx.x; e < R * C; e += blockDim.x) o[e] = 0.f;  // padding -> exp(-inf) = 0
  __syncthreads();
  if (nr <= 0 || nc <= 0) return;
  SkProb pb;
  pb.src = s + bi * sb; pb.splane = 0; pb.nplanes = 1; pb.bias = 0.f; pb.scale = TTDG_LOG2E / tau;
  bool flip = (C < R);            // step 1
  int fr = flip ? nc : nr, fc = flip ? nr : nc;  // valid sizes in the flipped frame
  if (fr > fc) flip = !flip;      // step 3
  if (!flip) { pb.r = nr; pb.c = nc; pb.sp = sr; pb.sq = sc; pb.op = C; pb.oq = 1; }
  else       { pb.r = nc; pb.c = nr; pb.sp = sc; pb.sq = sr; pb.op = 1; pb.oq = C; }
  pb.out = o; pb.mir = nullptr; pb.mp = pb.mq = 0;
  pb.mult = dummy ? pb.c - pb.r : 0;
  pb.potld = (R > C ? R : C) + 1;
  pb.pot = pot ? pot + (size_t)bi * iters * pb.potld : nullptr;
  sk_forward<kLds>(pb, smem, iters);
}

// backward of the stand-alone operator: ds = d loss / d s from dout = d loss / d out and the logged potentials
template <bool kLds>
__global__ void sinkhorn_batched_bwd_kernel(const float* __restrict__ s, int64_t sb, int64_t sr, int64_t sc, int R, int C,
                                            const int32_t* __restrict__ n1, const int32_t* __restrict__ n2, int dummy,
                                            float tau, int iters, const float* __restrict__ pot,
                                            const float* __restrict__ dout, float* __restrict__ ds) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bi = blockIdx.x;
  int nr = n1 ? n1[bi] : R, nc = n2 ? n2[bi] : C;
  float* d = ds + (size_t)bi * R * C;
  for (int e = threadIdx.x; e < R * C; e += blockDim.x) d[e] = 0.f;      // padding takes no gradient
  __syncthreads();
  if (nr <= 0 || nc <= 0) return;
  SkProb pb;
  pb.src = s + bi * sb; pb.splane = 0; pb.nplanes = 1; pb.bias = 0.f; pb.scale = TTDG_LOG2E / tau;
  bool flip = (C < R);
  int fr = flip ? nc : nr, fc = flip ? nr : nc;
  if (fr > fc) flip = !flip;
  int64_t op, oq;      // strides of the contiguous (R, C) gradient buffers in oriented (p, q) coordinates
  if (!flip) { pb.r = nr; pb.c = nc; pb.sp = sr; pb.sq = sc; op = C; oq = 1; }
  else       { pb.r = nc; pb.c = nr; pb.sp = sc; pb.sq = sr; op = 1; oq = C; }
  pb.mult = dummy ? pb.c - pb.r : 0;
  const int potld = (R > C ? R : C) + 1;
  sk_backward<kLds>(pb, pot + (size_t)bi * iters * potld, potld, dout + (size_t)bi * R * C, op, oq, d, op, oq, smem, iters, tau);
}

extern "C" int ttdg_sinkhorn_batched_bwd(const float* s, int64_t sb, int64_t sr, int64_t sc, int b, int r, int c,
                                         const int32_t* n1, const int32_t* n2, int dummy_row, float tau, int iters,
                                         const float* pot, const float* dout, float* ds, ttdg_stream_t stream) {
  TTDG_REQUIRE(s && pot && dout && ds && b >= 0 && r > 0 && c > 0 && tau > 0.f, "sinkhorn_batched_bwd: bad arguments");
  TTDG_REQUIRE(iters >= 1 && iters <= SK_MAXK, "sinkhorn_batched_bwd: iters out of range (the potentials of every sweep are logged)");
  if (b == 0) return 0;
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  const size_t base = (size_t)(4 * (hi + 1) + 4) * sizeof(float);
  const size_t need = base + (size_t)2 * lo * (hi | 1) * sizeof(float);
  const bool lds = need <= SK_LDS_CAP;
  const size_t bytes = lds ? need : base;
  const int threads = hi <= 64 ? 256 : 1024;
  hipStream_t st = (hipStream_t)stream;
  if (lds) {
    TTDG_ALLOW_LDS((sinkhorn_batched_bwd_kernel<true>), bytes);
    hipLaunchKernelGGL((sinkhorn_batched_bwd_kernel<true>), dim3(b), dim3(threads), bytes, st, s, sb, sr, sc, r, c, n1, n2, dummy_row,
                       tau, iters, pot, dout, ds);
  } else {
    hipLaunchKernelGGL((sinkhorn_batched_bwd_kernel<false>), dim3(b), dim3(threads), bytes, st, s, sb, sr, sc, r, c, n1, n2,
                       dummy_row, tau, iters, pot, dout, ds);
  }
  return ttdg_launch_status("sinkhorn_batched_bwd");
}

extern "C" int ttdg_sinkhorn_batched_fwd(const float* s, int64_t sb, int64_t sr, int64_t sc, int b, int r, int c,
                                         const int32_t* n1, const int32_t* n2, int dummy_row, float tau, int iters,
                                         float* out, float* pot, ttdg_stream_t stream) {
  TTDG_REQUIRE(s && out && b >= 0 && r > 0 && c > 0 && tau > 0.f, "sinkhorn_batched: bad arguments");
  TTDG_REQUIRE(iters >= 0 && iters <= 4096, "sinkhorn_batched: iters out of range");
  TTDG_REQUIRE(!pot || iters <= SK_MAXK, "sinkhorn_batched: potentials can be logged for at most 64 sweeps");
  if (b == 0) return 0;
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  const bool lds = sk_lds_bytes(lo, hi, true, 1) <= SK_LDS_CAP;
  const size_t bytes = sk_lds_bytes(lo, hi, lds, 1);
  const int threads = hi <= 64 ? 256 : 1024;
  hipStream_t st = (hipStream_t)stream;
  if (lds) {
    TTDG_ALLOW_LDS((sinkhorn_batched_kernel<true>), bytes);
    hipLaunchKernelGGL((sinkhorn_batched_kernel<true>), dim3(b), dim3(threads), bytes, st, s, sb, sr, sc, r, c, n1, n2,
                       dummy_row, tau, iters, out, pot);
  } else {
    hipLaunchKernelGGL((sinkhorn_batched_kernel<false>), dim3(b), dim3(threads), bytes, st, s, sb, sr, sc, r, c, n1, n2,
                       dummy_row, tau, iters, out, pot);
  }
  return ttdg_launch_status("sinkhorn_batched");
}
