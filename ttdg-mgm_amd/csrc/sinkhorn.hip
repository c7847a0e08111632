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
#include "common.h"

#define SK_MAXK 64
#define SK_DUMMY (-100.0f * TTDG_LOG2E)

struct SkProb {
  // oriented problem: r <= c; element (p,q) of the input is sum_s src[s*splane + p*sp + q*sq] + bias
  const float* src;
  int64_t sp, sq, splane;
  int nplanes;
  float bias, scale;  // L2 = (x + bias) * scale, scale = log2(e)/tau
  int r, c, mult;     // mult = number of dummy rows (0 when dummy_row is off)
  float* out;         // out[p*op + q*oq] = exp(y)
  int64_t op, oq;
  float* mir;         // optional mirror (transposed copy), may be null
  int64_t mp, mq;
  float* pot;         // optional potentials log: pot[k*(cmax+1) + idx]
  int potld;
};

__device__ __forceinline__ float sk_load(const SkProb& pb, int p, int q) {
  float v = pb.bias;
  const float* s = pb.src + p * pb.sp + q * pb.sq;
  for (int k = 0; k < pb.nplanes; ++k) v += s[k * pb.splane];
  return v * pb.scale;
}

__device__ __forceinline__ float sub_max(float v, int sg) {
  for (int o = sg >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float sub_sum(float v, int sg) {
  for (int o = sg >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LDS carve (dynamic): [f: cmax+1][g: cmax][mat: r*ldm  (kLds only)]
template <bool kLds>
__device__ void sk_forward(const SkProb& pb, float* smem, int iters) {
  const int r = pb.r, c = pb.c, mult = pb.mult;
  float* f = smem;           // r real rows + 1 dummy
  float* g = smem + c + 1;   // c
  float* mat = g + c;        // r x ldm
  const int ldm = c | 1;     // odd stride: column walks are conflict-free
  const int tid = threadIdx.x, nthr = blockDim.x;

  for (int q = tid; q < c; q += nthr) g[q] = 0.f;
  for (int p = tid; p <= r; p += nthr) f[p] = 0.f;
  if (kLds)
    for (int e = tid; e < r * c; e += nthr) {
      const int p = e / c, q = e - p * c;
      mat[p * ldm + q] = sk_load(pb, p, q);
    }
  __syncthreads();

  const int sg = (c > 32) ? 64 : (c > 16 ? 32 : 16);      // lanes per line
  const int sl = tid & (sg - 1), sgi = tid / sg, nsg = nthr / sg;
  for (int it = 0; it < iters; ++it) {
    if ((it & 1) == 0) {
      // rows: f_p = lse_q(L_pq - g_q); the dummy row uses the constant fill
      const int nlines = r + (mult > 0 ? 1 : 0);
      for (int p = sgi; p < nlines; p += nsg) {
        const bool dum = (p == r);
        float m = -INFINITY;
        for (int q = sl; q < c; q += sg) {
          const float t = (dum ? SK_DUMMY : (kLds ? mat[p * ldm + q] : sk_load(pb, p, q))) - g[q];
          m = fmaxf(m, t);
        }
        m = sub_max(m, sg);
        float s = 0.f;
        for (int q = sl; q < c; q += sg) {
          const float t = (dum ? SK_DUMMY : (kLds ? mat[p * ldm + q] : sk_load(pb, p, q))) - g[q];
          s += fast_exp2(t - m);
        }
        s = sub_sum(s, sg);
        if (sl == 0) {
          const float v = m + fast_log2(s);
          f[p] = v;
          if (pb.pot) pb.pot[it * pb.potld + p] = v;
        }
      }
    } else {
      // cols: g_q = lse over the r real rows and `mult` copies of the dummy row
      const float td0 = SK_DUMMY - f[r];
      for (int q = sgi; q < c; q += nsg) {
        float m = (mult > 0) ? td0 : -INFINITY;
        for (int p = sl; p < r; p += sg) m = fmaxf(m, (kLds ? mat[p * ldm + q] : sk_load(pb, p, q)) - f[p]);
        m = sub_max(m, sg);
        float s = 0.f;
        for (int p = sl; p < r; p += sg) s += fast_exp2((kLds ? mat[p * ldm + q] : sk_load(pb, p, q)) - f[p] - m);
        s = sub_sum(s, sg);
        if (mult > 0) s += (float)mult * fast_exp2(td0 - m);
        if (sl == 0) {
          const float v = m + fast_log2(s);
          g[q] = v;
          if (pb.pot) pb.pot[it * pb.potld + q] = v;
        }
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < r * c; e += nthr) {
    const int p = e / c, q = e - p * c;
    const float y = (kLds ? mat[p * ldm + q] : sk_load(pb, p, q)) - f[p] - g[q];
    const float v = fast_exp2(y);
    pb.out[p * pb.op + q * pb.oq] = v;
    if (pb.mir) pb.mir[p * pb.mp + q * pb.mq] = v;
  }
}

// ---- pair stage (multi_graph_matching.py:504-525) --------------------------------------------------
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
  const int r = pb.r, c = pb.c, mult = pb.mult, potld = cmax + 1;
  const float* pt = pot + (size_t)pair_fwd * iters * potld;

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

  const int sg = (c > 32) ? 64 : (c > 16 ? 32 : 16);
  const int sl = tid & (sg - 1), sgi = tid / sg, nsg = nthr / sg;
  for (int k = iters - 1; k >= 0; --k) {
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
                                        float tau, int iters, float* __restrict__ out) {
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
  pb.pot = nullptr; pb.potld = 0;
  sk_forward<kLds>(pb, smem, iters);
}

extern "C" int ttdg_sinkhorn_batched_fwd(const float* s, int64_t sb, int64_t sr, int64_t sc, int b, int r, int c,
                                         const int32_t* n1, const int32_t* n2, int dummy_row, float tau, int iters,
                                         float* out, ttdg_stream_t stream) {
  TTDG_REQUIRE(s && out && b >= 0 && r > 0 && c > 0 && tau > 0.f, "sinkhorn_batched: bad arguments");
  TTDG_REQUIRE(iters >= 0 && iters <= 4096, "sinkhorn_batched: iters out of range");
  if (b == 0) return 0;
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  const bool lds = sk_lds_bytes(lo, hi, true, 1) <= SK_LDS_CAP;
  const size_t bytes = sk_lds_bytes(lo, hi, lds, 1);
  const int threads = hi <= 64 ? 256 : 1024;
  hipStream_t st = (hipStream_t)stream;
  if (lds) {
    TTDG_ALLOW_LDS((sinkhorn_batched_kernel<true>), bytes);
    hipLaunchKernelGGL((sinkhorn_batched_kernel<true>), dim3(b), dim3(threads), bytes, st, s, sb, sr, sc, r, c, n1, n2,
                       dummy_row, tau, iters, out);
  } else {
    hipLaunchKernelGGL((sinkhorn_batched_kernel<false>), dim3(b), dim3(threads), bytes, st, s, sb, sr, sc, r, c, n1, n2,
                       dummy_row, tau, iters, out);
  }
  return ttdg_launch_status("sinkhorn_batched");
}
