// Grouped fp32 GEMM: up to 8 independent products in ONE launch.
//
// The matching step of a TTA batch issues fifteen small GEMMs (M = sum n_g ~ 120 rows; N, K in {32, 256, 512}): the five
// projections of the stacked node features (affinity.py:46-47, attentions.py:72-74, multi_graph_matching.py:531), the two
// hidden-layer halves (affinity.py:55 decomposed), and their eight gradient products.  As separate launches of 64 x 64
// tiles each filled 8..32 of the 256 CUs and cost 20-48 us apiece (0.49 ms per step, half of the solver).  Here the products
// that do not depend on each other share a launch, the output tile is 32 x 32 (M ~ 120, N = 256 -> 32 workgroups per
// product, 130+ per group), and the four wavefronts of a workgroup split K: every 64-deep slab is staged once through LDS
// ([k][m] order, stride 33: transposing stores and fragment reads are both conflict-free) and wavefront w runs the MFMAs of
// k in [16w, 16w + 16); the four 32 x 32 partial accumulators meet in LDS in a fixed order (deterministic).
//     C[m,n] = alpha * ( sum_k A(m,k) B(n,k)  +  sum_k A2(m,k) B2(n,k) ) + bias[n] + beta * C[m,n]
// with arbitrary element strides (the same descriptor serves x W^T, dY W and dY^T X) and an optional second K segment
// (dX = dXs Psr + dXt Ptg in one pass instead of a beta = 1 second launch).  v_mfma_f32_32x32x2_f32: exact fp32.
#include "common.h"

typedef float gg_f32x16 __attribute__((ext_vector_type(16)));
#define GG_T 32
#define GG_BK 64
#define GG_LD 33

struct GgGroup {
  int n;
  int tile_end[TTDG_GEMM_GROUP_MAX];
  ttdg_gemm_desc_t p[TTDG_GEMM_GROUP_MAX];
};

template <bool kcontig>
__device__ __forceinline__ void gg_stage(float (*T)[GG_LD], const float* __restrict__ X, int64_t sm, int64_t sk, int m0, int k0, int Mlim,
                                         int Klim, int tid) {
  if (kcontig) {
    const int k = tid & 63, mb = tid >> 6;       // 64 lanes walk k (256 contiguous bytes), 4 row groups
    const int gk = k0 + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = mb + 4 * r, gm = m0 + m;
      T[k][m] = (gm < Mlim && gk < Klim) ? X[gm * sm + gk * sk] : 0.f;
    }
  } else {
    const int m = tid & 31, kb = tid >> 5;       // 32 lanes walk m
    const int gm = m0 + m;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = kb + 8 * r, gk = k0 + k;
      T[k][m] = (gm < Mlim && gk < Klim) ? X[gm * sm + gk * sk] : 0.f;
    }
  }
}

__device__ __forceinline__ void gg_stage_any(float (*T)[GG_LD], const float* X, int64_t sm, int64_t sk, int m0, int k0, int Mlim, int Klim,
                                             int tid) {
  if (sk == 1) gg_stage<true>(T, X, sm, sk, m0, k0, Mlim, Klim, tid);
  else gg_stage<false>(T, X, sm, sk, m0, k0, Mlim, Klim, tid);
}

__global__ __launch_bounds__(256) void gemm_grouped_kernel(GgGroup g) {
  __shared__ float As[2][GG_BK][GG_LD];
  __shared__ float Bs[2][GG_BK][GG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int pi = 0;
  while (pi + 1 < g.n && (int)blockIdx.x >= g.tile_end[pi]) ++pi;
  const ttdg_gemm_desc_t& d = g.p[pi];
  const int t = blockIdx.x - (pi ? g.tile_end[pi - 1] : 0);
  const int tn = (d.N + GG_T - 1) / GG_T;
  const int m0 = (t / tn) * GG_T, n0 = (t % tn) * GG_T;

  gg_f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int kh = lane >> 5, mi = lane & 31;
  for (int seg = 0; seg < 2; ++seg) {
    const float* A = seg ? d.A2 : d.A;
    const float* B = seg ? d.B2 : d.B;
    const int K = seg ? d.K2 : d.K;
    if (!A || K <= 0) continue;
    const int64_t sam = seg ? d.sam2 : d.sam, sak = seg ? d.sak2 : d.sak, sbn = seg ? d.sbn2 : d.sbn, sbk = seg ? d.sbk2 : d.sbk;
    const int nk = (K + GG_BK - 1) / GG_BK;
    __syncthreads();          // the previous segment's last slab may still be read
    gg_stage_any(As[0], A, sam, sak, m0, 0, d.M, K, tid);
    gg_stage_any(Bs[0], B, sbn, sbk, n0, 0, d.N, K, tid);
    __syncthreads();
    for (int s = 0; s < nk; ++s) {
      const int cur = s & 1;
      if (s + 1 < nk) {
        gg_stage_any(As[cur ^ 1], A, sam, sak, m0, (s + 1) * GG_BK, d.M, K, tid);
        gg_stage_any(Bs[cur ^ 1], B, sbn, sbk, n0, (s + 1) * GG_BK, d.N, K, tid);
      }
      const int kb = wave * 16;
#pragma unroll
      for (int kk = 0; kk < 16; kk += 2) {
        const float a = As[cur][kb + kk + kh][mi];
        const float b = Bs[cur][kb + kk + kh][mi];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      __syncthreads();
    }
  }
  // the four K-partials meet in LDS (aliasing the staging buffers: everyone is past the last slab), fixed order
  float* red = &As[0][0][0];                      // 4 x 32 x 33 floats <= 2 x 64 x 33
  // C/D fragment: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
    red[(wave * GG_T + row) * GG_LD + mi] = acc[r];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tid + 256 * j, row = e >> 5, col = e & 31;
    const int m = m0 + row, n = n0 + col;
    if (m < d.M && n < d.N) {
      float v = (red[(0 * GG_T + row) * GG_LD + col] + red[(1 * GG_T + row) * GG_LD + col]) +
                (red[(2 * GG_T + row) * GG_LD + col] + red[(3 * GG_T + row) * GG_LD + col]);
      v = d.alpha * v + (d.bias ? d.bias[n] : 0.f);
      float* c = d.C + m * d.scm + n * d.scn;
      if (d.beta != 0.f) v += d.beta * (*c);
      *c = v;
    }
  }
}

extern "C" int ttdg_gemm_f32_grouped(const ttdg_gemm_desc_t* descs, int n, ttdg_stream_t stream) {
  TTDG_REQUIRE(descs && n >= 1 && n <= TTDG_GEMM_GROUP_MAX, "gemm_grouped: 1..8 products per launch");
  GgGroup g;
  g.n = 0;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    const ttdg_gemm_desc_t& d = descs[i];
    TTDG_REQUIRE(d.A && d.B && d.C, "gemm_grouped: null operand");
    TTDG_REQUIRE(d.M >= 0 && d.N >= 0 && d.K >= 0 && d.K2 >= 0, "gemm_grouped: negative size");
    TTDG_REQUIRE(d.K2 == 0 || (d.A2 && d.B2), "gemm_grouped: second K segment without operands");
    if (d.M == 0 || d.N == 0) continue;
    tiles += ((d.M + GG_T - 1) / GG_T) * ((d.N + GG_T - 1) / GG_T);
    g.p[g.n] = d;
    g.tile_end[g.n] = tiles;
    ++g.n;
  }
  if (g.n == 0) return 0;
  hipLaunchKernelGGL(gemm_grouped_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, g);
  return ttdg_launch_status("gemm_grouped");
}
