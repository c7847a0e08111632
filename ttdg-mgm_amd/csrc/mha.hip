// A3 — intra-graph attention adjacency (reference utils/attentions.py:60-86, version 'v2', one head;
// caller multi_graph_matching.py:496-502,571-574).  Only the attention map is used downstream: the
// value/context/output-projection/LayerNorm branch of the reference is dead code on this path.
//
// q, k are the (M, d) projections of the stacked nodes (gemm.hip).  One wavefront per attention row:
// the row's q lives in LDS (broadcast ds_read_b128), every lane dots it with its own k_j (rows of k
// stay L1/L2 resident: a graph's k block is n*1 KiB), then a wavefront softmax over the n scores.
// The block-diagonal A is stored packed, graph after graph, with its diagonal zeroed
// (A.fill_diagonal_(0), multi_graph_matching.py:502).
#include "common.h"

// Philox4x32-10 (Salmon et al. 2011): counter-based, so the dropout mask depends only on
// (seed, graph, row, col) and not on launch geometry.
__device__ __forceinline__ uint32_t philox_uniform_bits(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = 0x9E3779B9u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
    const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1;
    const uint32_t y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
    x0 = y0; x1 = y1; x2 = y2; x3 = y3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return x0;
}

#define MHA_ROWS 4  // rows (wavefronts) per workgroup

__global__ __launch_bounds__(256) void mha_adjacency_kernel(const float* __restrict__ q, const float* __restrict__ k, int d,
                                                            ttdg_graphs_t gr, float scale, float drop_p, uint64_t seed,
                                                            int zero_diag, float* __restrict__ Apack, int nmax) {
  extern __shared__ __attribute__((aligned(16))) float mha_smem[];  // [MHA_ROWS][d] q rows + [MHA_ROWS][nmax] scores
  const int g = blockIdx.y;
  const int n = gr.off[g + 1] - gr.off[g];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * MHA_ROWS + wave;
  if (blockIdx.x * MHA_ROWS >= n) return;
  float* qs = mha_smem + wave * d;
  float* sc = mha_smem + MHA_ROWS * d + wave * nmax;
  size_t aoff = 0;
  for (int h = 0; h < g; ++h) { const size_t m = gr.off[h + 1] - gr.off[h]; aoff += m * m; }
  const bool active = i < n;
  if (active)
    for (int e = lane * 4; e < d; e += 256)
      *reinterpret_cast<float4*>(qs + e) = *reinterpret_cast<const float4*>(q + (size_t)(gr.off[g] + i) * d + e);
  __syncthreads();
  if (!active) return;

  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64) {
    const float* kr = k + (size_t)(gr.off[g] + j) * d;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int e = 0; e < d; e += 4) {
      const float4 kv = *reinterpret_cast<const float4*>(kr + e);
      const float4 qv = *reinterpret_cast<const float4*>(qs + e);
      a0 = fmaf(qv.x, kv.x, a0); a1 = fmaf(qv.y, kv.y, a1); a2 = fmaf(qv.z, kv.z, a2); a3 = fmaf(qv.w, kv.w, a3);
    }
    const float s = ((a0 + a1) + (a2 + a3)) * scale;
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) {
    const float e = fast_exp2((sc[j] - mx) * TTDG_LOG2E);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  float* arow = Apack + aoff + (size_t)i * n;
  for (int j = lane; j < n; j += 64) {
    float v = sc[j] * inv;
    if (drop_p > 0.f) {
      const float u = (float)(philox_uniform_bits(seed, (uint32_t)g, (uint32_t)i, (uint32_t)j) >> 8) * (1.f / 16777216.f);
      v = (u < drop_p) ? 0.f : v * keep_scale;
    }
    arow[j] = (zero_diag && j == i) ? 0.f : v;
  }
}

// [r5] graphs of 128+ nodes (BASELINE cfg-3: 8 x 256): the one-wavefront-per-row kernel above has every lane stream its OWN k row
// (64 cache lines per load instruction: 61 us at cfg-3, address-path bound, VALU 7 % active).  There the scores are a plain
// product: q_g k_g^T * scale by the grouped MFMA GEMM (gemm_grouped.hip), written straight into the graph's packed block of
// Apack, and this kernel turns every row into its softmax IN PLACE - one wavefront per row, coalesced, the same exponent / sum /
// dropout / diagonal arithmetic as above.
__global__ __launch_bounds__(256) void mha_softmax_rows_kernel(ttdg_graphs_t gr, float drop_p, uint64_t seed, int zero_diag,
                                                               float* __restrict__ Apack) {
  const int g = blockIdx.y;
  const int n = gr.off[g + 1] - gr.off[g];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * MHA_ROWS + wave;
  if (i >= n) return;
  size_t aoff = 0;
  for (int h = 0; h < g; ++h) { const size_t m = gr.off[h + 1] - gr.off[h]; aoff += m * m; }
  float* arow = Apack + aoff + (size_t)i * n;
  float sc[16];                                     // n <= 1024: sixteen scores per lane
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int j = lane + 64 * t;
    sc[t] = j < n ? arow[j] : -INFINITY;
    mx = fmaxf(mx, sc[t]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int j = lane + 64 * t;
    sc[t] = j < n ? fast_exp2((sc[t] - mx) * TTDG_LOG2E) : 0.f;
    sum += sc[t];
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int j = lane + 64 * t;
    if (j < n) {
      float v = sc[t] * inv;
      if (drop_p > 0.f) {
        const float u = (float)(philox_uniform_bits(seed, (uint32_t)g, (uint32_t)i, (uint32_t)j) >> 8) * (1.f / 16777216.f);
        v = (u < drop_p) ? 0.f : v * keep_scale;
      }
      arow[j] = (zero_diag && j == i) ? 0.f : v;
    }
  }
}

extern "C" int ttdg_mha_adjacency(const float* q, const float* k, int d, ttdg_graphs_t gr, float scale, float drop_p,
                                  uint64_t seed, int zero_diag, float* Apack, ttdg_stream_t stream) {
  TTDG_REQUIRE(q && k && Apack && d > 0 && d % 4 == 0, "mha_adjacency: bad arguments");
  TTDG_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "mha_adjacency: dropout probability out of range");
  if (int e = ttdg_validate_graphs(gr)) return e;
  int nmax = 0;
  for (int g = 0; g < gr.G; ++g) nmax = gr.off[g + 1] - gr.off[g] > nmax ? gr.off[g + 1] - gr.off[g] : nmax;
  if (nmax >= 128 && nmax <= 1024) {
    // scores by the grouped GEMM, eight graphs per launch, then the in-place row softmax
    size_t aoff = 0;
    ttdg_gemm_desc_t desc[TTDG_GEMM_GROUP_MAX];
    int nd = 0;
    for (int g = 0; g < gr.G; ++g) {
      const int n = gr.off[g + 1] - gr.off[g];
      ttdg_gemm_desc_t& p = desc[nd];
      p.A = q + (size_t)gr.off[g] * d; p.sam = d; p.sak = 1;
      p.B = k + (size_t)gr.off[g] * d; p.sbn = d; p.sbk = 1;
      p.C = Apack + aoff; p.scm = n; p.scn = 1;
      p.bias = nullptr; p.A2 = nullptr; p.B2 = nullptr; p.sam2 = p.sak2 = p.sbn2 = p.sbk2 = 0;
      p.M = n; p.N = n; p.K = d; p.K2 = 0; p.alpha = scale; p.beta = 0.f;
      aoff += (size_t)n * n;
      if (++nd == TTDG_GEMM_GROUP_MAX || g == gr.G - 1) {
        if (int e = ttdg_gemm_f32_grouped(desc, nd, stream)) return e;
        nd = 0;
      }
    }
    hipLaunchKernelGGL(mha_softmax_rows_kernel, dim3((nmax + MHA_ROWS - 1) / MHA_ROWS, gr.G), dim3(256), 0, (hipStream_t)stream, gr,
                       drop_p, seed, zero_diag, Apack);
    return ttdg_launch_status("mha_softmax_rows");
  }
  const size_t bytes = (size_t)MHA_ROWS * (d + nmax) * sizeof(float);
  TTDG_LIMIT(bytes <= 150 * 1024, "mha_adjacency: graph too large");
  TTDG_ALLOW_LDS(mha_adjacency_kernel, bytes);
  hipLaunchKernelGGL(mha_adjacency_kernel, dim3((nmax + MHA_ROWS - 1) / MHA_ROWS, gr.G), dim3(256), bytes,
                     (hipStream_t)stream, q, k, d, gr, scale, drop_p, seed, zero_diag, Apack, nmax);
  return ttdg_launch_status("mha_adjacency");
}
