// A8/A9 — pseudo-label permutation loss, forward and gradient in one pass.
// Reference: multi_graph_matching.py:535-564 (+ collect_intra_class_matching_wrapper :594-633),
// utils/losses.py:83-103 (BCEFocalLoss, alpha .25, gamma 2, eps 1e-6) and :419-455 (PermutationLoss).
//
// For every unordered graph pair a < b:  s = Wds[a-rows, b-cols],  t = U_a U_b^T  (pseudo ground truth),
//   p = clamp(s, eps, 1-eps);  l = -alpha (1-p)^2 t log p - (1-alpha) p^2 (1-t) log(1-p)
//   loss = (1/#pairs) * sum_pairs mean_elements(l)
// The Hungarian solve the reference also runs on every s (:626) feeds only the unused 'perm_mat_list'
// and is skipped.  One workgroup per 64 x 64 tile of a pair block (the two 64 x 32 slabs of U staged in LDS, the
// thread's U_b row in registers, coalesced Wds / dWds rows), fixed-order reductions (deterministic): per-tile partial
// sums, then a one-workgroup epilogue adds tiles per pair and the per-pair means.  dWds gets d loss / d Wds on the
// a<b blocks (clamp gradient is 1 on [eps, 1-eps], 0 outside, as torch.clamp).
#include "common.h"

#define PL_TILE 64

static inline int pl_tiles(int n) { return (n + PL_TILE - 1) / PL_TILE; }

__device__ __forceinline__ int pl_tiles_d(int n) { return (n + PL_TILE - 1) / PL_TILE; }

// workgroup index -> (pair, tile): pairs (a<b) ordered (0,1),(0,2),(1,2),(0,3)..., tiles row-major inside a pair
__device__ __forceinline__ void pl_locate(const ttdg_graphs_t& gr, int wg, int& a, int& b, int& ti, int& tj, int& pair) {
  pair = 0;
  for (b = 1; b < gr.G; ++b)
    for (a = 0; a < b; ++a, ++pair) {
      const int nt = pl_tiles_d(gr.off[a + 1] - gr.off[a]) * pl_tiles_d(gr.off[b + 1] - gr.off[b]);
      if (wg < nt) {
        const int tb = pl_tiles_d(gr.off[b + 1] - gr.off[b]);
        ti = wg / tb; tj = wg - ti * tb;
        return;
      }
      wg -= nt;
    }
}

__global__ __launch_bounds__(256) void perm_loss_pair_kernel(const float* __restrict__ Wds, const float* __restrict__ U,
                                                             ttdg_graphs_t gr, float alpha, float eps,
                                                             float* __restrict__ tile_loss, float* __restrict__ dWds,
                                                             int32_t* __restrict__ flag) {
  __shared__ __attribute__((aligned(16))) float ua[PL_TILE * TTDG_UNIV];   // rows of graph a: broadcast reads
  __shared__ float ub[PL_TILE * (TTDG_UNIV + 1)];                          // rows of graph b: one row per lane, odd stride
  __shared__ float red[4];
  int a, b, ti, tj, pair;
  pl_locate(gr, blockIdx.x, a, b, ti, tj, pair);
  const int M = gr.off[gr.G];
  const int na = gr.off[a + 1] - gr.off[a], nb = gr.off[b + 1] - gr.off[b];
  const int npairs = gr.G * (gr.G - 1) / 2;
  const float gscale = 1.f / ((float)npairs * (float)na * (float)nb);
  const int i0 = ti * PL_TILE, j0 = tj * PL_TILE;
  const int tid = threadIdx.x;
  for (int e = tid; e < PL_TILE * TTDG_UNIV; e += 256) {
    const int r = e >> 5, u = e & 31;
    ua[e] = (i0 + r < na) ? U[(size_t)(gr.off[a] + i0 + r) * TTDG_UNIV + u] : 0.f;
    ub[r * (TTDG_UNIV + 1) + u] = (j0 + r < nb) ? U[(size_t)(gr.off[b] + j0 + r) * TTDG_UNIV + u] : 0.f;
  }
  __syncthreads();
  const int tx = tid & 63, ty = tid >> 6;
  float uj[TTDG_UNIV];
#pragma unroll
  for (int u = 0; u < TTDG_UNIV; ++u) uj[u] = ub[tx * (TTDG_UNIV + 1) + u];
  float acc = 0.f;
  bool bad = false;
  const int j = j0 + tx;
  // [r5] the thread's sixteen entries of Wds are requested together (clamped addresses, used only where valid): with the load inside
  // the loop every row paid its own round trip behind the previous row's logarithms (22 us for the 448 tiles of cfg-3)
  float sv[PL_TILE / 4];
#pragma unroll
  for (int rr = 0; rr < PL_TILE / 4; ++rr) {
    const int i = min(i0 + ty * (PL_TILE / 4) + rr, na - 1);
    sv[rr] = Wds[(size_t)(gr.off[a] + i) * M + gr.off[b] + min(j, nb - 1)];
  }
#pragma unroll
  for (int rr = 0; rr < PL_TILE / 4; ++rr) {
    const int il = ty * (PL_TILE / 4) + rr, i = i0 + il;
    if (i >= na || j >= nb) continue;
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int u = 0; u < TTDG_UNIV; u += 4) {
      const float4 x = *reinterpret_cast<const float4*>(ua + il * TTDG_UNIV + u);
      t0 = fmaf(x.x, uj[u], t0); t1 = fmaf(x.y, uj[u + 1], t1);
      t0 = fmaf(x.z, uj[u + 2], t0); t1 = fmaf(x.w, uj[u + 3], t1);
    }
    const float t = t0 + t1;
    const size_t o = (size_t)(gr.off[a] + i) * M + gr.off[b] + j;
    const float s = sv[rr];
    bad |= !(s >= 0.f && s <= 1.f) || !(t >= 0.f && t <= 1.f);
    const float p = fminf(fmaxf(s, eps), 1.f - eps);
    const float lp = logf(p), lq = logf(1.f - p);
    const float omp = 1.f - p;
    acc += -alpha * omp * omp * t * lp - (1.f - alpha) * p * p * (1.f - t) * lq;
    float d = -alpha * t * (-2.f * omp * lp + omp * omp / p) - (1.f - alpha) * (1.f - t) * (2.f * p * lq - p * p / omp);
    if (s < eps || s > 1.f - eps) d = 0.f;
    dWds[o] = d * gscale;
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  if (__any(bad) && (tid & 63) == 0) atomicOr(flag, 1);
  __syncthreads();
  if (tid == 0) tile_loss[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// loss = (1/#pairs) * sum_pairs (sum of the pair's tiles, in tile order) / (na nb)
// [r5] one workgroup: thread 0 lays out the pairs' tile ranges (arithmetic only), every thread adds the tiles of its pairs in tile
// order, thread 0 adds the pairs in pair order - the same sums in the same order as the one-thread walk over all tiles it replaces
// (448 dependent loads at cfg-3: 22 us).
#define PL_MAXPAIRS (TTDG_MAX_GRAPHS * (TTDG_MAX_GRAPHS - 1) / 2)
__global__ __launch_bounds__(256) void perm_loss_finish_kernel(const float* __restrict__ tile_loss, ttdg_graphs_t gr, float* __restrict__ loss) {
  __shared__ int pbeg[PL_MAXPAIRS + 1];
  __shared__ float pinv[PL_MAXPAIRS], psum[PL_MAXPAIRS];
  __shared__ int s_np;
  if (threadIdx.x == 0) {
    int wg = 0, npairs = 0;
    for (int b = 1; b < gr.G; ++b)
      for (int a = 0; a < b; ++a, ++npairs) {
        const int na = gr.off[a + 1] - gr.off[a], nb = gr.off[b + 1] - gr.off[b];
        pbeg[npairs] = wg;
        pinv[npairs] = (float)na * (float)nb;
        wg += pl_tiles_d(na) * pl_tiles_d(nb);
      }
    pbeg[npairs] = wg;
    s_np = npairs;
  }
  __syncthreads();
  const int npairs = s_np;
  for (int p = threadIdx.x; p < npairs; p += 256) {
    float ps = 0.f;
    for (int k = pbeg[p]; k < pbeg[p + 1]; ++k) ps += tile_loss[k];
    psum[p] = ps / pinv[p];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int p = 0; p < npairs; ++p) s += psum[p];
    *loss = s / (float)npairs;
  }
}

static int pl_total_tiles(const ttdg_graphs_t& gr) {
  int t = 0;
  for (int b = 1; b < gr.G; ++b)
    for (int a = 0; a < b; ++a) t += pl_tiles(gr.off[a + 1] - gr.off[a]) * pl_tiles(gr.off[b + 1] - gr.off[b]);
  return t;
}

extern "C" size_t ttdg_perm_loss_workspace_bytes(ttdg_graphs_t gr) {
  if (gr.G < 2 || gr.G > TTDG_MAX_GRAPHS) return 0;
  return (size_t)pl_total_tiles(gr) * sizeof(float);
}

extern "C" int ttdg_perm_loss_fwd_bwd(const float* Wds, const float* U, ttdg_graphs_t gr, float alpha, float eps,
                                      float* loss, float* dWds, int32_t* flag, float* pair_ws, ttdg_stream_t stream) {
  TTDG_REQUIRE(Wds && U && loss && dWds && flag && pair_ws, "perm_loss: null pointer");
  if (int e = ttdg_validate_graphs(gr)) return e;
  TTDG_REQUIRE(gr.G >= 2, "perm_loss: needs at least two graphs");
  const int M = gr.off[gr.G];
  hipStream_t st = (hipStream_t)stream;
  TTDG_HIP(hipMemsetAsync(dWds, 0, (size_t)M * M * sizeof(float), st));
  TTDG_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(perm_loss_pair_kernel, dim3(pl_total_tiles(gr)), dim3(256), 0, st, Wds, U, gr, alpha, eps, pair_ws, dWds, flag);
  hipLaunchKernelGGL(perm_loss_finish_kernel, dim3(1), dim3(256), 0, st, pair_ws, gr, loss);
  return ttdg_launch_status("perm_loss");
}
