// A8/A9 — pseudo-label permutation loss, forward and gradient in one pass.
// Reference: multi_graph_matching.py:535-564 (+ collect_intra_class_matching_wrapper :594-633),
// utils/losses.py:83-103 (BCEFocalLoss, alpha .25, gamma 2, eps 1e-6) and :419-455 (PermutationLoss).
//
// For every unordered graph pair a < b:  s = Wds[a-rows, b-cols],  t = U_a U_b^T  (pseudo ground truth),
//   p = clamp(s, eps, 1-eps);  l = -alpha (1-p)^2 t log p - (1-alpha) p^2 (1-t) log(1-p)
//   loss = (1/#pairs) * sum_pairs mean_elements(l)
// The Hungarian solve the reference also runs on every s (:626) feeds only the unused 'perm_mat_list'
// and is skipped.  One workgroup per pair, fixed-order tree reduction (deterministic); a one-thread
// epilogue adds the per-pair means.  dWds gets d loss / d Wds on the a<b blocks (clamp gradient is 1
// on [eps, 1-eps], 0 outside, as torch.clamp).
#include "common.h"

__global__ __launch_bounds__(256) void perm_loss_pair_kernel(const float* __restrict__ Wds, const float* __restrict__ U,
                                                             ttdg_graphs_t gr, float alpha, float eps,
                                                             float* __restrict__ pair_loss, float* __restrict__ dWds,
                                                             int32_t* __restrict__ flag) {
  int b = 1, idx = blockIdx.x;  // pairs (a<b) ordered (0,1),(0,2),(1,2),(0,3)...
  while (idx >= b) { idx -= b; ++b; }
  const int a = idx;
  const int M = gr.off[gr.G];
  const int na = gr.off[a + 1] - gr.off[a], nb = gr.off[b + 1] - gr.off[b];
  const int npairs = gr.G * (gr.G - 1) / 2;
  const float gscale = 1.f / ((float)npairs * (float)na * (float)nb);
  __shared__ float red[4];
  float acc = 0.f;
  bool bad = false;
  for (int e = threadIdx.x; e < na * nb; e += 256) {
    const int i = e / nb, j = e - i * nb;
    const float* ui = U + (size_t)(gr.off[a] + i) * TTDG_UNIV;
    const float* uj = U + (size_t)(gr.off[b] + j) * TTDG_UNIV;
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < TTDG_UNIV; u += 4) {
      const float4 x = *reinterpret_cast<const float4*>(ui + u);
      const float4 y = *reinterpret_cast<const float4*>(uj + u);
      t += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    const size_t o = (size_t)(gr.off[a] + i) * M + gr.off[b] + j;
    const float s = Wds[o];
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
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
  __syncthreads();
  if (threadIdx.x == 0) pair_loss[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / ((float)na * (float)nb);
}

__global__ void perm_loss_finish_kernel(const float* __restrict__ pair_loss, int npairs, float* __restrict__ loss) {
  float s = 0.f;
  for (int p = 0; p < npairs; ++p) s += pair_loss[p];
  *loss = s / (float)npairs;
}

extern "C" int ttdg_perm_loss_fwd_bwd(const float* Wds, const float* U, ttdg_graphs_t gr, float alpha, float eps,
                                      float* loss, float* dWds, int32_t* flag, float* pair_ws, ttdg_stream_t stream) {
  TTDG_REQUIRE(Wds && U && loss && dWds && flag && pair_ws, "perm_loss: null pointer");
  if (int e = ttdg_validate_graphs(gr)) return e;
  TTDG_REQUIRE(gr.G >= 2, "perm_loss: needs at least two graphs");
  const int M = gr.off[gr.G];
  const int npairs = gr.G * (gr.G - 1) / 2;
  hipStream_t st = (hipStream_t)stream;
  TTDG_HIP(hipMemsetAsync(dWds, 0, (size_t)M * M * sizeof(float), st));
  TTDG_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(perm_loss_pair_kernel, dim3(npairs), dim3(256), 0, st, Wds, U, gr, alpha, eps, pair_ws, dWds, flag);
  hipLaunchKernelGGL(perm_loss_finish_kernel, dim3(1), dim3(1), 0, st, pair_ws, npairs, loss);
  return ttdg_launch_status("perm_loss");
}
