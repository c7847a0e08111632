// A12 — DiceEvaluator reductions (reference evaluation/dice_metric.py:25-92, 110-240): for every (predicted mask,
// same-class ground-truth mask) pair of an eval batch, the 12 integer counts every score is a function of.
//
// The reference thresholds the scores, copies each kept mask to the host and runs numpy over the full H x W arrays
// three times (Dice :54-66, E-measure :110-143, S-measure :147-240).  For BOOLEAN maps all three are closed forms of
//      n(p AND g), n(p), n(g)   per quadrant of the image split at the ground truth's centroid
// (Dice and the E-measure need the image totals; the S-measure's object term needs the totals, its region term the four
// quadrants - evaluation/__init__.py:measures_from_counts restates the arithmetic in float64).  So the device work of
// the whole Dice pass of a batch is ONE launch that reads every mask pair once: HBM-bound, 2 bytes per pixel pair.
//
// Layout: masks are H x W bytes (0/1), rows contiguous; pred[pair] / gt[pair] are device addresses (the predictions of a
// batch live in one pasted tensor, the ground truth in per-image tensors).  counts[pair][q*3 + {0,1,2}] =
// {n(p&g), n(p), n(g)} of quadrant q = 2*(row >= cy) + (col >= cx).  Integer atomics: deterministic.
#include "common.h"

#define EV_THREADS 256
#define EV_ROWS 32          // image rows per workgroup

__global__ __launch_bounds__(EV_THREADS) void mask_pair_counts_kernel(const unsigned long long* __restrict__ pred,
                                                                      const unsigned long long* __restrict__ gt,
                                                                      const int* __restrict__ cy, const int* __restrict__ cx,
                                                                      int H, int W, int* __restrict__ counts) {
  const int pair = blockIdx.x;
  const unsigned char* p = reinterpret_cast<const unsigned char*>(pred[pair]);
  const unsigned char* g = reinterpret_cast<const unsigned char*>(gt[pair]);
  const int ycut = cy[pair], xcut = cx[pair];
  const int r0 = blockIdx.y * EV_ROWS, r1 = min(H, r0 + EV_ROWS);
  // a workgroup's rows lie on one side of ycut unless the cut falls inside: keep two accumulators sets (top, bottom)
  int acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0;
  const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g)) % 4 == 0);
  if (vec) {
    const int W4 = W >> 2;
    const int n4 = (r1 - r0) * W4;
    for (int e = threadIdx.x; e < n4; e += EV_THREADS) {
      const int r = r0 + e / W4, c4 = e % W4;
      const unsigned pw = reinterpret_cast<const unsigned*>(p + (size_t)r * W)[c4] & 0x01010101u;
      const unsigned gw = reinterpret_cast<const unsigned*>(g + (size_t)r * W)[c4] & 0x01010101u;
      // bytes left of the column cut: columns c4*4 + b < xcut
      const int nl = min(4, max(0, xcut - c4 * 4));
      const unsigned lm = nl >= 4 ? 0xffffffffu : ((1u << (8 * nl)) - 1u);
      const unsigned pg = pw & gw;
      const int qb = (r >= ycut) ? 6 : 0;
      acc[qb + 0] += __popc(pg & lm);  acc[qb + 1] += __popc(pw & lm);  acc[qb + 2] += __popc(gw & lm);
      acc[qb + 3] += __popc(pg & ~lm); acc[qb + 4] += __popc(pw & ~lm); acc[qb + 5] += __popc(gw & ~lm);
    }
  } else {
    const int n = (r1 - r0) * W;
    for (int e = threadIdx.x; e < n; e += EV_THREADS) {
      const int r = r0 + e / W, c = e % W;
      const int pv = p[(size_t)r * W + c] & 1, gv = g[(size_t)r * W + c] & 1;
      const int q = ((r >= ycut) ? 6 : 0) + ((c >= xcut) ? 3 : 0);
      acc[q + 0] += pv & gv; acc[q + 1] += pv; acc[q + 2] += gv;
    }
  }
  __shared__ int red[12];
  if (threadIdx.x < 12) red[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    int v = acc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&red[k], v);
  }
  __syncthreads();
  if (threadIdx.x < 12 && red[threadIdx.x]) atomicAdd(&counts[(size_t)pair * 12 + threadIdx.x], red[threadIdx.x]);
}

extern "C" int ttdg_mask_pair_counts(const unsigned long long* pred, const unsigned long long* gt, const int32_t* cy,
                                     const int32_t* cx, int npairs, int H, int W, int32_t* counts, ttdg_stream_t stream) {
  TTDG_REQUIRE(npairs >= 0 && H > 0 && W > 0, "mask_pair_counts: bad sizes");
  if (npairs == 0) return 0;
  TTDG_REQUIRE(pred && gt && cy && cx && counts, "mask_pair_counts: null pointer");
  TTDG_LIMIT(npairs <= 65535 * 32, "mask_pair_counts: too many pairs");
  hipStream_t st = (hipStream_t)stream;
  TTDG_HIP(hipMemsetAsync(counts, 0, (size_t)npairs * 12 * sizeof(int32_t), st));
  dim3 grid(npairs, (H + EV_ROWS - 1) / EV_ROWS);
  hipLaunchKernelGGL(mask_pair_counts_kernel, grid, dim3(EV_THREADS), 0, st, pred, gt, cy, cx, H, W, counts);
  return ttdg_launch_status("mask_pair_counts");
}

// ---- the closed forms, on the device too ------------------------------------------------------------------------------------
// Dice (dice_metric.py:54-66), E-measure (:110-143) and S-measure (:147-240) of one BOOLEAN pair from its twelve counts, in
// float64, one thread per pair - the arithmetic of evaluation/__init__.py:measures_from_counts term by term (which stays as the
// host-side statement the tests compare against).  As ~150 tiny float64 torch launches per batch this was 2 % of an adapted
// batch's launches for a few hundred flops.  best[owner[pair]][0..2] <- max(best, 100 x measure): the maximum over the same-class
// ground truths of a prediction (dice_metric.py:76-92); the measures are >= 0, so the maximum of doubles is the maximum of their
// bit patterns (one 64-bit integer atomic; order independent, hence deterministic).
__device__ __forceinline__ double ev_val(double f, double g, double mf, double mg) {
  const double af = f - mf, ag = g - mg;
  const double t = (2.0 * (ag * af) / (ag * ag + af * af + 1e-8)) + 1.0;
  return t * t / 4.0;
}
__device__ __forceinline__ double ev_s_object(double hit, double cnt) {
  const double x = hit / cnt;
  const double sig = sqrt((hit * ((1.0 - x) * (1.0 - x)) + (cnt - hit) * x * x) / cnt);
  return 2.0 * x / (x * x + 1.0 + sig + 1e-8);
}

__global__ __launch_bounds__(64) void mask_measures_kernel(const int* __restrict__ counts, const int* __restrict__ cy_,
                                                           const int* __restrict__ cx_, const int* __restrict__ owner, int n, int H,
                                                           int W, double alpha, double* __restrict__ best) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  double c[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int k = 0; k < 3; ++k) c[q][k] = (double)counts[(size_t)i * 12 + q * 3 + k];
  const double N = (double)H * (double)W;
  const double n11 = ((c[0][0] + c[1][0]) + c[2][0]) + c[3][0];
  const double npd = ((c[0][1] + c[1][1]) + c[2][1]) + c[3][1];
  const double ng = ((c[0][2] + c[1][2]) + c[2][2]) + c[3][2];
  const double dice = 2.0 * n11 / (npd + ng + 1e-6);
  // E-measure: fm = p, or all ones when the prediction is empty
  const double nf = npd == 0.0 ? N : npd, n11f = npd == 0.0 ? ng : n11;
  const double mf = nf / N, mg = ng / N;
  const double general = n11f * ev_val(1.0, 1.0, mf, mg) + (nf - n11f) * ev_val(1.0, 0.0, mf, mg) + (ng - n11f) * ev_val(0.0, 1.0, mf, mg) +
                         (N - nf - ng + n11f) * ev_val(0.0, 0.0, mf, mg);
  const double em = (ng == 0.0 ? N - nf : (ng == N ? nf : general)) / (N - 1.0 + 1e-8);
  // S-measure: object term from the totals, region term from the quadrants
  const double y = ng / N;
  const double obj = y * ev_s_object(n11, ng) + (1.0 - y) * ev_s_object(N - npd - ng + n11, N - ng);
  const double cy = fmin(fmax((double)cy_[i], 0.0), (double)H), cx = fmin(fmax((double)cx_[i], 0.0), (double)W);
  const double areas[4] = {cy * cx, cy * (W - cx), (H - cy) * cx, (H - cy) * (W - cx)};
  double reg = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double a = fmax(areas[q], 1.0);
    const double x = c[q][1] / a, yq = c[q][2] / a;
    const double sx = x * (1.0 - x), sy = yq * (1.0 - yq);
    const double sxy = areas[q] > 1.0 ? (c[q][0] - a * x * yq) / fmax(a - 1.0, 1.0) : nan("");
    const double al = 4.0 * x * yq * sxy, be = (x * x + yq * yq) * (sx + sy);
    const double ssim = al != 0.0 ? al / (be + 1e-8) : (be == 0.0 ? 1.0 : 0.0);
    reg += areas[q] > 0.0 ? areas[q] / N * ssim : 0.0;
  }
  const double sm = y == 0.0 ? 1.0 - npd / N : (y == 1.0 ? npd / N : alpha * obj + (1.0 - alpha) * reg);
  const double v[3] = {dice * 100.0, em * 100.0, sm * 100.0};
  unsigned long long* row = reinterpret_cast<unsigned long long*>(best + (size_t)owner[i] * 3);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // a one-pixel quadrant has no sample covariance: the reference's S-measure is NaN there, and torch.maximum propagates it - so
    // does this: the pattern of a positive quiet NaN is above every finite double's
    if (isnan(v[k])) atomicMax(row + k, 0x7FF8000000000000ull);
    else if (v[k] > 0.0) atomicMax(row + k, (unsigned long long)__double_as_longlong(v[k]));
  }
}

extern "C" int ttdg_mask_measures(const int32_t* counts, const int32_t* cy, const int32_t* cx, const int32_t* owner, int npairs,
                                  int H, int W, double alpha, double* best, ttdg_stream_t stream) {
  TTDG_REQUIRE(npairs >= 0 && H > 0 && W > 0, "mask_measures: bad sizes");
  if (npairs == 0) return 0;
  TTDG_REQUIRE(counts && cy && cx && owner && best, "mask_measures: null pointer");
  hipLaunchKernelGGL(mask_measures_kernel, dim3((npairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, counts, cy, cx, owner, npairs, H, W,
                     alpha, best);
  return ttdg_launch_status("mask_measures");
}
