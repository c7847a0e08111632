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
