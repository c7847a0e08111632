// Detection helpers the torch-native Mask R-CNN stand-in needs (torchvision is absent):
// ROIAlign (aligned, adaptive sampling — detectron2 ROIAlignV2 semantics [3P]) forward, and greedy NMS
// (IoU bit-mask matrix + one-wavefront sequential sweep).  Both are forward-only on the TTA path: proposals
// and detections carry no gradient (rcnn.py:333-345 feeds them to the node sampler as constants).
#include "common.h"

// rois: (R, 5) = (batch index, x1, y1, x2, y2) in image coordinates; feat: (B, C, H, W); out: (R, C, P, P)
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(const float* __restrict__ feat, int C, int H, int W,
                                                            const float* __restrict__ rois, int R, float scale, int P,
                                                            float* __restrict__ out) {
  const long total = (long)R * C * P * P;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int pw = idx % P, ph = (idx / P) % P, c = (idx / ((long)P * P)) % C;
    const int r = idx / ((long)P * P * C);
    const float* roi = rois + (size_t)r * 5;
    const int b = (int)roi[0];
    const float x1 = roi[1] * scale - 0.5f, y1 = roi[2] * scale - 0.5f;
    const float rw = roi[3] * scale - 0.5f - x1, rh = roi[4] * scale - 0.5f - y1;
    const float bw = rw / P, bh = rh / P;
    const int gh = max(1, (int)ceilf(rh / P)), gw = max(1, (int)ceilf(rw / P));
    const float* f = feat + ((size_t)b * C + c) * H * W;
    float acc = 0.f;
    for (int iy = 0; iy < gh; ++iy) {
      float y = y1 + ph * bh + (iy + 0.5f) * bh / gh;
      for (int ix = 0; ix < gw; ++ix) {
        float x = x1 + pw * bw + (ix + 0.5f) * bw / gw;
        if (y < -1.f || y > H || x < -1.f || x > W) continue;
        float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
        int y0 = (int)yy, x0 = (int)xx, y1i, x1i;
        if (y0 >= H - 1) { y0 = y1i = H - 1; yy = (float)y0; } else y1i = y0 + 1;
        if (x0 >= W - 1) { x0 = x1i = W - 1; xx = (float)x0; } else x1i = x0 + 1;
        const float ly = yy - y0, lx = xx - x0, hy = 1.f - ly, hx = 1.f - lx;
        acc += hy * hx * f[y0 * W + x0] + hy * lx * f[y0 * W + x1i] + ly * hx * f[y1i * W + x0] + ly * lx * f[y1i * W + x1i];
      }
    }
    out[idx] = acc / (float)(gh * gw);
  }
}

extern "C" int ttdg_roi_align_fwd(const float* feat, int B, int C, int H, int W, const float* rois, int R, float scale,
                                  int P, float* out, ttdg_stream_t stream) {
  TTDG_REQUIRE(feat && R >= 0 && (R == 0 || (rois && out)) && C > 0 && P > 0, "roi_align: bad arguments");      // (R == 0: empty tensors carry null pointers)
  if (R == 0) return 0;
  const long total = (long)R * C * P * P;
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(roi_align_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, feat, C, H, W, rois, R, scale, P, out);
  return ttdg_launch_status("roi_align_fwd");
}

// ---- NMS: boxes (N,4) sorted by descending score; group ids make it "batched" (boxes of different groups never
// suppress each other).  mask[i][w] bit j: box 64w+j (j>i) overlaps box i above the threshold.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ group,
                                                      int N, float thr, unsigned long long* __restrict__ mask, int words) {
  const int i = blockIdx.x, w = blockIdx.y, lane = threadIdx.x;
  const int j = w * 64 + lane;
  bool hit = false;
  if (j < N && j > i && group[i] == group[j]) {
    const float4 a = reinterpret_cast<const float4*>(boxes)[i], b = reinterpret_cast<const float4*>(boxes)[j];
    const float iw = fminf(a.z, b.z) - fmaxf(a.x, b.x), ih = fminf(a.w, b.w) - fmaxf(a.y, b.y);
    const float inter = fmaxf(iw, 0.f) * fmaxf(ih, 0.f);
    const float ua = (a.z - a.x) * (a.w - a.y) + (b.z - b.x) * (b.w - b.y) - inter;
    hit = inter > thr * ua;
  }
  const unsigned long long m = __ballot(hit);
  if (lane == 0) mask[(size_t)i * words + w] = m;
}

// One wavefront, blocked sweep.  The removed-set lives in registers (lane l owns words l, l+64, ...).  Boxes are
// visited in blocks of 64: the 64x64 diagonal sub-matrix is resolved sequentially from registers (readlane, no
// memory), then only the rows of the boxes that were KEPT are OR-ed into the later words, four rows of loads in
// flight at a time.  (A row-at-a-time sweep pays one dependent L2 round trip per kept box: 1.7 ms at N = 8400.)
__global__ __launch_bounds__(64) void nms_sweep_kernel(const unsigned long long* __restrict__ mask, int N, int words,
                                                       int32_t* __restrict__ keep, int32_t* __restrict__ nkeep) {
  const int lane = threadIdx.x;
  unsigned long long rem[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // up to 8*64*64 = 32768 boxes
  int cnt = 0;
  for (int b = 0; b < words; ++b) {
    const int owner = b & 63, slot = b >> 6;
    unsigned long long word = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) if (s == slot) word = rem[s];
    const unsigned long long remw = ((unsigned long long)__builtin_amdgcn_readlane((unsigned)(word >> 32), owner) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane((unsigned)word, owner);
    const int box = b * 64 + lane;
    const unsigned long long diag = (box < N) ? mask[(size_t)box * words + b] : 0ull;
    const int nbox = min(64, N - b * 64);
    unsigned long long alive = ~remw;
    if (nbox < 64) alive &= (1ull << nbox) - 1ull;
    unsigned long long kept = 0;
    while (alive) {
      const int i = __builtin_ctzll(alive);
      kept |= 1ull << i;
      const unsigned long long d = ((unsigned long long)__builtin_amdgcn_readlane((unsigned)(diag >> 32), i) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((unsigned)diag, i);
      alive &= ~d;
      alive &= ~(1ull << i);
    }
    if ((kept >> lane) & 1ull) keep[cnt + __builtin_popcountll(kept & ((1ull << lane) - 1ull))] = box;
    cnt += __builtin_popcountll(kept);
    // propagate the kept rows to the later words
    unsigned long long k = kept;
    while (k) {
      int r[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { r[q] = k ? __builtin_ctzll(k) : -1; if (k) k &= k - 1; }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int w = lane + 64 * s;
        if (64 * s >= words) break;
        if (w > b && w < words) {
          unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;
          m0 = mask[(size_t)(b * 64 + r[0]) * words + w];
          if (r[1] >= 0) m1 = mask[(size_t)(b * 64 + r[1]) * words + w];
          if (r[2] >= 0) m2 = mask[(size_t)(b * 64 + r[2]) * words + w];
          if (r[3] >= 0) m3 = mask[(size_t)(b * 64 + r[3]) * words + w];
          rem[s] |= (m0 | m1) | (m2 | m3);
        }
      }
    }
  }
  if (lane == 0) *nkeep = cnt;
}

extern "C" int ttdg_nms(const float* boxes, const int32_t* group, int N, float thr, void* mask_ws, int32_t* keep,
                        int32_t* nkeep, ttdg_stream_t stream) {
  TTDG_REQUIRE(boxes && group && keep && nkeep && N >= 0, "nms: bad arguments");
  TTDG_LIMIT(N <= 32768, "nms: more than 32768 boxes");
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) { TTDG_HIP(hipMemsetAsync(nkeep, 0, sizeof(int32_t), st)); return 0; }
  TTDG_REQUIRE(mask_ws, "nms: null workspace");
  const int words = (N + 63) / 64;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(N, words), dim3(64), 0, st, boxes, group, N, thr, (unsigned long long*)mask_ws, words);
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), 0, st, (const unsigned long long*)mask_ws, N, words, keep, nkeep);
  return ttdg_launch_status("nms");
}

// ---- grouped NMS: boxes sorted by (group, descending score); seg[g]..seg[g+1] delimits group g (device array).
// Groups never interact, so every group gets its own wavefront (own CU) for both phases - on the RPN path the five
// FPN levels of an image are swept concurrently instead of one after another.  Output: keep flags per box.
// Mask phase: one wavefront per (64-row tile, 64-column word) of a group's upper triangle; lane = column box j (loaded once),
// the 64 row boxes come in as wavefront-uniform loads; lane ii keeps row ii's word and the tile is written by one store per
// lane.  (Round 1 launched one 64-thread workgroup PER ROW AND WORD - 1.3 M workgroups of ~20 instructions for the RPN's
// 20 groups of 2000 candidates: 99 us of workgroup dispatch.)  Same IoU arithmetic, same words.  Words left of the diagonal
// tile are never read by the sweep and are not written.
__global__ __launch_bounds__(64) void nms_group_mask_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ seg,
                                                            int words, float thr, unsigned long long* __restrict__ mask) {
  const int g = blockIdx.z;
  const int s0 = seg[g], n = seg[g + 1] - s0;
  const int it = blockIdx.x, w = blockIdx.y, lane = threadIdx.x;
  if (it * 64 >= n || w * 64 >= n || w < it) return;
  const int j = w * 64 + lane;
  const float4* bx = reinterpret_cast<const float4*>(boxes) + s0;
  const float4 b = bx[min(j, n - 1)];
  const float barea = (b.z - b.x) * (b.w - b.y);
  const int rows = min(64, n - it * 64);
  unsigned long long mine = 0ull;
  for (int ii = 0; ii < rows; ++ii) {
    const int i = it * 64 + ii;
    const float4 a = bx[i];
    const float iw = fminf(a.z, b.z) - fmaxf(a.x, b.x), ih = fminf(a.w, b.w) - fmaxf(a.y, b.y);
    const float inter = fmaxf(iw, 0.f) * fmaxf(ih, 0.f);
    const float ua = (a.z - a.x) * (a.w - a.y) + barea - inter;
    const bool hit = j < n && j > i && inter > thr * ua;
    const unsigned long long m = __ballot(hit);
    if (lane == ii) mine = m;
  }
  if (lane < rows) mask[(size_t)(s0 + it * 64 + lane) * words + w] = mine;
}

__global__ __launch_bounds__(64) void nms_group_sweep_kernel(const unsigned long long* __restrict__ mask,
                                                             const int32_t* __restrict__ seg, int words,
                                                             unsigned char* __restrict__ flags) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int s0 = seg[g], N = seg[g + 1] - s0;
  const int nw = (N + 63) / 64;
  const unsigned long long* mk = mask + (size_t)s0 * words;
  unsigned long long rem[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = 0; b < nw; ++b) {
    const int owner = b & 63, slot = b >> 6;
    unsigned long long word = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) if (s == slot) word = rem[s];
    const unsigned long long remw = ((unsigned long long)__builtin_amdgcn_readlane((unsigned)(word >> 32), owner) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane((unsigned)word, owner);
    const int box = b * 64 + lane;
    const unsigned long long diag = (box < N) ? mk[(size_t)box * words + b] : 0ull;
    const int nbox = min(64, N - b * 64);
    unsigned long long alive = ~remw;
    if (nbox < 64) alive &= (1ull << nbox) - 1ull;
    unsigned long long kept = 0;
    while (alive) {
      const int i = __builtin_ctzll(alive);
      kept |= 1ull << i;
      const unsigned long long d = ((unsigned long long)__builtin_amdgcn_readlane((unsigned)(diag >> 32), i) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((unsigned)diag, i);
      alive &= ~d;
      alive &= ~(1ull << i);
    }
    if (box < N) flags[s0 + box] = (unsigned char)((kept >> lane) & 1ull);
    unsigned long long k = kept;
    while (k) {
      int r[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { r[q] = k ? __builtin_ctzll(k) : -1; if (k) k &= k - 1; }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int w = lane + 64 * s;
        if (64 * s >= nw) break;
        if (w > b && w < nw) {
          unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;
          m0 = mk[(size_t)(b * 64 + r[0]) * words + w];
          if (r[1] >= 0) m1 = mk[(size_t)(b * 64 + r[1]) * words + w];
          if (r[2] >= 0) m2 = mk[(size_t)(b * 64 + r[2]) * words + w];
          if (r[3] >= 0) m3 = mk[(size_t)(b * 64 + r[3]) * words + w];
          rem[s] |= (m0 | m1) | (m2 | m3);
        }
      }
    }
  }
}

extern "C" int ttdg_nms_grouped(const float* boxes, const int32_t* seg, int ngroups, int N, int max_group, float thr,
                                void* mask_ws, unsigned char* flags, ttdg_stream_t stream) {
  TTDG_REQUIRE(boxes && seg && flags && N >= 0 && ngroups >= 1 && max_group >= 0, "nms_grouped: bad arguments");
  if (N == 0) return 0;
  TTDG_REQUIRE(mask_ws, "nms_grouped: null workspace");
  TTDG_LIMIT(max_group <= 32768, "nms_grouped: more than 32768 boxes in one group");
  const int words = (max_group + 63) / 64;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(nms_group_mask_kernel, dim3(words, words, ngroups), dim3(64), 0, st, boxes, seg, words, thr,
                     (unsigned long long*)mask_ws);
  hipLaunchKernelGGL(nms_group_sweep_kernel, dim3(ngroups), dim3(64), 0, st, (const unsigned long long*)mask_ws, seg, words, flags);
  return ttdg_launch_status("nms_grouped");
}

// ---------------------------------------------------------------------------------------------------
// Fused box pipelines of the stand-in detector.  The torch formulation spends ~40 tiny launches per FPN level and
// per image on slicing / exp / clamp / isfinite / boolean indexing; the eval pass is bound by exactly those launches.

#define DET_SCALE_CLAMP 4.135166556742356f   /* log(1000 / 16), detectron2 Box2BoxTransform [3P] */

// returns "every coordinate finite before the clip"; `nonan` = "no coordinate is NaN" (what survives torch.clamp)
__device__ __forceinline__ bool det_decode(float ax1, float ay1, float ax2, float ay2, float dx, float dy, float dw, float dh,
                                           float wx, float wy, float ww, float wh, float img_h, float img_w, float4& out,
                                           bool* nonan = nullptr) {
  const float w = ax2 - ax1, h = ay2 - ay1;
  const float cx = ax1 + 0.5f * w, cy = ay1 + 0.5f * h;
  dx /= wx; dy /= wy;
  dw = fminf(dw / ww, DET_SCALE_CLAMP); dh = fminf(dh / wh, DET_SCALE_CLAMP);
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  const float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
  const bool fin = isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2);
  if (nonan) *nonan = !(isnan(x1) || isnan(y1) || isnan(x2) || isnan(y2));
  out = make_float4(fminf(fmaxf(x1, 0.f), img_w), fminf(fmaxf(y1, 0.f), img_h), fminf(fmaxf(x2, 0.f), img_w), fminf(fmaxf(y2, 0.f), img_h));
  return fin;
}

// RPN, one FPN level: the top-k anchors of every image (idx into the (h, w, a) raster) are decoded straight from the
// NCHW head output, clipped to the image, and tested (finite, non-empty); invalid candidates get score -inf.
//   deltas (B, A*4, H, W), anchors (H*W*A, 4), idx / score (B, k) -> boxes (B, K, 4) and scores (B, K) at column `col0`
__global__ __launch_bounds__(256) void rpn_decode_kernel(const float* __restrict__ deltas, const float* __restrict__ anchors,
                                                         const int64_t* __restrict__ idx, const float* __restrict__ score,
                                                         const float* __restrict__ sizes, int B, int k, int A, int H, int W,
                                                         int K, int col0, float* __restrict__ boxes, float* __restrict__ scores) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * k) return;
  const int b = e / k, j = e - b * k;
  const int64_t id = idx[e];
  const int a = (int)(id % A), hw = (int)(id / A);
  const size_t plane = (size_t)H * W;
  const float* d = deltas + ((size_t)b * A * 4 + (size_t)a * 4) * plane + hw;
  const float4 an = reinterpret_cast<const float4*>(anchors)[id];
  float4 o;
  const bool fin = det_decode(an.x, an.y, an.z, an.w, d[0], d[plane], d[2 * plane], d[3 * plane], 1.f, 1.f, 1.f, 1.f,
                              sizes[2 * b], sizes[2 * b + 1], o);
  const float s = score[e];
  const bool ok = fin && isfinite(s) && (o.z - o.x > 0.f) && (o.w - o.y > 0.f);
  const size_t oi = (size_t)b * K + col0 + j;
  // a rejected candidate keeps its slot (static shapes) but loses its area: the sort-free NMS sweeps it with its group, and a box
  // without area overlaps nothing - it cannot suppress a live candidate (the reference removes it before the NMS)
  reinterpret_cast<float4*>(boxes)[oi] = ok ? o : make_float4(0.f, 0.f, 0.f, 0.f);
  scores[oi] = ok ? s : -INFINITY;
}

extern "C" int ttdg_rpn_decode(const float* deltas, const float* anchors, const int64_t* idx, const float* score,
                               const float* sizes, int B, int k, int A, int H, int W, int K, int col0, float* boxes,
                               float* scores, ttdg_stream_t stream) {
  TTDG_REQUIRE(deltas && anchors && idx && score && sizes && boxes && scores, "rpn_decode: null pointer");
  TTDG_REQUIRE(B >= 0 && k >= 0 && A > 0 && H > 0 && W > 0 && col0 >= 0 && col0 + k <= K, "rpn_decode: bad sizes");
  if (B * k == 0) return 0;
  hipLaunchKernelGGL(rpn_decode_kernel, dim3((B * k + 255) / 256), dim3(256), 0, (hipStream_t)stream, deltas, anchors, idx, score,
                     sizes, B, k, A, H, W, K, col0, boxes, scores);
  return ttdg_launch_status("rpn_decode");
}

// ---- RPN selection, all FPN levels and all images in ONE launch -------------------------------------------------------------
// find_top_rpn_proposals [3P] up to the NMS: per (image, level) the k best objectness logits in descending order, their anchors
// decoded, clipped and tested.  With stock PyTorch that is, per level, a permute copy + topk (a radix select of ~8 kernels and a
// sort of the winners) + rpn_decode: ~55 launches per forward pass, 110 per adapted batch.  Here one workgroup of 1024 threads
// owns one (image, level) row of the NCHW head output:
//   1. exact k-th largest key by a 3-pass radix select (11 + 11 + 10 bits, histogram in LDS) over order-preserving integer keys;
//   2. gather: every key above the threshold, plus as many keys EQUAL to it as the rank needs (lowest (h, w, a) raster index first when there
//      are more ties than needed - deterministic; torch.topk leaves the order of ties unspecified);
//   3. bitonic sort of the <= 2048 winners in LDS on (key, ~raster index): descending score, ascending (h, w, a) index on ties;
//   4. decode + clip + validity straight into boxes (B, K, 4) / scores (B, K) at the level's column block.
// The row is read four times by one CU (p2: 480 KB per pass, L2-resident after the first), all rows concurrently.  Integer / comparison work only until step 4: the selection is exact.
#define RS_THREADS 1024
#define RS_BINS 2048
#define RS_MAXK 2048
// The logits of a level share a few exponents: the first histogram pass sends most of a row to a handful of bins, and LDS atomics
// of one wavefront to one address are served one after the other (measured: 198 us per launch with one histogram = 64 cycles per
// wavefront instruction).  RS_REPL copies, chosen by lane and shifted by one bank each, cut that to 64 / RS_REPL.
#define RS_REPL 8
#define RS_HSTRIDE (RS_BINS + 1)
#define RS_LDS_BYTES ((size_t)RS_REPL * RS_HSTRIDE * 4 + (size_t)RS_MAXK * 8)

struct RsLevels {
  ttdg_rpn_level_t l[TTDG_RPN_LEVELS_MAX];
  int n;
};

__device__ __forceinline__ unsigned rs_key(float x) {
  unsigned b = __float_as_uint(x);
  b = b == 0x80000000u ? 0u : b;                                  // -0.0 == +0.0, as in a floating-point comparison
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);             // larger float <=> larger key (NaN with a clear sign bit sorts first, as in torch)
}
__device__ __forceinline__ float rs_unkey(unsigned k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// sum of v over the threads ABOVE this one (exclusive suffix); wsum: RS_THREADS / 64 words of LDS
__device__ __forceinline__ unsigned rs_suffix_excl(unsigned v, unsigned* wsum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned s = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_down(s, off);
    if (lane + off < 64) s += o;
  }
  __syncthreads();
  if (lane == 0) wsum[wave] = s;
  __syncthreads();
  unsigned above = 0;
  for (int w = wave + 1; w < RS_THREADS / 64; ++w) above += wsum[w];
  return above + s - v;
}
// sum of v over the threads BELOW this one (exclusive prefix)
__device__ __forceinline__ unsigned rs_prefix_excl(unsigned v, unsigned* wsum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned s = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(s, off);
    if (lane >= off) s += o;
  }
  __syncthreads();
  if (lane == 63) wsum[wave] = s;
  __syncthreads();
  unsigned below = 0;
  for (int w = 0; w < wave; ++w) below += wsum[w];
  return below + s - v;
}

// One pass over a row: f(index, value, in_range) for every element, EIGHT independent loads in flight per thread (a plain loop keeps
// one: 118 dependent L2 round trips per pass for a p2 row - measured 37 us per pass, 0.3 us per iteration).  Every thread makes the
// same number of calls (f may use wavefront-wide votes); the tail calls carry in_range = false beyond the end.
template <class F>
__device__ __forceinline__ void rs_for_each(const float* __restrict__ x, int n, F f) {
  const int tid = threadIdx.x;
  int i0 = 0;
  for (; i0 + 8 * RS_THREADS <= n; i0 += 8 * RS_THREADS) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[i0 + u * RS_THREADS + tid];
#pragma unroll
    for (int u = 0; u < 8; ++u) f(i0 + u * RS_THREADS + tid, v[u], true);
  }
  for (; i0 < n; i0 += RS_THREADS) {
    const int i = i0 + tid;
    const bool in = i < n;
    f(i, in ? x[i] : 0.f, in);
  }
}

// kNhwc: the head outputs are channels-last - logits (B, H, W, A), deltas (B, H, W, 4A) in storage: a row of logits IS the (h, w, a)
// raster (memory index == raster index), a candidate's four deltas are adjacent.
template <bool kNhwc>
__global__ __launch_bounds__(RS_THREADS) void rpn_select_kernel(RsLevels lv, int B, int A, const float* __restrict__ sizes, int K,
                                                                float* __restrict__ boxes, float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
  unsigned long long* sel = reinterpret_cast<unsigned long long*>(rs_smem);                       // RS_MAXK winners
  unsigned* hist = reinterpret_cast<unsigned*>(rs_smem + (size_t)RS_MAXK * 8);                    // RS_REPL x RS_HSTRIDE
  __shared__ unsigned wsum[RS_THREADS / 64];
  __shared__ unsigned sh_bin, sh_need, sh_eq, sh_cnt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int b = blockIdx.x / lv.n, li = blockIdx.x - b * lv.n;
  const ttdg_rpn_level_t L = lv.l[li];
  const int HW = L.H * L.W, n = A * HW, k = L.k;
  const float* x = L.logits + (size_t)b * n;
  if (k <= 0) return;

  // 1. radix select of the k-th largest key
  unsigned prefix = 0, mask = 0, need = (unsigned)k;
#pragma unroll 1
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
    const unsigned bm = pass == 2 ? 1023u : 2047u;
    for (int i = tid; i < RS_REPL * RS_HSTRIDE; i += RS_THREADS) hist[i] = 0;
    __syncthreads();
    unsigned* myhist = hist + (lane & (RS_REPL - 1)) * RS_HSTRIDE;
    rs_for_each(x, n, [&](int, float v, bool in) {
      const unsigned key = rs_key(v);
      if (in && (key & mask) == prefix) atomicAdd(&myhist[(key >> shift) & bm], 1u);
    });
    __syncthreads();
    unsigned h0 = 0, h1 = 0;
#pragma unroll
    for (int r = 0; r < RS_REPL; ++r) { h0 += hist[r * RS_HSTRIDE + 2 * tid]; h1 += hist[r * RS_HSTRIDE + 2 * tid + 1]; }
    const unsigned above = rs_suffix_excl(h0 + h1, wsum);        // keys in bins > 2 tid + 1
    if (above < need && need <= above + h1) { sh_bin = 2 * tid + 1; sh_need = need - above; sh_eq = h1; }
    else if (above + h1 < need && need <= above + h1 + h0) { sh_bin = 2 * tid; sh_need = need - above - h1; sh_eq = h0; }
    __syncthreads();
    prefix |= sh_bin << shift;
    mask |= bm << shift;
    need = sh_need;
    __syncthreads();
  }
  const unsigned theta = prefix;            // the k-th largest key; `need` of the sh_eq keys equal to it belong to the top k
  const unsigned eq_total = sh_eq;
  const unsigned nabove = (unsigned)k - need;

  // 2. gather
  for (int i = tid; i < RS_MAXK; i += RS_THREADS) sel[i] = 0ull;
  if (tid == 0) sh_cnt = 0;
  __syncthreads();
  const bool ordered_ties = eq_total > need;
  rs_for_each(x, n, [&](int i, float v, bool in) {
    const unsigned key = rs_key(v);
    const bool take = in && (key > theta || (!ordered_ties && key == theta));
    const unsigned long long m = __ballot(take);
    if (m) {                                                        // one LDS atomic per wavefront that has winners
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(&sh_cnt, (unsigned)__popcll(m));
      base = __shfl(base, 0);
      if (take) {
        unsigned id = (unsigned)i;
        if (!kNhwc) { const int a = i / HW, hw = i - a * HW; id = (unsigned)(hw * A + a); }
        sel[base + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - id);
      }
    }
  });
  if (ordered_ties) {
    // more keys equal to the threshold than the rank needs: the `need` lowest (h, w, a) raster indices (a contiguous chunk of the
    // raster per thread + a block scan of the tie counts).  Strided reads, but only when the k-th rank falls inside a tie.
    const int C = (n + RS_THREADS - 1) / RS_THREADS;
    const int lo = tid * C, hi = min(n, lo + C);
    unsigned c = 0;
    for (int id = lo; id < hi; ++id) c += rs_key(x[kNhwc ? id : (id % A) * HW + id / A]) == theta;
    unsigned r = rs_prefix_excl(c, wsum);
    for (int id = lo; id < hi && r < need; ++id)
      if (rs_key(x[kNhwc ? id : (id % A) * HW + id / A]) == theta) {
        sel[nabove + r] = ((unsigned long long)theta << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)id);
        ++r;
      }
  }
  __syncthreads();

  // 3. bitonic sort, descending, of the smallest power of two >= k entries (the padding is 0: below every real entry)
  int SN = 64;
  while (SN < k) SN <<= 1;
  for (int size = 2; size <= SN; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (SN >> 1); t += RS_THREADS) {
        const int pos = 2 * t - (t & (stride - 1));
        const unsigned long long u = sel[pos], v = sel[pos + stride];
        const bool desc = (pos & size) == 0;
        if ((u < v) == desc) { sel[pos] = v; sel[pos + stride] = u; }
      }
      __syncthreads();
    }

  // 4. decode, clip, validity
  const float ih = sizes[2 * b], iw = sizes[2 * b + 1];
  const size_t plane = (size_t)HW;
  for (int j = tid; j < k; j += RS_THREADS) {
    const unsigned long long e = sel[j];
    const float sc = rs_unkey((unsigned)(e >> 32));
    const unsigned id = 0xFFFFFFFFu - (unsigned)(e & 0xFFFFFFFFull);
    const int a = (int)(id % (unsigned)A), hw = (int)(id / (unsigned)A);
    float d0, d1, d2, d3;
    if (kNhwc) {
      const float4 dv = reinterpret_cast<const float4*>(L.deltas + ((size_t)b * plane + hw) * (size_t)(4 * A))[a];
      d0 = dv.x; d1 = dv.y; d2 = dv.z; d3 = dv.w;
    } else {
      const float* d = L.deltas + ((size_t)b * A * 4 + (size_t)a * 4) * plane + hw;
      d0 = d[0]; d1 = d[plane]; d2 = d[2 * plane]; d3 = d[3 * plane];
    }
    const float4 an = reinterpret_cast<const float4*>(L.anchors)[id];
    float4 o;
    const bool fin = det_decode(an.x, an.y, an.z, an.w, d0, d1, d2, d3, 1.f, 1.f, 1.f, 1.f, ih, iw, o);
    const bool ok = fin && isfinite(sc) && (o.z - o.x > 0.f) && (o.w - o.y > 0.f);
    const size_t oi = (size_t)b * K + L.col0 + j;
    reinterpret_cast<float4*>(boxes)[oi] = ok ? o : make_float4(0.f, 0.f, 0.f, 0.f);
    scores[oi] = ok ? sc : -INFINITY;
  }
}

static int rpn_select_launch(const ttdg_rpn_level_t* levels, int nlevels, int B, int A, const float* sizes, int K, float* boxes,
                             float* scores, ttdg_stream_t stream, bool nhwc) {
  TTDG_REQUIRE(levels && sizes && boxes && scores, "rpn_select: null pointer");
  TTDG_REQUIRE(nlevels >= 1 && nlevels <= TTDG_RPN_LEVELS_MAX && B >= 0 && A > 0 && K >= 0, "rpn_select: bad sizes");
  RsLevels lv;
  lv.n = nlevels;
  for (int i = 0; i < nlevels; ++i) {
    const ttdg_rpn_level_t& l = levels[i];
    TTDG_REQUIRE(l.logits && l.deltas && l.anchors && l.H > 0 && l.W > 0, "rpn_select: bad level");
    TTDG_REQUIRE(l.k >= 0 && l.k <= RS_MAXK && (int64_t)l.k <= (int64_t)A * l.H * l.W && l.col0 >= 0 && l.col0 + l.k <= K,
                 "rpn_select: k must be <= min(2048, anchors of the level) and fit its column block");
    TTDG_REQUIRE((int64_t)A * l.H * l.W < (1ll << 31), "rpn_select: level too large");
    lv.l[i] = l;
  }
  if (B == 0) return 0;
  if (nhwc) {
    TTDG_ALLOW_LDS(rpn_select_kernel<true>, RS_LDS_BYTES);
    hipLaunchKernelGGL(rpn_select_kernel<true>, dim3(B * nlevels), dim3(RS_THREADS), RS_LDS_BYTES, (hipStream_t)stream, lv, B, A, sizes, K, boxes, scores);
  } else {
    TTDG_ALLOW_LDS(rpn_select_kernel<false>, RS_LDS_BYTES);
    hipLaunchKernelGGL(rpn_select_kernel<false>, dim3(B * nlevels), dim3(RS_THREADS), RS_LDS_BYTES, (hipStream_t)stream, lv, B, A, sizes, K, boxes, scores);
  }
  return ttdg_launch_status("rpn_select");
}

extern "C" int ttdg_rpn_select(const ttdg_rpn_level_t* levels, int nlevels, int B, int A, const float* sizes, int K, float* boxes,
                               float* scores, ttdg_stream_t stream) {
  return rpn_select_launch(levels, nlevels, B, A, sizes, K, boxes, scores, stream, false);
}
extern "C" int ttdg_rpn_select_nhwc(const ttdg_rpn_level_t* levels, int nlevels, int B, int A, const float* sizes, int K, float* boxes,
                                    float* scores, ttdg_stream_t stream) {
  for (int i = 0; levels && i < nlevels && i < TTDG_RPN_LEVELS_MAX; ++i)
    TTDG_REQUIRE(((uintptr_t)levels[i].deltas & 15) == 0, "rpn_select_nhwc: deltas must be 16-byte aligned");
  return rpn_select_launch(levels, nlevels, B, A, sizes, K, boxes, scores, stream, true);
}

// Box head inference (detectron2 fast_rcnn_inference [3P] up to the NMS): per proposal softmax over C+1 logits, per
// class box decode with the head's weights, clip, validity (finite row, as the reference drops the whole proposal),
// score threshold.  Candidates that fail get score -inf.  rois (N, 5) = (image index, x1, y1, x2, y2).
__global__ __launch_bounds__(256) void box_inference_kernel(const float* __restrict__ logits, const float* __restrict__ deltas,
                                                            const float* __restrict__ rois, const float* __restrict__ sizes,
                                                            int N, int C, float wx, float wy, float ww, float wh, float thr,
                                                            float* __restrict__ boxes, float* __restrict__ scores) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float* lg = logits + (size_t)n * (C + 1);
  float m = lg[0];
  for (int c = 1; c <= C; ++c) m = fmaxf(m, lg[c]);
  float den = 0.f;
  for (int c = 0; c <= C; ++c) den += expf(lg[c] - m);
  const float* r = rois + (size_t)n * 5;
  const int b = (int)r[0];
  const float ih = sizes[2 * b], iw = sizes[2 * b + 1];
  bool allfin = true;
  for (int c = 0; c < C; ++c) {
    const float* d = deltas + (size_t)n * 4 * C + 4 * c;
    float4 o;
    bool nonan;    // the reference tests isfinite AFTER the clamp: +-inf is clipped to the border and survives, NaN does not
    det_decode(r[1], r[2], r[3], r[4], d[0], d[1], d[2], d[3], wx, wy, ww, wh, ih, iw, o, &nonan);
    allfin &= nonan;
    reinterpret_cast<float4*>(boxes)[(size_t)n * C + c] = o;
    const float p = expf(lg[c] - m) / den;
    allfin &= isfinite(p);
    scores[(size_t)n * C + c] = p;
  }
  for (int c = 0; c < C; ++c) {
    const float p = scores[(size_t)n * C + c];
    if (!allfin || !(p > thr)) scores[(size_t)n * C + c] = -INFINITY;
  }
}

extern "C" int ttdg_box_inference(const float* logits, const float* deltas, const float* rois, const float* sizes, int N,
                                  int C, float wx, float wy, float ww, float wh, float score_thresh, float* boxes,
                                  float* scores, ttdg_stream_t stream) {
  TTDG_REQUIRE(N >= 0 && (N == 0 || (logits && deltas && rois && sizes && boxes && scores)) && C >= 1, "box_inference: bad arguments");
  if (N == 0) return 0;
  hipLaunchKernelGGL(box_inference_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, deltas, rois, sizes, N,
                     C, wx, wy, ww, wh, score_thresh, boxes, scores);
  return ttdg_launch_status("box_inference");
}

// Mask paste (detectron2 paste_masks_in_image [3P], grid-sample form): soft masks (R, S, S) are resampled bilinearly
// (align_corners = False, zeros outside) onto the H x W image grid inside their boxes and thresholded.
__global__ __launch_bounds__(256) void paste_masks_kernel(const float* __restrict__ masks, const float* __restrict__ boxes, int R, int S,
                                                          int H, int W, float thr, unsigned char* __restrict__ out) {
  const int r = blockIdx.y;
  const float4 b = reinterpret_cast<const float4*>(boxes)[r];
  const float* m = masks + (size_t)r * S * S;
  unsigned char* o = out + (size_t)r * H * W;
  const float sx = (float)S / (b.z - b.x), sy = (float)S / (b.w - b.y);
  for (int e = blockIdx.x * 256 + threadIdx.x; e < H * W; e += gridDim.x * 256) {
    const int y = e / W, x = e - y * W;
    // grid_sample coordinates: g = (p + 0.5 - b0) / (b1 - b0) * 2 - 1;  pixel = ((g + 1) * S - 1) / 2
    const float gx = ((float)x + 0.5f - b.x) / (b.z - b.x) * 2.f - 1.f, gy = ((float)y + 0.5f - b.y) / (b.w - b.y) * 2.f - 1.f;
    const float fx = ((gx + 1.f) * (float)S - 1.f) * 0.5f, fy = ((gy + 1.f) * (float)S - 1.f) * 0.5f;
    (void)sx; (void)sy;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float lx = fx - x0f, ly = fy - y0f;
    float v = 0.f;
    if (fx > -1.f && fx < (float)S && fy > -1.f && fy < (float)S) {
      const bool xa = x0 >= 0, xb = x0 + 1 < S, ya = y0 >= 0, yb = y0 + 1 < S;
      const float v00 = (xa && ya) ? m[y0 * S + x0] : 0.f, v01 = (xb && ya) ? m[y0 * S + x0 + 1] : 0.f;
      const float v10 = (xa && yb) ? m[(y0 + 1) * S + x0] : 0.f, v11 = (xb && yb) ? m[(y0 + 1) * S + x0 + 1] : 0.f;
      v = (v00 * (1.f - lx) + v01 * lx) * (1.f - ly) + (v10 * (1.f - lx) + v11 * lx) * ly;
    }
    o[e] = v >= thr ? 1 : 0;
  }
}

extern "C" int ttdg_paste_masks(const float* masks, const float* boxes, int R, int S, int H, int W, float threshold,
                                unsigned char* out, ttdg_stream_t stream) {
  TTDG_REQUIRE(R >= 0 && (R == 0 || (masks && boxes && out)) && S > 0 && H > 0 && W > 0, "paste_masks: bad arguments");
  if (R == 0) return 0;
  const int bx = (H * W + 255) / 256 < 64 ? (H * W + 255) / 256 : 64;
  hipLaunchKernelGGL(paste_masks_kernel, dim3(bx, R), dim3(256), 0, (hipStream_t)stream, masks, boxes, R, S, H, W, threshold, out);
  return ttdg_launch_status("paste_masks");
}

// ---------------------------------------------------------------------------------------------------
// y <- act(y + bias[c] (+ residual) (+ bias2[c])) in place on an NCHW activation: the per-channel shift of a folded
// FrozenBN (or a conv bias), the residual add and the ReLU of a bottleneck in ONE pass over the tensor instead of three
// or four (each pass over a 164 MB res2 activation costs ~65 us of HBM time).  In place; where gradients flow (res3 - res5 of
// a TTA step) ops.BiasActFn pairs it with relu_bwd_kernel below.
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                       const float* __restrict__ res, const float* __restrict__ bias2, int C, int HW,
                                                       size_t total, int relu) {
  // evaluation order of the unfused formulation, (y + bias) + (residual + bias2): bit-identical to conv-with-bias, add, ReLU
  if ((HW & 3) == 0) {
    const size_t nvec = total >> 2;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) {
      const int c = (int)((v * 4 / HW) % C);
      const float b = bias ? bias[c] : 0.f;
      float4 t = reinterpret_cast<float4*>(y)[v];
      if (bias) { t.x += b; t.y += b; t.z += b; t.w += b; }
      if (res) {
        float4 r = reinterpret_cast<const float4*>(res)[v];
        if (bias2) { const float b2 = bias2[c]; r.x += b2; r.y += b2; r.z += b2; r.w += b2; }
        t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
      } else if (bias2) {
        const float b2 = bias2[c];
        t.x += b2; t.y += b2; t.z += b2; t.w += b2;
      }
      if (relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
      reinterpret_cast<float4*>(y)[v] = t;
    }
  } else {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
      const int c = (int)((e / HW) % C);
      float t = y[e];
      if (bias) t += bias[c];
      if (res) t += bias2 ? res[e] + bias2[c] : res[e];
      else if (bias2) t += bias2[c];
      y[e] = relu ? fmaxf(t, 0.f) : t;
    }
  }
}

// Round 3: the same pass with the channel as a workgroup constant.  The flat kernel above pays a 64-bit division per float4
// to find its channel (~100 VALU instructions for 16 bytes of traffic) and keeps one vector per lane in flight: 3.5 TB/s at the
// median launch, 2x slower on average.  Here blockIdx.y is the (image, channel) plane, so bias[c] /
// bias2[c] are two scalar loads per workgroup, and every lane has four 16-byte loads (eight with a residual) in flight before
// its first store.  Needs HW % 4 == 0 (plane starts are then 16-byte aligned) and N * C <= 65535.
__global__ __launch_bounds__(256) void bias_act_plane_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                             const float* __restrict__ res, const float* __restrict__ bias2, int C, int HW4,
                                                             int relu) {
  const int plane = blockIdx.y, c = plane % C;
  const float b = bias ? bias[c] : 0.f, b2 = bias2 ? bias2[c] : 0.f;
  float4* yp = reinterpret_cast<float4*>(y) + (size_t)plane * HW4;
  const float4* rp = res ? reinterpret_cast<const float4*>(res) + (size_t)plane * HW4 : nullptr;
  const int v0 = blockIdx.x * 1024 + threadIdx.x;
  float4 t[4], r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int v = v0 + 256 * k;
    t[k] = v < HW4 ? yp[v] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (rp) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = v0 + 256 * k;
      r[k] = v < HW4 ? rp[v] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int v = v0 + 256 * k;
    if (v >= HW4) continue;
    float4 a = t[k];
    if (bias) { a.x += b; a.y += b; a.z += b; a.w += b; }        // (y + bias) + (residual + bias2): the unfused order
    if (rp) {
      float4 q = r[k];
      if (bias2) { q.x += b2; q.y += b2; q.z += b2; q.w += b2; }
      a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
    } else if (bias2) {
      a.x += b2; a.y += b2; a.z += b2; a.w += b2;
    }
    if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    yp[v] = a;
  }
}

// The same epilogue on a CHANNELS-LAST activation (N, H, W, C contiguous: what the backbone runs in since MIOpen's fastest fp32
// kernels on gfx950 are its NHWC implicit GEMMs - in NCHW it wraps them in transposes, 6 % of an adapted batch).  The channel is
// the fastest index: a float4 covers four consecutive channels, so bias / bias2 are float4 loads from a C-float table that stays in
// L1, no division (the lane's channel group advances by a constant), four vectors in flight per lane.  C % 4 == 0.
__global__ __launch_bounds__(256) void bias_act_nhwc_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                            const float* __restrict__ res, const float* __restrict__ bias2, int C4,
                                                            size_t nvec, int relu) {
  const size_t stride = (size_t)gridDim.x * 256;
  const int cstep = (int)(stride % (size_t)C4);
  size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
  int c4 = (int)(v % (size_t)C4);
  const float4* b4 = reinterpret_cast<const float4*>(bias);
  const float4* b24 = reinterpret_cast<const float4*>(bias2);
  for (; v + 3 * stride < nvec; v += 4 * stride) {
    float4 t[4], r[4];
    int cc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      t[u] = reinterpret_cast<float4*>(y)[v + u * stride];
      if (res) r[u] = reinterpret_cast<const float4*>(res)[v + u * stride];
      cc[u] = c4;
      c4 += cstep; c4 -= c4 >= C4 ? C4 : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (bias) { const float4 b = b4[cc[u]]; t[u].x += b.x; t[u].y += b.y; t[u].z += b.z; t[u].w += b.w; }
      if (res) {
        if (bias2) { const float4 b = b24[cc[u]]; r[u].x += b.x; r[u].y += b.y; r[u].z += b.z; r[u].w += b.w; }
        t[u].x += r[u].x; t[u].y += r[u].y; t[u].z += r[u].z; t[u].w += r[u].w;
      } else if (bias2) { const float4 b = b24[cc[u]]; t[u].x += b.x; t[u].y += b.y; t[u].z += b.z; t[u].w += b.w; }
      if (relu) { t[u].x = fmaxf(t[u].x, 0.f); t[u].y = fmaxf(t[u].y, 0.f); t[u].z = fmaxf(t[u].z, 0.f); t[u].w = fmaxf(t[u].w, 0.f); }
      reinterpret_cast<float4*>(y)[v + u * stride] = t[u];
    }
  }
  for (; v < nvec; v += stride) {
    float4 t = reinterpret_cast<float4*>(y)[v];
    if (bias) { const float4 b = b4[c4]; t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w; }
    if (res) {
      float4 r = reinterpret_cast<const float4*>(res)[v];
      if (bias2) { const float4 b = b24[c4]; r.x += b.x; r.y += b.y; r.z += b.z; r.w += b.w; }
      t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
    } else if (bias2) { const float4 b = b24[c4]; t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w; }
    if (relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
    reinterpret_cast<float4*>(y)[v] = t;
    c4 += cstep; c4 -= c4 >= C4 ? C4 : 0;
  }
}

extern "C" int ttdg_bias_act_nhwc(float* y, const float* bias, const float* residual, const float* bias2, int64_t rows, int C,
                                  int relu, ttdg_stream_t stream) {
  TTDG_REQUIRE(y && rows >= 0 && C > 0 && (C & 3) == 0, "bias_act_nhwc: C must be a positive multiple of 4");
  TTDG_REQUIRE((((uintptr_t)y | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)bias2) & 15) == 0, "bias_act_nhwc: 16-byte alignment");
  const size_t nvec = (size_t)rows * (size_t)(C >> 2);
  if (nvec == 0) return 0;
  const size_t want = (nvec + 1023) / 1024;                  // ~4 vectors per lane
  const int blocks = (int)(want < 1 ? 1 : (want > 16384 ? 16384 : want));
  hipLaunchKernelGGL(bias_act_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, bias, residual, bias2, C >> 2, nvec, relu);
  return ttdg_launch_status("bias_act_nhwc");
}

extern "C" int ttdg_bias_act(float* y, const float* bias, const float* residual, const float* bias2, int N, int C, int HW,
                             int relu, ttdg_stream_t stream) {
  TTDG_REQUIRE(y && N >= 0 && C > 0 && HW > 0, "bias_act: bad arguments");
  const size_t total = (size_t)N * C * HW;
  if (total == 0) return 0;
  if ((HW & 3) == 0 && (size_t)N * C <= 65535 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)residual & 15) == 0) {
    const int HW4 = HW >> 2;
    hipLaunchKernelGGL(bias_act_plane_kernel, dim3((HW4 + 1023) / 1024, N * C), dim3(256), 0, (hipStream_t)stream, y, bias, residual, bias2,
                       C, HW4, relu);
    return ttdg_launch_status("bias_act");
  }
  const size_t work = (HW & 3) == 0 ? total / 4 : total;
  const size_t want = (work + 255) / 256;
  const int blocks = (int)(want < 8192 ? want : 8192);
  hipLaunchKernelGGL(bias_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, bias, residual, bias2, C, HW, total, relu);
  return ttdg_launch_status("bias_act");
}

// Backward of the ReLU epilogue above where gradients do flow (the adapted res3 - res5 blocks): gin = out > 0 ? gout : 0,
// one pass; the same tensor is the gradient of the convolution output AND of the residual branch.
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ out,
                                                       float* __restrict__ gin, size_t total) {
  const size_t nvec = total >> 2;
  // two vectors of each operand in flight per lane before the first store
  for (size_t v = (size_t)blockIdx.x * 512 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 512) {
    const size_t v1 = v + 256;
    const bool has1 = v1 < nvec;
    const float4 g0 = reinterpret_cast<const float4*>(gout)[v], o0 = reinterpret_cast<const float4*>(out)[v];
    float4 g1 = g0, o1 = o0;
    if (has1) { g1 = reinterpret_cast<const float4*>(gout)[v1]; o1 = reinterpret_cast<const float4*>(out)[v1]; }
    float4 r;
    r.x = o0.x > 0.f ? g0.x : 0.f; r.y = o0.y > 0.f ? g0.y : 0.f; r.z = o0.z > 0.f ? g0.z : 0.f; r.w = o0.w > 0.f ? g0.w : 0.f;
    reinterpret_cast<float4*>(gin)[v] = r;
    if (has1) {
      r.x = o1.x > 0.f ? g1.x : 0.f; r.y = o1.y > 0.f ? g1.y : 0.f; r.z = o1.z > 0.f ? g1.z : 0.f; r.w = o1.w > 0.f ? g1.w : 0.f;
      reinterpret_cast<float4*>(gin)[v1] = r;
    }
  }
  for (size_t e = (nvec << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256)
    gin[e] = out[e] > 0.f ? gout[e] : 0.f;
}

extern "C" int ttdg_relu_bwd(const float* gout, const float* out, float* gin, size_t total, ttdg_stream_t stream) {
  TTDG_REQUIRE(gout && out && gin, "relu_bwd: null pointer");
  if (total == 0) return 0;
  const size_t want = (total / 4 + 511) / 512 + 1;
  const int blocks = (int)(want < 16384 ? want : 16384);
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gout, out, gin, total);
  return ttdg_launch_status("relu_bwd");
}

// ---------------------------------------------------------------------------------------------------
// detectron2 ROIPooler [3P] in ONE launch: every ROI picks its FPN level inside the kernel
//   level = clamp(floor(canonical_level + log2(sqrt(area) / canonical_size + 1e-8)), min_level, max_level) - min_level
// and is sampled from that level's map (same adaptive bilinear sampling as roi_align_fwd_kernel).  The torch formulation
// is a per-level nonzero() (a host read each) + index + ROIAlign + index_put: 4 syncs and ~40 launches per call.
// Work mapping (round 2): the eight XCDs of the chip have private 4 MB L2s and workgroup b is observed to run on XCD
// b % 8 (placement affects speed only).  With the flat (roi, channel, bin) order every XCD touched every channel plane of
// every ROI patch: the rocprofv3 FETCH_SIZE of one box-head call (4000 proposals of a trained RPN, all stacked on the same
// few objects) was 2.1 - 4.0 GB for 218 MB of feature maps.  Now XCD x owns the channel slice [x C/8, (x+1) C/8): its L2
// only ever sees 1/8 of the planes, and the proposals of an image - consecutive in the roi list - re-read them from there.
template <bool kSliced>
__global__ __launch_bounds__(256) void roi_align_ml_kernel(ttdg_fpn_t fp, ttdg_levels_t lv, const float* __restrict__ rois, int R, int P,
                                                           float canon_size, int canon_level, int min_level,
                                                           float* __restrict__ out) {
  const int C = fp.C;
  const int CS = kSliced ? C >> 3 : C;                       // channels per slice
  const int slice = kSliced ? (int)(blockIdx.x & 7) : 0;
  const long per = (long)R * CS * P * P;                     // outputs per slice
  const long nblk = kSliced ? (long)(gridDim.x >> 3) : (long)gridDim.x;
  for (long e = (long)(kSliced ? blockIdx.x >> 3 : blockIdx.x) * 256 + threadIdx.x; e < per; e += nblk * 256) {
    const int pw = e % P, ph = (e / P) % P, c = slice * CS + (int)((e / ((long)P * P)) % CS);
    const int r = e / ((long)P * P * CS);
    const long idx = (((long)r * C + c) * P + ph) * P + pw;
    const float* roi = rois + (size_t)r * 5;
    const int b = (int)roi[0];
    const float area = fmaxf((roi[3] - roi[1]) * (roi[4] - roi[2]), 0.f);
    int l = (int)floorf((float)canon_level + log2f(sqrtf(area) / canon_size + 1e-8f));
    l = min(max(l, min_level), min_level + fp.n - 1) - min_level;
    const int H = fp.h[l], W = fp.w[l];
    const float scale = 1.f / (float)lv.stride[l];
    const float x1 = roi[1] * scale - 0.5f, y1 = roi[2] * scale - 0.5f;
    const float rw = roi[3] * scale - 0.5f - x1, rh = roi[4] * scale - 0.5f - y1;
    const float bw = rw / P, bh = rh / P;
    const int gh = max(1, (int)ceilf(rh / P)), gw = max(1, (int)ceilf(rw / P));
    const float* f = fp.feat[l] + ((size_t)b * C + c) * H * W;
    float acc = 0.f;
    for (int iy = 0; iy < gh; ++iy) {
      float y = y1 + ph * bh + (iy + 0.5f) * bh / gh;
      for (int ix = 0; ix < gw; ++ix) {
        float x = x1 + pw * bw + (ix + 0.5f) * bw / gw;
        if (y < -1.f || y > H || x < -1.f || x > W) continue;
        float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
        int y0 = (int)yy, x0 = (int)xx, y1i, x1i;
        if (y0 >= H - 1) { y0 = y1i = H - 1; yy = (float)y0; } else y1i = y0 + 1;
        if (x0 >= W - 1) { x0 = x1i = W - 1; xx = (float)x0; } else x1i = x0 + 1;
        const float ly = yy - y0, lx = xx - x0, hy = 1.f - ly, hx = 1.f - lx;
        acc += hy * hx * f[y0 * W + x0] + hy * lx * f[y0 * W + x1i] + ly * hx * f[y1i * W + x0] + ly * lx * f[y1i * W + x1i];
      }
    }
    out[idx] = acc / (float)(gh * gw);
  }
}

// ---- separable ROIPooler (round 2) ---------------------------------------------------------------------------------
// ROIAlign (aligned, adaptive sampling) is a separable linear map of the ROI's patch:
//     out[ph][pw] = sum_yy sum_xx Wy[ph][yy] Wx[pw][xx] F[ylo + yy][xlo + xx]
// where Wy[ph][.] collects, over the g_h samples of bin row ph, the two bilinear row weights of every sample (the skip rule
// `y < -1 || y > H` and the border clamp are per-axis, so they live in the tables too), likewise Wx, and 1 / (g_h g_w) is
// folded into Wy.  The direct kernel above spends ~300 VALU instructions per output on addresses and weights that are the
// same for all C channels of a ROI; here one workgroup owns (ROI, channel slice = XCD), builds the two tables ONCE in LDS
// (<= g + 2 weights per bin: sample spacing b / ceil(b) <= 1 pixel), and every wavefront then streams channels: the patch
// rows come in coalesced, T[yy][pw] = sum_xx Wx F is contracted first, then out = sum_yy Wy T.  ~760 FMAs and ~600 loaded
// floats per channel instead of 784 scattered taps x 4 + their arithmetic.
#define RA_MAXP 14         /* pooled size handled (7: box head, 14: mask head) */
#define RA_MAXW 10         /* weights per bin and axis: g + 2 with g <= 8 */
#define RA_MAXPATCH 48     /* patch rows / columns handled by the table path (48 KB of LDS per workgroup: 3 per CU) */
#define RA_WAVES 4

struct RaAxis {            // one axis of one ROI
  float w[RA_MAXP][RA_MAXW];
  int start[RA_MAXP];      // first patch index (absolute pixel index) of bin p
  int cnt[RA_MAXP];        // number of weights (0 = every sample of the bin was skipped)
};

__device__ __forceinline__ void ra_build_axis(RaAxis& ax, int p, float v1, float b, int g, int L) {
  // bin p of an axis of length L: samples v = v1 + p b + (s + 0.5) b / g, s < g
  float w[RA_MAXW];
#pragma unroll
  for (int k = 0; k < RA_MAXW; ++k) w[k] = 0.f;
  int first = -1, last = -1;
  for (int s = 0; s < g; ++s) {
    float v = v1 + p * b + (s + 0.5f) * b / g;
    if (v < -1.f || v > (float)L) continue;
    float vv = fmaxf(v, 0.f);
    int i0 = (int)vv, i1;
    if (i0 >= L - 1) { i0 = i1 = L - 1; vv = (float)i0; } else i1 = i0 + 1;
    const float l = vv - i0, h = 1.f - l;
    if (first < 0) first = i0;                  // samples are visited in increasing order: i0 is non-decreasing
    last = i1;
    const int k0 = i0 - first, k1 = i1 - first;
#pragma unroll
    for (int k = 0; k < RA_MAXW; ++k) { if (k == k0) w[k] += h; if (k == k1) w[k] += l; }
  }
  ax.start[p] = first < 0 ? 0 : first;
  ax.cnt[p] = first < 0 ? 0 : last - first + 1;
#pragma unroll
  for (int k = 0; k < RA_MAXW; ++k) ax.w[p][k] = w[k];
}

__global__ __launch_bounds__(64 * RA_WAVES) void roi_align_sep_kernel(ttdg_fpn_t fp, ttdg_levels_t lv, const float* __restrict__ rois, int R, int P,
                                                                       int nslice, float canon_size, int canon_level, int min_level,
                                                                       float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float ra_smem[];
  __shared__ RaAxis s_y, s_x;
  __shared__ int s_geo[8];      // ylo, hp, xlo, wp, ok
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = blockIdx.x / nslice, slice = blockIdx.x % nslice;
  const int C = fp.C, CS = C / nslice;
  const float* roi = rois + (size_t)r * 5;
  const int b = (int)roi[0];
  const float area = fmaxf((roi[3] - roi[1]) * (roi[4] - roi[2]), 0.f);
  int l = (int)floorf((float)canon_level + log2f(sqrtf(area) / canon_size + 1e-8f));
  l = min(max(l, min_level), min_level + fp.n - 1) - min_level;
  const int H = fp.h[l], W = fp.w[l];
  const float scale = 1.f / (float)lv.stride[l];
  const float x1 = roi[1] * scale - 0.5f, y1 = roi[2] * scale - 0.5f;
  const float rw = roi[3] * scale - 0.5f - x1, rh = roi[4] * scale - 0.5f - y1;
  const float bw = rw / P, bh = rh / P;
  const int gh = max(1, (int)ceilf(rh / P)), gw = max(1, (int)ceilf(rw / P));
  const bool tables = gh <= RA_MAXW - 2 && gw <= RA_MAXW - 2;
  if (tables) {
    if (tid < P) ra_build_axis(s_y, tid, y1, bh, gh, H);
    else if (tid >= 64 && tid < 64 + P) ra_build_axis(s_x, tid - 64, x1, bw, gw, W);
  }
  __syncthreads();
  if (tid == 0) {
    int ylo = 1 << 30, yhi = -1, xlo = 1 << 30, xhi = -1;
    if (tables)
      for (int p = 0; p < P; ++p) {
        if (s_y.cnt[p]) { ylo = min(ylo, s_y.start[p]); yhi = max(yhi, s_y.start[p] + s_y.cnt[p] - 1); }
        if (s_x.cnt[p]) { xlo = min(xlo, s_x.start[p]); xhi = max(xhi, s_x.start[p] + s_x.cnt[p] - 1); }
      }
    const bool empty = yhi < 0 || xhi < 0;
    s_geo[0] = empty ? 0 : ylo; s_geo[1] = empty ? 0 : yhi - ylo + 1;
    s_geo[2] = empty ? 0 : xlo; s_geo[3] = empty ? 0 : xhi - xlo + 1;
    s_geo[4] = tables && (empty || (yhi - ylo + 1 <= RA_MAXPATCH && xhi - xlo + 1 <= RA_MAXPATCH));
  }
  __syncthreads();
  const int ylo = s_geo[0], hp = s_geo[1], xlo = s_geo[2], wp = s_geo[3];
  float* obase = out + ((size_t)r * C + (size_t)slice * CS) * P * P;
  const float* fbase = fp.feat[l] + ((size_t)b * C + (size_t)slice * CS) * H * W;
  if (s_geo[4] && (hp == 0 || wp == 0)) {
    // every sample of one axis lies outside the map (the other axis may still have taps): all samples are skipped and
    // every bin is 0 - the T tile below would never be written
    for (int e = tid; e < CS * P * P; e += 64 * RA_WAVES) obase[e] = 0.f;
    return;
  }
  if (!s_geo[4]) {
    // a ROI outside the table limits (more than 8 samples per bin, or a patch above 64 pixels): the direct formula
    const float inv = 1.f / (float)(gh * gw);
    for (int e = tid; e < CS * P * P; e += 64 * RA_WAVES) {
      const int pw = e % P, ph = (e / P) % P, cc = e / (P * P);
      const float* f = fbase + (size_t)cc * H * W;
      float acc = 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        float y = y1 + ph * bh + (iy + 0.5f) * bh / gh;
        for (int ix = 0; ix < gw; ++ix) {
          float x = x1 + pw * bw + (ix + 0.5f) * bw / gw;
          if (y < -1.f || y > H || x < -1.f || x > W) continue;
          float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
          int y0 = (int)yy, x0 = (int)xx, y1i, x1i;
          if (y0 >= H - 1) { y0 = y1i = H - 1; yy = (float)y0; } else y1i = y0 + 1;
          if (x0 >= W - 1) { x0 = x1i = W - 1; xx = (float)x0; } else x1i = x0 + 1;
          const float ly = yy - y0, lx = xx - x0, hy = 1.f - ly, hx = 1.f - lx;
          acc += hy * hx * f[y0 * W + x0] + hy * lx * f[y0 * W + x1i] + ly * hx * f[y1i * W + x0] + ly * lx * f[y1i * W + x1i];
        }
      }
      obase[e] = acc * inv;
    }
    return;
  }
  // per-wavefront scratch: patch (hp x wpad) and T (hp x P)
  const int wpad = wp | 1;
  float* patch = ra_smem + (size_t)wave * (RA_MAXPATCH * (RA_MAXPATCH + 1) + RA_MAXPATCH * RA_MAXP);
  float* T = patch + RA_MAXPATCH * (RA_MAXPATCH + 1);
  const float inv = 1.f / (float)(gh * gw);
  const int rows_per = wp <= 32 ? 2 : 1;                     // patch rows fetched per load round
  const int lx_ = rows_per == 2 ? (lane & 31) : lane, lr_ = rows_per == 2 ? (lane >> 5) : 0;
  for (int cc = wave; cc < CS; cc += RA_WAVES) {
    const float* f = fbase + (size_t)cc * H * W + (size_t)ylo * W + xlo;
    // eight load rounds in flight before the first LDS store (a load -> wait -> store loop pays one L2 round trip per row)
    for (int y0 = lr_; y0 < hp; y0 += 8 * rows_per) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int yy = y0 + k * rows_per;
        v[k] = (yy < hp && lx_ < wp) ? f[(size_t)yy * W + lx_] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int yy = y0 + k * rows_per;
        if (yy < hp && lx_ < wp) patch[yy * wpad + lx_] = v[k];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int it = lane; it < hp * P; it += 64) {
      const int yy = it / P, pw = it - yy * P;
      const int n = s_x.cnt[pw], st = s_x.start[pw] - xlo;
      const float* pr = patch + yy * wpad + st;
      float t = 0.f;
      for (int k = 0; k < n; ++k) t = fmaf(s_x.w[pw][k], pr[k], t);
      T[it] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* o = obase + (size_t)cc * P * P;
    for (int it = lane; it < P * P; it += 64) {
      const int ph = it / P, pw = it - ph * P;
      const int n = s_y.cnt[ph], st = s_y.start[ph] - ylo;
      float a = 0.f;
      for (int k = 0; k < n; ++k) a = fmaf(s_y.w[ph][k], T[(st + k) * P + pw], a);
      o[it] = a * inv;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- channels-last ROIPooler (round 2) ------------------------------------------------------------------------------
// Same tables as roi_align_sep_kernel, but the feature maps are NHWC copies (ttdg_nchw_to_nhwc) and a LANE IS A CHANNEL:
// every tap of a bin is one fully coalesced 256-byte read per wavefront, its weight wy[ph][ky] * wx[pw][kx] is uniform
// (LDS broadcast), so a wavefront spends ~(g+1)^2 FMAs + as many loads per bin for 64 channels - against ~600 wave
// instructions per CHANNEL in the patch-staging kernel, which the profile showed to be VALU-issue bound (0.87 ms for
// 4000 ROIs).  The (64 channels x 49 bins) result block is contiguous in the (R, C, P, P) output: it is transposed through
// LDS and written coalesced.
/* bins staged per output flush: template parameter RN_CH of roi_align_nhwc_kernel (P = 7: all 49 of them, or 25; P = 14: four / eight flushes) */

// RN_CH: bins staged per output flush.  49 = the whole 7 x 7 block (one fully coalesced flush per 64 channels, 51 KB of LDS per
// workgroup: 3 workgroups per CU); 25 halves the tile (26 KB: 6 workgroups per CU - the kernel waits on L2 round trips, so
// occupancy is throughput) at the price of two 100-byte runs per channel instead of one 196-byte run.
template <int RN_CH>
__global__ __launch_bounds__(256) void roi_align_nhwc_kernel(ttdg_fpn_t fp, ttdg_levels_t lv, const float* __restrict__ rois, int R, int P,
                                                             float canon_size, int canon_level, int min_level, float* __restrict__ out,
                                                             int g_roi_xcd_chunks) {
  constexpr int RN_CHUNK = RN_CH;
  __shared__ RaAxis s_y, s_x;
  __shared__ float s_tile[4][64 * (RN_CHUNK + 1)];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // Round 3: ROI -> XCD placement.  Workgroup b is observed on XCD b % 8 (speed only), and the ROI list is sorted by image with
  // the proposals of one object next to each other.  With r = b every XCD's private L2 saw every image's hot patches: the
  // FETCH_SIZE of a box-head call was 3.3-6.5x the bytes of the maps - eight L2s each fetching the same
  // lines through the fabric.  Now XCD x owns the contiguous eighth [x * chunk, (x + 1) * chunk) of the list: one image's
  // (half an image's) patches per L2.
  const int chunk = (R + 7) >> 3;
  const int r = (g_roi_xcd_chunks ? ((int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3)) : (int)blockIdx.x);
  if (r >= R || (g_roi_xcd_chunks && (int)(blockIdx.x >> 3) >= chunk)) return;
  const int C = fp.C;
  const float* roi = rois + (size_t)r * 5;
  const int b = (int)roi[0];
  const float area = fmaxf((roi[3] - roi[1]) * (roi[4] - roi[2]), 0.f);
  int l = (int)floorf((float)canon_level + log2f(sqrtf(area) / canon_size + 1e-8f));
  l = min(max(l, min_level), min_level + fp.n - 1) - min_level;
  const int H = fp.h[l], W = fp.w[l];
  const float scale = 1.f / (float)lv.stride[l];
  const float x1 = roi[1] * scale - 0.5f, y1 = roi[2] * scale - 0.5f;
  const float rw = roi[3] * scale - 0.5f - x1, rh = roi[4] * scale - 0.5f - y1;
  const float bw = rw / P, bh = rh / P;
  const int gh = max(1, (int)ceilf(rh / P)), gw = max(1, (int)ceilf(rw / P));
  const bool tables = gh <= RA_MAXW - 2 && gw <= RA_MAXW - 2;
  if (tables) {
    if (tid < P) ra_build_axis(s_y, tid, y1, bh, gh, H);
    else if (tid >= 64 && tid < 64 + P) ra_build_axis(s_x, tid - 64, x1, bw, gw, W);
  }
  __syncthreads();
  const float inv = 1.f / (float)(gh * gw);
  const int PP = P * P;
  float* tile = s_tile[wave];
  for (int c0 = wave * 64; c0 < C; c0 += 256) {
    const int c = c0 + lane;
    const bool cok = c < C;
    const float* f = fp.feat[l] + (size_t)b * H * W * C + (cok ? c : 0);
    float* o = out + ((size_t)r * C + c0) * PP;
    const int nch = min(64, C - c0);
    for (int q0 = 0; q0 < PP; q0 += RN_CHUNK) {
      const int nb = min(RN_CHUNK, PP - q0);
      for (int j = 0; j < nb; ++j) {
        const int bin = q0 + j, ph = bin / P, pw = bin - ph * P;
        float acc = 0.f;
        if (tables) {
          // all taps of four patch rows are issued before the first FMA consumes one (a tap-by-tap loop is one L2 round trip
          // per tap: 4 us per bin measured); rows / columns beyond the bin's extent are predicated off, their weights are 0
          const int ny = s_y.cnt[ph], nx = s_x.cnt[pw], ys = s_y.start[ph], xs = s_x.start[pw];
          float wxr[RA_MAXW];
#pragma unroll
          for (int kx = 0; kx < RA_MAXW; ++kx) wxr[kx] = s_x.w[pw][kx];
          for (int ky0 = 0; ky0 < ny; ky0 += 4) {
            float v[4][RA_MAXW];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              const float* rowp = f + ((size_t)(ys + ky0 + a) * W + xs) * C;
#pragma unroll
              for (int kx = 0; kx < RA_MAXW; ++kx) v[a][kx] = (ky0 + a < ny && kx < nx) ? rowp[(size_t)kx * C] : 0.f;
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              const float wyv = (ky0 + a < ny) ? s_y.w[ph][ky0 + a] : 0.f;
#pragma unroll
              for (int kx = 0; kx < RA_MAXW; ++kx) acc = fmaf(wyv * wxr[kx], v[a][kx], acc);
            }
          }
        } else {      // more than 8 samples per bin and axis: the direct formula, still one channel per lane
          for (int iy = 0; iy < gh; ++iy) {
            float y = y1 + ph * bh + (iy + 0.5f) * bh / gh;
            for (int ix = 0; ix < gw; ++ix) {
              float x = x1 + pw * bw + (ix + 0.5f) * bw / gw;
              if (y < -1.f || y > H || x < -1.f || x > W) continue;
              float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
              int y0 = (int)yy, x0 = (int)xx, y1i, x1i;
              if (y0 >= H - 1) { y0 = y1i = H - 1; yy = (float)y0; } else y1i = y0 + 1;
              if (x0 >= W - 1) { x0 = x1i = W - 1; xx = (float)x0; } else x1i = x0 + 1;
              const float ly = yy - y0, lx = xx - x0, hy = 1.f - ly, hx = 1.f - lx;
              acc += hy * hx * f[((size_t)y0 * W + x0) * C] + hy * lx * f[((size_t)y0 * W + x1i) * C] +
                     ly * hx * f[((size_t)y1i * W + x0) * C] + ly * lx * f[((size_t)y1i * W + x1i) * C];
            }
          }
        }
        tile[lane * (RN_CHUNK + 1) + j] = acc * inv;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // channel ch of the block owns out[ch * PP + q0 .. + nb): runs of nb floats, the whole block when nb == PP
      for (int idx = lane; idx < nch * nb; idx += 64) {
        const int ch = idx / nb, jj = idx - ch * nb;
        o[(size_t)ch * PP + q0 + jj] = tile[ch * (RN_CHUNK + 1) + jj];
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---- [r6] the same pooler with FOUR channels per lane ---------------------------------------------------------------------------------
// roi_align_nhwc_kernel issues one dword per lane and tap: 256 B per wavefront instruction, 3.9 GB of tap traffic through the L1 / texture
// addresser per box-head call (4000 ROIs x 49 bins x ~20 taps x 1 KB) - the kernel is bound by that instruction rate, not by HBM or the
// fabric (the XCD-chunk A/B of round 3 moved it by 2 %, FETCH_SIZE 3.6 x the algorithmic bytes is L2 traffic between overlapping proposals).
// Here a lane owns four consecutive channels (one 16-byte load per tap: 1 KB per wavefront instruction, a quarter of the instructions for
// the same bytes), a wavefront covers 256 channels, and the four wavefronts of the workgroup split the BINS of the ROI (bin = wave + 4 i).
// The results meet in one LDS tile [bin][256 + 1] and leave as ONE contiguous run per 49-bin chunk and channel: for P = 7 the ROI's whole
// (C, 7, 7) block is a single 50 KB contiguous store.  Same taps, same weights, same order of the additions per output value: bit-identical
// to roi_align_nhwc_kernel.
#define RN4_CH 49
__global__ __launch_bounds__(256) void roi_align_nhwc4_kernel(ttdg_fpn_t fp, ttdg_levels_t lv, const float* __restrict__ rois, int R, int P,
                                                              float canon_size, int canon_level, int min_level, float* __restrict__ out) {
  __shared__ RaAxis s_y, s_x;
  __shared__ __attribute__((aligned(16))) float s_t[RN4_CH * 257];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = blockIdx.x;
  if (r >= R) return;
  const int C = fp.C;
  const float* roi = rois + (size_t)r * 5;
  const int b = (int)roi[0];
  const float area = fmaxf((roi[3] - roi[1]) * (roi[4] - roi[2]), 0.f);
  int l = (int)floorf((float)canon_level + log2f(sqrtf(area) / canon_size + 1e-8f));
  l = min(max(l, min_level), min_level + fp.n - 1) - min_level;
  const int H = fp.h[l], W = fp.w[l];
  const float scale = 1.f / (float)lv.stride[l];
  const float x1 = roi[1] * scale - 0.5f, y1 = roi[2] * scale - 0.5f;
  const float rw = roi[3] * scale - 0.5f - x1, rh = roi[4] * scale - 0.5f - y1;
  const float bw = rw / P, bh = rh / P;
  const int gh = max(1, (int)ceilf(rh / P)), gw = max(1, (int)ceilf(rw / P));
  const bool tables = gh <= RA_MAXW - 2 && gw <= RA_MAXW - 2;
  if (tables) {
    if (tid < P) ra_build_axis(s_y, tid, y1, bh, gh, H);
    else if (tid >= 64 && tid < 64 + P) ra_build_axis(s_x, tid - 64, x1, bw, gw, W);
  }
  __syncthreads();
  const float inv = 1.f / (float)(gh * gw);
  const int PP = P * P;
  typedef float rn4_f4 __attribute__((ext_vector_type(4)));
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + 4 * lane;
    const bool cok = c < C;                                   // (C % 4 == 0: a lane's four channels are inside or outside)
    const float* f = fp.feat[l] + (size_t)b * H * W * C + (cok ? c : 0);
    const int nch = min(256, C - c0);
    for (int q0 = 0; q0 < PP; q0 += RN4_CH) {
      const int nb = min(RN4_CH, PP - q0);
      for (int j = wave; j < nb; j += 4) {
        const int bin = q0 + j, ph = bin / P, pw = bin - ph * P;
        rn4_f4 acc = {0.f, 0.f, 0.f, 0.f};
        if (tables) {
          const int ny = s_y.cnt[ph], nx = s_x.cnt[pw], ys = s_y.start[ph], xs = s_x.start[pw];
          float wxr[RA_MAXW];
#pragma unroll
          for (int kx = 0; kx < RA_MAXW; ++kx) wxr[kx] = s_x.w[pw][kx];
          for (int ky0 = 0; ky0 < ny; ky0 += 2) {
            rn4_f4 v[2][RA_MAXW];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              const float* rowp = f + ((size_t)(ys + ky0 + a) * W + xs) * C;
#pragma unroll
              for (int kx = 0; kx < RA_MAXW; ++kx) {
                if (ky0 + a < ny && kx < nx) v[a][kx] = *reinterpret_cast<const rn4_f4*>(rowp + (size_t)kx * C);
                else v[a][kx] = rn4_f4{0.f, 0.f, 0.f, 0.f};
              }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              const float wyv = (ky0 + a < ny) ? s_y.w[ph][ky0 + a] : 0.f;
#pragma unroll
              for (int kx = 0; kx < RA_MAXW; ++kx) {
                const float wgt = wyv * wxr[kx];
                acc.x = fmaf(wgt, v[a][kx].x, acc.x); acc.y = fmaf(wgt, v[a][kx].y, acc.y);
                acc.z = fmaf(wgt, v[a][kx].z, acc.z); acc.w = fmaf(wgt, v[a][kx].w, acc.w);
              }
            }
          }
        } else {      // more than 8 samples per bin and axis: the direct formula
          for (int iy = 0; iy < gh; ++iy) {
            float y = y1 + ph * bh + (iy + 0.5f) * bh / gh;
            for (int ix = 0; ix < gw; ++ix) {
              float x = x1 + pw * bw + (ix + 0.5f) * bw / gw;
              if (y < -1.f || y > H || x < -1.f || x > W) continue;
              float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
              int y0 = (int)yy, x0 = (int)xx, y1i, x1i;
              if (y0 >= H - 1) { y0 = y1i = H - 1; yy = (float)y0; } else y1i = y0 + 1;
              if (x0 >= W - 1) { x0 = x1i = W - 1; xx = (float)x0; } else x1i = x0 + 1;
              const float ly = yy - y0, lx = xx - x0, hy = 1.f - ly, hx = 1.f - lx;
              const rn4_f4 a00 = *reinterpret_cast<const rn4_f4*>(f + ((size_t)y0 * W + x0) * C), a01 = *reinterpret_cast<const rn4_f4*>(f + ((size_t)y0 * W + x1i) * C);
              const rn4_f4 a10 = *reinterpret_cast<const rn4_f4*>(f + ((size_t)y1i * W + x0) * C), a11 = *reinterpret_cast<const rn4_f4*>(f + ((size_t)y1i * W + x1i) * C);
              acc.x += hy * hx * a00.x + hy * lx * a01.x + ly * hx * a10.x + ly * lx * a11.x;
              acc.y += hy * hx * a00.y + hy * lx * a01.y + ly * hx * a10.y + ly * lx * a11.y;
              acc.z += hy * hx * a00.z + hy * lx * a01.z + ly * hx * a10.z + ly * lx * a11.z;
              acc.w += hy * hx * a00.w + hy * lx * a01.w + ly * hx * a10.w + ly * lx * a11.w;
            }
          }
        }
        float* t = s_t + j * 257 + 4 * lane;
        t[0] = acc.x * inv, t[1] = acc.y * inv, t[2] = acc.z * inv, t[3] = acc.w * inv;
      }
      __syncthreads();
      // channel ch of the block owns out[ch * PP + q0 .. + nb): one contiguous run per channel; all of (C, P, P) when nb == PP
      float* o = out + ((size_t)r * C + c0) * PP + q0;
      for (int e = tid; e < nch * nb; e += 256) {
        const int ch = e / nb, jj = e - ch * nb;
        o[(size_t)ch * PP + jj] = s_t[jj * 257 + ch];
      }
      __syncthreads();
    }
  }
}

// NCHW -> NHWC copy of one feature level: (B, C, HW) -> (B, HW, C), 32 x 32 tiles through LDS (both sides coalesced)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW) {
  __shared__ float t[32][33];
  const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  const float* s = src + (size_t)b * C * HW;
  float* d = dst + (size_t)b * HW * C;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, p = p0 + tx;
    if (c < C && p < HW) t[ty + 8 * k][tx] = s[(size_t)c * HW + p];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = p0 + ty + 8 * k, c = c0 + tx;
    if (c < C && p < HW) d[(size_t)p * C + c] = t[tx][ty + 8 * k];
  }
}

extern "C" int ttdg_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, ttdg_stream_t stream) {
  TTDG_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad arguments");
  TTDG_LIMIT(B <= 65535 && (C + 31) / 32 <= 65535, "nchw_to_nhwc: too many images / channels");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((H * W + 31) / 32, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, src, dst, C, H * W);
  return ttdg_launch_status("nchw_to_nhwc");
}

static int g_roi_nhwc_chunk = 25;   // bins per output flush: 25 (default: 355 vs 421 us per call, profiles/r03_roi_align_ab.txt); ttdg_debug_set_roi_align_sliced(mode | 32) = 49, | 64 = 13
static int g_roi_nhwc_wide = 1;     // [r6] 1 = four channels per lane (roi_align_nhwc4_kernel, C % 4 == 0), 0 = one channel per lane; ttdg_debug_set_roi_align_sliced(mode | 128) selects 0
static int g_roi_nhwc_xcd = 0;      // 1 = XCD x owns a contiguous eighth of the ROI list, 0 = ROI r on workgroup r (default: measured 427 vs 437 us per call,
                                    // profiles/r03_roi_align_ab.txt - the pooler is not bound by fabric traffic); ttdg_debug_set_roi_align_sliced(mode | 16) selects 1
extern "C" int ttdg_roi_align_multilevel_nhwc(ttdg_fpn_t fp, ttdg_levels_t lv, const float* rois, int R, int P, float canonical_size,
                                              int canonical_level, int min_level, float* out, ttdg_stream_t stream) {
  TTDG_REQUIRE(R >= 0 && (R == 0 || (rois && out)) && P > 0 && P <= RA_MAXP && fp.n >= 1 && fp.n <= TTDG_MAX_LEVELS && fp.C > 0 && lv.n == fp.n,
               "roi_align_multilevel_nhwc: bad arguments");
  if (R == 0) return 0;
  bool aligned = (fp.C & 3) == 0;
  for (int l = 0; l < fp.n; ++l) aligned = aligned && (((uintptr_t)fp.feat[l]) & 15) == 0;
  if (g_roi_nhwc_wide && aligned && !g_roi_nhwc_xcd) {
    hipLaunchKernelGGL(roi_align_nhwc4_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, fp, lv, rois, R, P, canonical_size, canonical_level,
                       min_level, out);
    return ttdg_launch_status("roi_align_multilevel_nhwc (4 channels per lane)");
  }
  const int grid = g_roi_nhwc_xcd ? 8 * ((R + 7) / 8) : R;
  if (g_roi_nhwc_chunk == 13)
    hipLaunchKernelGGL((roi_align_nhwc_kernel<13>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fp, lv, rois, R, P, canonical_size,
                       canonical_level, min_level, out, g_roi_nhwc_xcd);
  else if (g_roi_nhwc_chunk == 25)
    hipLaunchKernelGGL((roi_align_nhwc_kernel<25>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fp, lv, rois, R, P, canonical_size,
                       canonical_level, min_level, out, g_roi_nhwc_xcd);
  else
    hipLaunchKernelGGL((roi_align_nhwc_kernel<49>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fp, lv, rois, R, P, canonical_size,
                       canonical_level, min_level, out, g_roi_nhwc_xcd);
  return ttdg_launch_status("roi_align_multilevel_nhwc");
}

static int g_roi_align_sliced = 1;   // 2 = separable table kernel (default below), 1 = direct kernel with the XCD-sliced mapping, 0 = direct, flat
static int g_roi_align_mode = 2;
// bits 0-1: NCHW kernel variant (2 = separable tables, 1 = direct XCD-sliced, 0 = direct flat); bit 4 (value 16): the channels-last
// kernel gives every XCD one contiguous eighth of the ROI list instead of ROI r on workgroup r
extern "C" int ttdg_debug_set_roi_align_sliced(int on) {
  g_roi_nhwc_xcd = (on >= 0 && (on & 16)) ? 1 : 0;
  g_roi_nhwc_wide = (on >= 0 && (on & 128)) ? 0 : 1;
  g_roi_nhwc_chunk = (on >= 0 && (on & 64)) ? 13 : ((on >= 0 && (on & 32)) ? 49 : 25);
  if (on >= 0) on &= 7;
  g_roi_align_mode = on < 0 ? 2 : (on > 2 ? 2 : on);
  g_roi_align_sliced = on != 0;
  return 0;
}

extern "C" int ttdg_roi_align_multilevel(ttdg_fpn_t fp, ttdg_levels_t lv, const float* rois, int R, int P, float canonical_size,
                                         int canonical_level, int min_level, float* out, ttdg_stream_t stream) {
  TTDG_REQUIRE(R >= 0 && (R == 0 || (rois && out)) && P > 0 && fp.n >= 1 && fp.n <= TTDG_MAX_LEVELS && fp.C > 0 && lv.n == fp.n,
               "roi_align_multilevel: bad arguments");
  if (R == 0) return 0;
  if (g_roi_align_mode == 2 && P <= RA_MAXP) {   // separable table kernel: one workgroup per (ROI, channel slice = XCD)
    const int nslice = fp.C % 8 == 0 ? 8 : 1;
    const size_t bytes = (size_t)RA_WAVES * (RA_MAXPATCH * (RA_MAXPATCH + 1) + RA_MAXPATCH * RA_MAXP) * sizeof(float);
    TTDG_ALLOW_LDS(roi_align_sep_kernel, bytes);
    TTDG_LIMIT((long)R * nslice < 2147483647L, "roi_align_multilevel: too many ROIs");
    hipLaunchKernelGGL(roi_align_sep_kernel, dim3(R * nslice), dim3(64 * RA_WAVES), bytes, (hipStream_t)stream, fp, lv, rois, R, P, nslice,
                       canonical_size, canonical_level, min_level, out);
    return ttdg_launch_status("roi_align_multilevel");
  }
  if (fp.C % 8 == 0 && g_roi_align_sliced) {      // XCD-sliced mapping: 8 x (blocks per slice)
    const long per = (long)R * (fp.C / 8) * P * P;
    const int bps = (int)((per + 255) / 256 < 16384 ? (per + 255) / 256 : 16384);
    hipLaunchKernelGGL(roi_align_ml_kernel<true>, dim3(8 * bps), dim3(256), 0, (hipStream_t)stream, fp, lv, rois, R, P, canonical_size,
                       canonical_level, min_level, out);
    return ttdg_launch_status("roi_align_multilevel");
  }
  const long total = (long)R * fp.C * P * P;
  const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(roi_align_ml_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, fp, lv, rois, R, P, canonical_size,
                     canonical_level, min_level, out);
  return ttdg_launch_status("roi_align_multilevel");
}
