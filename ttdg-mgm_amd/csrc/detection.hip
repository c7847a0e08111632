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
  TTDG_REQUIRE(feat && rois && out && R >= 0 && C > 0 && P > 0, "roi_align: bad arguments");
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
__global__ __launch_bounds__(64) void nms_group_mask_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ seg,
                                                            int words, float thr, unsigned long long* __restrict__ mask) {
  const int g = blockIdx.z;
  const int s0 = seg[g], n = seg[g + 1] - s0;
  const int i = blockIdx.x, w = blockIdx.y, lane = threadIdx.x;
  if (i >= n || w * 64 >= n) return;
  const int j = w * 64 + lane;
  bool hit = false;
  if (j < n && j > i) {
    const float4 a = reinterpret_cast<const float4*>(boxes)[s0 + i], b = reinterpret_cast<const float4*>(boxes)[s0 + j];
    const float iw = fminf(a.z, b.z) - fmaxf(a.x, b.x), ih = fminf(a.w, b.w) - fmaxf(a.y, b.y);
    const float inter = fmaxf(iw, 0.f) * fmaxf(ih, 0.f);
    const float ua = (a.z - a.x) * (a.w - a.y) + (b.z - b.x) * (b.w - b.y) - inter;
    hit = inter > thr * ua;
  }
  const unsigned long long m = __ballot(hit);
  if (lane == 0) mask[(size_t)(s0 + i) * words + w] = m;
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
  hipLaunchKernelGGL(nms_group_mask_kernel, dim3(max_group, words, ngroups), dim3(64), 0, st, boxes, seg, words, thr,
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
  reinterpret_cast<float4*>(boxes)[oi] = o;
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
  TTDG_REQUIRE(logits && deltas && rois && sizes && boxes && scores && N >= 0 && C >= 1, "box_inference: bad arguments");
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
  TTDG_REQUIRE(masks && boxes && out && R >= 0 && S > 0 && H > 0 && W > 0, "paste_masks: bad arguments");
  if (R == 0) return 0;
  const int bx = (H * W + 255) / 256 < 64 ? (H * W + 255) / 256 : 64;
  hipLaunchKernelGGL(paste_masks_kernel, dim3(bx, R), dim3(256), 0, (hipStream_t)stream, masks, boxes, R, S, H, W, threshold, out);
  return ttdg_launch_status("paste_masks");
}

// ---------------------------------------------------------------------------------------------------
// y <- act(y + bias[c] (+ residual) (+ bias2[c])) in place on an NCHW activation: the per-channel shift of a folded
// FrozenBN (or a conv bias), the residual add and the ReLU of a bottleneck in ONE pass over the tensor instead of three
// or four (each pass over a 164 MB res2 activation costs ~65 us of HBM time).  Forward-only: used where no gradient
// flows (frozen stem / res2, the eval pass, the RPN head).
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                       const float* __restrict__ res, const float* __restrict__ bias2, int C, int HW,
                                                       size_t total, int relu) {
  if ((HW & 3) == 0) {
    const size_t nvec = total >> 2;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) {
      const int c = (int)((v * 4 / HW) % C);
      float b = bias ? bias[c] : 0.f;
      if (bias2) b += bias2[c];
      float4 t = reinterpret_cast<float4*>(y)[v];
      if (res) {
        const float4 r = reinterpret_cast<const float4*>(res)[v];
        t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
      }
      t.x += b; t.y += b; t.z += b; t.w += b;
      if (relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
      reinterpret_cast<float4*>(y)[v] = t;
    }
  } else {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
      const int c = (int)((e / HW) % C);
      float t = y[e] + (bias ? bias[c] : 0.f) + (bias2 ? bias2[c] : 0.f);
      if (res) t += res[e];
      y[e] = relu ? fmaxf(t, 0.f) : t;
    }
  }
}

extern "C" int ttdg_bias_act(float* y, const float* bias, const float* residual, const float* bias2, int N, int C, int HW,
                             int relu, ttdg_stream_t stream) {
  TTDG_REQUIRE(y && N >= 0 && C > 0 && HW > 0, "bias_act: bad arguments");
  const size_t total = (size_t)N * C * HW;
  if (total == 0) return 0;
  const size_t work = (HW & 3) == 0 ? total / 4 : total;
  const size_t want = (work + 255) / 256;
  const int blocks = (int)(want < 8192 ? want : 8192);
  hipLaunchKernelGGL(bias_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, bias, residual, bias2, C, HW, total, relu);
  return ttdg_launch_status("bias_act");
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

static int g_roi_align_sliced = 1;
extern "C" int ttdg_debug_set_roi_align_sliced(int on) { g_roi_align_sliced = on != 0; return 0; }

extern "C" int ttdg_roi_align_multilevel(ttdg_fpn_t fp, ttdg_levels_t lv, const float* rois, int R, int P, float canonical_size,
                                         int canonical_level, int min_level, float* out, ttdg_stream_t stream) {
  TTDG_REQUIRE(rois && out && R >= 0 && P > 0 && fp.n >= 1 && fp.n <= TTDG_MAX_LEVELS && fp.C > 0 && lv.n == fp.n,
               "roi_align_multilevel: bad arguments");
  if (R == 0) return 0;
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
