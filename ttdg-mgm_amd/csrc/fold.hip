// A0 glue — FrozenBN scale folded into MANY convolution filters in one launch.
// Reference: detectron2's FrozenBatchNorm2d [3P] behind every ResNet convolution (rcnn.py:219 -> backbone); the frozen affine
// map y = conv(x, w) * scale[c] + shift[c] is evaluated here as conv(x, w * scale[c]) + shift[c].  The adapted filters move at
// every TTA step, so the fold is recomputed per step: with stock PyTorch that is one 3-microsecond elementwise kernel per
// filter in the forward pass, one in the backward pass (grad_w = grad_wf * scale) and one more in the Dice pass - 137 launches
// per adapted batch for 94 MB of filters.  One launch per stage instead.
//   out_t[r][j] = in_t[r][j] * scale_t[r]        for every tensor t of the group (row r = output channel)
// HBM-bound streaming (8 B per filter value); the tensor table travels by value in the kernel arguments.
#include "common.h"

#define RS_CHUNK 8192            // elements per workgroup: 256 threads x 8 float4

struct RsGroup {
  ttdg_row_scale_t t[TTDG_ROW_SCALE_MAX];
  int chunk_end[TTDG_ROW_SCALE_MAX];    // exclusive prefix of workgroups per tensor
  int n;
};

__global__ __launch_bounds__(256) void row_scale_multi_kernel(RsGroup g) {
  int ti = 0;
  while (ti + 1 < g.n && (int)blockIdx.x >= g.chunk_end[ti]) ++ti;
  const ttdg_row_scale_t t = g.t[ti];
  const int64_t off = (int64_t)((int)blockIdx.x - (ti ? g.chunk_end[ti - 1] : 0)) * RS_CHUNK;
  const int64_t total = (int64_t)t.rows * t.rowlen;
  const int64_t rem = total - off;
  const int n = rem < RS_CHUNK ? (int)rem : RS_CHUNK;
  const float* in = t.in + off;
  float* out = t.out + off;
  const bool vec = (t.rowlen & 3) == 0 && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0;
  if (vec) {
    const int rl4 = t.rowlen >> 2;
    const int64_t e0 = off >> 2;
    for (int i = threadIdx.x; i < (n >> 2); i += 256) {
      const float s = t.scale[(e0 + i) / rl4];
      float4 v = reinterpret_cast<const float4*>(in)[i];
      v.x *= s; v.y *= s; v.z *= s; v.w *= s;
      reinterpret_cast<float4*>(out)[i] = v;
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) out[i] = in[i] * t.scale[(off + i) / t.rowlen];
  }
}

extern "C" int ttdg_row_scale_multi(const ttdg_row_scale_t* items, int n, ttdg_stream_t stream) {
  TTDG_REQUIRE(items && n >= 1 && n <= TTDG_ROW_SCALE_MAX, "row_scale_multi: 1..64 tensors per launch");
  RsGroup g;
  g.n = 0;
  int chunks = 0;
  for (int i = 0; i < n; ++i) {
    const ttdg_row_scale_t& t = items[i];
    TTDG_REQUIRE(t.rows >= 0 && t.rowlen >= 0, "row_scale_multi: negative size");
    const int64_t total = (int64_t)t.rows * t.rowlen;
    if (total == 0) continue;
    TTDG_REQUIRE(t.in && t.scale && t.out, "row_scale_multi: null tensor");
    chunks += (int)((total + RS_CHUNK - 1) / RS_CHUNK);
    g.t[g.n] = t;
    g.chunk_end[g.n] = chunks;
    ++g.n;
  }
  if (g.n == 0) return 0;
  hipLaunchKernelGGL(row_scale_multi_kernel, dim3(chunks), dim3(256), 0, (hipStream_t)stream, g);
  return ttdg_launch_status("row_scale_multi");
}
