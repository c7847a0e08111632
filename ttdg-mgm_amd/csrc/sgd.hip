// A11 — fused multi-tensor SGD step (momentum, weight decay), one launch for every parameter tensor.
// Reference: optimizer.step() at engine/trainer.py:482 on the torch.optim.SGD that detectron2's
// build_optimizer [3P] creates (train_net.py:65): one param-group per tensor, momentum .9, wd 1e-4
// (0 for norm layers), no nesterov/dampening; ~65 tensors / ~27 M values carry gradients on the TTA
// path, which stock PyTorch updates with several small kernels per tensor.
//   d = g + wd*p;  buf = first ? d : momentum*buf + d;  p -= lr*buf
// HBM-bound: 12 B read + 8 B written per parameter (20 B/param).  Chunks of `chunk`
// elements map thread blocks to tensors; 16-byte vector accesses in the body, scalar tail.
#include "common.h"

__global__ __launch_bounds__(256) void sgd_multi_tensor_kernel(const ttdg_sgd_tensor_t* __restrict__ table,
                                                               const int32_t* __restrict__ chunk_tensor,
                                                               const int64_t* __restrict__ chunk_off, int chunk, float lr,
                                                               float momentum) {
  const ttdg_sgd_tensor_t t = table[chunk_tensor[blockIdx.x]];
  const int64_t off = chunk_off[blockIdx.x];
  const int64_t rem = t.n - off;
  const int n = rem < chunk ? (int)rem : chunk;
  float* p = t.p + off;
  const float* g = t.g + off;
  float* b = t.buf + off;
  const float wd = t.wd;
  const bool first = t.first != 0;
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)b)) & 15) == 0;
  const int nv = vec ? (n >> 2) : 0;
  for (int i = threadIdx.x; i < nv; i += 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 bv;
    const float4 d = make_float4(fmaf(wd, pv.x, gv.x), fmaf(wd, pv.y, gv.y), fmaf(wd, pv.z, gv.z), fmaf(wd, pv.w, gv.w));
    if (first) bv = d;
    else {
      bv = reinterpret_cast<float4*>(b)[i];
      bv = make_float4(fmaf(momentum, bv.x, d.x), fmaf(momentum, bv.y, d.y), fmaf(momentum, bv.z, d.z), fmaf(momentum, bv.w, d.w));
    }
    pv = make_float4(fmaf(-lr, bv.x, pv.x), fmaf(-lr, bv.y, pv.y), fmaf(-lr, bv.z, pv.z), fmaf(-lr, bv.w, pv.w));
    reinterpret_cast<float4*>(b)[i] = bv;
    reinterpret_cast<float4*>(p)[i] = pv;
  }
  for (int i = nv * 4 + threadIdx.x; i < n; i += 256) {
    const float pv = p[i];
    const float d = fmaf(wd, pv, g[i]);
    const float bv = first ? d : fmaf(momentum, b[i], d);
    b[i] = bv;
    p[i] = fmaf(-lr, bv, pv);
  }
}

extern "C" int ttdg_sgd_multi_tensor(const ttdg_sgd_tensor_t* table, const int32_t* chunk_tensor, const int64_t* chunk_off,
                                     int nchunks, int chunk, float lr, float momentum, ttdg_stream_t stream) {
  TTDG_REQUIRE(table && chunk_tensor && chunk_off && nchunks >= 0 && chunk > 0 && chunk % 4 == 0, "sgd: bad arguments");
  if (nchunks == 0) return 0;
  hipLaunchKernelGGL(sgd_multi_tensor_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, chunk_tensor,
                     chunk_off, chunk, lr, momentum);
  return ttdg_launch_status("sgd_multi_tensor");
}
