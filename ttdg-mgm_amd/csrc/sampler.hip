// A2 — graph-node sampler (reference build_graph.py:27-250, PrototypeComputation).
//
// The reference builds (npoints x k) ltrb tensors per image, boolean-mask gathers over NHWC-permuted
// copies of every FPN level (p2 alone is a 41 MB copy per image at 800^2) and Python-level strided
// slicing.  Here: (1) one thread per FPN location assigns the minimum-area box that contains the point
// and cares about the level; (2) one workgroup per image walks the five levels in raster order with a
// wavefront-ballot prefix scan and keeps every `step`-th positive (step = count // sample_dist when > 1);
// (3) one wavefront per selected node gathers its 256 channels straight out of the NCHW map (and
// scatters the gradient back the same way).  HBM-bound, tiny: 4*d*M gathered bytes + the label pass.
#include "common.h"

#define SAMPLER_INF 100000000.0f

__global__ __launch_bounds__(256) void node_labels_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ classes,
                                                          const int32_t* __restrict__ nbox, int kmax, ttdg_levels_t lv,
                                                          int npts, int32_t* __restrict__ labels) {
  extern __shared__ float sb[];  // kmax x (x0,y0,x1,y1,area) + classes
  const int b = blockIdx.y;
  const int k = nbox[b];
  float* bx = sb;
  int* cl = (int*)(sb + 5 * kmax);
  for (int t = threadIdx.x; t < k; t += 256) {
    const float* p = boxes + ((size_t)b * kmax + t) * 4;
    bx[5 * t + 0] = p[0]; bx[5 * t + 1] = p[1]; bx[5 * t + 2] = p[2]; bx[5 * t + 3] = p[3];
    bx[5 * t + 4] = (p[2] - p[0] + 1.f) * (p[3] - p[1] + 1.f);     // build_graph.py:117-124
    cl[t] = classes[(size_t)b * kmax + t] + 1;                      // label = class + 1 (:85)
  }
  __syncthreads();
  const int gp = blockIdx.x * 256 + threadIdx.x;
  if (gp >= npts) return;
  int l = 0, base = 0;
  while (l + 1 < lv.n && gp >= base + lv.h[l] * lv.w[l]) { base += lv.h[l] * lv.w[l]; ++l; }
  const int p = gp - base;
  const int s = lv.stride[l];
  const float x = (float)((p % lv.w[l]) * s + s / 2), y = (float)((p / lv.w[l]) * s + s / 2);   // :144-157
  const float lo = lv.lo[l], hi = lv.hi[l];
  float best = SAMPLER_INF;
  int lab = 0;
  for (int t = 0; t < k; ++t) {
    const float le = x - bx[5 * t + 0], to = y - bx[5 * t + 1], ri = bx[5 * t + 2] - x, bo = bx[5 * t + 3] - y;
    const float mn = fminf(fminf(le, to), fminf(ri, bo)), mx = fmaxf(fmaxf(le, to), fmaxf(ri, bo));
    const bool ok = (mn > 0.f) && (mx >= lo) && (mx <= hi);          // :94-99
    const float a = ok ? bx[5 * t + 4] : SAMPLER_INF;
    if (a < best) { best = a; lab = cl[t]; }                         // min area, first index on ties (:107)
  }
  labels[(size_t)b * npts + gp] = (best == SAMPLER_INF) ? 0 : lab;   // :111
}

extern "C" int ttdg_node_labels(const float* boxes, const int32_t* classes, const int32_t* nbox, int B, int kmax,
                                ttdg_levels_t lv, int32_t* labels, ttdg_stream_t stream) {
  TTDG_REQUIRE(boxes && classes && nbox && labels && B > 0 && kmax > 0, "node_labels: bad arguments");
  TTDG_REQUIRE(lv.n >= 1 && lv.n <= TTDG_MAX_LEVELS, "node_labels: level count out of range");
  int npts = 0;
  for (int l = 0; l < lv.n; ++l) npts += lv.h[l] * lv.w[l];
  const size_t bytes = (size_t)kmax * 6 * sizeof(float);
  TTDG_LIMIT(bytes <= 48 * 1024, "node_labels: too many boxes per image");
  hipLaunchKernelGGL(node_labels_kernel, dim3((npts + 255) / 256, B), dim3(256), bytes, (hipStream_t)stream, boxes,
                     classes, nbox, kmax, lv, npts, labels);
  return ttdg_launch_status("node_labels");
}

// one workgroup (1024 threads) per image; two passes per level: count, then ordered strided selection
__global__ __launch_bounds__(1024) void node_select_kernel(const int32_t* __restrict__ labels, ttdg_levels_t lv, int npts,
                                                           int sample_dist, int cap, int32_t* __restrict__ sel_idx,
                                                           int32_t* __restrict__ sel_lab, int32_t* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int s_total;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t* lab = labels + (size_t)b * npts;
  int out = 0, base = 0;
  for (int l = 0; l < lv.n; ++l) {
    const int n = lv.h[l] * lv.w[l];
    // pass 1: number of positives
    int c = 0;
    for (int p = tid; p < n; p += 1024) c += (lab[base + p] > 0);
    c = (int)wave_sum((float)c);   // counts < 2^24: exact in fp32
    if (lane == 0) wsum[wave] = c;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; s_total = t; }
    __syncthreads();
    const int cnt = s_total;
    int step = cnt / sample_dist;                      // build_graph.py:189
    if (step <= 1) step = 1;                           // :190-195
    const int nsel = (cnt + step - 1) / step;
    // pass 2: raster-order rank via ballot prefix; keep rank % step == 0
    int run = 0;
    for (int p0 = 0; p0 < n; p0 += 1024) {
      const int p = p0 + tid;
      const int lv_ = (p < n) ? lab[base + p] : 0;
      const bool pos = lv_ > 0;
      const unsigned long long m = __ballot(pos);
      const int within = __popcll(m & ((1ull << lane) - 1ull));
      __syncthreads();
      if (lane == 0) wsum[wave] = __popcll(m);
      __syncthreads();
      int pre = 0, tot = 0;
      for (int w = 0; w < 16; ++w) { const int v = wsum[w]; if (w < wave) pre += v; tot += v; }
      if (pos) {
        const int rank = run + pre + within;
        if (rank % step == 0) {
          const int o = out + rank / step;
          if (o < cap) { sel_idx[(size_t)b * cap + o] = p | (l << 28); sel_lab[(size_t)b * cap + o] = lv_; }
        }
      }
      run += tot;
    }
    out += nsel;
    base += n;
    __syncthreads();
  }
  if (tid == 0) count[b] = out < cap ? out : cap;
}

extern "C" int ttdg_node_select(const int32_t* labels, int B, ttdg_levels_t lv, int sample_dist, int cap,
                                int32_t* sel_idx, int32_t* sel_lab, int32_t* count, ttdg_stream_t stream) {
  TTDG_REQUIRE(labels && sel_idx && sel_lab && count && B > 0 && sample_dist > 0, "node_select: bad arguments");
  TTDG_REQUIRE(lv.n >= 1 && lv.n <= TTDG_MAX_LEVELS, "node_select: level count out of range");
  TTDG_REQUIRE(cap >= lv.n * (2 * sample_dist - 1), "node_select: cap below the worst-case node count");
  int npts = 0;
  for (int l = 0; l < lv.n; ++l) {
    npts += lv.h[l] * lv.w[l];
    TTDG_LIMIT(lv.h[l] * lv.w[l] < (1 << 28), "node_select: level too large for the packed id");
  }
  hipLaunchKernelGGL(node_select_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, labels, lv, npts, sample_dist, cap,
                     sel_idx, sel_lab, count);
  return ttdg_launch_status("node_select");
}

// gather / scatter: one wavefront per node, lanes stride over channels
template <bool kBackward>
__global__ __launch_bounds__(256) void node_gather_kernel(ttdg_fpn_t fp, const int32_t* __restrict__ img,
                                                          const int32_t* __restrict__ pid, int n, float* __restrict__ rows) {
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (node >= n) return;
  const int id = pid[node], l = id >> 28, p = id & ((1 << 28) - 1);
  const size_t hw = (size_t)fp.h[l] * fp.w[l];
  float* f = fp.feat[l] + (size_t)img[node] * fp.C * hw + p;
  float* r = rows + (size_t)node * fp.C;
  for (int c = lane; c < fp.C; c += 64) {
    if (kBackward) atomicAdd(f + c * hw, r[c]);
    else r[c] = f[c * hw];
  }
}

// the same on channels-last maps (N, H, W, C): a node's C values are ONE contiguous row - a coalesced read / atomic row
template <bool kBackward>
__global__ __launch_bounds__(256) void node_gather_nhwc_kernel(ttdg_fpn_t fp, const int32_t* __restrict__ img,
                                                               const int32_t* __restrict__ pid, int n, float* __restrict__ rows) {
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (node >= n) return;
  const int id = pid[node], l = id >> 28, p = id & ((1 << 28) - 1);
  const size_t hw = (size_t)fp.h[l] * fp.w[l];
  float* f = fp.feat[l] + ((size_t)img[node] * hw + p) * fp.C;
  float* r = rows + (size_t)node * fp.C;
  for (int c = lane; c < fp.C; c += 64) {
    if (kBackward) atomicAdd(f + c, r[c]);
    else r[c] = f[c];
  }
}

static int check_fpn(const ttdg_fpn_t& fp) {
  if (fp.n < 1 || fp.n > TTDG_MAX_LEVELS || fp.C <= 0) return ttdg_fail(TTDG_EINVAL, "node_gather: bad pyramid descriptor");
  for (int l = 0; l < fp.n; ++l)
    if (!fp.feat[l]) return ttdg_fail(TTDG_EINVAL, "node_gather: null level pointer");
  return 0;
}

extern "C" int ttdg_node_gather_fwd(ttdg_fpn_t fp, const int32_t* img, const int32_t* pid, int n, float* out,
                                    ttdg_stream_t stream) {
  TTDG_REQUIRE(n >= 0 && (n == 0 || (img && pid && out)), "node_gather_fwd: bad arguments");      // n == 0 (no node selected in the whole batch): empty tensors carry null pointers
  if (int e = check_fpn(fp)) return e;
  if (n == 0) return 0;
  hipLaunchKernelGGL((node_gather_kernel<false>), dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, fp, img, pid, n, out);
  return ttdg_launch_status("node_gather_fwd");
}

extern "C" int ttdg_node_gather_bwd(ttdg_fpn_t dfp, const int32_t* img, const int32_t* pid, int n, const float* dout,
                                    ttdg_stream_t stream) {
  TTDG_REQUIRE(n >= 0 && (n == 0 || (img && pid && dout)), "node_gather_bwd: bad arguments");      // n == 0 (no node selected in the whole batch): empty tensors carry null pointers
  if (int e = check_fpn(dfp)) return e;
  if (n == 0) return 0;
  hipLaunchKernelGGL((node_gather_kernel<true>), dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, dfp, img, pid, n,
                     const_cast<float*>(dout));
  return ttdg_launch_status("node_gather_bwd");
}

extern "C" int ttdg_node_gather_fwd_nhwc(ttdg_fpn_t fp, const int32_t* img, const int32_t* pid, int n, float* out,
                                         ttdg_stream_t stream) {
  TTDG_REQUIRE(n >= 0 && (n == 0 || (img && pid && out)), "node_gather_fwd_nhwc: bad arguments");      // n == 0 (no node selected in the whole batch): empty tensors carry null pointers
  if (int e = check_fpn(fp)) return e;
  if (n == 0) return 0;
  hipLaunchKernelGGL((node_gather_nhwc_kernel<false>), dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, fp, img, pid, n, out);
  return ttdg_launch_status("node_gather_fwd_nhwc");
}

extern "C" int ttdg_node_gather_bwd_nhwc(ttdg_fpn_t dfp, const int32_t* img, const int32_t* pid, int n, const float* dout,
                                         ttdg_stream_t stream) {
  TTDG_REQUIRE(n >= 0 && (n == 0 || (img && pid && dout)), "node_gather_bwd_nhwc: bad arguments");      // n == 0 (no node selected in the whole batch): empty tensors carry null pointers
  if (int e = check_fpn(dfp)) return e;
  if (n == 0) return 0;
  hipLaunchKernelGGL((node_gather_nhwc_kernel<true>), dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, dfp, img, pid, n,
                     const_cast<float*>(dout));
  return ttdg_launch_status("node_gather_bwd_nhwc");
}
