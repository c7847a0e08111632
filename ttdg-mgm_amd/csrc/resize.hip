// Test-mapper resize on the device (reference data path: DatasetMapper(is_train=False) -> ResizeShortestEdge [3P], driven by
// data/build.py:122-154 and iterated at engine/trainer.py:470,485).  Round 2 resized on the HOST (F.interpolate on 3 x 512 x 512
// floats per image) and uploaded the 800 x 800 result: 7.7 MB of PCIe per batch and ~10 ms of single-threaded host work per
// image inside the loader.  Here the raw uint8 image is uploaded (0.8 MB per 512 x 512 image) and resized by the GPU.
//
// The arithmetic is the host mapper's (ttdg_mgm_amd.data.map_for_test = torch F.interpolate on float32, bilinear,
// align_corners = False, antialias when shrinking; then round-half-even, clamp, uint8):
//   enlarging / same size:  src = scale * (dst + 0.5) - 0.5 (clamped at 0), i0 = floor, i1 = min(i0 + 1, in - 1), l = src - i0;
//                           v = (1-ly) ((1-lx) p00 + lx p01) + ly ((1-lx) p10 + lx p11),  scale = in / out in float
//   shrinking (antialias):  separable triangle filter of support `scale` per axis: center = scale (i + 0.5),
//                           taps [max(int(center - support + 0.5), 0), min(int(center + support + 0.5), in)), weights
//                           1 - |(j - center + 0.5) / scale| normalised to sum 1; horizontal pass into float, then vertical.
// Held to <= 1 LSB of the host result by tests/test_gpu_parity.py (the summation order of the taps differs).
#include "common.h"

__device__ __forceinline__ unsigned char rz_to_u8(float v) {
  v = rintf(v);                         // round half to even, as torch.round
  v = fminf(fmaxf(v, 0.f), 255.f);
  return (unsigned char)v;
}

__global__ __launch_bounds__(256) void resize_bilinear_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                                 int planes, int H, int W, int OH, int OW, float sy, float sx) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox >= OW || oy >= OH) return;
  float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
  for (int p = blockIdx.z; p < planes; p += gridDim.z) {
    const unsigned char* s = src + (size_t)p * H * W;
    const float p00 = s[(size_t)y0 * W + x0], p01 = s[(size_t)y0 * W + x1], p10 = s[(size_t)y1 * W + x0], p11 = s[(size_t)y1 * W + x1];
    const float v = hy * (hx * p00 + lx * p01) + ly * (hx * p10 + lx * p11);
    dst[((size_t)p * OH + oy) * OW + ox] = rz_to_u8(v);
  }
}

// one axis of the antialiased filter: out[.., o, ..] = sum_j w_j in[.., xmin + j, ..]
// kVertical = false: rows of a (planes*H, W) uint8 image -> float (planes*H, OW);  true: float (planes, H, OW) -> uint8 (planes, OH, OW)
template <bool kVertical>
__global__ __launch_bounds__(256) void resize_aa_axis_kernel(const void* __restrict__ srcv, void* __restrict__ dstv, int planes, int H, int W,
                                                             int OH, int OW, float scale) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int row = blockIdx.y * 4 + (threadIdx.x >> 6);          // horizontal: input row (planes*H); vertical: output row (planes*OH)
  const int in_size = kVertical ? H : W;
  if (ox >= OW) return;
  if (row >= (kVertical ? planes * OH : planes * H)) return;
  const int o = kVertical ? row % OH : ox;
  const float support = scale >= 1.f ? scale : 1.f, invscale = scale >= 1.f ? 1.f / scale : 1.f;
  const float center = scale * ((float)o + 0.5f);
  int xmin = (int)(center - support + 0.5f);
  xmin = xmin < 0 ? 0 : xmin;
  int xend = (int)(center + support + 0.5f);
  xend = xend > in_size ? in_size : xend;
  float total = 0.f;
  for (int j = xmin; j < xend; ++j) {
    float w = fabsf(((float)(j) - center + 0.5f) * invscale);
    total += w < 1.f ? 1.f - w : 0.f;
  }
  float acc = 0.f;
  if (kVertical) {
    const float* s = (const float*)srcv + (size_t)(row / OH) * H * OW + ox;
    for (int j = xmin; j < xend; ++j) {
      float w = fabsf(((float)(j) - center + 0.5f) * invscale);
      w = w < 1.f ? 1.f - w : 0.f;
      acc += (w / total) * s[(size_t)j * OW];
    }
    ((unsigned char*)dstv)[(size_t)row * OW + ox] = rz_to_u8(acc);
  } else {
    const unsigned char* s = (const unsigned char*)srcv + (size_t)row * W;
    for (int j = xmin; j < xend; ++j) {
      float w = fabsf(((float)(j) - center + 0.5f) * invscale);
      w = w < 1.f ? 1.f - w : 0.f;
      acc += (w / total) * (float)s[j];
    }
    ((float*)dstv)[(size_t)row * OW + ox] = acc;
  }
}

extern "C" size_t ttdg_resize_u8_workspace_bytes(int planes, int H, int W, int OH, int OW) {
  return (OH < H || OW < W) ? (size_t)planes * H * OW * sizeof(float) : 0;
}

extern "C" int ttdg_resize_bilinear_u8(const unsigned char* src, unsigned char* dst, int planes, int H, int W, int OH, int OW, void* ws,
                                       ttdg_stream_t stream) {
  TTDG_REQUIRE(src && dst && planes >= 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "resize_u8: bad arguments");
  if (planes == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
  if (OH < H || OW < W) {
    // shrinking on either axis: the host mapper switches antialiasing on for BOTH axes (an enlarged axis then degenerates to
    // plain two-tap weights: support 1)
    TTDG_REQUIRE(ws, "resize_u8: shrinking needs the workspace of ttdg_resize_u8_workspace_bytes");
    hipLaunchKernelGGL((resize_aa_axis_kernel<false>), dim3((OW + 63) / 64, (planes * H + 3) / 4), dim3(256), 0, st, (const void*)src, ws, planes,
                       H, W, OH, OW, sx);
    hipLaunchKernelGGL((resize_aa_axis_kernel<true>), dim3((OW + 63) / 64, (planes * OH + 3) / 4), dim3(256), 0, st, (const void*)ws, (void*)dst,
                       planes, H, W, OH, OW, sy);
    return ttdg_launch_status("resize_aa_u8");
  }
  const int gz = planes < 64 ? planes : 64;
  hipLaunchKernelGGL(resize_bilinear_u8_kernel, dim3((OW + 63) / 64, (OH + 3) / 4, gz), dim3(256), 0, st, src, dst, planes, H, W, OH, OW, sy, sx);
  return ttdg_launch_status("resize_bilinear_u8");
}
