/* oracle/detection.c — CPU restatements of ROIAlign (aligned, adaptive sampling; detectron2 ROIAlignV2 semantics
 * [3P, absent]) and greedy NMS (torchvision.ops.nms semantics [3P, absent]).  TEST INFRASTRUCTURE (see
 * oracle/__init__.py): checks csrc/detection.hip and serves the CPU baseline of bench.py.  OpenMP over ROIs. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void ttdg_oracle_roi_align(const float* feat, int B, int C, int H, int W, const float* rois, int R, float scale, int P,
                           float* out) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int r = 0; r < R; ++r) {
    const float* roi = rois + (size_t)r * 5;
    const int b = (int)roi[0];
    const float x1 = roi[1] * scale - 0.5f, y1 = roi[2] * scale - 0.5f;
    const float rw = roi[3] * scale - 0.5f - x1, rh = roi[4] * scale - 0.5f - y1;
    const float bw = rw / P, bh = rh / P;
    int gh = (int)ceilf(rh / P), gw = (int)ceilf(rw / P);
    if (gh < 1) gh = 1;
    if (gw < 1) gw = 1;
    for (int c = 0; c < C; ++c) {
      const float* f = feat + ((size_t)b * C + c) * H * W;
      for (int ph = 0; ph < P; ++ph)
        for (int pw = 0; pw < P; ++pw) {
          float acc = 0.f;
          for (int iy = 0; iy < gh; ++iy) {
            const float y = y1 + ph * bh + (iy + 0.5f) * bh / gh;
            for (int ix = 0; ix < gw; ++ix) {
              const float x = x1 + pw * bw + (ix + 0.5f) * bw / gw;
              if (y < -1.f || y > H || x < -1.f || x > W) continue;
              float yy = y < 0.f ? 0.f : y, xx = x < 0.f ? 0.f : x;
              int y0 = (int)yy, x0 = (int)xx, y1i, x1i;
              if (y0 >= H - 1) { y0 = y1i = H - 1; yy = (float)y0; } else y1i = y0 + 1;
              if (x0 >= W - 1) { x0 = x1i = W - 1; xx = (float)x0; } else x1i = x0 + 1;
              const float ly = yy - y0, lx = xx - x0, hy = 1.f - ly, hx = 1.f - lx;
              acc += hy * hx * f[y0 * W + x0] + hy * lx * f[y0 * W + x1i] + ly * hx * f[y1i * W + x0] + ly * lx * f[y1i * W + x1i];
            }
          }
          out[(((size_t)r * C + c) * P + ph) * P + pw] = acc / (float)(gh * gw);
        }
    }
  }
}

/* boxes (N,4) sorted by descending score; returns the number of kept boxes, their indices in keep */
int ttdg_oracle_nms(const float* boxes, const int32_t* group, int N, float thr, int32_t* keep) {
  char* removed = calloc(N > 0 ? N : 1, 1);
  int cnt = 0;
  for (int i = 0; i < N; ++i) {
    if (removed[i]) continue;
    keep[cnt++] = i;
    const float* a = boxes + (size_t)i * 4;
    for (int j = i + 1; j < N; ++j) {
      if (removed[j] || group[j] != group[i]) continue;
      const float* b = boxes + (size_t)j * 4;
      const float iw = fminf(a[2], b[2]) - fmaxf(a[0], b[0]), ih = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
      const float inter = (iw > 0.f ? iw : 0.f) * (ih > 0.f ? ih : 0.f);
      const float ua = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter;
      if (inter > thr * ua) removed[j] = 1;
    }
  }
  free(removed);
  return cnt;
}
