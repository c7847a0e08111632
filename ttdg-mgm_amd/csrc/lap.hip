// A7 — batched LAP, the operator behind GModule.utils.hungarian.hungarian (utils/hungarian.py:8-66).
// One wavefront (one 64-thread workgroup) per matrix; algorithm in lap_device.h.
#include "lap_device.h"

static int g_lap_variant = 0;   // 0 = default (compiler-lowered fp64 DPP min, cost column in registers: 14-17 % faster than reading the costs
                                // per step, tools/bench_lap.py), 1 = hand-scheduled inline-asm min, 2 = costs read from memory per step (A/B only)
extern "C" int ttdg_debug_set_lap_variant(int v) { g_lap_variant = v; return 0; }

__global__ __launch_bounds__(64) void lap_batched_kernel(const float* __restrict__ s, int R, int C, float* __restrict__ x, int variant) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lap_smem[];
  const float* m = s + (size_t)blockIdx.x * R * C;
  float* o = x + (size_t)blockIdx.x * R * C;
  const int lane = threadIdx.x;
  for (int e = lane; e < R * C; e += 64) o[e] = 0.f;
  const bool tr = C < R;  // tall matrices are solved transposed (scipy does the same)
  const int nr = tr ? C : R, nc = tr ? R : C;
  if (nc <= 64) {   // register-resident solver
    int j;
    if (variant == 0 && nr <= 32) j = lap_wave_solve_reg<0, true>(nr, nc, m, tr ? 1 : C, tr ? C : 1);     // default: cost column in registers
    else if (variant == 1) j = lap_wave_solve_reg<1>(nr, nc, m, tr ? 1 : C, tr ? C : 1);
    else j = lap_wave_solve_reg<0>(nr, nc, m, tr ? 1 : C, tr ? C : 1);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane < nr) { if (tr) o[(size_t)j * C + lane] = 1.f; else o[(size_t)lane * C + j] = 1.f; }
    return;
  }
  if (nc <= 256 && nr <= 64) {   // wide register-resident solver: 2 or 4 columns per lane
    const int j = nc <= 128 ? lap_wave_solve_regw<2>(nr, nc, m, tr ? 1 : C, tr ? C : 1) : lap_wave_solve_regw<4>(nr, nc, m, tr ? 1 : C, tr ? C : 1);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane < nr) { if (tr) o[(size_t)j * C + lane] = 1.f; else o[(size_t)lane * C + j] = 1.f; }
    return;
  }
  LapScratch sc = lap_carve(lap_smem, nr, nc);
  lap_wave_solve(nr, nc, m, tr ? 1 : C, tr ? C : 1, sc);
  wave_sync();
  for (int i = lane; i < nr; i += 64) {
    const int j = sc.col4row[i];
    if (tr) o[(size_t)j * C + i] = 1.f; else o[(size_t)i * C + j] = 1.f;
  }
}

extern "C" int ttdg_lap_batched(const float* s, int b, int r, int c, float* x, ttdg_stream_t stream) {
  TTDG_REQUIRE(s && x && b >= 0 && r > 0 && c > 0, "lap_batched: bad arguments");
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  TTDG_LIMIT(lap_scratch_bytes(lo, hi) <= 60 * 1024, "lap_batched: matrix too large for one wavefront's LDS scratch");
  if (b == 0) return 0;
  const size_t bytes = lap_scratch_bytes(lo, hi);
  TTDG_ALLOW_LDS((lap_batched_kernel), bytes);
  hipLaunchKernelGGL(lap_batched_kernel, dim3(b), dim3(64), bytes, (hipStream_t)stream, s, r, c, x, g_lap_variant);
  return ttdg_launch_status("lap_batched");
}
