// Shared device/host helpers for libttdg_mgm.so (gfx950 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ttdg_mgm.h"

#define TTDG_WAVE 64

extern thread_local char g_ttdg_err[512];

static inline int ttdg_fail(int code, const char* what) {
  snprintf(g_ttdg_err, sizeof(g_ttdg_err), "%s", what);
  return code;
}

#define TTDG_REQUIRE(cond, msg)                       \
  do {                                                \
    if (!(cond)) return ttdg_fail(TTDG_EINVAL, msg);  \
  } while (0)

#define TTDG_LIMIT(cond, msg)                         \
  do {                                                \
    if (!(cond)) return ttdg_fail(TTDG_ELIMIT, msg);  \
  } while (0)

#define TTDG_HIP(expr)                                                            \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      snprintf(g_ttdg_err, sizeof(g_ttdg_err), "%s: %s", #expr, hipGetErrorString(e_)); \
      return (int)e_;                                                             \
    }                                                                             \
  } while (0)

// opt a kernel into more than 48 KiB of dynamic LDS (gfx950 has 160 KiB per CU)
#define TTDG_ALLOW_LDS(kernel, bytes)                                                                     \
  do {                                                                                                    \
    if ((bytes) > 48 * 1024)                                                                              \
      TTDG_HIP(hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
  } while (0)

// call after every launch: reports launch-configuration errors without synchronising
static inline int ttdg_launch_status(const char* kernel) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_ttdg_err, sizeof(g_ttdg_err), "%s: %s", kernel, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

static inline int ttdg_validate_graphs(const ttdg_graphs_t& gr) {
  if (gr.G < 1 || gr.G > TTDG_MAX_GRAPHS) return ttdg_fail(TTDG_EINVAL, "graphs: G out of range");
  if (gr.off[0] != 0) return ttdg_fail(TTDG_EINVAL, "graphs: off[0] != 0");
  for (int g = 0; g < gr.G; ++g)
    if (gr.off[g + 1] <= gr.off[g]) return ttdg_fail(TTDG_EINVAL, "graphs: empty or unordered graph");
  return 0;
}

#ifdef __HIPCC__
__device__ __forceinline__ int graph_of(const ttdg_graphs_t& gr, int i) {
  int g = 0;
  while (g + 1 < gr.G && i >= gr.off[g + 1]) ++g;
  return g;
}

// ---- wavefront reductions (all 64 lanes participate, result in every lane) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

// base-2 exp/log on the transcendental unit (v_exp_f32 / v_log_f32, ~1 ulp)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
#define TTDG_LOG2E 1.4426950408889634f
#define TTDG_LN2 0.6931471805599453f
#endif
