// Dense fp32 GEMM on the matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).
// C[m,n] = alpha * sum_k A(m,k) B(n,k) + bias[n] + beta * C[m,n], arbitrary element strides so the
// same kernel serves x W^T (nn.Linear fwd), dY W (input grad) and dY^T X (weight grad).
//
// Workgroup = 256 threads = 4 wavefronts in a 2x2 arrangement over a 64x64 output tile; each
// wavefront owns one 32x32 accumulator (16 fp32 per lane).  K is consumed in BK=32 slabs staged
// through LDS in [k][m] order, so a wavefront's A/B fragment read (lane l -> m = l&31, k = l>>5)
// touches 32 consecutive words per half-wave: conflict-free ds_read_b32, two per 64-cycle MFMA.
//
// [r4] Software pipeline.  The matrices on this path give ONE workgroup per CU (2048 x 512 output = 256 tiles; everything
// smaller is less), i.e. one wavefront per SIMD and nobody to hide a load behind: round 3's loop (load -> LDS -> barrier -> 8
// MFMAs, 16-byte-per-row scalar loads) spent 1.2 us per slab on 0.2 us of MFMA work - 15 % of the fp32 matrix peak at
// 2048 x 512 x 256.  Now the next slab travels global -> REGISTERS (16-byte loads: 64 B per row per wave instruction for a
// k-contiguous operand, 256 B for an m-contiguous one) while the current slab's first 8 MFMAs run, is written to the other
// LDS buffer behind them (transposing ds_write_b32, <= 2-way = free; ds_write_b128 for m-contiguous operands), and the last
// 8 MFMAs cover the stores: per slab one barrier and one LDS round trip are exposed, not a global round trip.  The k order
// of every accumulator is unchanged: results are bit-identical to round 3's kernel.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 64
#define BN 64
#define BK 32
#define LDS_LD (BM + 4)  // +4 words: rows stay 16-byte aligned (ds_write_b128) and transposing stores are <= 2-way

// One operand slab (64 x BK) on its way global -> registers -> LDS T[k][m].  `kcontig` selects the thread -> element map that
// keeps the global reads wide along whichever index has unit stride; `vec` = 16-byte loads are legal (base pointer 16-byte
// aligned, the other stride a multiple of 4): otherwise - and on the ragged edge - guarded scalar loads, zero-filled.
template <bool kcontig, bool vec>
struct Slab {
  float4 r[2];
  __device__ __forceinline__ void load(const float* __restrict__ X, int64_t sm, int64_t sk, int m0, int k0, int Mlim, int Klim,
                                       int tid) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int gm, gk;
      if (kcontig) {  // lane: row (tid & 15) + 16 * wave, 16-byte chunk ((tid >> 4) & 3) + 4 p of the row's 128-byte slab
        gm = m0 + (tid & 15) + 16 * (tid >> 6);
        gk = k0 + 4 * (((tid >> 4) & 3) + 4 * p);
      } else {        // lane: k row (tid >> 4) + 16 p, four consecutive m
        gm = m0 + 4 * (tid & 15);
        gk = k0 + (tid >> 4) + 16 * p;
      }
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kcontig) {
        if (gm < Mlim) {
          const float* src = X + gm * sm + gk;
          if (vec && gk + 3 < Klim) v = *reinterpret_cast<const float4*>(src);
          else {
            if (gk < Klim) v.x = src[0];
            if (gk + 1 < Klim) v.y = src[1];
            if (gk + 2 < Klim) v.z = src[2];
            if (gk + 3 < Klim) v.w = src[3];
          }
        }
      } else {
        if (gk < Klim) {
          const float* src = X + gk * sk + gm;
          if (vec && gm + 3 < Mlim) v = *reinterpret_cast<const float4*>(src);
          else {
            if (gm < Mlim) v.x = src[0];
            if (gm + 1 < Mlim) v.y = src[1];
            if (gm + 2 < Mlim) v.z = src[2];
            if (gm + 3 < Mlim) v.w = src[3];
          }
        }
      }
      r[p] = v;
    }
  }
  __device__ __forceinline__ void store(float (*T)[LDS_LD], int tid) const {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (kcontig) {
        const int m = (tid & 15) + 16 * (tid >> 6), k = 4 * (((tid >> 4) & 3) + 4 * p);
        T[k][m] = r[p].x, T[k + 1][m] = r[p].y, T[k + 2][m] = r[p].z, T[k + 3][m] = r[p].w;
      } else {
        *reinterpret_cast<float4*>(&T[(tid >> 4) + 16 * p][4 * (tid & 15)]) = r[p];
      }
    }
  }
};

// general strides (neither index has unit stride): element-wise, as round 3 staged everything
template <bool vec>
struct SlabAny {
  float r[8];
  __device__ __forceinline__ void load(const float* __restrict__ X, int64_t sm, int64_t sk, int m0, int k0, int Mlim, int Klim, int tid) {
    const int m = tid & 63, kb = tid >> 6;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int gm = m0 + m, gk = k0 + kb + 4 * q;
      r[q] = (gm < Mlim && gk < Klim) ? X[gm * sm + gk * sk] : 0.f;
    }
  }
  __device__ __forceinline__ void store(float (*T)[LDS_LD], int tid) const {
    const int m = tid & 63, kb = tid >> 6;
#pragma unroll
    for (int q = 0; q < 8; ++q) T[kb + 4 * q][m] = r[q];
  }
};

// operand layout codes: 0 = k has unit stride, 1 = m (n) has unit stride, 2 = neither
template <int layout, bool vec> struct SlabOf { typedef Slab<layout == 0, vec> type; };
template <bool vec> struct SlabOf<2, vec> { typedef SlabAny<vec> type; };

template <int la, int lb, bool vec>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t sam, int64_t sak,
                                                       const float* __restrict__ B, int64_t sbn, int64_t sbk,
                                                       float* __restrict__ C, int64_t scm, int64_t scn,
                                                       const float* __restrict__ bias, int M, int N, int K, float alpha,
                                                       float beta, int Kc, float* __restrict__ part) {
  // split-K (part != nullptr): blockIdx.z owns k in [z*Kc, (z+1)*Kc) and writes alpha * partial into its own M x N plane
  // (deterministic; gemm_splitk_reduce_kernel adds the planes, the bias and beta * C)
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDS_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  const int kbeg = part ? blockIdx.z * Kc : 0;
  const int kend = part ? min(K, kbeg + Kc) : K;
  const int nk = (kend - kbeg + BK - 1) / BK;
  typename SlabOf<la, vec>::type sa;
  typename SlabOf<lb, vec>::type sb;
  if (nk > 0) {
    sa.load(A, sam, sak, m0, kbeg, M, kend, tid);
    sb.load(B, sbn, sbk, n0, kbeg, N, kend, tid);
    sa.store(As[0], tid);
    sb.store(Bs[0], tid);
  }
  __syncthreads();
  const int kh = lane >> 5, mi = lane & 31;
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const bool more = t + 1 < nk;
    if (more) {  // the next slab starts its trip to the registers now and is needed only after the first 8 MFMAs
      sa.load(A, sam, sak, m0, kbeg + (t + 1) * BK, M, kend, tid);
      sb.load(B, sbn, sbk, n0, kbeg + (t + 1) * BK, N, kend, tid);
    }
    // the slab's 16 + 16 fragment words are requested up front (lgkmcnt retires in order: MFMA j waits for read j only),
    // so one LDS latency is exposed per slab instead of one per MFMA pair
    float fa[BK / 2], fb[BK / 2];
#pragma unroll
    for (int j = 0; j < BK / 2; ++j) {
      fa[j] = As[cur][2 * j + kh][wm + mi];
      fb[j] = Bs[cur][2 * j + kh][wn + mi];
    }
    __builtin_amdgcn_sched_barrier(0);       // keep the reads ahead of the MFMAs (the scheduler otherwise sinks each pair to its use)
#pragma unroll
    for (int j = 0; j < BK / 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) {  // the other buffer was last read in slab t - 1: every wavefront has passed that slab's barrier
      sa.store(As[cur ^ 1], tid);
      sb.store(Bs[cur ^ 1], tid);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = BK / 4; j < BK / 2; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc, 0, 0, 0);
    __syncthreads();
  }

  // C/D fragment: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int n = n0 + wn + (lane & 31);
  if (n < N) {
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < M) {
        if (part) { part[((size_t)blockIdx.z * M + m) * N + n] = alpha * acc[r]; continue; }
        float* c = C + m * scm + n * scn;
        float v = alpha * acc[r] + bv;
        if (beta != 0.f) v += beta * (*c);
        *c = v;
      }
    }
  }
}

static inline int gemm_layout(int64_t s_outer, int64_t s_k) { return s_k == 1 ? 0 : (s_outer == 1 ? 1 : 2); }
// 16-byte loads are legal for an operand when its base is 16-byte aligned and its non-unit stride is a multiple of 4 elements
static inline bool gemm_vec_ok(const float* X, int64_t s_outer, int64_t s_k, int layout) {
  if (layout == 2) return true;    // element-wise anyway
  const int64_t other = layout == 0 ? s_outer : s_k;
  return ((uintptr_t)X & 15) == 0 && (other & 3) == 0;
}

#define GEMM_DISPATCH(LA, LB, VEC, ...)                                                                     \
  do {                                                                                                      \
    if (VEC) hipLaunchKernelGGL((gemm_f32_kernel<LA, LB, true>), __VA_ARGS__);                              \
    else hipLaunchKernelGGL((gemm_f32_kernel<LA, LB, false>), __VA_ARGS__);                                 \
  } while (0)
#define GEMM_LAUNCH(la, lb, vec, ...)                                                                       \
  do {                                                                                                      \
    switch ((la) * 3 + (lb)) {                                                                              \
      case 0: GEMM_DISPATCH(0, 0, vec, __VA_ARGS__); break;                                                 \
      case 1: GEMM_DISPATCH(0, 1, vec, __VA_ARGS__); break;                                                 \
      case 2: GEMM_DISPATCH(0, 2, vec, __VA_ARGS__); break;                                                 \
      case 3: GEMM_DISPATCH(1, 0, vec, __VA_ARGS__); break;                                                 \
      case 4: GEMM_DISPATCH(1, 1, vec, __VA_ARGS__); break;                                                 \
      case 5: GEMM_DISPATCH(1, 2, vec, __VA_ARGS__); break;                                                 \
      case 6: GEMM_DISPATCH(2, 0, vec, __VA_ARGS__); break;                                                 \
      case 7: GEMM_DISPATCH(2, 1, vec, __VA_ARGS__); break;                                                 \
      default: GEMM_DISPATCH(2, 2, vec, __VA_ARGS__); break;                                                \
    }                                                                                                       \
  } while (0)

extern "C" int ttdg_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                             float* C, int64_t scm, int64_t scn, const float* bias, int M, int N, int K, float alpha,
                             float beta, ttdg_stream_t stream) {
  TTDG_REQUIRE(A && B && C, "gemm: null operand");
  TTDG_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: negative size");
  if (M == 0 || N == 0) return 0;
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
  hipStream_t st = (hipStream_t)stream;
  const int la = gemm_layout(sam, sak), lb = gemm_layout(sbn, sbk);
  const bool vec = gemm_vec_ok(A, sam, sak, la) && gemm_vec_ok(B, sbn, sbk, lb);
  GEMM_LAUNCH(la, lb, vec, grid, dim3(256), 0, st, A, sam, sak, B, sbn, sbk, C, scm, scn, bias, M, N, K, alpha, beta, 0, (float*)nullptr);
  return ttdg_launch_status("gemm_f32");
}

// ---- split-K variant for weight-gradient shapes (few output tiles, long K: dW = dY^T X with K = sum n_g) ------------
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, int kslices, float* __restrict__ C,
                                                                 int64_t scm, int64_t scn, const float* __restrict__ bias, int M,
                                                                 int N, float beta) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)M * N) return;
  const int m = (int)(e / N), n = (int)(e - (size_t)m * N);
  float v = 0.f;
  for (int z = 0; z < kslices; ++z) v += part[(size_t)z * M * N + e];    // fixed order: deterministic
  if (bias) v += bias[n];
  float* c = C + m * scm + n * scn;
  if (beta != 0.f) v += beta * (*c);
  *c = v;
}

extern "C" size_t ttdg_gemm_splitk_workspace_bytes(int M, int N, int kslices) {
  return (size_t)(kslices > 0 ? kslices : 0) * M * N * sizeof(float);
}

extern "C" int ttdg_gemm_f32_splitk(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                                    float* C, int64_t scm, int64_t scn, const float* bias, int M, int N, int K, float alpha,
                                    float beta, int kslices, void* ws, ttdg_stream_t stream) {
  TTDG_REQUIRE(A && B && C && ws, "gemm_splitk: null operand");
  TTDG_REQUIRE(M >= 0 && N >= 0 && K >= 0 && kslices >= 1 && kslices <= 64, "gemm_splitk: bad size");
  if (M == 0 || N == 0) return 0;
  const int Kc = (((K + kslices - 1) / kslices) + BK - 1) / BK * BK;
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, kslices);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  const int la = gemm_layout(sam, sak), lb = gemm_layout(sbn, sbk);
  const bool vec = gemm_vec_ok(A, sam, sak, la) && gemm_vec_ok(B, sbn, sbk, lb);
  GEMM_LAUNCH(la, lb, vec, grid, dim3(256), 0, st, A, sam, sak, B, sbn, sbk, C, scm, scn, bias, M, N, K, alpha, beta, Kc, part);
  if (int e = ttdg_launch_status("gemm_f32_splitk")) return e;
  const size_t total = (size_t)M * N;
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, kslices, C, scm, scn, bias,
                     M, N, beta);
  return ttdg_launch_status("gemm_splitk_reduce");
}

// ---- column sums (bias gradients): out[n] = sum_m X[m, n] ------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t ld, float* __restrict__ out,
                                                     int M, int N) {
  // 64 columns per workgroup (one per lane, coalesced), 4 wavefronts split the rows
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (n < N)
    for (int m = wave; m < M; m += 4) s += X[m * ld + n];
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && n < N) out[n] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

extern "C" int ttdg_colsum_f32(const float* X, int64_t ld, float* out, int M, int N, ttdg_stream_t stream) {
  TTDG_REQUIRE(X && out && N > 0 && M >= 0, "colsum: bad arguments");
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, (hipStream_t)stream, X, ld, out, M, N);
  return ttdg_launch_status("colsum");
}
