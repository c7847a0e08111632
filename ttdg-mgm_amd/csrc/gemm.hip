// Dense fp32 GEMM on the matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).
// C[m,n] = alpha * sum_k A(m,k) B(n,k) + bias[n] + beta * C[m,n], arbitrary element strides so the
// same kernel serves x W^T (nn.Linear fwd), dY W (input grad) and dY^T X (weight grad).
//
// Workgroup = 256 threads = 4 wavefronts in a 2x2 arrangement over a 64x64 output tile; each
// wavefront owns one 32x32 accumulator (16 fp32 per lane).  K is consumed in BK=16 slabs staged
// through LDS in [k][m] order, so a wavefront's A/B fragment read (lane l -> m = l&31, k = l>>5)
// touches 32 consecutive words per half-wave: conflict-free ds_read_b32.
// The matrices on this path are small (M = sum n_g ~ 100..2048, N,K in {32,256,512}); the kernel is
// latency/launch bound at those sizes, so it favours many small tiles over deep pipelining.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 64
#define BN 64
#define BK 16
#define LDS_LD (BM + 4)  // +4 words: transposing stores of k-contiguous operands are <=2-way

// Stage a (64 x BK) operand slab into LDS as T[k][m].  `kcontig` selects the thread->element map that
// keeps the global reads coalesced along whichever index has unit stride.
template <bool kcontig>
__device__ __forceinline__ void stage(float (*T)[LDS_LD], const float* __restrict__ X, int64_t sm, int64_t sk, int m0,
                                      int k0, int Mlim, int Klim, int tid) {
  if (kcontig) {
    const int k = tid & 15, mb = tid >> 4;  // 16 lanes walk k, 16 row groups
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mb + 16 * r;
      const int gm = m0 + m, gk = k0 + k;
      T[k][m] = (gm < Mlim && gk < Klim) ? X[gm * sm + gk * sk] : 0.f;
    }
  } else {
    const int m = tid & 63, kb = tid >> 6;  // 64 lanes walk m
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kb + 4 * r;
      const int gm = m0 + m, gk = k0 + k;
      T[k][m] = (gm < Mlim && gk < Klim) ? X[gm * sm + gk * sk] : 0.f;
    }
  }
}

template <bool a_kc, bool b_kc>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t sam, int64_t sak,
                                                       const float* __restrict__ B, int64_t sbn, int64_t sbk,
                                                       float* __restrict__ C, int64_t scm, int64_t scn,
                                                       const float* __restrict__ bias, int M, int N, int K, float alpha,
                                                       float beta, int Kc, float* __restrict__ part) {
  // split-K (part != nullptr): blockIdx.z owns k in [z*Kc, (z+1)*Kc) and writes alpha * partial into its own M x N plane
  // (deterministic; gemm_splitk_reduce_kernel adds the planes, the bias and beta * C)
  __shared__ float As[2][BK][LDS_LD];
  __shared__ float Bs[2][BK][LDS_LD];
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
  stage<a_kc>(As[0], A, sam, sak, m0, kbeg, M, kend, tid);
  stage<b_kc>(Bs[0], B, sbn, sbk, n0, kbeg, N, kend, tid);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    if (t + 1 < nk) {  // prefetch the next slab into the other buffer while this one feeds the MFMAs
      stage<a_kc>(As[cur ^ 1], A, sam, sak, m0, kbeg + (t + 1) * BK, M, kend, tid);
      stage<b_kc>(Bs[cur ^ 1], B, sbn, sbk, n0, kbeg + (t + 1) * BK, N, kend, tid);
    }
    const int kh = lane >> 5, mi = lane & 31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = As[cur][kk + kh][wm + mi];
      const float b = Bs[cur][kk + kh][wn + mi];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
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

extern "C" int ttdg_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                             float* C, int64_t scm, int64_t scn, const float* bias, int M, int N, int K, float alpha,
                             float beta, ttdg_stream_t stream) {
  TTDG_REQUIRE(A && B && C, "gemm: null operand");
  TTDG_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: negative size");
  if (M == 0 || N == 0) return 0;
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
  hipStream_t st = (hipStream_t)stream;
  const bool akc = (sak == 1), bkc = (sbk == 1);
#define LAUNCH(a, b) \
  hipLaunchKernelGGL((gemm_f32_kernel<a, b>), grid, dim3(256), 0, st, A, sam, sak, B, sbn, sbk, C, scm, scn, bias, M, N, K, alpha, beta, 0, (float*)nullptr)
  if (akc && bkc) LAUNCH(true, true);
  else if (akc) LAUNCH(true, false);
  else if (bkc) LAUNCH(false, true);
  else LAUNCH(false, false);
#undef LAUNCH
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
  const bool akc = (sak == 1), bkc = (sbk == 1);
  float* part = (float*)ws;
#define LAUNCH(a, b) \
  hipLaunchKernelGGL((gemm_f32_kernel<a, b>), grid, dim3(256), 0, st, A, sam, sak, B, sbn, sbk, C, scm, scn, bias, M, N, K, alpha, beta, Kc, part)
  if (akc && bkc) LAUNCH(true, true);
  else if (akc) LAUNCH(true, false);
  else if (bkc) LAUNCH(false, true);
  else LAUNCH(false, false);
#undef LAUNCH
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
