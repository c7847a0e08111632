// Dense fp32 GEMM on the matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).
// C[m,n] = alpha * sum_k A(m,k) B(n,k) + bias[n] + beta * C[m,n], arbitrary element strides so the
// same kernel serves x W^T (nn.Linear fwd), dY W (input grad) and dY^T X (weight grad).
//
// Workgroup = 256 threads = 4 wavefronts in a 2x2 arrangement over a 64x64 output tile; each wavefront owns one 32x32
// accumulator (16 fp32 per lane).  K is consumed in BK = 32 slabs staged through LDS in [k][m] order.
//
// The matrices on this path give ONE workgroup per CU (2048 x 512 output = 256 tiles; everything smaller is less): one wavefront
// per SIMD, nobody else to hide anything behind, so the wavefront's own instruction stream has to keep its MFMA pipe fed:
//   * global -> registers: three slabs are in flight (register ring of three, 16-byte loads, UNCONDITIONAL - a load under a
//     branch is waited for at the join; out-of-range chunks read element 0 and are zeroed on their way to LDS);
//   * registers -> LDS (double-buffered) between the two MFMA halves of a slab; one barrier per slab;
//   * LDS -> fragments one HALF slab ahead: the 16 fragment words of half 1 are requested before the 8 MFMAs of half 0 issue,
//     those of the next slab's half 0 right behind the barrier, under the last 4 MFMAs of half 1.  (Round 3 / the first round-4
//     loop requested all 32 words and then waited: ~400 exposed cycles per slab in front of 1024 cycles of MFMA - SQ counters of
//     profiles/r04_cfg3_sq_pmc.json: wavefront life 15.9 k cycles for 8.2 k cycles of MFMA.)
//   * LDS rows are rotated, element (k, m) sits in column (m + 16 ((k >> 2) & 3) + 32 (k & 1)) mod 64 of row k: the fragment
//     read of an MFMA (rows 2j and 2j+1, 32 consecutive m each) covers 64 distinct banks, and so does the transposing store of a
//     k-contiguous operand (16 m x 4 k-chunks per wavefront instruction).  The +4-word padding of round 3 left both 2-way.
//   * tiles are dealt to the XCDs in contiguous runs (workgroup id mod 8 = XCD): an XCD's L2 then fetches 1/8 of A instead of
//     all of it (2048 x 512 x 256: 768 KB instead of 2.06 MB per XCD).
// The k order of every accumulator is unchanged since round 1: results are bit-identical across all of these variants.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 64
#define BN 64
#define BK 32
#define LDS_LD BM
#define CT_LD (BN + 4)    /* row stride of the output tile on its way out (epilogue): 16-byte aligned rows */
// column of element (k, m) inside LDS row k (see the header)
#define LDS_COL(k, m) (((m) + 16 * (((k) >> 2) & 3) + 32 * ((k) & 1)) & 63)

// One operand slab (64 x BK) on its way global -> registers -> LDS T[k][m].  `kcontig` selects the thread -> element map that
// keeps the global reads wide along whichever index has unit stride; `vec` = 16-byte loads are legal (base pointer 16-byte
// aligned, the other stride a multiple of 4): otherwise - and on the ragged edge - guarded scalar loads, zero-filled.
template <bool kcontig, bool vec>
struct Slab {
  float4 r[2];
  bool ok[2];
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
      if (vec) {
        // `vec` also promises that the extent along the unit-stride index is a multiple of 4: a 16-byte chunk is inside the
        // matrix or outside it, never across the edge.  The load itself is UNCONDITIONAL (an outside chunk reads element 0 and
        // is zeroed on its way to LDS): a load under a branch makes the compiler wait for it at the join (s_waitcnt vmcnt(0)
        // right behind every load - which is what serialised round 3's and the first round-4 pipeline).
        ok[p] = kcontig ? (gm < Mlim && gk < Klim) : (gk < Klim && gm < Mlim);
        const int64_t off = kcontig ? gm * sm + gk : gk * sk + gm;
        r[p] = *reinterpret_cast<const float4*>(X + (ok[p] ? off : 0));
        continue;
      }
      ok[p] = true;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kcontig) {
        if (gm < Mlim) {
          const float* src = X + gm * sm + gk;
          if (gk < Klim) v.x = src[0];
          if (gk + 1 < Klim) v.y = src[1];
          if (gk + 2 < Klim) v.z = src[2];
          if (gk + 3 < Klim) v.w = src[3];
        }
      } else {
        if (gk < Klim) {
          const float* src = X + gk * sk + gm;
          if (gm < Mlim) v.x = src[0];
          if (gm + 1 < Mlim) v.y = src[1];
          if (gm + 2 < Mlim) v.z = src[2];
          if (gm + 3 < Mlim) v.w = src[3];
        }
      }
      r[p] = v;
    }
  }
  __device__ __forceinline__ void store(float (*T)[LDS_LD], int tid) const {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float4 v = ok[p] ? r[p] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (kcontig) {
        const int m = (tid & 15) + 16 * (tid >> 6), c = (tid >> 4) & 3, k = 4 * (c + 4 * p);
        const int ce = (m + 16 * c) & 63, co = ce ^ 32;       // even / odd k of the chunk: LDS_COL(k + e, m)
        T[k][ce] = v.x, T[k + 1][co] = v.y, T[k + 2][ce] = v.z, T[k + 3][co] = v.w;
      } else {
        const int k = (tid >> 4) + 16 * p;
        *reinterpret_cast<float4*>(&T[k][LDS_COL(k, 4 * (tid & 15))]) = v;     // rotation by a multiple of 16 words: the four m stay together
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
    for (int q = 0; q < 8; ++q) T[kb + 4 * q][LDS_COL(kb + 4 * q, m)] = r[q];
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
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BK * LDS_LD > BM * CT_LD ? 2 * 2 * BK * LDS_LD : BM * CT_LD];
  float (*As)[BK][LDS_LD] = reinterpret_cast<float (*)[BK][LDS_LD]>(smem);
  float (*Bs)[BK][LDS_LD] = As + 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  // XCD-aware tile order: the hardware deals workgroup ids round-robin to the 8 XCDs; give every XCD a contiguous run of tiles
  // (whole tile rows of C when the tile count is a multiple of 8) so that its L2 holds one slice of A, not all of it
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int T = gridDim.x * gridDim.y;
    if ((T & 7) == 0) {
      const int L = blockIdx.y * gridDim.x + blockIdx.x;
      const int L2 = (L & 7) * (T >> 3) + (L >> 3);
      by = L2 / gridDim.x; bx = L2 - by * gridDim.x;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  const int kbeg = part ? blockIdx.z * Kc : 0;
  const int kend = part ? min(K, kbeg + Kc) : K;
  const int nk = (kend - kbeg + BK - 1) / BK;
  typename SlabOf<la, vec>::type sa0, sa1, sa2;
  typename SlabOf<lb, vec>::type sb0, sb1, sb2;
#define GEMM_LOAD(X, Y, T)      /* beyond the last slab: the last one again (harmless, never stored) - no branch around a load */ \
  {                                                                         \
    const int tt = (T) < nk ? (T) : nk - 1;                                 \
    X.load(A, sam, sak, m0, kbeg + tt * BK, M, kend, tid);                  \
    Y.load(B, sbn, sbk, n0, kbeg + tt * BK, N, kend, tid);                  \
  }
  const int kh = lane >> 5, mi = lane & 31;
  // fragment columns of this lane for the four row rotations (rows 2j + kh: rotation (j >> 1) & 3, odd rows + 32)
  int ca[4], cb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    ca[r] = (wm + mi + 32 * kh + 16 * r) & 63;
    cb[r] = (wn + mi + 32 * kh + 16 * r) & 63;
  }
  float fa0[BK / 4], fb0[BK / 4], fa1[BK / 4], fb1[BK / 4];     // fragments of the two half slabs
#define GEMM_FRAGS(FA, FB, BUF, HALF)                                                   \
  _Pragma("unroll") for (int j = 0; j < BK / 4; ++j) {                                  \
    const int jj = (HALF) * (BK / 4) + j;                                               \
    FA[j] = As[BUF][2 * jj + kh][ca[(jj >> 1) & 3]];                                    \
    FB[j] = Bs[BUF][2 * jj + kh][cb[(jj >> 1) & 3]];                                    \
  }
  if (nk > 0) {
    GEMM_LOAD(sa0, sb0, 0)
    GEMM_LOAD(sa1, sb1, 1)
    GEMM_LOAD(sa2, sb2, 2)
    sa0.store(As[0], tid);
    sb0.store(Bs[0], tid);
  }
  __syncthreads();
  if (nk > 0) { GEMM_FRAGS(fa0, fb0, 0, 0) }
  // one slab: LOADX/LOADY = the ring slot slab t + 3 goes into (it held slab t, which is in LDS), STX/STY = the slot of slab t + 1
#define GEMM_SLAB(T, PREFETCH, LOADX, LOADY, STX, STY)                                                              \
  {                                                                                                                 \
    const int cur = (T) & 1;                                                                                        \
    const bool more = (T) + 1 < nk;                                                                                 \
    GEMM_FRAGS(fa1, fb1, cur, 1)                                                                                    \
    if (PREFETCH) GEMM_LOAD(LOADX, LOADY, (T) + 3)                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    _Pragma("unroll") for (int j = 0; j < BK / 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[j], fb0[j], acc, 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    if (more) {                                                                                                     \
      STX.store(As[cur ^ 1], tid);                                                                                  \
      STY.store(Bs[cur ^ 1], tid);                                                                                  \
    }                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    _Pragma("unroll") for (int j = 0; j < BK / 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[j], fb1[j], acc, 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    __syncthreads();                                                                                                \
    if (more) { GEMM_FRAGS(fa0, fb0, cur ^ 1, 0) }                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    _Pragma("unroll") for (int j = BK / 8; j < BK / 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[j], fb1[j], acc, 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }
  // whole triples run without a branch around any load; the last one or two slabs need no prefetch at all
  int t = 0;
  for (; t + 3 <= nk; t += 3) {
    GEMM_SLAB(t, true, sa0, sb0, sa1, sb1)
    GEMM_SLAB(t + 1, true, sa1, sb1, sa2, sb2)
    GEMM_SLAB(t + 2, true, sa2, sb2, sa0, sb0)
  }
  if (t < nk) GEMM_SLAB(t, false, sa0, sb0, sa1, sb1)
  if (t + 1 < nk) GEMM_SLAB(t + 1, false, sa1, sb1, sa2, sb2)
#undef GEMM_SLAB
#undef GEMM_FRAGS
#undef GEMM_LOAD

  // C/D fragment: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  // [r4] A whole tile of a row-major C leaves through LDS: 16 lanes write one 256-byte row segment per instruction (16-byte
  // stores).  Straight from the fragment layout a wavefront instruction writes two 128-byte pieces of two rows, one float per
  // lane: measured 2.8 us of the 12.7 us of the 2048 x 512 x 256 product for 4 MB of output (the K loop: 7.6 us, launch 2.2 us).
  // Same arithmetic per element (alpha * acc + bias, + beta * C): the bits do not depend on the path.
  const bool whole = !part && scn == 1 && m0 + BM <= M && n0 + BN <= N && (scm & 3) == 0 && (((uintptr_t)C | (uintptr_t)(bias ? bias : C)) & 15) == 0;
  if (whole) {
    float (*Ct)[CT_LD] = reinterpret_cast<float (*)[CT_LD]>(smem);      // (every wavefront is past the last slab's barrier)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ct[wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][wn + (lane & 31)] = acc[r];
    __syncthreads();
    const int c4 = 4 * (tid & 15);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4*>(bias + n0 + c4);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = 16 * p + (tid >> 4);
      const float4 a = *reinterpret_cast<const float4*>(&Ct[row][c4]);
      float* c = C + (m0 + row) * scm + n0 + c4;
      float4 v = make_float4(alpha * a.x + bv.x, alpha * a.y + bv.y, alpha * a.z + bv.z, alpha * a.w + bv.w);
      if (beta != 0.f) {
        const float4 o = *reinterpret_cast<const float4*>(c);
        v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w;
      }
      *reinterpret_cast<float4*>(c) = v;
    }
    return;
  }
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
// and the extent along the unit-stride index is a multiple of 4 (a chunk never straddles the edge: Slab::load)
static inline bool gemm_vec_ok(const float* X, int64_t s_outer, int64_t s_k, int layout, int outer, int K) {
  if (layout == 2) return true;    // element-wise anyway
  const int64_t other = layout == 0 ? s_outer : s_k;
  const int unit_extent = layout == 0 ? K : outer;
  return ((uintptr_t)X & 15) == 0 && (other & 3) == 0 && (unit_extent & 3) == 0;
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
  const bool vec = gemm_vec_ok(A, sam, sak, la, M, K) && gemm_vec_ok(B, sbn, sbk, lb, N, K);
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
  const bool vec = gemm_vec_ok(A, sam, sak, la, M, K) && gemm_vec_ok(B, sbn, sbk, lb, N, K);
  GEMM_LAUNCH(la, lb, vec, grid, dim3(256), 0, st, A, sam, sak, B, sbn, sbk, C, scm, scn, bias, M, N, K, alpha, beta, Kc, part);
  if (int e = ttdg_launch_status("gemm_f32_splitk")) return e;
  const size_t total = (size_t)M * N;
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, kslices, C, scm, scn, bias,
                     M, N, beta);
  return ttdg_launch_status("gemm_splitk_reduce");
}

// ---- column sums (bias gradients): out[n] = sum_m X[m, n] ------------------------------------------
// [r4] 16 columns x 64 row groups per workgroup of 1024 threads (round 3: 64 columns x 4 row groups, N / 64 = 8 workgroups for
// the 2048 x 512 bias gradient of cfg-3 - 512 dependent loads per thread, 121 us for 4 MB).  A thread keeps eight independent
// loads in flight; the 64 partial sums of a column meet in LDS and are added in a fixed order (deterministic).
#define CS_COLS 16
#define CS_GROUPS 64
__global__ __launch_bounds__(CS_COLS * CS_GROUPS) void colsum_kernel(const float* __restrict__ X, int64_t ld, float* __restrict__ out,
                                                                     int M, int N) {
  __shared__ float part[CS_GROUPS][CS_COLS + 1];
  const int c = threadIdx.x & (CS_COLS - 1), rg = threadIdx.x / CS_COLS;
  const int n = blockIdx.x * CS_COLS + c;
  float s = 0.f;
  if (n < N) {
    const float* col = X + n;
    int m = rg;
    for (; m + 7 * CS_GROUPS < M; m += 8 * CS_GROUPS) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = col[(int64_t)(m + u * CS_GROUPS) * ld];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; m < M; m += CS_GROUPS) s += col[(int64_t)m * ld];
  }
  part[rg][c] = s;
  __syncthreads();
  if (threadIdx.x < CS_COLS && n < N) {
    float t = 0.f;
#pragma unroll 8
    for (int g = 0; g < CS_GROUPS; ++g) t += part[g][threadIdx.x];
    out[n] = t;
  }
}

extern "C" int ttdg_colsum_f32(const float* X, int64_t ld, float* out, int M, int N, ttdg_stream_t stream) {
  TTDG_REQUIRE(X && out && N > 0 && M >= 0, "colsum: bad arguments");
  hipLaunchKernelGGL(colsum_kernel, dim3((N + CS_COLS - 1) / CS_COLS), dim3(CS_COLS * CS_GROUPS), 0, (hipStream_t)stream, X, ld, out, M, N);
  return ttdg_launch_status("colsum");
}
