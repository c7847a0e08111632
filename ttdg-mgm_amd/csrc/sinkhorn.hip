// A5 — log-space Sinkhorn (reference utils/sinkhorn.py:85-87 -> pygmtools.sinkhorn [3P, absent];
// specification: SURVEY.md Appendix B; oracle: oracle/sinkhorn_spec.py).
//
// Formulation.  The reference rewrites the whole matrix at every sweep (log_s -= logsumexp).  We keep the
// scaled input L = s/tau constant and carry the dual potentials instead:
//     row sweep:  f_p = lse_q(L_pq - g_q)      col sweep:  g_q = lse_p(L_pq - f_p)      y = L - f (+) g
// which is the same map in exact arithmetic, reads the matrix once per sweep, writes only r+c numbers,
// and lets the backward pass rebuild every intermediate y^(k) from K small vectors instead of K matrices.
// Dummy rows (dummy_row=True): the (c - r) appended rows are identical (all -100 after tau scaling) and
// stay identical under both sweeps, so they are ONE virtual row with multiplicity c - r.
// Everything runs in base 2 (L2 = L*log2 e) on v_exp_f32 / v_log_f32.
//
// One workgroup per matrix; a line (row or column) is reduced by a sub-group of 16/32/64 lanes with
// xor-shuffles; the matrix lives in LDS when it fits (<= 36k floats), otherwise it is re-read from L2.
#include "sinkhorn_device.h"

// ---- pair stage (multi_graph_matching.py:504-525) --------------------------------------------------
// (Round 2 tried the solver's one-wavefront block-layout projector here: equally accurate - max |Wds - oracle| 1.4e-6 vs 1.9e-6
// on planted case p4 - but not faster for the ten 20..40-node pairs of a TTA step (116 vs 102 us under rocprofv3: neither kernel
// is bound by its sweeps), and its different rounding alone moved p4's solve off the reference's permutation in the
// tau = 0.00625 stage; removed.)
__device__ __forceinline__ void pair_of(int idx, int G, int& a, int& b) {
  a = 0;
  while ((a + 1) * (a + 2) / 2 <= idx) ++a;  // pairs ordered (0,0),(1,0),(1,1),(2,0)...
  b = idx - a * (a + 1) / 2;
}

template <bool kLds>
__global__ void sinkhorn_pairs_fwd_kernel(const float* __restrict__ part, int ksplit, const float* __restrict__ b2,
                                          ttdg_graphs_t gr, float tau, int iters, float* __restrict__ Wds,
                                          float* __restrict__ pot, int cmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int a, b;
  pair_of(blockIdx.x, gr.G, a, b);
  const int M = gr.off[gr.G];
  const int na = gr.off[a + 1] - gr.off[a], nb = gr.off[b + 1] - gr.off[b];
  const float* blk = part + (size_t)gr.off[a] * M + gr.off[b];  // element (i in a, j in b) at blk[i*M + j]
  float* wab = Wds + (size_t)gr.off[a] * M + gr.off[b];
  float* wba = Wds + (size_t)gr.off[b] * M + gr.off[a];
  SkProb pb;
  pb.src = blk;
  pb.splane = (int64_t)M * M;
  pb.nplanes = ksplit;
  pb.bias = b2 ? *b2 : 0.f;
  pb.scale = TTDG_LOG2E / tau;
  if (nb >= na) {  // rows = a-nodes (reference: end_y - start_y >= end_x - start_x -> no transpose)
    pb.r = na; pb.c = nb; pb.sp = M; pb.sq = 1;
    pb.out = wab; pb.op = M; pb.oq = 1;
    pb.mir = (a != b) ? wba : nullptr; pb.mp = 1; pb.mq = M;
  } else {         // rows = b-nodes (transposed before, transposed back after)
    pb.r = nb; pb.c = na; pb.sp = 1; pb.sq = M;
    pb.out = wab; pb.op = 1; pb.oq = M;
    pb.mir = wba; pb.mp = M; pb.mq = 1;
  }
  pb.mult = pb.c - pb.r;
  pb.potld = cmax + 1;
  pb.pot = pot ? pot + (size_t)blockIdx.x * iters * pb.potld : nullptr;
  sk_forward<kLds>(pb, smem, iters);
}


// ---- register-resident pair stage for 128 < c <= 256 (BASELINE cfg-3: 256-node graphs) ---------------------------------
// A 256 x 256 fp32 matrix (256 KB) does not fit the 160 KB LDS, but it fits the register file of ONE workgroup:
// 1024 threads x 64 VGPRs.  Wavefront w owns rows [16w, 16w+16), lane l owns columns 4l..4l+3, so
//   row sweep : 4 in-lane terms per row, the sixteen rows of a wavefront reduced together, no LDS, no barrier;
//   col sweep : 16 in-lane terms per column, the 16 wavefront partials meet in LDS (one barrier), and every wavefront
//               finishes all 256 columns redundantly (no second barrier, identical values everywhere).
// (Rounds 1-4 ran the log-domain sweeps in this layout - 20 exponentials per entry, 144 us at 36 x 256 x 256; round 5: the
// scaling form below.)
#define SKR_THREADS 1024
#define SKR_WAVES 16
#define SKR_C 256
#define SKR_RW (SKR_C / SKR_WAVES)   /* rows per wavefront */
#define SKR_BIG 1.2e24f
#define SKR_SMALL 8.3e-25f

__device__ __forceinline__ float skr_sgpr(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ bool skr_sane(float s) { return s > SKR_SMALL && s < SKR_BIG; }

// element (p, q) through a wavefront-uniform row pointer + a 32-bit lane offset (scalar base + vector offset addressing)
__device__ __forceinline__ float skr_load(const SkProb& pb, const float* rowp, int qo) {
  float v = pb.bias;
  for (int k = 0; k < pb.nplanes; ++k) v += rowp[(size_t)k * pb.splane + qo];
  return v * pb.scale;
}

__device__ __forceinline__ void sk_pair_problem(const float* part, int ksplit, const float* b2, const ttdg_graphs_t& gr, float tau,
                                                int a, int b, SkProb& pb) {
  const int M = gr.off[gr.G];
  const int na = gr.off[a + 1] - gr.off[a], nb = gr.off[b + 1] - gr.off[b];
  pb.src = part + (size_t)gr.off[a] * M + gr.off[b];
  pb.splane = (int64_t)M * M;
  pb.nplanes = ksplit;
  pb.bias = b2 ? *b2 : 0.f;
  pb.scale = TTDG_LOG2E / tau;
  if (nb >= na) { pb.r = na; pb.c = nb; pb.sp = M; pb.sq = 1; }
  else          { pb.r = nb; pb.c = na; pb.sp = 1; pb.sq = M; }
  pb.mult = pb.c - pb.r;
}


// ---- [r5] scaling-form pair stage for 128 < c <= 256: ONE exponential per entry for the whole problem ----------------------------
// (the log-domain kernel of rounds 1-4 spent 144 us on 36 blocks of 256 x 256 - 20 exponentials per entry, sixteen
// branch-separated wavefront reductions per row sweep.)  With f0 = the row potentials after sweep 0 (g = 0) and
// K_pq = exp2(L_pq - f0_p) (the row-normalised matrix, K <= 1), every later state of the reference's iteration is
//     exp2(L - f - g) = K_pq u_p v_q,      u_p = exp2(f0_p - f_p),   v_q = exp2(-g_q)
// and the sweeps are   row: s_p = sum_q K_pq v_q,  f_p = f0_p + log2 s_p,  u_p = 1 / s_p
//                      col: c_q = sum_p K_pq u_p,  g_q = log2 c_q,         v_q = 1 / c_q
// - the same map in exact arithmetic (sinkhorn.py:58-87 -> pygmtools), one FMA per entry per sweep, the potentials
// are f0 + log2(sum) with no error carried from sweep to sweep, and the output is K u v (no final exponential).  The dummy rows
// (one virtual row of multiplicity c - r, constant fill) are K_dq = 1 / c exactly.
// What the log domain buys - range - is kept by a fall-back: a live line sum outside [2^-80, 2^80] (e.g. a column whose every entry
// is 2^-126 below its row's maximum) sends the WHOLE workgroup into sk_forward<false> - the log-domain sweeps of sinkhorn_device.h on
// the input itself, streamed from L2 (status through LDS before the next barrier: every wavefront takes the branch at the same sweep).
// Layout as above (wavefront w: rows 16 w .. 16 w + 15, lane l: columns 4 l .. 4 l + 3, K in 64 registers); the sixteen row sums of
// a sweep are reduced TOGETHER: v_permlane32_swap / v_permlane16_swap transpose while they add (16 -> 8 -> 4 registers), four DPP
// stages finish inside the 16-lane rows: 40 instructions instead of 96 + 16 broadcasts, and no branch between the rows.
struct SkrAdd { static __device__ __forceinline__ float op(float a, float b) { return a + b; } };
struct SkrMax { static __device__ __forceinline__ float op(float a, float b) { return fmaxf(a, b); } };

// a[ri] = this lane's partial of local row ri  ->  t[k]: every lane of the 16-lane row j holds the total of local row k + 4 j
template <class Op>
__device__ __forceinline__ void skr_reduce16(const float (&a)[SKR_RW], float (&t)[4]) {
  float b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {      // lanes 0-31: row i over both halves, lanes 32-63: row i + 8
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(a[i]), __float_as_int(a[i + 8]), false, false);
    b[i] = Op::op(__int_as_float(r[0]), __int_as_float(r[1]));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {      // 16-lane rows 0..3: local rows i, i + 4, i + 8, i + 12
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(b[i]), __float_as_int(b[i + 4]), false, false);
    t[i] = Op::op(__int_as_float(r[0]), __int_as_float(r[1]));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    t[i] = Op::op(t[i], dpp_f<0xB1>(t[i]));
    t[i] = Op::op(t[i], dpp_f<0x4E>(t[i]));
    t[i] = Op::op(t[i], dpp_f<0x141>(t[i]));
    t[i] = Op::op(t[i], dpp_f<0x140>(t[i]));
  }
}
// the same with the partials produced on demand, two rows at a time (the sixteen partials never live together: the backward kernel
// keeps 64 registers of K next to them)
template <class Op, class F>
__device__ __forceinline__ void skr_reduce16_stream(F&& partial, float (&t)[4]) {
  float b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = partial(i), y = partial(i + 8);
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(x), __float_as_int(y), false, false);
    b[i] = Op::op(__int_as_float(r[0]), __int_as_float(r[1]));
    if (i & 1) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(b[i]), __float_as_int(b[i + 4]), false, false);
    t[i] = Op::op(__int_as_float(r[0]), __int_as_float(r[1]));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    t[i] = Op::op(t[i], dpp_f<0xB1>(t[i]));
    t[i] = Op::op(t[i], dpp_f<0x4E>(t[i]));
    t[i] = Op::op(t[i], dpp_f<0x141>(t[i]));
    t[i] = Op::op(t[i], dpp_f<0x140>(t[i]));
  }
}
// local row ri of the layout skr_reduce16 returns, as a wavefront-uniform scalar
#define SKR_ROW_SCALAR(t, ri) __int_as_float(__builtin_amdgcn_readlane(__float_as_int((t)[(ri) & 3]), 16 * ((ri) >> 2)))

// ---- 16-byte tile accesses of the register layout (wavefront: rows p0 .. p0 + 15, lane: columns q0 .. q0 + 3) -----------------------
// A block's element (p, q) lives at base + p * sp + q * sq with one of the two strides equal to 1 (which one depends on the pair's
// orientation).  kAlongQ: the lane's four columns are contiguous (one dwordx4 per row, a wavefront instruction covers 1 KB);
// otherwise four consecutive rows are (dwordx4 per column and row quad, 16-byte pieces one matrix row apart).  Blocks start at
// arbitrary node offsets: the vector type is 4-byte aligned (gfx950 global accesses need dword alignment only).  Lanes / row quads
// that straddle the block's edge use clamped scalar loads, stores touch valid elements only.
typedef float skr_f4u __attribute__((ext_vector_type(4), aligned(4)));

// rows pp .. pp + 3 (pp wavefront-uniform), columns q0 .. q0 + 3
template <bool kAlongQ>
__device__ __forceinline__ void skr_quad_load(const float* __restrict__ base, int64_t sp, int64_t sq, int pp, int q0, int r, int c,
                                              float (&D)[4][4]) {
  if (kAlongQ) {
    // every lane issues the 16-byte load from an in-bounds address (lanes at or beyond the block's last columns read the block's last
    // four columns: those values are never used); the one lane that straddles the edge then fetches its valid elements one by one
    // (a block of fewer than four columns has no in-bounds 16-byte piece at all: every lane takes the clamped scalar loads)
    if (c < 4) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float* rowp = base + (int64_t)min(pp + t, r - 1) * sp;
#pragma unroll
        for (int j = 0; j < 4; ++j) D[t][j] = rowp[min(q0 + j, c - 1)];
      }
      return;
    }
    const int qs = min(q0, c - 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const skr_f4u x = *reinterpret_cast<const skr_f4u*>(base + (int64_t)min(pp + t, r - 1) * sp + qs);
      D[t][0] = x.x; D[t][1] = x.y; D[t][2] = x.z; D[t][3] = x.w;
    }
    if (q0 < c && q0 + 3 >= c) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float* rowp = base + (int64_t)min(pp + t, r - 1) * sp;
#pragma unroll
        for (int j = 0; j < 3; ++j) D[t][j] = rowp[min(q0 + j, c - 1)];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* colp = base + (int64_t)min(q0 + j, c - 1) * sq;
      if (pp + 3 < r) {              // wavefront-uniform
        const skr_f4u x = *reinterpret_cast<const skr_f4u*>(colp + pp);
        D[0][j] = x.x; D[1][j] = x.y; D[2][j] = x.z; D[3][j] = x.w;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) D[t][j] = colp[min(pp + t, r - 1)];
      }
    }
  }
}

// val(t, j) = the value of element (pp + t, q0 + j); only elements inside the r x c block are written
template <bool kAlongQ, class F>
__device__ __forceinline__ void skr_quad_store(float* __restrict__ base, int64_t sp, int64_t sq, int pp, int q0, int r, int c, F&& val) {
  if (kAlongQ) {
    if (q0 + 3 < c) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (pp + t < r) {
          skr_f4u x;
          x.x = val(t, 0); x.y = val(t, 1); x.z = val(t, 2); x.w = val(t, 3);
          *reinterpret_cast<skr_f4u*>(base + (int64_t)(pp + t) * sp + q0) = x;
        }
    }
    if (q0 < c && q0 + 3 >= c) {       // the lane that straddles the edge (at most one per wavefront)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (pp + t < r) {
          float* rowp = base + (int64_t)(pp + t) * sp;
#pragma unroll
          for (int j = 0; j < 3; ++j) if (q0 + j < c) rowp[q0 + j] = val(t, j);
        }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (q0 + j < c) {
        float* colp = base + (int64_t)(q0 + j) * sq;
        if (pp + 3 < r) {
          skr_f4u x;
          x.x = val(0, j); x.y = val(1, j); x.z = val(2, j); x.w = val(3, j);
          *reinterpret_cast<skr_f4u*>(colp + pp) = x;
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) if (pp + t < r) colp[pp + t] = val(t, j);
        }
      }
    }
  }
}

template <bool kAlongQ>
__device__ __forceinline__ void skr_tile_load(const float* __restrict__ base, int64_t sp, int64_t sq, int p0, int q0, int r, int c,
                                              float (&A)[SKR_RW][4]) {
#pragma unroll
  for (int k = 0; k < SKR_RW / 4; ++k) {
    float D[4][4];
    skr_quad_load<kAlongQ>(base, sp, sq, p0 + 4 * k, q0, r, c, D);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) A[4 * k + t][j] = D[t][j];
  }
}

// val(i, j) = the value of element (p0 + i, q0 + j)
template <bool kAlongQ, class F>
__device__ __forceinline__ void skr_tile_store(float* __restrict__ base, int64_t sp, int64_t sq, int p0, int q0, int r, int c, F&& val) {
#pragma unroll
  for (int k = 0; k < SKR_RW / 4; ++k)
    skr_quad_store<kAlongQ>(base, sp, sq, p0 + 4 * k, q0, r, c, [&](int t, int j) { return val(4 * k + t, j); });
}

// [r5] column totals of a sweep: every wavefront has written its 256 partials to s_part[wave][.] and the workgroup has met.  Rounds 1-4
// (and the first scaling-form build) let EVERY wavefront add all sixteen partials of its lanes' columns - 16 x 16 KB of LDS reads per
// sweep, ~2000 cycles of the LDS port, more than the sweep's arithmetic.  Here the first four wavefronts add one column per thread
// (the sixteen partials in the same order as before: bit-identical totals), the totals go through 1 KB of LDS and a second barrier.
__device__ __forceinline__ void skr_column_totals(const float* __restrict__ sp, float* __restrict__ s_col, int tid, int q0, float (&cs)[4]) {
  if (tid < SKR_C) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SKR_WAVES; ++w) t += sp[w * SKR_C + tid];
    s_col[tid] = t;
  }
  __syncthreads();
  const float4 t4 = *reinterpret_cast<const float4*>(&s_col[q0]);
  cs[0] = t4.x; cs[1] = t4.y; cs[2] = t4.z; cs[3] = t4.w;
}

// kFlip: the instantiation handles the blocks of that orientation only (rows <= cols after transposition or not) and returns at once
// on the others - the orientation decides which index is contiguous in memory, and one compiled path per kernel keeps the 64 registers
// of K next to everything else without spills.  The host launches the instantiation(s) the batch needs (cfg-3: equal sizes, one launch).
template <bool kFlip>
__global__ __launch_bounds__(SKR_THREADS) void sinkhorn_pairs_fwd_scale_kernel(const float* __restrict__ part, int ksplit,
                                                                               const float* __restrict__ b2, ttdg_graphs_t gr, float tau,
                                                                               int iters, float* __restrict__ Wds,
                                                                               float* __restrict__ pot, int cmax) {
  __shared__ __attribute__((aligned(16))) float s_part[2][SKR_WAVES * SKR_C];
  __shared__ __attribute__((aligned(16))) float s_col[SKR_C];
  __shared__ int s_bad;
  int a, b;
  pair_of(blockIdx.x, gr.G, a, b);
  const int M = gr.off[gr.G];
  SkProb pb;
  sk_pair_problem(part, ksplit, b2, gr, tau, a, b, pb);
  float* wab = Wds + (size_t)gr.off[a] * M + gr.off[b];
  float* wba = Wds + (size_t)gr.off[b] * M + gr.off[a];
  if ((pb.sq != 1) != kFlip) return;
  constexpr bool flip = kFlip;
  pb.out = wab; pb.op = flip ? 1 : M; pb.oq = flip ? M : 1;
  pb.mir = (flip || a != b) ? wba : nullptr; pb.mp = flip ? M : 1; pb.mq = flip ? 1 : M;
  const int potld = cmax + 1;
  float* pt = pot ? pot + (size_t)blockIdx.x * iters * potld : nullptr;
  const int r = pb.r, c = pb.c, mult = pb.mult;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int p0 = wave * SKR_RW, q0 = lane * 4;
  if (tid == 0) s_bad = 0;
  __syncthreads();

  // ---- load (16-byte accesses along the block's contiguous direction), L = (x + b2) log2(e) / tau, -inf outside the block
  float A[SKR_RW][4];
  skr_tile_load<!kFlip>(pb.src, pb.sp, pb.sq, p0, q0, r, c, A);
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) A[i][j] = (p0 + i < r && q0 + j < c) ? (A[i][j] + pb.bias) * pb.scale : -INFINITY;
  // row liveness in the layout of skr_reduce16: register k, 16-lane row j <-> local row k + 4 j
  bool rlive[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) rlive[k] = p0 + k + 4 * (lane >> 4) < r;
  bool clive[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) clive[j] = q0 + j < c;

  // ---- sweep 0 (rows, g = 0), exact two-pass form:  f0 = max + log2 sum exp2(L - max);   K = exp2(L - f0)
  float part16[SKR_RW], mt[4], st[4], f0t[4];
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i) part16[i] = fmaxf(fmaxf(A[i][0], A[i][1]), fmaxf(A[i][2], A[i][3]));
  skr_reduce16<SkrMax>(part16, mt);
#pragma unroll
  for (int k = 0; k < 4; ++k) mt[k] = rlive[k] ? mt[k] : 0.f;
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i) {
    const float m = SKR_ROW_SCALAR(mt, i);
#pragma unroll
    for (int j = 0; j < 4; ++j) A[i][j] = fast_exp2(A[i][j] - m);
    part16[i] = (A[i][0] + A[i][1]) + (A[i][2] + A[i][3]);
  }
  skr_reduce16<SkrAdd>(part16, st);
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    st[k] = rlive[k] ? st[k] : 1.f;
    bad |= !(st[k] >= 1.f && st[k] < SKR_BIG);                 // a live row holds its maximum: the sum is in [1, c]; NaN input fails here
    f0t[k] = mt[k] + fast_log2(st[k]);
    if (pt && iters > 0 && (lane & 15) == 0 && rlive[k]) pt[p0 + k + 4 * (lane >> 4)] = f0t[k];
    st[k] = __builtin_amdgcn_rcpf(st[k]);
  }
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i) {
    const float u = SKR_ROW_SCALAR(st, i);
#pragma unroll
    for (int j = 0; j < 4; ++j) A[i][j] *= u;
  }
  const float kd = mult > 0 ? 1.f / (float)c : 0.f;                                   // the dummy row of K: uniform
  const float fd0 = fast_log2((float)c);             // [r6] the dummy row's potential is logged WITHOUT the constant fill (include/ttdg_mgm.h: pot)
  if (pt && iters > 0 && mult > 0 && tid == 0) pt[r] = fd0;
  if (__ballot(bad) != 0ull && lane == 0) s_bad = 1;
  float usc[SKR_RW];                     // u_p of this wavefront's rows (wavefront-uniform: SGPRs)
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i) usc[i] = 1.f;
  float ud = 1.f, v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = clive[j] ? 1.f : 0.f;
  int buf = 0;
  if (iters == 0 && tid == 0) s_bad = 1; // no sweep at all: the output is exp2(L) - only the exact form has it

  for (int it = 1; it < iters; ++it) {
    if (it & 1) {
      // ---- columns: c_q = sum_p K_pq u_p (+ mult * kd * u_d) ----
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < SKR_RW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(A[i][j], usc[i], acc[j]);
      *reinterpret_cast<float4*>(&s_part[buf][wave * SKR_C + q0]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      __syncthreads();
      if (s_bad) break;                  // set before this barrier by some wavefront's row sweep: uniform over the workgroup
      float cs[4];
      skr_column_totals(s_part[buf], s_col, tid, q0, cs);
      buf ^= 1;
      bool cbad = false;
      float gq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (mult > 0) cs[j] = __builtin_fmaf((float)mult * kd, ud, cs[j]);
        cbad |= clive[j] && !skr_sane(cs[j]);
        gq[j] = fast_log2(cs[j]);
        v[j] = clive[j] ? __builtin_amdgcn_rcpf(cs[j]) : 0.f;
      }
      if (__ballot(cbad) != 0ull) {      // the same columns, the same sums in every wavefront: the same decision
        if (tid == 0) s_bad = 1;
        break;
      }
      if (pt && wave == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (clive[j]) pt[it * potld + q0 + j] = gq[j];
      }
    } else {
      // ---- rows: s_p = sum_q K_pq v_q ----
      skr_reduce16_stream<SkrAdd>([&](int i) { return __builtin_fmaf(A[i][0], v[0], A[i][1] * v[1]) + __builtin_fmaf(A[i][2], v[2], A[i][3] * v[3]); }, st);
      bool rbad = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        st[k] = rlive[k] ? st[k] : 1.f;
        rbad |= !skr_sane(st[k]);
        if (pt && (lane & 15) == 0 && rlive[k]) pt[it * potld + p0 + k + 4 * (lane >> 4)] = f0t[k] + fast_log2(st[k]);
        st[k] = __builtin_amdgcn_rcpf(st[k]);
      }
#pragma unroll
      for (int i = 0; i < SKR_RW; ++i) usc[i] = SKR_ROW_SCALAR(st, i);
      if (mult > 0) {                    // the dummy row: s_d = kd * sum_q v_q  (every wavefront, redundantly)
        const float sd = kd * wave_sum_f32_dpp((v[0] + v[1]) + (v[2] + v[3]));
        rbad |= !skr_sane(sd);
        ud = __builtin_amdgcn_rcpf(sd);
        if (pt && tid == 0) pt[it * potld + r] = fd0 + fast_log2(sd);
      }
      if (__ballot(rbad) != 0ull && lane == 0) s_bad = 1;      // read by everybody behind the next column sweep's barrier
    }
  }
  __syncthreads();                       // the last row sweep's status (iters odd), and the break paths meet here
  if (s_bad) {                           // cold path: the log-domain sweeps on the input itself, re-read from L2 at every sweep
    __syncthreads();                     // (nobody still reads s_part: its first c + 1 + c floats become the potentials)
    pb.pot = pt; pb.potld = potld;
    sk_forward<false>(pb, &s_part[0][0], iters);
    return;
  }

  // ---- output: Wds = K u v in the block's own orientation; the mirrored blocks are written by skr_block_transpose_kernel ----
  skr_tile_store<!kFlip>(pb.out, pb.op, pb.oq, p0, q0, r, c, [&](int i, int j) { return (A[i][j] * usc[i]) * v[j]; });
}

// Transposed copy of every off-diagonal pair block: src block (a, b) [src_lower] or (b, a) -> dst block of the opposite triangle.
// Forward: Wds(b, a) = Wds(a, b)^T (the symmetric fill of multi_graph_matching.py:523-525); backward: the incoming gradient of
// Wds(b, a) laid out like the (a, b) block it belongs to.  64 x 64 tiles through LDS, both sides in 256-byte row segments.
__global__ __launch_bounds__(256) void skr_block_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, ttdg_graphs_t gr,
                                                                  int src_lower) {
  __shared__ float tile[64][65];
  int a = 1, idx = blockIdx.y;
  while (idx >= a) { idx -= a; ++a; }
  const int b = idx;
  const int M = gr.off[gr.G];
  const int na = gr.off[a + 1] - gr.off[a], nb = gr.off[b + 1] - gr.off[b];
  const int nr = src_lower ? na : nb, nc = src_lower ? nb : na;            // shape of the source block
  const int tcols = (nc + 63) >> 6, ntiles = ((nr + 63) >> 6) * tcols;
  if ((int)blockIdx.x >= ntiles) return;
  const int r0 = ((int)blockIdx.x / tcols) << 6, c0 = ((int)blockIdx.x % tcols) << 6;
  const float* sb = src + (size_t)(src_lower ? gr.off[a] : gr.off[b]) * M + (src_lower ? gr.off[b] : gr.off[a]);
  float* db = dst + (size_t)(src_lower ? gr.off[b] : gr.off[a]) * M + (src_lower ? gr.off[a] : gr.off[b]);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int rr = r0 + ty + 4 * k, cc = c0 + tx;
    if (rr < nr && cc < nc) tile[ty + 4 * k][tx] = sb[(size_t)rr * M + cc];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int cc = c0 + ty + 4 * k, rr = r0 + tx;                          // dst row = source column
    if (rr < nr && cc < nc) db[(size_t)cc * M + rr] = tile[tx][ty + 4 * k];
  }
}

static inline void skr_launch_block_transpose(const float* src, float* dst, const ttdg_graphs_t& gr, int cmax, bool src_lower, hipStream_t st) {
  if (gr.G < 2) return;
  const int t = (cmax + 63) >> 6;
  hipLaunchKernelGGL(skr_block_transpose_kernel, dim3(t * t, gr.G * (gr.G - 1) / 2), dim3(256), 0, st, src, dst, gr, src_lower ? 1 : 0);
}

// which orientations the pairs a >= b of a batch have (host side)
static inline void skr_orientations(const ttdg_graphs_t& gr, bool& any_plain, bool& any_flip) {
  any_plain = any_flip = false;
  for (int a = 0; a < gr.G; ++a)
    for (int b = 0; b <= a; ++b) {
      const int na = gr.off[a + 1] - gr.off[a], nb = gr.off[b + 1] - gr.off[b];
      if (nb >= na) any_plain = true; else any_flip = true;
    }
}

static inline size_t sk_lds_bytes(int rmax, int cmax, bool mat, int nmat) {
  size_t n = (size_t)(cmax + 1) + cmax + 4;
  if (mat) n += (size_t)nmat * rmax * (cmax | 1);
  return n * sizeof(float);
}
#define SK_LDS_CAP (150 * 1024)

static inline int max_graph(const ttdg_graphs_t& gr) {
  int m = 0;
  for (int g = 0; g < gr.G; ++g) m = gr.off[g + 1] - gr.off[g] > m ? gr.off[g + 1] - gr.off[g] : m;
  return m;
}

extern "C" int ttdg_sinkhorn_pairs_fwd(const float* part, int ksplit, const float* b2, ttdg_graphs_t gr, float tau,
                                       int iters, float* Wds, float* pot, ttdg_stream_t stream) {
  TTDG_REQUIRE(part && Wds && ksplit >= 1 && tau > 0.f, "sinkhorn_pairs_fwd: bad arguments");
  TTDG_REQUIRE(iters >= 0 && iters <= SK_MAXK, "sinkhorn_pairs_fwd: iters out of range");
  if (int e = ttdg_validate_graphs(gr)) return e;
  const int cmax = max_graph(gr);
  const int npairs = gr.G * (gr.G + 1) / 2;
  const bool lds = sk_lds_bytes(cmax, cmax, true, 1) <= SK_LDS_CAP;
  const size_t bytes = sk_lds_bytes(cmax, cmax, lds, 1);
  const int threads = cmax <= 64 ? 256 : 1024;
  hipStream_t st = (hipStream_t)stream;
  if (cmax > 128 && cmax <= SKR_C && ksplit == 1) {
    bool plain, flipped;
    skr_orientations(gr, plain, flipped);
    if (plain) hipLaunchKernelGGL(sinkhorn_pairs_fwd_scale_kernel<false>, dim3(npairs), dim3(SKR_THREADS), 0, st, part, ksplit, b2, gr, tau,
                                  iters, Wds, pot, cmax);
    if (flipped) hipLaunchKernelGGL(sinkhorn_pairs_fwd_scale_kernel<true>, dim3(npairs), dim3(SKR_THREADS), 0, st, part, ksplit, b2, gr, tau,
                                    iters, Wds, pot, cmax);
    skr_launch_block_transpose(Wds, Wds, gr, cmax, true, st);
    return ttdg_launch_status("sinkhorn_pairs_fwd_scale");
  }
  if (lds) {
    TTDG_ALLOW_LDS((sinkhorn_pairs_fwd_kernel<true>), bytes);
    hipLaunchKernelGGL((sinkhorn_pairs_fwd_kernel<true>), dim3(npairs), dim3(threads), bytes, st, part, ksplit, b2, gr, tau,
                       iters, Wds, pot, cmax);
  } else {
    hipLaunchKernelGGL((sinkhorn_pairs_fwd_kernel<false>), dim3(npairs), dim3(threads), bytes, st, part, ksplit, b2, gr,
                       tau, iters, Wds, pot, cmax);
  }
  return ttdg_launch_status("sinkhorn_pairs_fwd");
}

// ---- backward of the pair stage ------------------------------------------------------------------
// dY = dOut * out on the real rows, 0 on the dummy row; for k = K-1 .. 0:
//   line sums  S = sum over the line of dY   (dummy row counted `mult` times in column lines)
//   dY -= exp(y^(k)) * S,   y^(k) = L - f^(k) - g^(k) rebuilt from the logged potentials.
// dL = dY after sweep 0;  dM = dL / tau.
// Backward of one oriented Sinkhorn problem (shared by the pair stage and the stand-alone batched operator).
// pt: logged potentials of the forward (iters x potld); dout(p,q) / dm(p,q) addressed with the given strides.
// smem carve: [f: c+1][g: c][ls: c+1][dd: c][L: r*ldm (kLds)][dY: r*ldm (kLds)]
template <bool kLds>
// dout and dm MAY ALIAS (the scaling-form backward's cold path passes the dM block as both: every thread reads exactly the
// elements it overwrites at the end) - no __restrict__ on either.
__device__ void sk_backward(const SkProb& pb, const float* __restrict__ pt, int potld, const float* dout, int64_t dop,
                            int64_t doq, float* dm, int64_t dmp, int64_t dmq, float* smem, int iters, float tau) {
  const int r = pb.r, c = pb.c, mult = pb.mult;
  const int ldm = c | 1;
  float* f = smem;            // r+1
  float* g = f + c + 1;       // c
  float* ls = g + c;          // line sums, c+1
  float* dd = ls + c + 1;     // dY of the dummy row, c
  float* mat = dd + c;        // L   (kLds)
  float* dy = kLds ? mat + r * ldm : nullptr;  // dY  (kLds) else lives in the dM block
  const int tid = threadIdx.x, nthr = blockDim.x;
  auto DY = [&](int p, int q) -> float& { return kLds ? dy[p * ldm + q] : dm[p * dmp + q * dmq]; };
  auto LV = [&](int p, int q) -> float { return kLds ? mat[p * ldm + q] : sk_load(pb, p, q); };

  // final potentials: last row sweep / last col sweep
  const int klast_row = (iters - 1) & ~1, klast_col = ((iters - 1) & 1) ? iters - 1 : iters - 2;
  for (int p = tid; p <= r; p += nthr) f[p] = (iters >= 1 && (p < r || mult > 0)) ? pt[klast_row * potld + p] : 0.f;
  for (int q = tid; q < c; q += nthr) { g[q] = (klast_col >= 1) ? pt[klast_col * potld + q] : 0.f; dd[q] = 0.f; }
  if (kLds)
    for (int e = tid; e < r * c; e += nthr) { const int p = e / c, q = e - p * c; mat[p * ldm + q] = sk_load(pb, p, q); }
  __syncthreads();
  for (int e = tid; e < r * c; e += nthr) {
    const int p = e / c, q = e - p * c;
    const float out = fast_exp2(LV(p, q) - f[p] - g[q]);
    DY(p, q) = dout[p * dop + q * doq] * out;
  }
  __syncthreads();

  for (int k = iters - 1; k >= 0; --k) {
    const int sg = sk_group((k & 1) ? r : c);
    const int sl = tid & (sg - 1), sgi = tid / sg, nsg = nthr / sg;
    // potentials as of just after sweep k
    if ((k & 1) == 0) { for (int p = tid; p <= r; p += nthr) if (p < r || mult > 0) f[p] = pt[k * potld + p]; }
    else              { for (int q = tid; q < c; q += nthr) g[q] = pt[k * potld + q]; }
    if ((k & 1) == 0) { for (int q = tid; q < c; q += nthr) g[q] = (k >= 1) ? pt[(k - 1) * potld + q] : 0.f; }
    else              { for (int p = tid; p <= r; p += nthr) if (p < r || mult > 0) f[p] = pt[(k - 1) * potld + p]; }
    __syncthreads();
    if ((k & 1) == 0) {
      const int nlines = r + (mult > 0 ? 1 : 0);
      for (int p = sgi; p < nlines; p += nsg) {
        float s = 0.f;
        if (p == r) { for (int q = sl; q < c; q += sg) s += dd[q]; }
        else        { for (int q = sl; q < c; q += sg) s += DY(p, q); }
        s = sub_sum(s, sg);
        if (sl == 0) ls[p] = s;
      }
      __syncthreads();
      for (int e = tid; e < r * c; e += nthr) {
        const int p = e / c, q = e - p * c;
        DY(p, q) -= fast_exp2(LV(p, q) - f[p] - g[q]) * ls[p];
      }
      if (mult > 0)
        for (int q = tid; q < c; q += nthr) dd[q] -= fast_exp2(-f[r] - g[q]) * ls[r];
    } else {
      for (int q = sgi; q < c; q += nsg) {
        float s = 0.f;
        for (int p = sl; p < r; p += sg) s += DY(p, q);
        s = sub_sum(s, sg);
        if (sl == 0) ls[q] = s + (float)mult * dd[q];
      }
      __syncthreads();
      for (int e = tid; e < r * c; e += nthr) {
        const int p = e / c, q = e - p * c;
        DY(p, q) -= fast_exp2(LV(p, q) - f[p] - g[q]) * ls[q];
      }
      if (mult > 0)
        for (int q = tid; q < c; q += nthr) dd[q] -= fast_exp2(-f[r] - g[q]) * ls[q];
    }
    __syncthreads();
  }
  const float inv_tau = 1.f / tau;
  for (int e = tid; e < r * c; e += nthr) {
    const int p = e / c, q = e - p * c;
    dm[p * dmp + q * dmq] = DY(p, q) * inv_tau;
  }
}

template <bool kLds>
__global__ void sinkhorn_pairs_bwd_kernel(const float* __restrict__ part, int ksplit, const float* __restrict__ b2,
                                          const float* __restrict__ pot, const float* __restrict__ dWds,
                                          ttdg_graphs_t gr, float tau, int iters, float* __restrict__ dM, int cmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // pairs with a > b only: index over (1,0),(2,0),(2,1),...
  int a = 1, idx = blockIdx.x;
  while (idx >= a) { idx -= a; ++a; }
  const int b = idx;
  const int pair_fwd = a * (a + 1) / 2 + b;
  const int M = gr.off[gr.G];
  const int na = gr.off[a + 1] - gr.off[a], nb = gr.off[b + 1] - gr.off[b];
  const float* blk = part + (size_t)gr.off[a] * M + gr.off[b];
  const float* dout = dWds + (size_t)gr.off[b] * M + gr.off[a];  // loss reads Wds[b-rows, a-cols] = out^T
  float* dm = dM + (size_t)gr.off[a] * M + gr.off[b];
  SkProb pb;
  pb.src = blk; pb.splane = (int64_t)M * M; pb.nplanes = ksplit;
  pb.bias = b2 ? *b2 : 0.f; pb.scale = TTDG_LOG2E / tau;
  int64_t dop, doq, dmp, dmq;  // strides of dOut / dM in oriented (p,q) coordinates
  if (nb >= na) { pb.r = na; pb.c = nb; pb.sp = M; pb.sq = 1; dop = 1; doq = M; dmp = M; dmq = 1; }
  else          { pb.r = nb; pb.c = na; pb.sp = 1; pb.sq = M; dop = M; doq = 1; dmp = 1; dmq = M; }
  pb.mult = pb.c - pb.r;
  const int potld = cmax + 1;
  sk_backward<kLds>(pb, pot + (size_t)pair_fwd * iters * potld, potld, dout, dop, doq, dm, dmp, dmq, smem, iters, tau);
}

// register-resident backward for 128 < c <= 256: dY lives in registers (16 x 4 per lane, same ownership as the forward
// kernel), L is streamed from L2 every sweep in groups of 4 rows (the next group's 16 loads are in flight under the
// current group's exps), potentials come from the forward log; row sums are DPP wavefront reductions, column sums meet
// in LDS with one barrier per column sweep (double-buffered, every wavefront finishes all columns).
#define SKR_GROUP 4
__device__ __forceinline__ void skr_load_group(const SkProb& pb, int p0, int grp, int r, const unsigned (&qo)[4], float (&dst)[SKR_GROUP][4]) {
#pragma unroll
  for (int a = 0; a < SKR_GROUP; ++a) {
    const char* rowp = reinterpret_cast<const char*>(pb.src + (int64_t)min(p0 + grp * SKR_GROUP + a, r - 1) * pb.sp);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[a][j] = *reinterpret_cast<const float*>(rowp + qo[j]);
  }
}

__global__ __launch_bounds__(SKR_THREADS) void sinkhorn_pairs_bwd_reg_kernel(const float* __restrict__ part, int ksplit,
                                                                             const float* __restrict__ b2, const float* __restrict__ pot,
                                                                             const float* __restrict__ dWds, ttdg_graphs_t gr, float tau,
                                                                             int iters, float* __restrict__ dM, int cmax) {
  __shared__ __attribute__((aligned(16))) float s_part[2][SKR_WAVES * SKR_C];
  int a = 1, idx = blockIdx.x;
  while (idx >= a) { idx -= a; ++a; }
  const int b = idx;
  const int pair_fwd = a * (a + 1) / 2 + b;
  const int M = gr.off[gr.G];
  SkProb pb;
  sk_pair_problem(part, ksplit, b2, gr, tau, a, b, pb);
  const bool flip = pb.sq != 1;
  const float* dout = dWds + (size_t)gr.off[b] * M + gr.off[a];
  float* dm = dM + (size_t)gr.off[a] * M + gr.off[b];
  const int64_t dop = flip ? M : 1, dmp = flip ? 1 : M;
  const int doq = flip ? 1 : M, dmq = flip ? M : 1;
  const int r = pb.r, c = pb.c, mult = pb.mult, potld = cmax + 1;
  const float* pt = pot + (size_t)pair_fwd * iters * potld;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave: provably uniform -> scalar row pointers
  const int p0 = wave * SKR_RW, q0 = lane * 4;
  const float bias = pb.bias, scale = pb.scale;
  unsigned qo[4], qd[4], qm[4];   // 32-bit byte offsets of the lane's columns in L / dOut / dM
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = min(q0 + j, c - 1);
    qo[j] = (unsigned)(q * (int)pb.sq) * 4u; qd[j] = (unsigned)(q * doq) * 4u; qm[j] = (unsigned)(q * dmq) * 4u;
  }

  float f[SKR_RW], g[4], fd, dY[SKR_RW][4], dd[4] = {0.f, 0.f, 0.f, 0.f};
  // potentials of sweep index kf (rows) / kg (cols); a negative index means "still zero"
#define SKR_LOAD_POT(kf, kg)                                                                             \
  {                                                                                                      \
    const float fv = ((kf) >= 0 && lane < SKR_RW && p0 + lane < r) ? pt[(kf) * potld + p0 + lane] : 0.f;     \
    _Pragma("unroll") for (int i = 0; i < SKR_RW; ++i) f[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fv), i)); \
    fd = ((kf) >= 0 && mult > 0) ? pt[(kf) * potld + r] : 0.f;                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) g[j] = ((kg) >= 0 && q0 + j < c) ? pt[(kg) * potld + q0 + j] : 0.f; \
  }
  {
    const int klast_row = (iters - 1) & ~1, klast_col = ((iters - 1) & 1) ? iters - 1 : iters - 2;
    SKR_LOAD_POT(klast_row, (klast_col >= 1 ? klast_col : -1))
  }
  float Lb[2][SKR_GROUP][4];
  // dY = dOut * out on the real rows
  skr_load_group(pb, p0, 0, r, qo, Lb[0]);
#pragma unroll
  for (int grp = 0; grp < SKR_RW / SKR_GROUP; ++grp) {
    if (grp + 1 < SKR_RW / SKR_GROUP) skr_load_group(pb, p0, grp + 1, r, qo, Lb[(grp + 1) & 1]);
#pragma unroll
    for (int a2 = 0; a2 < SKR_GROUP; ++a2) {
      const int i = grp * SKR_GROUP + a2, p = p0 + i;
      const char* drow = reinterpret_cast<const char*>(dout + (int64_t)min(p, r - 1) * dop);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float o = fast_exp2((Lb[grp & 1][a2][j] + bias) * scale - f[i] - g[j]);
        dY[i][j] = (p < r && q0 + j < c) ? *reinterpret_cast<const float*>(drow + qd[j]) * o : 0.f;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  int buf = 0;
  for (int k = iters - 1; k >= 0; --k) {
    if ((k & 1) == 0) { SKR_LOAD_POT(k, (k >= 1 ? k - 1 : -1)) } else { SKR_LOAD_POT(k - 1, k) }
    float de[4];   // exp(y) of the dummy row in the lane's columns
#pragma unroll
    for (int j = 0; j < 4; ++j) de[j] = (mult > 0 && q0 + j < c) ? fast_exp2(-fd - g[j]) : 0.f;      // (fd: logged without the fill)
    float ls[4] = {0.f, 0.f, 0.f, 0.f};
    const bool rows = (k & 1) == 0;
    if (!rows) {
      float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < SKR_RW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[j] += dY[i][j];
      *reinterpret_cast<float4*>(&s_part[buf][wave * SKR_C + q0]) = make_float4(cs[0], cs[1], cs[2], cs[3]);
      __syncthreads();
#pragma unroll
      for (int w = 0; w < SKR_WAVES; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(&s_part[buf][w * SKR_C + q0]);
        ls[0] += v.x; ls[1] += v.y; ls[2] += v.z; ls[3] += v.w;
      }
      buf ^= 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) ls[j] += (float)mult * dd[j];
    }
    skr_load_group(pb, p0, 0, r, qo, Lb[0]);
#pragma unroll
    for (int grp = 0; grp < SKR_RW / SKR_GROUP; ++grp) {
      if (grp + 1 < SKR_RW / SKR_GROUP) skr_load_group(pb, p0, grp + 1, r, qo, Lb[(grp + 1) & 1]);
#pragma unroll
      for (int a2 = 0; a2 < SKR_GROUP; ++a2) {
        const int i = grp * SKR_GROUP + a2;
        if (p0 + i < r) {
          float lrow = 0.f;
          if (rows) lrow = wave_sum_f32_dpp((dY[i][0] + dY[i][1]) + (dY[i][2] + dY[i][3]));
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float e = fast_exp2((Lb[grp & 1][a2][j] + bias) * scale - f[i] - g[j]);
            if (q0 + j < c) dY[i][j] -= e * (rows ? lrow : ls[j]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (rows) {
      if (mult > 0) {
        const float lsd = wave_sum_f32_dpp((dd[0] + dd[1]) + (dd[2] + dd[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) dd[j] -= de[j] * lsd;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) dd[j] -= de[j] * ls[j];
    }
  }
#undef SKR_LOAD_POT
  const float inv_tau = 1.f / tau;
#pragma unroll
  for (int i = 0; i < SKR_RW; ++i) {
    const int p = p0 + i;
    if (p < r) {
      char* mrow = reinterpret_cast<char*>(dm + (int64_t)p * dmp);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (q0 + j < c) *reinterpret_cast<float*>(mrow + qm[j]) = dY[i][j] * inv_tau;
    }
  }
}

// ---- [r5] scaling-form backward for 128 < c <= 256: no dY matrix, one matrix-vector product per sweep ---------------------------
// The recurrence above needs exp(y^(k)) of EVERY sweep: rounds 1-4 rebuilt it with one exponential per entry per sweep from L streamed
// out of L2 twenty times (310 us at 28 blocks of 256 x 256).  In the scaling form (see sinkhorn_pairs_fwd_scale_kernel)
//     P^(k) = exp2(y^(k)) = K o (u^(k) x v^(k)),   K = exp2(L - f0_p)  (one exponential per entry, once),
//     u^(k)_p = exp2(f0_p - f_p),  v^(k)_q = exp2(-g_q)  with (f, g) the logged potentials as of just after sweep k,
// every update is a rank-one multiple of K: dY = K o Z with Z = D - sum_k a^(k) x b^(k), D = dOut o (u_last x v_last) and
//     row sweep k:  S_p = sum_q dY_pq,  a^(k) = u S,   b^(k) = v          column sweep k:  S_q = sum_p dY_pq + mult dd_q,  a^(k) = u,  b^(k) = v S.
// The line sums themselves obey a recurrence that never touches dY: with rs_p = sum_q dY_pq and cs_q = sum over the REAL rows of dY_pq
//     row sweep:     cs_q -= v_q sum_p K_pq (u_p rs_p);   rs_p = 0           (the rows of P^(k) sum to one after a row sweep)
//     column sweep:  rs_p -= u_p sum_q K_pq (v_q S_q);    cs_q -= S_q (1 - mult Pd_q)     (the columns of P^(k), dummy rows included, sum to one)
// and the dummy row (constant fill: K_dq = 1 / c, multiplicity mult) keeps its own row vector dd_q as before.  So a sweep is ONE
// matrix-vector product with K (one FMA per entry) and the state is K alone: the layout of the forward kernel (1024 threads, K in 64
// registers per lane).  The twenty (a, b) pairs are logged in LDS and applied at the end: dM = K o (D - sum_k a^(k) x b^(k)) / tau.
// Range: the products u v enter Z without K, so every potential of the pair is checked up front (|f - f0| and |g| below 60 in log2
// units, all finite); otherwise the workgroup takes sk_backward<false> (log-domain, dY in the dM block) - the cold path.

template <bool kFlip>
__global__ __launch_bounds__(SKR_THREADS) void sinkhorn_pairs_bwd_scale_kernel(const float* __restrict__ part, int ksplit,
                                                                               const float* __restrict__ b2, const float* __restrict__ pot,
                                                                               const float* __restrict__ dWds, ttdg_graphs_t gr, float tau,
                                                                               int iters, float* __restrict__ dM, int cmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_part = smem;                                // [2][wave][256] column partials
  float* s_a = smem + 2 * SKR_WAVES * SKR_C;           // [iters][256]      a^(k)_p
  float* s_b = s_a + iters * SKR_C;                    // [iters][256]      b^(k)_q
  float* s_u = s_b + iters * SKR_C;                    // [iters + 1][256]  u^(k)_p;  index `iters` = the final state
  float* s_v = s_u + (iters + 1) * SKR_C;              // [iters + 1][256]  v^(k)_q
  float* s_ud = s_v + (iters + 1) * SKR_C;             // [iters + 1]       u^(k) of the dummy row
  float* s_col = s_ud + ((iters + 1 + 3) & ~3);        // [256]             column totals of a sweep (skr_column_totals)
  int a = 1, idx = blockIdx.x;
  while (idx >= a) { idx -= a; ++a; }
  const int b = idx;
  const int pair_fwd = a * (a + 1) / 2 + b;
  const int M = gr.off[gr.G];
  SkProb pb;
  sk_pair_problem(part, ksplit, b2, gr, tau, a, b, pb);
  if ((pb.sq != 1) != kFlip) return;
  constexpr bool flip = kFlip;
  // The incoming gradient belongs to Wds(b, a) = out^T; skr_block_transpose_kernel has laid it out like the (a, b) block INSIDE the dM
  // block this workgroup is about to produce (dM is unspecified outside the a > b blocks, and every thread reads exactly the elements
  // it overwrites at the end): dOut and dM share addresses and the orientation of L - every access is a 16-byte piece of a row.
  float* dm = dM + (size_t)gr.off[a] * M + gr.off[b];
  const float* dout = dm;
  const int64_t dmp = flip ? 1 : M, dmq = flip ? M : 1;
  const int64_t dop = dmp, doq = dmq;
  const int r = pb.r, c = pb.c, mult = pb.mult, potld = cmax + 1;
  const float* pt = pot + (size_t)pair_fwd * iters * potld;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int p0 = wave * SKR_RW, q0 = lane * 4;

  // ---- every sweep's multipliers, once: u^(k) = exp2(f0 - f), v^(k) = exp2(-g) with (f, g) as of just after sweep k, and the range
  //      check of every logged potential of this pair (uniform over the workgroup) ----
  int bad = 0;
  for (int e = tid; e < (iters + 1) * SKR_C; e += SKR_THREADS) {
    const int k = e >> 8, i = e & (SKR_C - 1);
    int kf, kg;
    if (k == iters) { kf = (iters - 1) & ~1; kg = ((iters - 1) & 1) ? iters - 1 : iters - 2; }
    else { kf = (k & 1) ? k - 1 : k; kg = (k & 1) ? k : k - 1; }
    float uu = 0.f, vv = 0.f;
    if (i < r) { const float d = pt[i] - pt[kf * potld + i]; bad |= !(fabsf(d) < 60.f); uu = fast_exp2(d); }
    if (i < c) { const float gq = kg >= 1 ? pt[kg * potld + i] : 0.f; bad |= !(fabsf(gq) < 60.f); vv = fast_exp2(-gq); }
    s_u[e] = uu; s_v[e] = vv;
    if (i == 0) {
      float d = 0.f;
      if (mult > 0) { d = pt[r] - pt[kf * potld + r]; bad |= !(fabsf(d) < 60.f); }
      s_ud[k] = fast_exp2(d);
    }
  }
  bad = __syncthreads_or(bad);
  if (bad) {
    sk_backward<false>(pb, pt, potld, dout, dop, doq, dm, dmp, dmq, smem, iters, tau);
    return;
  }
  const float kd = mult > 0 ? 1.f / (float)c : 0.f;
  const int prow0 = p0 + 4 * (lane >> 4);              // this lane's row in register k of the skr_reduce16 layout: prow0 + k
  bool rlive[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) rlive[k] = prow0 + k < r;

  // ---- K = exp2(L - f0) ----
  float A[SKR_RW][4];
  skr_tile_load<!kFlip>(pb.src, pb.sp, pb.sq, p0, q0, r, c, A);
  {
    const float f0v = (lane < SKR_RW && p0 + lane < r) ? pt[p0 + lane] : 0.f;       // lane i: f0 of local row i
#pragma unroll
    for (int i = 0; i < SKR_RW; ++i) {
      const float f0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f0v), i));
#pragma unroll
      for (int j = 0; j < 4; ++j) A[i][j] = (p0 + i < r && q0 + j < c) ? fast_exp2((A[i][j] + pb.bias) * pb.scale - f0) : 0.f;
    }
  }

  float ut[4], v[4], ud;
#define SKB_LOAD_UV(k)                                                                          \
  {                                                                                             \
    _Pragma("unroll") for (int k2 = 0; k2 < 4; ++k2) ut[k2] = s_u[(k) * SKR_C + ((prow0 + k2) & (SKR_C - 1))]; \
    const float4 t_ = *reinterpret_cast<const float4*>(&s_v[(k) * SKR_C + q0]);                 \
    v[0] = t_.x; v[1] = t_.y; v[2] = t_.z; v[3] = t_.w;                                         \
    ud = s_ud[(k)];                                                                             \
  }
  SKB_LOAD_UV(iters)
  // ---- line sums of dY = K o D,  D = dOut o (u_last x v_last):  rs_p (layout of skr_reduce16), cs_q (every wavefront: its 4 columns) ----
  float rs[4] = {0.f, 0.f, 0.f, 0.f}, cs[4], dd[4] = {0.f, 0.f, 0.f, 0.f};
  int buf = 0;
  {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k4 = 0; k4 < SKR_RW / 4; ++k4) {     // four rows at a time: the register peak stays low
      float D[4][4];
      skr_quad_load<!kFlip>(dout, dop, doq, p0 + 4 * k4, q0, r, c, D);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = 4 * k4 + t;
        const float ui = SKR_ROW_SCALAR(ut, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) { D[t][j] = (D[t][j] * (ui * v[j])) * A[i][j]; acc[j] += D[t][j]; }     // K is zero outside the block
        const float srow = wave_sum_f32_dpp((D[t][0] + D[t][1]) + (D[t][2] + D[t][3]));
        rs[i & 3] = ((lane >> 4) == (i >> 2)) ? srow : rs[i & 3];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    *reinterpret_cast<float4*>(&s_part[(buf * SKR_WAVES + wave) * SKR_C + q0]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    skr_column_totals(&s_part[(buf * SKR_WAVES) * SKR_C], s_col, tid, q0, cs);
    buf ^= 1;
  }

  for (int k = iters - 1; k >= 0; --k) {
    SKB_LOAD_UV(k)
    float de[4];                          // P^(k) of the dummy row in the lane's columns
#pragma unroll
    for (int j = 0; j < 4; ++j) de[j] = (kd * ud) * v[j];
    if (k & 1) {
      // ---- column sweep: S_q = cs_q + mult dd_q;  rs_p -= u_p sum_q K_pq (v_q S_q);  cs_q -= S_q (1 - mult Pd_q);  dd_q -= Pd_q S_q ----
      float bq[4], t4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float S = __builtin_fmaf((float)mult, dd[j], cs[j]);
        bq[j] = v[j] * S;
        cs[j] = __builtin_fmaf(-S, 1.f - (float)mult * de[j], cs[j]);
        dd[j] = __builtin_fmaf(-de[j], S, dd[j]);
      }
      skr_reduce16_stream<SkrAdd>([&](int i) { return __builtin_fmaf(A[i][0], bq[0], A[i][1] * bq[1]) + __builtin_fmaf(A[i][2], bq[2], A[i][3] * bq[3]); }, t4);
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        rs[k2] = __builtin_fmaf(-ut[k2], t4[k2], rs[k2]);
        if ((lane & 15) == 0 && rlive[k2]) s_a[k * SKR_C + prow0 + k2] = ut[k2];
      }
      if (wave == 0) *reinterpret_cast<float4*>(&s_b[k * SKR_C + q0]) = make_float4(bq[0], bq[1], bq[2], bq[3]);
    } else {
      // ---- row sweep: a_p = u_p rs_p;  cs_q -= v_q sum_p K_pq a_p;  rs_p = 0;  dummy row: dd_q -= Pd_q sum_q dd_q ----
      float at[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        at[k2] = ut[k2] * rs[k2];
        rs[k2] = 0.f;
        if ((lane & 15) == 0 && rlive[k2]) s_a[k * SKR_C + prow0 + k2] = at[k2];
      }
      if (wave == 0) *reinterpret_cast<float4*>(&s_b[k * SKR_C + q0]) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int i = 0; i < SKR_RW; ++i) {
        const float ai = SKR_ROW_SCALAR(at, i);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(A[i][j], ai, acc[j]);
      }
      *reinterpret_cast<float4*>(&s_part[(buf * SKR_WAVES + wave) * SKR_C + q0]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      __syncthreads();
      float ks[4];
      skr_column_totals(&s_part[(buf * SKR_WAVES) * SKR_C], s_col, tid, q0, ks);
      buf ^= 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) cs[j] = __builtin_fmaf(-v[j], ks[j], cs[j]);
      if (mult > 0) {
        const float lsd = wave_sum_f32_dpp((dd[0] + dd[1]) + (dd[2] + dd[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) dd[j] = __builtin_fmaf(-de[j], lsd, dd[j]);
      }
    }
  }
  __syncthreads();                       // the (a, b) log is complete

  // ---- dM = K o (D - sum_k a^(k) x b^(k)) / tau, four rows at a time ----
  SKB_LOAD_UV(iters)
#undef SKB_LOAD_UV
  const float inv_tau = 1.f / tau;
#pragma unroll
  for (int g4 = 0; g4 < SKR_RW / 4; ++g4) {
    float z[4][4];
    skr_quad_load<!kFlip>(dout, dop, doq, p0 + 4 * g4, q0, r, c, z);
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2) {
      const float ui = SKR_ROW_SCALAR(ut, 4 * g4 + i2);
#pragma unroll
      for (int j = 0; j < 4; ++j) z[i2][j] *= ui * v[j];
    }
    for (int k = iters - 1; k >= 0; --k) {
      const float4 av = *reinterpret_cast<const float4*>(&s_a[k * SKR_C + p0 + 4 * g4]);     // wavefront-uniform address: broadcast
      const float4 bv = *reinterpret_cast<const float4*>(&s_b[k * SKR_C + q0]);
      const float aa[4] = {av.x, av.y, av.z, av.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
        for (int j = 0; j < 4; ++j) z[i2][j] = __builtin_fmaf(-aa[i2], bb[j], z[i2][j]);
    }
    auto dmv = [&](int t, int j) { return (A[4 * g4 + t][j] * z[t][j]) * inv_tau; };
    skr_quad_store<!kFlip>(dm, dmp, dmq, p0 + 4 * g4, q0, r, c, dmv);
    __builtin_amdgcn_sched_barrier(0);     // the next group's dOut loads stay behind this group (register pressure)
  }
}

extern "C" int ttdg_sinkhorn_pairs_bwd(const float* part, int ksplit, const float* b2, const float* pot,
                                       const float* dWds, ttdg_graphs_t gr, float tau, int iters, float* dM,
                                       ttdg_stream_t stream) {
  TTDG_REQUIRE(part && pot && dWds && dM && ksplit >= 1 && tau > 0.f, "sinkhorn_pairs_bwd: bad arguments");
  TTDG_REQUIRE(iters >= 1 && iters <= SK_MAXK, "sinkhorn_pairs_bwd: iters out of range");
  if (int e = ttdg_validate_graphs(gr)) return e;
  if (gr.G < 2) return 0;
  const int cmax = max_graph(gr);
  const int npairs = gr.G * (gr.G - 1) / 2;
  const size_t base = (size_t)(4 * (cmax + 1) + 4) * sizeof(float);
  const size_t need = base + (size_t)2 * cmax * (cmax | 1) * sizeof(float);
  const bool lds = need <= SK_LDS_CAP;
  const size_t bytes = lds ? need : base;
  const int threads = cmax <= 64 ? 256 : 1024;
  hipStream_t st = (hipStream_t)stream;
  if (cmax > 128 && cmax <= SKR_C && ksplit == 1 && iters > 28) {         // the scaling form's per-sweep tables would not fit next to the partials
    hipLaunchKernelGGL(sinkhorn_pairs_bwd_reg_kernel, dim3(npairs), dim3(SKR_THREADS), 0, st, part, ksplit, b2, pot, dWds, gr, tau,
                       iters, dM, cmax);
    return ttdg_launch_status("sinkhorn_pairs_bwd_reg");
  }
  if (cmax > 128 && cmax <= SKR_C && ksplit == 1) {
    const size_t skb = (size_t)(2 * SKR_WAVES * SKR_C + (4 * iters + 2) * SKR_C + 72 + SKR_C) * sizeof(float);   // column partials + the (a, b) log + the (u, v) tables (117 KB at iters = 20); the cold path's 4 (c + 1) floats fit inside
    bool plain, flipped;
    skr_orientations(gr, plain, flipped);
    skr_launch_block_transpose(dWds, dM, gr, cmax, false, st);             // dM(a, b) <- dWds(b, a)^T: the kernels below work in place
    if (plain) {
      TTDG_ALLOW_LDS(sinkhorn_pairs_bwd_scale_kernel<false>, skb);
      hipLaunchKernelGGL(sinkhorn_pairs_bwd_scale_kernel<false>, dim3(npairs), dim3(SKR_THREADS), skb, st, part, ksplit, b2, pot, dWds,
                         gr, tau, iters, dM, cmax);
    }
    if (flipped) {
      TTDG_ALLOW_LDS(sinkhorn_pairs_bwd_scale_kernel<true>, skb);
      hipLaunchKernelGGL(sinkhorn_pairs_bwd_scale_kernel<true>, dim3(npairs), dim3(SKR_THREADS), skb, st, part, ksplit, b2, pot, dWds,
                         gr, tau, iters, dM, cmax);
    }
    return ttdg_launch_status("sinkhorn_pairs_bwd_scale");
  }
  if (lds) {
    TTDG_ALLOW_LDS((sinkhorn_pairs_bwd_kernel<true>), bytes);
    hipLaunchKernelGGL((sinkhorn_pairs_bwd_kernel<true>), dim3(npairs), dim3(threads), bytes, st, part, ksplit, b2, pot, dWds,
                       gr, tau, iters, dM, cmax);
  } else {
    hipLaunchKernelGGL((sinkhorn_pairs_bwd_kernel<false>), dim3(npairs), dim3(threads), bytes, st, part, ksplit, b2, pot,
                       dWds, gr, tau, iters, dM, cmax);
  }
  return ttdg_launch_status("sinkhorn_pairs_bwd");
}

// ---- stand-alone batched operator (GModule.utils.sinkhorn.Sinkhorn.forward) ------------------------
template <bool kLds>
__global__ void sinkhorn_batched_kernel(const float* __restrict__ s, int64_t sb, int64_t sr, int64_t sc, int R, int C,
                                        const int32_t* __restrict__ n1, const int32_t* __restrict__ n2, int dummy,
                                        float tau, int iters, float* __restrict__ out, float* __restrict__ pot) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bi = blockIdx.x;
  // Appendix B steps 1-3: the valid block is nr x nc inside the (R, C) frame; orient it rows <= cols.
  // Whole-batch transposition (C < R) and per-matrix transposition (nr > nc) both reduce to "the shorter
  // valid side becomes the row axis"; on ties the frame orientation decides (rows stay rows).
  int nr = n1 ? n1[bi] : R, nc = n2 ? n2[bi] : C;
  float* o = out + (size_t)bi * R * C;
  for (int e = threadIdx.x; e < R * C; e += blockDim.x) o[e] = 0.f;  // padding -> exp(-inf) = 0
  __syncthreads();
  if (nr <= 0 || nc <= 0) return;
  SkProb pb;
  pb.src = s + bi * sb; pb.splane = 0; pb.nplanes = 1; pb.bias = 0.f; pb.scale = TTDG_LOG2E / tau;
  bool flip = (C < R);            // step 1
  int fr = flip ? nc : nr, fc = flip ? nr : nc;  // valid sizes in the flipped frame
  if (fr > fc) flip = !flip;      // step 3
  if (!flip) { pb.r = nr; pb.c = nc; pb.sp = sr; pb.sq = sc; pb.op = C; pb.oq = 1; }
  else       { pb.r = nc; pb.c = nr; pb.sp = sc; pb.sq = sr; pb.op = 1; pb.oq = C; }
  pb.out = o; pb.mir = nullptr; pb.mp = pb.mq = 0;
  pb.mult = dummy ? pb.c - pb.r : 0;
  pb.potld = (R > C ? R : C) + 1;
  pb.pot = pot ? pot + (size_t)bi * iters * pb.potld : nullptr;
  sk_forward<kLds>(pb, smem, iters);
}

// backward of the stand-alone operator: ds = d loss / d s from dout = d loss / d out and the logged potentials
template <bool kLds>
__global__ void sinkhorn_batched_bwd_kernel(const float* __restrict__ s, int64_t sb, int64_t sr, int64_t sc, int R, int C,
                                            const int32_t* __restrict__ n1, const int32_t* __restrict__ n2, int dummy,
                                            float tau, int iters, const float* __restrict__ pot,
                                            const float* __restrict__ dout, float* __restrict__ ds) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bi = blockIdx.x;
  int nr = n1 ? n1[bi] : R, nc = n2 ? n2[bi] : C;
  float* d = ds + (size_t)bi * R * C;
  for (int e = threadIdx.x; e < R * C; e += blockDim.x) d[e] = 0.f;      // padding takes no gradient
  __syncthreads();
  if (nr <= 0 || nc <= 0) return;
  SkProb pb;
  pb.src = s + bi * sb; pb.splane = 0; pb.nplanes = 1; pb.bias = 0.f; pb.scale = TTDG_LOG2E / tau;
  bool flip = (C < R);
  int fr = flip ? nc : nr, fc = flip ? nr : nc;
  if (fr > fc) flip = !flip;
  int64_t op, oq;      // strides of the contiguous (R, C) gradient buffers in oriented (p, q) coordinates
  if (!flip) { pb.r = nr; pb.c = nc; pb.sp = sr; pb.sq = sc; op = C; oq = 1; }
  else       { pb.r = nc; pb.c = nr; pb.sp = sc; pb.sq = sr; op = 1; oq = C; }
  pb.mult = dummy ? pb.c - pb.r : 0;
  const int potld = (R > C ? R : C) + 1;
  sk_backward<kLds>(pb, pot + (size_t)bi * iters * potld, potld, dout + (size_t)bi * R * C, op, oq, d, op, oq, smem, iters, tau);
}

extern "C" int ttdg_sinkhorn_batched_bwd(const float* s, int64_t sb, int64_t sr, int64_t sc, int b, int r, int c,
                                         const int32_t* n1, const int32_t* n2, int dummy_row, float tau, int iters,
                                         const float* pot, const float* dout, float* ds, ttdg_stream_t stream) {
  TTDG_REQUIRE(s && pot && dout && ds && b >= 0 && r > 0 && c > 0 && tau > 0.f, "sinkhorn_batched_bwd: bad arguments");
  TTDG_REQUIRE(iters >= 1 && iters <= SK_MAXK, "sinkhorn_batched_bwd: iters out of range (the potentials of every sweep are logged)");
  if (b == 0) return 0;
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  const size_t base = (size_t)(4 * (hi + 1) + 4) * sizeof(float);
  const size_t need = base + (size_t)2 * lo * (hi | 1) * sizeof(float);
  const bool lds = need <= SK_LDS_CAP;
  const size_t bytes = lds ? need : base;
  const int threads = hi <= 64 ? 256 : 1024;
  hipStream_t st = (hipStream_t)stream;
  if (lds) {
    TTDG_ALLOW_LDS((sinkhorn_batched_bwd_kernel<true>), bytes);
    hipLaunchKernelGGL((sinkhorn_batched_bwd_kernel<true>), dim3(b), dim3(threads), bytes, st, s, sb, sr, sc, r, c, n1, n2, dummy_row,
                       tau, iters, pot, dout, ds);
  } else {
    hipLaunchKernelGGL((sinkhorn_batched_bwd_kernel<false>), dim3(b), dim3(threads), bytes, st, s, sb, sr, sc, r, c, n1, n2,
                       dummy_row, tau, iters, pot, dout, ds);
  }
  return ttdg_launch_status("sinkhorn_batched_bwd");
}

extern "C" int ttdg_sinkhorn_batched_fwd(const float* s, int64_t sb, int64_t sr, int64_t sc, int b, int r, int c,
                                         const int32_t* n1, const int32_t* n2, int dummy_row, float tau, int iters,
                                         float* out, float* pot, ttdg_stream_t stream) {
  TTDG_REQUIRE(s && out && b >= 0 && r > 0 && c > 0 && tau > 0.f, "sinkhorn_batched: bad arguments");
  TTDG_REQUIRE(iters >= 0 && iters <= 4096, "sinkhorn_batched: iters out of range");
  TTDG_REQUIRE(!pot || iters <= SK_MAXK, "sinkhorn_batched: potentials can be logged for at most 64 sweeps");
  if (b == 0) return 0;
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  const bool lds = sk_lds_bytes(lo, hi, true, 1) <= SK_LDS_CAP;
  const size_t bytes = sk_lds_bytes(lo, hi, lds, 1);
  const int threads = hi <= 64 ? 256 : 1024;
  hipStream_t st = (hipStream_t)stream;
  if (lds) {
    TTDG_ALLOW_LDS((sinkhorn_batched_kernel<true>), bytes);
    hipLaunchKernelGGL((sinkhorn_batched_kernel<true>), dim3(b), dim3(threads), bytes, st, s, sb, sr, sc, r, c, n1, n2,
                       dummy_row, tau, iters, out, pot);
  } else {
    hipLaunchKernelGGL((sinkhorn_batched_kernel<false>), dim3(b), dim3(threads), bytes, st, s, sb, sr, sc, r, c, n1, n2,
                       dummy_row, tau, iters, out, pot);
  }
  return ttdg_launch_status("sinkhorn_batched");
}
