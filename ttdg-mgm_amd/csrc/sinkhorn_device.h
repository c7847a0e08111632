// Device-side log-Sinkhorn on dual potentials, shared by the pair stage / stand-alone operator (sinkhorn.hip) and
// the large-graph GA-MGM solver (gagm_large.hip).  Formulation and citations: see the header of sinkhorn.hip.
#pragma once
#include "common.h"
#include "lap_device.h"   // DPP row operations


#define SK_MAXK 64
#define SK_DUMMY (-100.0f * TTDG_LOG2E)

struct SkProb {
  // oriented problem: r <= c; element (p,q) of the input is sum_s src[s*splane + p*sp + q*sq] + bias
  const float* src;
  int64_t sp, sq, splane;
  int nplanes;
  float bias, scale;  // L2 = (x + bias) * scale, scale = log2(e)/tau
  int r, c, mult;     // mult = number of dummy rows (0 when dummy_row is off)
  float* out;         // out[p*op + q*oq] = exp(y)
  int64_t op, oq;
  float* mir;         // optional mirror (transposed copy), may be null
  int64_t mp, mq;
  float* pot;         // optional potentials log: pot[k*(cmax+1) + idx]
  int potld;
};

__device__ __forceinline__ float sk_load(const SkProb& pb, int p, int q) {
  float v = pb.bias;
  const float* s = pb.src + p * pb.sp + q * pb.sq;
  int k = 0;
  // eight planes requested before the first addition (the plain loop pays one memory round trip per plane; pair stage of a
  // TTA step 117 -> 102 us); the additions keep their order: bit-identical sums
  for (; k + 8 <= pb.nplanes; k += 8) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = s[(int64_t)(k + j) * pb.splane];
#pragma unroll
    for (int j = 0; j < 8; ++j) v += t[j];
  }
  for (; k < pb.nplanes; ++k) v += s[k * pb.splane];
  return v * pb.scale;
}

// Reductions over aligned sub-groups of sg = 16 / 32 / 64 lanes on DPP row operations (no ds_bpermute round trips):
// four in-row stages leave every lane of a 16-lane row with the row total; row_bcast15 / row_bcast31 extend it to the
// last lane of a 32 / 64 group, from where it is broadcast with v_readlane.  Result in every lane of the sub-group.
// Lanes of other sub-groups may be masked off (ragged line counts): every source lane is in the reader's own sub-group.
__device__ __forceinline__ float sub_bcast(float v, int sg) {
  if (sg == 64) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  if (sg == 32) {
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return (threadIdx.x & 32) ? b : a;
  }
  return v;
}
__device__ __forceinline__ float sub_max(float v, int sg) {
#define OP(C, R) v = fmaxf(v, __int_as_float(dpp_mov<C, R>(__float_as_int(v))));
  OP(0xB1, 0xf) OP(0x4E, 0xf) OP(0x141, 0xf) OP(0x140, 0xf)
  if (sg >= 32) { OP(0x142, 0xa) }
  if (sg == 64) { OP(0x143, 0xc) }
#undef OP
  return sub_bcast(v, sg);
}
__device__ __forceinline__ float sub_sum(float v, int sg) {
#define OP(C, R) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), C, R, 0xf, false));
  OP(0xB1, 0xf) OP(0x4E, 0xf) OP(0x141, 0xf) OP(0x140, 0xf)
  if (sg >= 32) { OP(0x142, 0xa) }
  if (sg == 64) { OP(0x143, 0xc) }
#undef OP
  return sub_bcast(v, sg);
}
__device__ __forceinline__ int sk_group(int len) { return len > 32 ? 64 : (len > 16 ? 32 : 16); }

// LDS carve (dynamic): [f: cmax+1][g: cmax][mat: r*ldm  (kLds only)]
template <bool kLds>
__device__ void sk_forward(const SkProb& pb, float* smem, int iters) {
  const int r = pb.r, c = pb.c, mult = pb.mult;
  float* f = smem;           // r real rows + 1 dummy
  float* g = smem + c + 1;   // c
  float* mat = g + c;        // r x ldm
  const int ldm = c | 1;     // odd stride: column walks are conflict-free
  const int tid = threadIdx.x, nthr = blockDim.x;

  for (int q = tid; q < c; q += nthr) g[q] = 0.f;
  for (int p = tid; p <= r; p += nthr) f[p] = 0.f;
  if (kLds)
    for (int e = tid; e < r * c; e += nthr) {
      const int p = e / c, q = e - p * c;
      mat[p * ldm + q] = sk_load(pb, p, q);
    }
  __syncthreads();

  for (int it = 0; it < iters; ++it) {
    const int sg = sk_group((it & 1) ? r : c);             // lanes per line: rows hold c entries, columns r
    const int sl = tid & (sg - 1), sgi = tid / sg, nsg = nthr / sg;
    if ((it & 1) == 0) {
      // rows: f_p = lse_q(L_pq - g_q); the dummy row uses the constant fill
      const int nlines = r + (mult > 0 ? 1 : 0);
      for (int p = sgi; p < nlines; p += nsg) {
        const bool dum = (p == r);
        float m = -INFINITY;
        // [r6] the dummy row is swept WITHOUT its constant (t = -g_q; the fill cancels in L - f): f[r] holds lse_q(-g_q), and that is what is logged.  Rounds 1-5 formed SK_DUMMY - g_q and SK_DUMMY - f[r]: two round trips through |144| per sweep pair, ulp(144) = 1.5e-5
        // relative on the dummy mass of every column sum (profiles/r06_pair_stage_accuracy.txt).
        for (int q = sl; q < c; q += sg) {
          const float t = (dum ? 0.f : (kLds ? mat[p * ldm + q] : sk_load(pb, p, q))) - g[q];
          m = fmaxf(m, t);
        }
        m = sub_max(m, sg);
        float s = 0.f;
        for (int q = sl; q < c; q += sg) {
          const float t = (dum ? 0.f : (kLds ? mat[p * ldm + q] : sk_load(pb, p, q))) - g[q];
          s += fast_exp2(t - m);
        }
        s = sub_sum(s, sg);
        if (sl == 0) {
          const float v = m + fast_log2(s);
          f[p] = v;
          if (pb.pot) pb.pot[it * pb.potld + p] = v;       // (the dummy row's entry: its potential without the fill, as every backward reads it)
        }
      }
    } else {
      // cols: g_q = lse over the r real rows and `mult` copies of the dummy row
      const float td0 = -f[r];                               // (f[r] is the dummy row's potential without the fill: see the row sweep)
      for (int q = sgi; q < c; q += nsg) {
        float m = (mult > 0) ? td0 : -INFINITY;
        for (int p = sl; p < r; p += sg) m = fmaxf(m, (kLds ? mat[p * ldm + q] : sk_load(pb, p, q)) - f[p]);
        m = sub_max(m, sg);
        float s = 0.f;
        for (int p = sl; p < r; p += sg) s += fast_exp2((kLds ? mat[p * ldm + q] : sk_load(pb, p, q)) - f[p] - m);
        s = sub_sum(s, sg);
        if (mult > 0) s += (float)mult * fast_exp2(td0 - m);
        if (sl == 0) {
          const float v = m + fast_log2(s);
          g[q] = v;
          if (pb.pot) pb.pot[it * pb.potld + q] = v;
        }
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < r * c; e += nthr) {
    const int p = e / c, q = e - p * c;
    const float y = (kLds ? mat[p * ldm + q] : sk_load(pb, p, q)) - f[p] - g[q];
    const float v = fast_exp2(y);
    pb.out[p * pb.op + q * pb.oq] = v;
    if (pb.mir) pb.mir[p * pb.mp + q * pb.mq] = v;
  }
}

