// A4 — learned pairwise affinity, decomposed form (reference utils/affinity.py:44-57).
//
// The reference materialises cat([X'_i ; Y'_j]) as an (n1, n2, 512) tensor and pushes it through
// Linear(512,512)+ReLU+Linear(512,1).  Splitting the first Linear by input half gives the exact identity
//     M_ij = sum_k w2[k] * relu(P[i,k] + Q[j,k]) + b2,   P = X' W1[:, :256]^T,  Q = Y' W1[:, 256:]^T + b1
// (SURVEY.md §8a A4).  P and Q are two small GEMMs (gemm.hip, MFMA); what remains is this pairwise
// reduction, arithmetic intensity ~100 FLOP/B -> fp32 VALU bound, not HBM and not MFMA (relu sits between the two
// contractions).  The forward kernel writes relu(x) = (x + |x|) / 2: the linear half separates,
//     M_ij = sum_k h_k |P_ik + Q_jk|  +  (sum_k h_k P_ik)  +  (sum_k h_k Q_jk) + b2,      h = w2 / 2,
// so the (i,j,k) loop is one packed add (v_pk_add_f32, two k per instruction) and one fma with the |.| source
// modifier: 1.5 VALU instructions per (i,j,k) instead of 3 (add, max, fma); the two row sums ride along with the
// LDS staging (+2 % work).
//
// Forward:  workgroup = 64(i) x 64(j) outputs x one K slice, 256 threads, 4x4 register tile per thread
//           (rows ty+16a, cols tx+16b so that the LDS row stride of 36 words is conflict-free for
//           ds_read_b128), P/Q slabs staged row-major [row][k] in LDS.  K may be split over blockIdx.z
//           into `ksplit` partial planes that the Sinkhorn load sums (deterministic, no atomics) so that
//           a batch of four ~30-node graphs still fills the chip.
// Backward: S[i,k] = sum_j dM[i,j] [P[i,k] + Q[j,k] > 0] and its mirror R[j,k]; same tiling with the
//           roles (row, reduced index) swapped by a template flag; 3 lane-ops per (i,j,k) per pass.
#include "common.h"

#define TILE 64
#define BK 32
#define LDK (BK + 4)  // 36-word row stride: 16 rows x 4 words hit 64 distinct banks
typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc + w * |x| in ONE instruction.  Left to itself hipcc materialises |x| with v_and_b32 so that it can pair the
// multiply-adds into v_pk_fma_f32 (packed operands have no abs modifier): 2 instructions per element instead of 1.5.
// exact 0/1 step of a packed pair: 1 where d > 0 (down to the smallest denormal), else 0 -- two packed multiplies whose
// [0,1] output clamp is a free modifier
__device__ __forceinline__ f32x2 relu_step2(f32x2 d) {
  const f32x2 big = {8.507059173023462e37f, 8.507059173023462e37f};     // 2^126
  f32x2 t;
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(t) : "v"(d), "v"(big));
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(t) : "v"(t), "v"(big));
  return t;
}

__device__ __forceinline__ float fma_abs(float w, float x, float acc) {
  asm("v_fma_f32 %0, %1, |%2|, %0" : "+v"(acc) : "v"(w), "v"(x));
  return acc;
}


__global__ __launch_bounds__(256) void affinity_fwd_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                           const float* __restrict__ w2, int H, ttdg_graphs_t gr,
                                                           int kslice, float* __restrict__ part) {
  const int M = gr.off[gr.G];
  // [r4] compact launch: blockIdx.x enumerates ONLY the wanted tiles, tile row by tile row (row ti wants the column tiles below
  // the end of its last row's graph).  Round 3 launched the full nt x nt grid and let 44 % of the workgroups return at once: all
  // 1024 were placed on the CUs in one static round (capacity 5 per CU), so a CU's share of WORKING tiles ranged from 0 to 4.
  int ti = 0, tj = blockIdx.x;
  for (;; ++ti) {
    const int nc = (gr.off[graph_of(gr, min(ti * TILE + TILE, M) - 1) + 1] + TILE - 1) / TILE;
    if (tj < nc) break;
    tj -= nc;
  }
  const int i0 = ti * TILE, j0 = tj * TILE;
  // [r4] the next K slab travels global -> registers while the current one is consumed from LDS (two LDS buffers, one barrier
  // per slab).  Round 3 loaded, stored, synchronised and only then computed: with ~2 workgroups per CU at cfg-3 (576 tiles on
  // 256 CUs) nobody covered the ~1 us round trip in front of every 1.3 us of arithmetic.  Loads are unconditional (a row
  // beyond M reads row 0 and is zeroed on its way to LDS): a load under a branch is waited for at the join.
  __shared__ __attribute__((aligned(16))) float Ps[2][TILE][LDK];
  __shared__ __attribute__((aligned(16))) float Qs[2][TILE][LDK];
  __shared__ __attribute__((aligned(16))) float Ws[2][BK];
  __shared__ float Arow[TILE], Brow[TILE];     // sum_k h_k P_ik / Q_jk over this K slice
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int kbeg = blockIdx.z * kslice, kend = kbeg + kslice;

  float acc[4][4], tot[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = tot[a][b] = 0.f;
  float pa[2] = {0.f, 0.f}, qa[2] = {0.f, 0.f};

  const int lrow = tid >> 3, lk = (tid & 7) * 4;  // staging map: 8 lanes x float4 cover one 32-wide row
  bool pok[2], qok[2];
  const float* prow[2];
  const float* qrow[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = lrow + 32 * r;
    pok[r] = i0 + row < M;
    qok[r] = j0 + row < M;
    prow[r] = P + (size_t)(pok[r] ? i0 + row : 0) * H + lk;
    qrow[r] = Q + (size_t)(qok[r] ? j0 + row : 0) * H + lk;
  }
  float4 pv[2], qv[2], wv;
#define AFF_LOAD(K0)                                                     \
  {                                                                      \
    wv = *reinterpret_cast<const float4*>(w2 + (K0) + lk);               \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) {                      \
      pv[r] = *reinterpret_cast<const float4*>(prow[r] + (K0));          \
      qv[r] = *reinterpret_cast<const float4*>(qrow[r] + (K0));          \
    }                                                                    \
  }
  AFF_LOAD(kbeg)
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK, buf ^= 1) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = lrow + 32 * r;
      const float4 p4 = pok[r] ? pv[r] : zero4, q4 = qok[r] ? qv[r] : zero4;
      *reinterpret_cast<float4*>(&Ps[buf][row][lk]) = p4;
      *reinterpret_cast<float4*>(&Qs[buf][row][lk]) = q4;
      pa[r] = fmaf(wv.x, p4.x, fmaf(wv.y, p4.y, fmaf(wv.z, p4.z, fmaf(wv.w, p4.w, pa[r]))));
      qa[r] = fmaf(wv.x, q4.x, fmaf(wv.y, q4.y, fmaf(wv.z, q4.z, fmaf(wv.w, q4.w, qa[r]))));
    }
    if (tid < 8) *reinterpret_cast<float4*>(&Ws[buf][lk]) = make_float4(0.5f * wv.x, 0.5f * wv.y, 0.5f * wv.z, 0.5f * wv.w);
    __syncthreads();          // slab k0 is in LDS; the other buffer was last read before the previous barrier
    {
      const int kn = k0 + BK < kend ? k0 + BK : k0;          // past the end: the same slab again (never stored)
      AFF_LOAD(kn)
    }
#pragma unroll 2
    for (int kk = 0; kk < BK; kk += 4) {
      f32x2 p[4][2], q[4][2];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float4 v = *reinterpret_cast<const float4*>(&Ps[buf][ty + 16 * a][kk]);
        p[a][0] = (f32x2){v.x, v.y}; p[a][1] = (f32x2){v.z, v.w};
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float4 v = *reinterpret_cast<const float4*>(&Qs[buf][tx + 16 * b][kk]);
        q[b][0] = (f32x2){v.x, v.y}; q[b][1] = (f32x2){v.z, v.w};
      }
      const float4 w = *reinterpret_cast<const float4*>(&Ws[buf][kk]);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const f32x2 x0 = p[a][0] + q[b][0], x1 = p[a][1] + q[b][1];      // v_pk_add_f32
          acc[a][b] = fma_abs(w.x, x0.x, acc[a][b]);                       // |.| is a free VOP3 source modifier
          acc[a][b] = fma_abs(w.y, x0.y, acc[a][b]);
          acc[a][b] = fma_abs(w.z, x1.x, acc[a][b]);
          acc[a][b] = fma_abs(w.w, x1.y, acc[a][b]);
        }
    }
    if (((k0 - kbeg) / BK & 3) == 3) {
      // [r6] every 128 hidden units the running sums move to a second register set and the chains restart (as in pair_stage.hip: a 512-long fp32
      // chain per entry lost 2 - 3 x what the fp32 reference loses)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { tot[a][b] += acc[a][b]; acc[a][b] = 0.f; }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] += tot[a][b];
#undef AFF_LOAD
  // row sums of the linear half: the 8 lanes that staged a row hold its partial dot products
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { pa[r] += __shfl_xor(pa[r], o, 64); qa[r] += __shfl_xor(qa[r], o, 64); }
    if ((tid & 7) == 0) { Arow[lrow + 32 * r] = 0.5f * pa[r]; Brow[lrow + 32 * r] = 0.5f * qa[r]; }
  }
  __syncthreads();
  float* out = part + (size_t)blockIdx.z * M * M;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = i0 + ty + 16 * a;
    if (i >= M) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = j0 + tx + 16 * b;
      if (j < M) out[(size_t)i * M + j] = acc[a][b] + (Arow[ty + 16 * a] + Brow[tx + 16 * b]);
    }
  }
}

extern "C" int ttdg_affinity_pairwise_fwd(const float* P, const float* Q, const float* w2, int H, ttdg_graphs_t gr,
                                          int ksplit, float* part, ttdg_stream_t stream) {
  TTDG_REQUIRE(P && Q && w2 && part, "affinity_fwd: null pointer");
  if (int e = ttdg_validate_graphs(gr)) return e;
  TTDG_REQUIRE(ksplit >= 1 && H % (ksplit * BK) == 0, "affinity_fwd: H must be a multiple of ksplit*32");
  const int M = gr.off[gr.G];
  const int nt = (M + TILE - 1) / TILE;
  int wanted = 0;                                   // same enumeration as the kernel's
  for (int ti = 0, g = 0; ti < nt; ++ti) {
    const int last = (ti * TILE + TILE < M ? ti * TILE + TILE : M) - 1;
    while (last >= gr.off[g + 1]) ++g;
    wanted += (gr.off[g + 1] + TILE - 1) / TILE;
  }
  hipLaunchKernelGGL(affinity_fwd_kernel, dim3(wanted, 1, ksplit), dim3(256), 0, (hipStream_t)stream, P, Q, w2, H, gr,
                     H / ksplit, part);
  return ttdg_launch_status("affinity_fwd");
}

// ---------------------------------------------------------------------------------------------------
// Backward accumulate:  O[r,k] = sum_c D(r,c) * [X[r,k] + Y[c,k] > 0]
//   kRowsAreSrc = true  (dP pass): r = i (src node), c = j over graphs strictly before graph(i);  D = dM[r][c]
//   kRowsAreSrc = false (dQ pass): r = j (tgt node), c = i over graphs strictly after graph(j);   D = dM[c][r]
// Workgroup = 64 rows x 64 k, thread tile 4 rows (ty*4+a) x 4 k (tx*4+b); the reduced index c is streamed in
// slabs of 32 through LDS: Dt[c][r] (row index contiguous -> one ds_read_b128 per c) and Ys[c][k] = -Y.
// [r5] ONE launch for both passes (blockIdx.y >= nrt: the dQ pass).  The reduced range of a row tile is block-triangular - a dP
// row of graph g reduces over the g graphs before it, a dQ row over the G - 1 - g behind it - so the range of every (pass, row
// tile) is cut into slices of at most `SL` slabs (nearly equal lengths inside a tile) and ONLY those slices exist: blockIdx.z
// beyond a tile's slice count returns at once, planes that no slice writes are never read (the finish kernel derives the same
// counts).  Round 4 cut EVERY tile into the same eight slices: workgroups of 0 .. 7 slabs, and 2 x 8 full planes (64 MB at cfg-3)
// written and read back where 2 x 3.5 carry data.  The two passes' ranges are complementary, so one launch holds the same work
// for every row tile.  The next slab travels global -> registers while the current one is consumed from LDS (two LDS buffers,
// one barrier per slab; unconditional clamped loads, the mask applied on the way to LDS), as in the forward kernel.
#define BC 32
#define LDR (TILE + 4)

// reduced range [cbeg, cend) of row tile `rt` in a pass, and the limit of every row (dP: first excluded index; dQ: first included)
__device__ __forceinline__ void aff_bwd_range(const ttdg_graphs_t& gr, int M, bool rows_are_src, int rt, int& cbeg, int& cend) {
  const int r0 = rt * TILE, rl = min(r0 + TILE, M) - 1;
  if (rows_are_src) { cbeg = 0; cend = gr.off[graph_of(gr, rl)]; }
  else { cbeg = gr.off[graph_of(gr, r0) + 1]; cend = M; }
}
__host__ __device__ __forceinline__ int aff_bwd_nslices(int nslab, int SL) { return (nslab + SL - 1) / SL; }

template <bool kRowsAreSrc>
__device__ __forceinline__ void affinity_bwd_body(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ dM,
                                                  int H, const ttdg_graphs_t& gr, int SL, int rt, float* __restrict__ Opart,
                                                  float (*Dt)[BC][LDR], float (*Ys)[BC][LDR], int* rlim) {
  const int M = gr.off[gr.G];
  const int r0 = rt * TILE, k0 = blockIdx.x * TILE;
  int cbeg, cend;
  aff_bwd_range(gr, M, kRowsAreSrc, rt, cbeg, cend);
  {
    const int nslab = (cend - cbeg + BC - 1) / BC, ns = aff_bwd_nslices(nslab, SL);
    if ((int)blockIdx.z >= ns) return;                       // this slice does not exist (its plane is never read)
    const int per = (nslab + ns - 1) / ns;                   // nearly equal slices inside the tile
    const int b0 = cbeg + blockIdx.z * per * BC;
    cend = min(cend, b0 + per * BC);
    cbeg = b0;
  }
  float* O = Opart + (size_t)blockIdx.z * M * H;             // partial plane of this slice
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;

  if (tid < TILE) {        // rlim, per row: first excluded (dP) / first included (dQ) reduced index
    const int r = r0 + tid;
    int lim = kRowsAreSrc ? 0 : M;
    if (r < M) {
      const int g = graph_of(gr, r);
      lim = kRowsAreSrc ? gr.off[g] : gr.off[g + 1];
    }
    rlim[tid] = lim;
  }
  __syncthreads();

  f32x2 x2[4][2], acc2[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = r0 + ty * 4 + a;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < M) v = *reinterpret_cast<const float4*>(X + (size_t)r * H + k0 + tx * 4);
    x2[a][0] = (f32x2){v.x, v.y}; x2[a][1] = (f32x2){v.z, v.w};
    acc2[a][0] = acc2[a][1] = (f32x2){0.f, 0.f};
  }

  // staging maps.  Y slab: 32 c x 64 k, thread -> (c, c + 16) x one float4 of k.  D slab as Dt[c][r]:
  //   dP: D(r,c) = dM[r][c], global rows are r, contiguous along c -> thread (r, r + 32) x four consecutive c, transposing store;
  //   dQ: D(r,c) = dM[c][r], global rows are c, contiguous along r -> thread r x eight c (coalesced).
  const int yc = tid >> 4, ykq = (tid & 15) * 4;
  const int pr = tid >> 3, pcq = (tid & 7) * 4;             // dP
  const int qr = tid & 63, qcb = tid >> 6;                  // dQ
  float4 yv[2];
  float dv8[8];
  int plim[2] = {0, 0};
  bool prok[2] = {false, false};
  if (kRowsAreSrc) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) { prok[rr] = r0 + pr + 32 * rr < M; plim[rr] = prok[rr] ? rlim[pr + 32 * rr] : 0; }
  }
  const int qlim = rlim[qr];
  const bool qrok = r0 + qr < M;
#define AFFB_LOAD(C0)                                                                                         \
  {                                                                                                            \
    _Pragma("unroll") for (int rr = 0; rr < 2; ++rr)                                                           \
      yv[rr] = *reinterpret_cast<const float4*>(Y + (size_t)min((C0) + yc + 16 * rr, M - 1) * H + k0 + ykq);   \
    if (kRowsAreSrc) {                                                                                         \
      _Pragma("unroll") for (int rr = 0; rr < 2; ++rr)                                                         \
      _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                            \
        dv8[4 * rr + e] = dM[(size_t)min(r0 + pr + 32 * rr, M - 1) * M + min((C0) + pcq + e, M - 1)];          \
    } else {                                                                                                   \
      _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                            \
        dv8[e] = dM[(size_t)min((C0) + qcb + 4 * e, M - 1) * M + min(r0 + qr, M - 1)];                         \
    }                                                                                                          \
  }
  if (cbeg < cend) AFFB_LOAD(cbeg)
  int buf = 0;
  for (int c0 = cbeg; c0 < cend; c0 += BC, buf ^= 1) {
    // registers -> LDS: -Y (relu'(x + y) = [x > -y] exactly: fp32 sums of finite values never round to 0 unless 0), and the D slab
    // masked to the wanted blocks (everything else contributes 0)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const bool ok = c0 + yc + 16 * rr < M;
      const float4 v = yv[rr];
      *reinterpret_cast<float4*>(&Ys[buf][yc + 16 * rr][ykq]) = ok ? make_float4(-v.x, -v.y, -v.z, -v.w) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (kRowsAreSrc) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int e = 0; e < 4; ++e) Dt[buf][pcq + e][pr + 32 * rr] = (prok[rr] && c0 + pcq + e < plim[rr]) ? dv8[4 * rr + e] : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int cg = c0 + qcb + 4 * e;
        Dt[buf][qcb + 4 * e][qr] = (qrok && cg < M && cg >= qlim) ? dv8[e] : 0.f;
      }
    }
    __syncthreads();          // slab c0 is in LDS; the other buffer was last read before the previous barrier
    {
      const int cn = c0 + BC < cend ? c0 + BC : c0;          // past the end: the same slab again (never stored)
      AFFB_LOAD(cn)
    }
#pragma unroll 4
    for (int cc = 0; cc < BC; ++cc) {
      const float4 d = *reinterpret_cast<const float4*>(&Dt[buf][cc][ty * 4]);
      const float4 y = *reinterpret_cast<const float4*>(&Ys[buf][cc][tx * 4]);
      const float dv[4] = {d.x, d.y, d.z, d.w};
      const f32x2 y01 = {y.x, y.y}, y23 = {y.z, y.w};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        // relu'(x + y) = [x > -y] as an exact 0/1 step WITHOUT a compare (no VCC round trip between v_cmp and
        // v_cndmask): step = clamp(clamp((x - (-y)) * 2^126) * 2^126), the clamps being the free [0,1] output modifier
        // of the packed multiply; positive differences down to the smallest denormal saturate to 1, the rest to 0.
        // acc += step * d is then bit-identical to the select-add, in 4 packed instructions per two k.
        acc2[a][0] = __builtin_elementwise_fma(relu_step2(x2[a][0] - y01), (f32x2){dv[a], dv[a]}, acc2[a][0]);
        acc2[a][1] = __builtin_elementwise_fma(relu_step2(x2[a][1] - y23), (f32x2){dv[a], dv[a]}, acc2[a][1]);
      }
    }
  }
#undef AFFB_LOAD
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = r0 + ty * 4 + a;
    if (r < M)
      *reinterpret_cast<float4*>(O + (size_t)r * H + k0 + tx * 4) = make_float4(acc2[a][0].x, acc2[a][0].y, acc2[a][1].x, acc2[a][1].y);
  }
}

// grid (H / 64, 2 * nrt, max slices): blockIdx.y < nrt -> S planes (dP pass: X = P, Y = Q), else R planes (dQ pass: X = Q, Y = P)
__global__ __launch_bounds__(256) void affinity_bwd_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                           const float* __restrict__ dM, int H, ttdg_graphs_t gr, int SL, int nrt,
                                                           float* __restrict__ Spart, float* __restrict__ Rpart) {
  __shared__ __attribute__((aligned(16))) float Dt[2][BC][LDR];
  __shared__ __attribute__((aligned(16))) float Ys[2][BC][LDR];
  __shared__ int rlim[TILE];
  if ((int)blockIdx.y < nrt) affinity_bwd_body<true>(P, Q, dM, H, gr, SL, blockIdx.y, Spart, Dt, Ys, rlim);
  else affinity_bwd_body<false>(Q, P, dM, H, gr, SL, blockIdx.y - nrt, Rpart, Dt, Ys, rlim);
}

// finish, stage 1: S = sum of the partial planes; dP = w2 * S, dQ = w2 * R; per 64-row chunk the partial column sums
// of P*S + Q*R (-> dw2) and of dM over the wanted blocks (-> db2).  Grid (H/64, ceil(M/FR)): a chunk of FR = 16 rows lies inside
// one row tile of the accumulate kernel, so the number of planes that carry its rows is one number per pass (aff_bwd_range /
// aff_bwd_nslices).  [r5] 16 rows per workgroup instead of 64: every wavefront walks its rows one after the other with one L2 / HBM
// round trip per row, and 256 workgroups of sixteen such trips took 56 us for 40 MB at cfg-3; now four trips on 1024 workgroups.
#define FR 16
__global__ __launch_bounds__(256) void affinity_bwd_finish_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                                  const float* __restrict__ w2, const float* __restrict__ dM,
                                                                  int H, ttdg_graphs_t gr, int SL,
                                                                  const float* __restrict__ Spart, const float* __restrict__ Rpart,
                                                                  float* __restrict__ dP, float* __restrict__ dQ,
                                                                  float* __restrict__ dw2part, float* __restrict__ db2part) {
  __shared__ float red[4][64];
  const int M = gr.off[gr.G];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const int m0 = blockIdx.y * FR, m1 = min(M, m0 + FR);
  int nS, nR;
  {
    int cb, ce;
    aff_bwd_range(gr, M, true, m0 / TILE, cb, ce);
    nS = aff_bwd_nslices((ce - cb + BC - 1) / BC, SL);
    aff_bwd_range(gr, M, false, m0 / TILE, cb, ce);
    nR = aff_bwd_nslices((ce - cb + BC - 1) / BC, SL);
  }
  const float w = w2[k];
  const size_t plane = (size_t)M * H;
  float s = 0.f;
  for (int m = m0 + wave; m < m1; m += 4) {
    const size_t o = (size_t)m * H + k;
    // all partial planes of this element in flight together (<= 16 per pass), then a fixed-order sum
    float sp[16], rp[16];
#pragma unroll
    for (int z = 0; z < 16; ++z) {
      sp[z] = (z < nS) ? Spart[z * plane + o] : 0.f;
      rp[z] = (z < nR) ? Rpart[z * plane + o] : 0.f;
    }
    float sv = 0.f, rv = 0.f;
#pragma unroll
    for (int z = 0; z < 16; ++z) { sv += sp[z]; rv += rp[z]; }
    s += P[o] * sv + Q[o] * rv;
    dP[o] = sv * w;
    dQ[o] = rv * w;
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0) dw2part[(size_t)blockIdx.y * H + k] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  {   // db2 partial of this (row chunk, column slice): sum of dM[i, j < off[graph(i)]] over j in the workgroup's slice
    __syncthreads();
    const int jper = (M + (int)gridDim.x - 1) / (int)gridDim.x;
    const int jb = blockIdx.x * jper, je = min(M, jb + jper);
    float t = 0.f;
    for (int m = m0 + wave; m < m1; m += 4) {
      const int lim = min(je, gr.off[graph_of(gr, m)]);
      for (int j = jb + lane; j < lim; j += 64) t += dM[(size_t)m * M + j];
    }
    t = wave_sum(t);
    if (lane == 0) red[wave][0] = t;
    __syncthreads();
    if (threadIdx.x == 0) db2part[blockIdx.y * gridDim.x + blockIdx.x] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
  }
}

// finish, stage 2: fixed-order sums over the row chunks.  Workgroup = 64 k x 4 wavefronts: wavefront w adds the chunks w, w + 4, ...
// in order, the four partials meet in LDS in a fixed tree; db2 (workgroup 0): thread t adds partials t, t + 256, ..., then a
// fixed tree over the 256 threads.  Deterministic, and no thread walks more than nchunk / 4 dependent loads.
__global__ __launch_bounds__(256) void affinity_bwd_reduce_kernel(const float* __restrict__ dw2part, const float* __restrict__ db2part,
                                                                 int H, int nchunk, float* __restrict__ dw2, float* __restrict__ db2) {
  __shared__ float red[4][64];
  __shared__ float redb[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (k < H)
    for (int c = wave; c < nchunk; c += 4) s += dw2part[(size_t)c * H + k];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && k < H) dw2[k] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  if (blockIdx.x == 0) {
    float t = 0.f;
    const int np = nchunk * (H / 64);                                  // one partial per (row chunk, column slice)
    for (int c = threadIdx.x; c < np; c += 256) t += db2part[c];
    redb[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) redb[threadIdx.x] += redb[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) *db2 = redb[0];
  }
}

// planes the workspace holds per pass (an upper bound of every tile's slice count: the slice length is chosen accordingly)
static int affinity_bwd_nsplit(int M, int H) {
  const int base = (H / TILE) * ((M + TILE - 1) / TILE);
  int ns = (2048 + base - 1) / base;                 // aim at ~8 workgroups per CU
  const int maxs = (M + BC - 1) / BC;                // at least one slab per slice
  if (ns > maxs) ns = maxs;
  if (ns > 16) ns = 16;
  return ns < 1 ? 1 : ns;
}

extern "C" size_t ttdg_affinity_bwd_workspace_bytes(int M, int H) {
  const int ns = affinity_bwd_nsplit(M, H), nchunk = (M + FR - 1) / FR;
  return ((size_t)2 * ns * M * H + (size_t)nchunk * H + (size_t)nchunk * (H / TILE) + 16) * sizeof(float);
}

extern "C" int ttdg_affinity_pairwise_bwd(const float* P, const float* Q, const float* w2, const float* dM, int H,
                                          ttdg_graphs_t gr, float* dP, float* dQ, float* dw2, float* db2, void* ws,
                                          ttdg_stream_t stream) {
  TTDG_REQUIRE(P && Q && w2 && dM && dP && dQ && dw2 && db2 && ws, "affinity_bwd: null pointer");
  if (int e = ttdg_validate_graphs(gr)) return e;
  TTDG_REQUIRE(H % TILE == 0, "affinity_bwd: H must be a multiple of 64");
  const int M = gr.off[gr.G];
  const int nsmax = affinity_bwd_nsplit(M, H), nchunk = (M + FR - 1) / FR, nrt = (M + TILE - 1) / TILE, nkt = H / TILE;
  // slice length (slabs): ~7 slices of work per CU over both passes, and no tile with more slices than the workspace has planes
  // (host-side restatement of aff_bwd_range)
  long total = 0;
  int longest = 0;
  for (int rt = 0, g0 = 0, g1 = 0; rt < nrt; ++rt) {
    const int r0 = rt * TILE, rl = (r0 + TILE < M ? r0 + TILE : M) - 1;
    while (r0 >= gr.off[g0 + 1]) ++g0;
    while (rl >= gr.off[g1 + 1]) ++g1;
    const int sp = (gr.off[g1] + BC - 1) / BC, sq = (M - gr.off[g0 + 1] + BC - 1) / BC;
    total += sp + sq;
    longest = sp > longest ? sp : longest;
    longest = sq > longest ? sq : longest;
  }
  int SL = (int)((total * nkt + 1791) / 1792);
  if (SL < (longest + nsmax - 1) / nsmax) SL = (longest + nsmax - 1) / nsmax;
  if (SL < 1) SL = 1;
  const int nz = longest > 0 ? (longest + SL - 1) / SL : 1;
  float* Spart = (float*)ws;
  float* Rpart = Spart + (size_t)nsmax * M * H;
  float* dw2part = Rpart + (size_t)nsmax * M * H;
  float* db2part = dw2part + (size_t)nchunk * H;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(affinity_bwd_kernel, dim3(nkt, 2 * nrt, nz), dim3(256), 0, st, P, Q, dM, H, gr, SL, nrt, Spart, Rpart);
  hipLaunchKernelGGL(affinity_bwd_finish_kernel, dim3(H / 64, nchunk), dim3(256), 0, st, P, Q, w2, dM, H, gr, SL, Spart, Rpart,
                     dP, dQ, dw2part, db2part);
  hipLaunchKernelGGL(affinity_bwd_reduce_kernel, dim3((H + 63) / 64), dim3(256), 0, st, dw2part, db2part, H, nchunk, dw2, db2);
  return ttdg_launch_status("affinity_bwd");
}
