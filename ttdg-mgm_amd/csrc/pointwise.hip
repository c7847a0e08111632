// Streaming fp32 GEMM for the backbone's pointwise (1 x 1) convolutions in channels-last memory, with the FrozenBN shift, the residual
// sum and the ReLU applied where the accumulators leave the matrix cores (rcnn.py:331-345 runs these layers through detectron2's
// Conv2d + FrozenBatchNorm2d + F.relu_ [3P]; trainer.py:469-485 runs them twice per adapted batch).
//
//   C[m, n] = act( sum_k A'(m, k) B(n, k) + bias[n] + (res[r(m), n] + bias2[n]) ),   A'(m, k) = A(m, k) or relu(A(m, k) + pbias[k])
//
// In NHWC a stride-1 pointwise convolution IS this product (m = pixel, k = input channel, n = output channel); a stride-2 one
// reads every other pixel of every other row (row map `a_map`), the FPN's top-down sum reads its residual from the coarser level
// (row map `r_up`: nearest-neighbour 2x), and the two backward products are the same kernel with other operand layouts:
//   dX = dY W            A = dY (k-contiguous), B(n, k) = W[k, n] (n-contiguous)
//   dW = dY^T X          A(n, m) = dY[m, n], B(k, m) = X[m, k] (both: the reduction index is the strided one), split over pixels.
//
// Design (gfx950):
//  * v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles per instruction per SIMD = the fp32 peak).  256 threads = 4 wavefronts in a
//    2 x 2 arrangement over a BM x BN tile (128 x 128: every wavefront owns 2 x 2 accumulator blocks of 32 x 32 = 64 registers);
//    two workgroups per CU (64 KB of LDS each), so one workgroup's barrier / epilogue hides behind the other's MFMAs.
//  * operands go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), one 32-deep K slab per
//    stage, two stages; ONE barrier per slab.  A wave-instruction writes 1 KiB lane-linearly, so the LDS image is chosen through
//    the per-lane SOURCE address:
//      - k-contiguous operand: image [row][8 chunks of 16 B], chunk c of row r stored at position c ^ ((r >> 1) & 7).  A lane
//        reads its fragment with ds_read_b128 (four consecutive k of one row): the 16 lanes the LDS serves per cycle then sit on
//        16 distinct 16-byte slots (conflict-free; unswizzled it is 16-way).  Eight lanes still read one whole 128-byte line.
//      - output-index-contiguous operand: image [k][R], fragments by ds_read_b32 (32 consecutive words per half wavefront).
//    The k order inside an accumulator is therefore (per 32-slab) 0-3, 8-11, 16-19, 24-27 interleaved with 4-7, ... in pairs -
//    a fixed order, the same for every launch (deterministic), not the ascending order of gemm.hip.
//  * the epilogue goes through LDS (64 rows at a time): 32 lanes write one 512-byte row segment with 16-byte stores; the residual
//    is read the same way; bias / residual / ReLU cost no extra pass over HBM.
//  * tiles are dealt to the XCDs in contiguous runs with the n tiles of one m tile adjacent: the A panel of an m tile is fetched
//    into ONE L2.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MM_BK 32
#define MM_PRO_MAXK 2048      /* longest reduction the fused input activation takes (its shift vector is staged in LDS) */

struct mm_args {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* res;
  const float* bias2;
  const float* pbias;
  int64_t lda, ldb, ldc, ldres;
  int M, N, K;
  int a_map, a_howo, a_wo, a_hw, a_w, a_s;   // A row m = (img, ho, wo) of an (Ho, Wo) map -> pixel (img, s ho, s wo) of the (H, W) input
  int r_up, r_hw, r_w;                       // residual row of pixel (img, h, w) of an (H, W) map: (img, h / 2, w / 2) of the (H/2, W/2) map
  int relu, prelu;
  int kslices, kc;                           // split of the reduction over blockIdx.y (kc: multiple of MM_BK); partial planes in `part`
  float* part;
  // second reduction segment (SEG2 kernels): + sum_k A2(m, k) B2(n, k), both k-contiguous, A2 with its own strided row map - the
  // projection shortcut of a bottleneck's first block accumulated into conv3's product (its output is never written or re-read)
  const float* A2;
  const float* B2;
  int64_t lda2, ldb2;
  int K2;
  int a2_map, a2_howo, a2_wo, a2_hw, a2_w, a2_s;
  int dbg;                                   // ablation (tile field bits 8-9; WRONG RESULTS): 1 = only the first slab is loaded, 2 = no MFMA
  // split output (C2 != NULL): columns [nsplit, N) go to C2[m * ldc2 + (n - nsplit)] - two heads that read one input in one pass
  float* C2;
  int64_t ldc2;
  int nsplit;
};

__device__ __forceinline__ void mm_glds16(const float* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// rows / columns of operand X that this lane stages, fixed for the whole K loop
template <int R, bool KC>
struct mm_loader {
  static constexpr int IPW = R / 32;                 // wave-instructions per stage and wavefront
  const float* p[IPW];                               // KC: row pointer + swizzled chunk; !KC: column pointer (k row added per slab)
  int krow[IPW];                                     // !KC: k row inside the slab
  int kchunk;                                        // KC: source chunk (0..7) of this lane
  int64_t ld;
  __device__ __forceinline__ void init(const float* X, int64_t ld_, int r0, int Rlim, int wave, int lane, int map, int howo, int wo, int hw, int w_, int st) {
    ld = ld_;
    if (KC) {
      const int sw = (wave * 4 + (lane >> 4)) & 7;
      kchunk = (lane & 7) ^ sw;
#pragma unroll
      for (int q = 0; q < IPW; ++q) {
        int r = r0 + (q * 4 + wave) * 8 + (lane >> 3);
        r = r < Rlim ? r : Rlim - 1;                 // beyond the edge: a valid row whose results are never stored
        int64_t row = r;
        if (map) {
          const int img = r / howo, rem = r - img * howo, ho = rem / wo, wq = rem - ho * wo;
          row = (int64_t)img * hw + (int64_t)(ho * st) * w_ + wq * st;
        }
        p[q] = X + row * ld;
      }
    } else {
      constexpr int CPR = R / 4;                     // 16-byte chunks per k row
      constexpr int KPI = 64 / CPR;                  // k rows per wave-instruction
      int c = r0 + 4 * (lane % CPR);
      c = c < Rlim ? c : (Rlim - 4 > 0 ? Rlim - 4 : 0);
#pragma unroll
      for (int q = 0; q < IPW; ++q) {
        krow[q] = (q * 4 + wave) * KPI + lane / CPR;
        p[q] = X + c;
      }
      kchunk = 0;
    }
  }
  // one stage: slab [k0, k0 + 32) -> lds (R * 32 floats)
  __device__ __forceinline__ void issue(float* lds, int k0, int Klim, int wave) const {
#pragma unroll
    for (int q = 0; q < IPW; ++q) {
      float* dst = lds + (q * 4 + wave) * 256;
      if (KC) {
        int k = k0 + 4 * kchunk;
        k = k < Klim ? k : 0;                        // (K % 4 == 0) a chunk is inside or outside; outside: finite filler, zeroed in the fragment
        mm_glds16(p[q] + k, dst);
      } else {
        int k = k0 + krow[q];
        k = k < Klim ? k : Klim - 1;
        mm_glds16(p[q] + (int64_t)k * ld, dst);
      }
    }
  }
};

// [r6] Measured and NOT adopted: requesting group q + 1's fragments under group q's MFMAs (two register sets, order pinned by sched_barrier,
// exact lgkmcnt waits): 61.2 vs 59.0 us on 40000 x 512 x 128 - with the loads of every slab but the first REMOVED the product still takes
// 54 of its 59 us (profiles/r06_pointwise_ablation.txt): five co-resident workgroups per CU already cover each other's LDS round trips, the K
// loop runs at the rate the matrix pipe sustains at this clock.
// the 64 (128 x 128 tile) MFMAs of one 32-deep slab: four groups of eight k; in a group the lower half wavefront owns k 8q .. 8q+3,
// the upper half 8q+4 .. 8q+7 (one ds_read_b128 per 32 x 32 block and operand)
template <int BM, int BN, bool AKC, bool BKC, bool PRO, bool TAIL>
__device__ __forceinline__ void mm_slab(const float* As, const float* Bs, const int* fa, const int* fb, int sw, int h, int k0, int kend,
                                        const mm_args& p, const float* pb_lds, f32x16 (*acc)[BN / 64]) {
  constexpr int MB = BM / 64, NB = BN / 64;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float af[MB][4], bf[NB][4];
    const int kq = 4 * (2 * q + h);                    // this lane's four k of the group
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      if (AKC) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(As + fa[b] + 4 * ((2 * q + h) ^ sw));
        af[b][0] = v.x, af[b][1] = v.y, af[b][2] = v.z, af[b][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) af[b][j] = As[(kq + j) * BM + fa[b]];
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (BKC) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(Bs + fb[b] + 4 * ((2 * q + h) ^ sw));
        bf[b][0] = v.x, bf[b][1] = v.y, bf[b][2] = v.z, bf[b][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[b][j] = Bs[(kq + j) * BN + fb[b]];
      }
    }
    if (PRO) {                                         // fused input activation: relu(A + shift[k]); the shift vector sits in LDS
      const f32x4 pb = *reinterpret_cast<const f32x4*>(pb_lds + (TAIL ? min(k0 + kq, p.K - 4) : k0 + kq));
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        af[b][0] += pb.x, af[b][1] += pb.y, af[b][2] += pb.z, af[b][3] += pb.w;
        if (p.prelu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) af[b][j] = fmaxf(af[b][j], 0.f);
        }
      }
    }
    if (TAIL) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool dead = k0 + kq + j >= kend;
#pragma unroll
        for (int b = 0; b < MB; ++b) af[b][j] = dead ? 0.f : af[b][j];
#pragma unroll
        for (int b = 0; b < NB; ++b) bf[b][j] = dead ? 0.f : bf[b][j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int bm = 0; bm < MB; ++bm)
#pragma unroll
        for (int bn = 0; bn < NB; ++bn) acc[bm][bn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[bm][j], bf[bn][j], acc[bm][bn], 0, 0, 0);
  }
}

// one (tile, reduction slice) of one product: `Lraw` of `T` tiles (dealt to the XCDs below), slice `zslice` of the pixel split
template <int BM, int BN, bool AKC, bool BKC, bool PRO, bool LONGK, bool SEG2>
__device__ __forceinline__ void mm_body(const mm_args& p, int Lraw, int T, int zslice) {
  constexpr int WM = BM / 2, WN = BN / 2, MB = WM / 32, NB = WN / 32;
  constexpr int ASZ = BM * MM_BK, BSZ = BN * MM_BK, STAGE = ASZ + BSZ;
  constexpr int CT_LD = BN + 4;
  constexpr int EPI = 64 * CT_LD;
  constexpr int PB = PRO ? MM_PRO_MAXK : 0;              // the input shift vector (PRO) lives behind the stages
  constexpr int SMEM = (2 * STAGE > EPI ? 2 * STAGE : EPI) + PB;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];     // the ONLY LDS object (a second one makes hipcc drain the DMA queue before every ds_read)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int h = lane >> 5, l32 = lane & 31;

  // tile of this workgroup: XCD x gets a contiguous run of the (m tile, n tile) list, n fastest
  const int NT = (p.N + BN - 1) / BN;
  int L = Lraw;
  {
    const int q = T >> 3, r = T & 7, x = L & 7, i = L >> 3;
    L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int mt = L / NT, nt = L - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;

  const int kbeg = p.part ? zslice * p.kc : 0;
  const int kend = p.part ? min(p.K, kbeg + p.kc) : p.K;
  const int nk1 = (kend - kbeg + MM_BK - 1) / MM_BK;
  const int nk = nk1 + (SEG2 ? p.K2 / MM_BK : 0);          // (SEG2: K and K2 are multiples of the slab depth, no split)

  mm_loader<BM, AKC> la;
  mm_loader<BN, BKC> lb;
  la.init(p.A, p.lda, m0, p.M, wave, lane, p.a_map, p.a_howo, p.a_wo, p.a_hw, p.a_w, p.a_s);
  lb.init(p.B, p.ldb, n0, p.N, wave, lane, 0, 1, 1, 0, 0, 1);
  mm_loader<BM, true> la2;
  mm_loader<BN, true> lb2;
  if (SEG2) {
    la2.init(p.A2, p.lda2, m0, p.M, wave, lane, p.a2_map, p.a2_howo, p.a2_wo, p.a2_hw, p.a2_w, p.a2_s);
    lb2.init(p.B2, p.ldb2, n0, p.N, wave, lane, 0, 1, 1, 0, 0, 1);
  }

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x16 tot[LONGK ? MB : 1][LONGK ? NB : 1];
  if (LONGK) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) tot[i][j][e] = 0.f;
  }

  // fragment addresses (floats, inside a stage)
  int fa[MB], fb[NB];
  const int sw = (l32 >> 1) & 7;
#pragma unroll
  for (int b = 0; b < MB; ++b) fa[b] = AKC ? (wr * WM + 32 * b + l32) * 32 : wr * WM + 32 * b + l32;
#pragma unroll
  for (int b = 0; b < NB; ++b) fb[b] = BKC ? (wc * WN + 32 * b + l32) * 32 : wc * WN + 32 * b + l32;

  const float* pb_lds = smem + SMEM - PB;
  if (PRO) {
    for (int k = tid; k < p.K; k += 256) smem[SMEM - PB + k] = p.pbias[k];
    // (made visible by the first barrier of the K loop; never overwritten: the epilogue buffer ends below it)
  }
  if (nk > 0) {
    la.issue(smem, kbeg, kend, wave);
    lb.issue(smem + ASZ, kbeg, kend, wave);
  }
  for (int t = 0; t < nk; ++t) {
    const float* As = smem + (t & 1) * STAGE;
    const float* Bs = As + ASZ;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wavefront's share of slab t has landed
    __syncthreads();                                     // ... everybody's has, and nobody still reads the other stage
    const int k0 = kbeg + t * MM_BK;
    if (t + 1 < nk && !(p.dbg & 1)) {
      float* nxt = smem + ((t + 1) & 1) * STAGE;
      if (!SEG2 || t + 1 < nk1) {
        la.issue(nxt, k0 + MM_BK, kend, wave);
        lb.issue(nxt + ASZ, k0 + MM_BK, kend, wave);
      } else {
        la2.issue(nxt, (t + 1 - nk1) * MM_BK, p.K2, wave);
        lb2.issue(nxt + ASZ, (t + 1 - nk1) * MM_BK, p.K2, wave);
      }
    }
    // ragged end of the reduction (only the last slab can be ragged): the filler is zeroed in registers
    if (p.dbg & 2) {
    } else if (SEG2 && t >= nk1) mm_slab<BM, BN, true, true, false, false>(As, Bs, fa, fb, sw, h, 0, MM_BK, p, pb_lds, acc);   // (no input activation on the second segment)
    else if (!SEG2 && kend - k0 < MM_BK) mm_slab<BM, BN, AKC, BKC, PRO, true>(As, Bs, fa, fb, sw, h, k0, kend, p, pb_lds, acc);
    else mm_slab<BM, BN, AKC, BKC, PRO, false>(As, Bs, fa, fb, sw, h, k0, kend, p, pb_lds, acc);
    if (LONGK && (t & 7) == 7) {
      // long reductions: every 256 k the running sums move to a second set of registers and the chains restart, so a chain is
      // 256 products long whatever K is (rounding of a 2048-long fp32 chain measured 4x a vendor kernel's against float64)
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          tot[i][j] += acc[i][j];
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
    }
  }
  if (LONGK) {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] += tot[i][j];
  }

  // ---- epilogue: 64 tile rows at a time through LDS; C/D fragment: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float* Ct = smem;
  const int cq = 4 * (tid & (BN / 4 - 1));               // column quad of this thread inside the tile
  constexpr int RPP = 256 / (BN / 4);                    // rows covered by the 256 threads in one sweep
  const int rsub = tid / (BN / 4);
  const bool ncol = n0 + cq < p.N;                       // (N % 4 == 0: a quad is inside or outside)
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (!p.part && ncol) {
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + cq);
    if (p.bias2) {
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.bias2 + n0 + cq);
      bv += b2;
    }
  }
#pragma unroll
  for (int bm = 0; bm < MB; ++bm) {
    __syncthreads();                                     // the last slab's readers (bm = 0) / the previous pass's readers are done
#pragma unroll
    for (int bn = 0; bn < NB; ++bn)
#pragma unroll
      for (int r = 0; r < 16; ++r) Ct[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * CT_LD + wc * WN + 32 * bn + l32] = acc[bm][bn][r];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 64 / RPP; ++s) {
      const int lr = s * RPP + rsub;                     // buffer row: wavefront row (lr >> 5), row (lr & 31) of its block bm
      const int m = m0 + (lr >> 5) * WM + 32 * bm + (lr & 31);
      if (m < p.M && ncol) {
        f32x4 v = *reinterpret_cast<const f32x4*>(Ct + lr * CT_LD + cq);
        if (p.part) {
          *reinterpret_cast<f32x4*>(p.part + ((size_t)zslice * p.M + m) * p.N + n0 + cq) = v;
          continue;
        }
        v += bv;
        if (p.res) {
          int64_t rr = m;
          if (p.r_up) {
            const int img = m / p.r_hw, rem = m - img * p.r_hw, hh = rem / p.r_w, ww = rem - hh * p.r_w;
            rr = (int64_t)img * (p.r_hw >> 2) + (int64_t)(hh >> 1) * (p.r_w >> 1) + (ww >> 1);
          }
          v += *reinterpret_cast<const f32x4*>(p.res + rr * p.ldres + n0 + cq);
        }
        if (p.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
        float* dst = p.C + (int64_t)m * p.ldc + n0 + cq;
        if (p.C2 && n0 + cq >= p.nsplit) dst = p.C2 + (int64_t)m * p.ldc2 + (n0 + cq - p.nsplit);
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    }
  }
}

template <int BM, int BN, bool AKC, bool BKC, bool PRO, bool LONGK, bool SEG2>
__global__ __launch_bounds__(256, 2) void mm_kernel(const mm_args p) {
  mm_body<BM, BN, AKC, BKC, PRO, LONGK, SEG2>(p, blockIdx.x, gridDim.x, blockIdx.y);
}

// Several products of one operand-layout class in ONE launch (64 x 64 tiles): the seventeen nn.Linear-shaped products of a matching
// step beyond 512 stacked nodes are 0.3 - 0.5 GFLOP each - 8.6 - 15 us of which ~5 are launch and first-load latency.  Workgroup b
// belongs to product d with start[d] <= b < start[d + 1]; inside a product the units are (slice, tile), tile fastest.
#define MM_GROUP_MAX 8
struct mm_group {
  int n;
  int start[MM_GROUP_MAX + 1];
  int tiles[MM_GROUP_MAX];
  mm_args a[MM_GROUP_MAX];
};
template <bool AKC, bool BKC, bool LONGK>
__global__ __launch_bounds__(256, 2) void mm_group_kernel(const mm_group g) {
  int d = 0;
  while (d + 1 < g.n && (int)blockIdx.x >= g.start[d + 1]) ++d;
  const int local = blockIdx.x - g.start[d], T = g.tiles[d];
  const int z = local / T;
  mm_body<64, 64, AKC, BKC, false, LONGK, false>(g.a[d], local - z * T, T, z);
}

struct mm_reduce_item { const float* part; float* C; const float* bias; int64_t ldc; int M, N, ks, start; };
struct mm_reduce_group { int n; mm_reduce_item it[MM_GROUP_MAX]; };
__global__ __launch_bounds__(256) void mm_group_reduce_kernel(const mm_reduce_group g) {
  int d = 0;
  while (d + 1 < g.n && (int)blockIdx.x >= g.it[d + 1].start) ++d;
  const mm_reduce_item& r = g.it[d];
  const size_t q = (size_t)(blockIdx.x - r.start) * 256 + threadIdx.x;
  const int nq = r.N >> 2;
  if (q >= (size_t)r.M * nq) return;
  const int m = (int)(q / nq), n = 4 * (int)(q - (size_t)m * nq);
  const size_t plane = (size_t)r.M * r.N, e = (size_t)m * r.N + n;
  f32x4 v = *reinterpret_cast<const f32x4*>(r.part + e);
  for (int z = 1; z < r.ks; ++z) v += *reinterpret_cast<const f32x4*>(r.part + z * plane + e);
  if (r.bias) v += *reinterpret_cast<const f32x4*>(r.bias + n);
  *reinterpret_cast<f32x4*>(r.C + (int64_t)m * r.ldc + n) = v;
}

// partial planes -> C (+ bias), fixed order: deterministic
__global__ __launch_bounds__(256) void mm_reduce_kernel(const float* __restrict__ part, int kslices, float* __restrict__ C, int64_t ldc,
                                                        const float* __restrict__ bias, int M, int N) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;          // one column quad each
  const int nq = N >> 2;
  if (q >= (size_t)M * nq) return;
  const int m = (int)(q / nq), n = 4 * (int)(q - (size_t)m * nq);
  const size_t plane = (size_t)M * N, e = (size_t)m * N + n;
  f32x4 v = *reinterpret_cast<const f32x4*>(part + e);
  for (int z = 1; z < kslices; ++z) v += *reinterpret_cast<const f32x4*>(part + z * plane + e);
  if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
  *reinterpret_cast<f32x4*>(C + (int64_t)m * ldc + n) = v;
}

template <bool AKC, bool BKC, bool PRO, bool LONGK>
static int mm_launch(int tile, const mm_args& a, dim3 grid_y, hipStream_t st) {
  const int bm = tile >> 1 ? 64 : 128, bn = tile & 1 ? 64 : 128;
  const int tiles = ((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn);
  dim3 grid(tiles, grid_y.y);
  switch (tile) {
    case 0: hipLaunchKernelGGL((mm_kernel<128, 128, AKC, BKC, PRO, LONGK, false>), grid, dim3(256), 0, st, a); break;
    case 1: hipLaunchKernelGGL((mm_kernel<128, 64, AKC, BKC, PRO, LONGK, false>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((mm_kernel<64, 128, AKC, BKC, PRO, LONGK, false>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((mm_kernel<64, 64, AKC, BKC, PRO, LONGK, false>), grid, dim3(256), 0, st, a); break;
  }
  return ttdg_launch_status("mm_f32");
}

// two reduction segments: 64 x 64 tiles, k-contiguous operands
template <bool PRO, bool LONGK>
static int mm_launch_seg2(const mm_args& a, hipStream_t st) {
  const int tiles = ((a.M + 63) / 64) * ((a.N + 63) / 64);
  hipLaunchKernelGGL((mm_kernel<64, 64, true, true, PRO, LONGK, true>), dim3(tiles), dim3(256), 0, st, a);
  return ttdg_launch_status("mm_f32 (two segments)");
}

// tile code: bit 1 = BM 64 (else 128), bit 0 = BN 64 (else 128).  Measured on every pointwise layer of the bench shape, forward and
// both backward products (profiles/r06_pointwise_ab_*.txt): the 64 x 64 tile - five workgroups per CU, 16 accumulator registers per
// wavefront - wins or ties everywhere (the 128 x 128 tile's two workgroups per CU expose the LDS-DMA latency of its two-stage ring);
// the larger tiles stay selectable per call for the A/B.
static int mm_pick_tile(int M, int N, int kslices) { return 3; }

extern "C" size_t ttdg_mm_workspace_bytes(int M, int N, int kslices) { return kslices > 1 ? (size_t)kslices * M * N * sizeof(float) : 0; }

// validation + the kernel-side argument block of one product; returns 1 for an empty product (nothing to launch)
static int mm_prepare(const ttdg_mm_t* d, mm_args& a, int& ks_out, int& empty) {
  empty = 0;
  TTDG_REQUIRE(d && d->A && d->B && d->C, "mm: null operand");
  TTDG_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, "mm: negative size");
  if (d->M == 0 || d->N == 0) { empty = 1; return 0; }
  TTDG_REQUIRE((d->N & 3) == 0 && (d->ldc & 3) == 0 && ((uintptr_t)d->C & 15) == 0, "mm: N, ldc must be multiples of 4 and C 16-byte aligned");
  TTDG_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0 && (d->lda & 3) == 0 && (d->ldb & 3) == 0, "mm: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  TTDG_REQUIRE(d->a_layout == 0 || d->a_layout == 1, "mm: a_layout");
  TTDG_REQUIRE(d->b_layout == 0 || d->b_layout == 1, "mm: b_layout");
  TTDG_REQUIRE(!(d->a_layout == 1 && d->b_layout == 0), "mm: (A output-contiguous, B k-contiguous) is not built");
  if (d->a_layout == 0 || d->b_layout == 0) TTDG_REQUIRE((d->K & 3) == 0, "mm: K must be a multiple of 4 for a k-contiguous operand");
  if (d->a_layout == 1) TTDG_REQUIRE((d->M & 3) == 0, "mm: M must be a multiple of 4 for an m-contiguous A");
  TTDG_REQUIRE(!d->a_stride || d->a_layout == 0, "mm: the strided row map needs a k-contiguous A");
  TTDG_REQUIRE(!d->res || ((d->ldres & 3) == 0 && ((uintptr_t)d->res & 15) == 0), "mm: residual alignment");
  TTDG_REQUIRE(!d->bias || ((uintptr_t)d->bias & 15) == 0, "mm: bias alignment");
  TTDG_REQUIRE(!d->bias2 || ((uintptr_t)d->bias2 & 15) == 0, "mm: bias2 alignment");
  TTDG_REQUIRE(!d->pbias || (d->a_layout == 0 && d->b_layout == 0 && d->K <= MM_PRO_MAXK), "mm: the input shift needs k-contiguous operands and K <= 2048");
  TTDG_REQUIRE(d->kslices >= 0 && d->kslices <= 1024, "mm: kslices");
  TTDG_REQUIRE(d->kslices <= 1 || (d->ws && !d->res && !d->relu), "mm: the split form takes a workspace and no residual / ReLU");
  a.A = d->A, a.B = d->B, a.C = d->C, a.bias = d->bias, a.res = d->res, a.bias2 = d->bias2, a.pbias = d->pbias;
  a.lda = d->lda, a.ldb = d->ldb, a.ldc = d->ldc, a.ldres = d->ldres;
  a.M = d->M, a.N = d->N, a.K = d->K;
  a.a_map = d->a_stride > 1;
  a.a_s = d->a_stride > 1 ? d->a_stride : 1;
  a.a_w = d->a_w, a.a_hw = d->a_h * d->a_w;
  if (a.a_map) {
    TTDG_REQUIRE(d->a_h > 0 && d->a_w > 0, "mm: strided row map without an input size");
    const int ho = (d->a_h - 1) / a.a_s + 1, wo = (d->a_w - 1) / a.a_s + 1;
    TTDG_REQUIRE(d->M % (ho * wo) == 0, "mm: M is not a whole number of strided maps");
    a.a_wo = wo, a.a_howo = ho * wo;
  } else {
    a.a_wo = a.a_howo = 1;
  }
  a.r_up = d->res_up != 0;
  a.r_w = d->res_w, a.r_hw = d->res_h * d->res_w;
  if (a.r_up) TTDG_REQUIRE(d->res && d->res_h > 0 && d->res_w > 0 && !(d->res_h & 1) && !(d->res_w & 1) && d->M % a.r_hw == 0, "mm: up-sampled residual needs an even (H, W) map");
  a.relu = d->relu, a.prelu = d->prelu;
  a.A2 = d->A2, a.B2 = d->B2, a.lda2 = d->lda2, a.ldb2 = d->ldb2, a.K2 = d->A2 ? d->K2 : 0;
  a.a2_map = d->a2_stride > 1;
  a.a2_s = d->a2_stride > 1 ? d->a2_stride : 1;
  a.a2_w = d->a2_w, a.a2_hw = d->a2_h * d->a2_w;
  a.a2_wo = a.a2_howo = 1;
  if (a.K2 > 0) {
    TTDG_REQUIRE(d->B2 && d->a_layout == 0 && d->b_layout == 0 && d->kslices <= 1, "mm: the second segment needs k-contiguous operands and no split");
    TTDG_REQUIRE((d->K % MM_BK) == 0 && (d->K2 % MM_BK) == 0, "mm: with a second segment both reductions must be multiples of 32");
    TTDG_REQUIRE(((uintptr_t)d->A2 & 15) == 0 && ((uintptr_t)d->B2 & 15) == 0 && (d->lda2 & 3) == 0 && (d->ldb2 & 3) == 0, "mm: second-segment operand alignment");
    TTDG_REQUIRE((d->tile & 255) == 0 || (d->tile & 255) == 4, "mm: the second segment is built for 64 x 64 tiles");
    if (a.a2_map) {
      TTDG_REQUIRE(d->a2_h > 0 && d->a2_w > 0, "mm: strided second-segment row map without an input size");
      const int ho = (d->a2_h - 1) / a.a2_s + 1, wo = (d->a2_w - 1) / a.a2_s + 1;
      TTDG_REQUIRE(d->M % (ho * wo) == 0, "mm: M is not a whole number of strided maps (second segment)");
      a.a2_wo = wo, a.a2_howo = ho * wo;
    }
  }
  a.C2 = d->C2, a.ldc2 = d->ldc2, a.nsplit = d->C2 ? d->nsplit : 0;
  if (d->C2) {
    TTDG_REQUIRE(d->nsplit > 0 && d->nsplit < d->N && (d->nsplit & 3) == 0 && (d->ldc2 & 3) == 0 && ((uintptr_t)d->C2 & 15) == 0 && d->kslices <= 1,
                 "mm: split output needs 0 < nsplit < N, multiples of 4, an aligned C2 and no split reduction");
  }
  const int ks = d->kslices > 1 ? d->kslices : 0;
  a.kslices = ks;
  a.kc = ks ? (((d->K + ks - 1) / ks) + MM_BK - 1) / MM_BK * MM_BK : 0;
  a.part = ks ? (float*)d->ws : nullptr;
  a.dbg = (d->tile >> 8) & 3;                                    // ablation bits (tools/ablate_pointwise.py); 0 on every product path
  ks_out = ks;
  return 0;
}

extern "C" int ttdg_mm_f32(const ttdg_mm_t* d, ttdg_stream_t stream) {
  mm_args a;
  int ks = 0, empty = 0;
  if (int e = mm_prepare(d, a, ks, empty)) return e;
  if (empty) return 0;
  hipStream_t st = (hipStream_t)stream;
  int tile = (d->tile & 255) > 0 ? (d->tile & 255) - 1 : mm_pick_tile(d->M, d->N, ks);
  TTDG_REQUIRE(tile >= 0 && tile < 4, "mm: tile code");
  dim3 gy(1, ks ? ks : 1);
  int rc;
  const bool longk = (ks ? a.kc : d->K + a.K2) >= 1024;           // (per workgroup: a split reduction restarts its chains per slice anyway)
  if (a.K2 > 0) {
    if (d->pbias) return longk ? mm_launch_seg2<true, true>(a, st) : mm_launch_seg2<true, false>(a, st);
    return longk ? mm_launch_seg2<false, true>(a, st) : mm_launch_seg2<false, false>(a, st);
  }
#define MM_GO(AKC, BKC, PRO) (longk ? mm_launch<AKC, BKC, PRO, true>(tile, a, gy, st) : mm_launch<AKC, BKC, PRO, false>(tile, a, gy, st))
  if (d->a_layout == 0 && d->b_layout == 0) rc = d->pbias ? MM_GO(true, true, true) : MM_GO(true, true, false);
  else if (d->a_layout == 0) rc = MM_GO(true, false, false);
  else rc = MM_GO(false, false, false);
#undef MM_GO
  if (rc || !ks) return rc;
  const size_t quads = (size_t)d->M * (d->N >> 2);
  hipLaunchKernelGGL(mm_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, a.part, ks, d->C, d->ldc, d->bias, d->M, d->N);
  return ttdg_launch_status("mm_reduce");
}

// up to MM_GROUP_MAX products of ONE operand-layout class (and one long-reduction class) in one launch, 64 x 64 tiles; split
// reductions are added by one grouped reduce launch.  No residual map / input activation / second segment in a group.
extern "C" int ttdg_mm_f32_grouped(const ttdg_mm_t* descs, int n, ttdg_stream_t stream) {
  TTDG_REQUIRE(descs && n >= 1 && n <= MM_GROUP_MAX, "mm_grouped: between 1 and 8 products");
  mm_group g;
  mm_reduce_group rg;
  g.n = 0, rg.n = 0;
  int units = 0, runits = 0, cls = -1, longk = -1;
  for (int i = 0; i < n; ++i) {
    const ttdg_mm_t* d = descs + i;
    mm_args a;
    int ks = 0, empty = 0;
    if (int e = mm_prepare(d, a, ks, empty)) return e;
    if (empty) continue;
    TTDG_REQUIRE(!d->pbias && !d->A2 && !d->C2 && !d->res_up && d->a_stride <= 1 && (d->tile & 255) == 0, "mm_grouped: plain products only (no input activation, second segment, split output, row maps, forced tile)");
    const int c = d->a_layout * 2 + d->b_layout, lk = (ks ? a.kc : d->K) >= 1024;
    TTDG_REQUIRE((cls < 0 || cls == c) && (longk < 0 || longk == lk), "mm_grouped: the products of a group share their operand layouts and reduction class");
    cls = c, longk = lk;
    const int tiles = ((d->M + 63) / 64) * ((d->N + 63) / 64);
    g.a[g.n] = a, g.tiles[g.n] = tiles, g.start[g.n] = units;
    units += tiles * (ks ? ks : 1);
    ++g.n;
    if (ks) {
      mm_reduce_item& r = rg.it[rg.n++];
      r.part = a.part, r.C = d->C, r.bias = d->bias, r.ldc = d->ldc, r.M = d->M, r.N = d->N, r.ks = ks, r.start = runits;
      runits += (int)(((size_t)d->M * (d->N >> 2) + 255) / 256);
    }
  }
  if (g.n == 0) return 0;
  g.start[g.n] = units;
  hipStream_t st = (hipStream_t)stream;
#define MM_GG(AKC, BKC)                                                                                        \
  do {                                                                                                         \
    if (longk) hipLaunchKernelGGL((mm_group_kernel<AKC, BKC, true>), dim3(units), dim3(256), 0, st, g);        \
    else hipLaunchKernelGGL((mm_group_kernel<AKC, BKC, false>), dim3(units), dim3(256), 0, st, g);             \
  } while (0)
  if (cls == 0) MM_GG(true, true);
  else if (cls == 1) MM_GG(true, false);
  else MM_GG(false, false);
#undef MM_GG
  if (int e = ttdg_launch_status("mm_f32_grouped")) return e;
  if (rg.n) {
    hipLaunchKernelGGL(mm_group_reduce_kernel, dim3(runits), dim3(256), 0, st, rg);
    return ttdg_launch_status("mm_grouped_reduce");
  }
  return 0;
}
