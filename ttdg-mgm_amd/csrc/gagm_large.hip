// A6 — GA-MGM solver for graphs beyond the single-workgroup kernel's 128-node limit
// (reference multi_graph_matching.py:300-389; BASELINE cfg-3: 8 graphs x 256 nodes, M = 2048).
//
// At this size the per-iteration operands no longer fit one CU: W is M x M (16 MB at cfg-3, MALL/L2 resident across
// iterations) and W U alone is 0.27 GFLOP.  One iteration = TWO launches, and the host is not in the loop:
//
//   gagm_large_mul_kernel      grid (row tiles of 32 nodes) x (ks K-slices + 1), 256 threads, fp32 MFMA 32x32x2:
//        slices 0..ks-1 : partial (W U)[tile] over their K range      -> WUp[z]           (deterministic planes, no atomics)
//        slice  ks      : B[tile] = A_g[tile,:] U_g, then the tile's share of S = U^T B   -> B, Sp[tile]
//   gagm_large_project_kernel  one workgroup per graph, 1024 threads:
//        S = sum of the tile shares; V_g = (2q B_g S + sum_z WUp[z]) / G; projector (whole-workgroup log-Sinkhorn from
//        sinkhorn_device.h, or the scipy-exact one-wavefront LAP of lap_device.h on an LDS copy of V_g);
//        ||U' - U||^2 and ||U' - U''||^2 of the graph.
//
// The stage machine (tau annealing, Sinkhorn -> Hungarian switch, per-stage iteration counters, both exits of :361) is
// evaluated ON THE DEVICE at the start of every mul launch from the norms the previous projection left behind: every
// workgroup derives the same control word, workgroup (0,0) stores it for the projection launch (two slots, ping-pong).
// Once `done` is set the remaining queued launches return immediately, so the host enqueues iterations in growing
// chunks and reads ONE flag per chunk (the reference reads two norms per iteration).  U / lastU / lastU2 are a ring of
// three buffers indexed by the device-side iteration counter.
//
// Hungarian stage: the exact cycle shortcut of the single-workgroup kernel (gagm.hip) -- the stage map is a deterministic
// map on a finite set and the reference only exits on periods 1 and 2, so a longer cycle burns all 200 iterations.  Every
// projection leaves a one-byte-per-node code of its state and a per-graph 64-bit hash; the control step looks the hash up
// in the stage's history, verifies the candidate byte for byte, and on a period p >= 3 jumps to the state iteration
// max_iter-1 would land on (bit-identical to running them all).
#include "lap_device.h"
#include "sinkhorn_device.h"
#include "lap_certified.h"

#define NU 32
#define GL_TILE 32
#define GL_MAXKS 8
#define GL_HIST 256   /* states remembered per Hungarian stage (>= the reference's 200-iteration cap) */
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GlCtl {   // control state at the START of an iteration; 16 words, 16-byte aligned in the workspace (gl_carve)
  int32_t done, stage, hung, it, total;
  float tau;
  int32_t iters[6];
  int32_t jump;          // 1 + history index of the final state when the cycle shortcut fired, else 0
  int32_t cyc_p, cyc_i;  // period and detection iteration (info[14], info[15])
  int32_t exec;          // iterations EXECUTED when `done` was set (= launches that did work; `total` also counts the iterations a cycle jump skipped)
};

struct GlWs {
  float* V0;     // first-iteration V        (M x 32)  -- same offsets as the single-workgroup kernel's workspace
  float* U1;     // first projected U        (M x 32)
  float* ring;   // U, lastU, lastU2         (3 x M x 32)
  float* B;      // A U                      (M x 32)
  float* V;      // current V                (M x 32)
  float* WUp;    // K-slice planes of W U    (ks x M x 32)
  float* Sp;     // per-tile shares of S     (ntiles x 32 x 32)
  float* dn;     // per-graph squared norms  (2 x 64)
  GlCtl* ctl;    // two slots
  int32_t* res;  // control word after the last enqueued iteration (host-visible copy source), 16 words
  unsigned long long* hg;      // per-graph state hashes of the latest projection (64)
  unsigned long long* hhash;   // whole-state hash per Hungarian-stage iteration (GL_HIST)
  double* lapv;                // Hungarian stage: column duals of every graph's last certified LAP (M), warm start of the next
  int32_t* lapok;              // per graph: lapv holds the duals of the previous iteration (64)
  int32_t* lapstat;            // [0] certified LAPs, [1] scipy-order fallbacks of this solve, [2] pricing rounds, [3] augmented rows (cfg.profile)
  unsigned* bar;               // persistent kernel: [0] grid-barrier arrivals, [1] barrier timeout flag
  unsigned long long* prof;    // cfg.profile: cycles summed over graphs and iterations - operands + S, V, Sinkhorn projector, certified LAP, scipy-order LAP, norms / hash
  unsigned char* hist;         // state codes, GL_HIST x M bytes
  int M, ntiles, ks, Kc;
};

static inline int gl_ntiles(const ttdg_graphs_t& gr) {
  int t = 0;
  for (int g = 0; g < gr.G; ++g) t += (gr.off[g + 1] - gr.off[g] + GL_TILE - 1) / GL_TILE;
  return t;
}

static inline size_t gl_ws_floats(int M, int ntiles, int ks) {
  return (size_t)M * NU * (size_t)(7 + ks) + (size_t)ntiles * NU * NU + 128 + 32 + 16 + 2 * (64 + GL_HIST) + 2 * (size_t)M + 64 + 8 + 16 + 8 +
         (size_t)GL_HIST * ((M + 3) / 4) + 8;
}

// upper bound over every partition of M nodes into <= 64 graphs (ttdg_gagm_workspace_bytes takes only M)
size_t ttdg_gagm_large_ws_bound(int M) { return gl_ws_floats(M, M / GL_TILE + TTDG_MAX_GRAPHS, GL_MAXKS) * sizeof(float); }

static GlWs gl_carve(float* ws, const ttdg_graphs_t& gr) {
  GlWs w;
  const int M = gr.off[gr.G];
  const size_t MU = (size_t)M * NU;
  w.M = M;
  w.ntiles = gl_ntiles(gr);
  int ks = 512 / w.ntiles;
  ks = ks < 1 ? 1 : (ks > GL_MAXKS ? GL_MAXKS : ks);
  int Kc = ((M + ks - 1) / ks + 127) & ~127;       // every wavefront of a slice gets whole 32-wide chunks
  ks = (M + Kc - 1) / Kc;
  w.ks = ks; w.Kc = Kc;
  w.V0 = ws; w.U1 = ws + MU; w.ring = ws + 2 * MU; w.B = ws + 5 * MU; w.V = ws + 6 * MU; w.WUp = ws + 7 * MU;
  w.Sp = w.WUp + (size_t)ks * MU;
  w.dn = w.Sp + (size_t)w.ntiles * NU * NU;
  w.ctl = (GlCtl*)(w.dn + 128);
  w.res = (int32_t*)(w.dn + 128 + 32);
  w.hg = (unsigned long long*)(w.dn + 128 + 32 + 16);      // 8-byte aligned: every block above is a multiple of 2 floats
  w.hhash = w.hg + 64;
  w.lapv = (double*)(w.hhash + GL_HIST);
  w.lapok = (int32_t*)(w.lapv + M);
  w.lapstat = w.lapok + 64;
  w.prof = (unsigned long long*)(w.lapstat + 8);
  w.bar = (unsigned*)(w.prof + 8);
  w.hist = (unsigned char*)(w.bar + 8);
  return w;
}

// Control word at the start of iteration t: the stored word of iteration t-1 advanced by the norms (and, in the Hungarian
// stage, the state hash) its projection left behind (:361-383).  Called by every thread of a workgroup; every workgroup
// of a launch derives the same word; `store` (one workgroup per launch) persists it for the projection launch.
__device__ __forceinline__ GlCtl gl_control(const GlWs& w, int G, const ttdg_gagm_cfg_t& cfg, int t, bool store) {
  __shared__ GlCtl s_ctl;
  __shared__ int s_prev, s_i, s_ok, s_scan;
  __shared__ unsigned long long s_h;
  const int tid = threadIdx.x;
  if (tid == 0) {
    GlCtl c = w.ctl[t & 1];
    s_prev = -1; s_ok = 1; s_i = 0; s_scan = 0;
    if (t > 0 && !c.done) {
      float s1 = 0.f, s2 = 0.f;
      for (int g = 0; g < G; ++g) { s1 += w.dn[2 * g]; s2 += w.dn[2 * g + 1]; }
      const int i = c.it;          // index of the iteration just finished inside its stage
      ++c.it; ++c.total;
      const bool conv = sqrtf(s1) < cfg.tol || s2 == 0.f;
      if (conv || c.it >= cfg.max_iter) {
        if (c.stage < 6) c.iters[c.stage] = c.it;
        ++c.stage; c.it = 0;
        if (c.hung) c.done = 1;                                                // :374-376
        else if (cfg.max_stages > 0 && c.stage >= cfg.max_stages) c.done = 1;
        else if (c.tau > cfg.min_tau) c.tau *= cfg.gamma;                      // :377-379
        else c.hung = 1;                                                       // :382-383
        if (c.done) c.exec = t;
      } else if (c.hung && !cfg.no_cycle_skip && i < GL_HIST) {
        unsigned long long h = 0ull;
        for (int g = 0; g < G; ++g) h ^= w.hg[g];
        if (store) w.hhash[i] = h;
        s_h = h; s_i = i; s_scan = 1;
      }
    }
    s_ctl = c;
  }
  __syncthreads();
  if (s_scan) {
    // most recent earlier state with this hash.  [r6] every thread compares its share of the history (rounds 2-5: thread 0 walked it
    // backwards, one dependent load per remembered state - the mul launch grew by 0.1 us per Hungarian-stage iteration, 13.8 -> 16.1 us
    // over the 22 of a cfg-3 solve); same answer: the largest matching index
    const int i = s_i;
    const unsigned long long h = s_h;
    int best = -1;
    for (int k = tid; k < i; k += blockDim.x) if (w.hhash[k] == h) best = k;     // ascending per thread: the last hit is its largest
    if (best >= 0) atomicMax(&s_prev, best);
    __syncthreads();
  }
  if (s_prev >= 0) {   // exact check of the candidate (a hash collision must not jump), all threads
    const unsigned char* ha = w.hist + (size_t)s_prev * w.M;
    const unsigned char* hb = w.hist + (size_t)s_i * w.M;
    for (int e = tid; e < w.M; e += blockDim.x) if (ha[e] != hb[e]) s_ok = 0;
    __syncthreads();
    if (tid == 0 && s_ok) {
      const int i = s_i, p = i - s_prev;
      if (p >= 3) {                                  // periods 1 and 2 are the reference's own exits
        GlCtl c = s_ctl;
        const int R = cfg.max_iter - 1 - i;          // iterations the reference would still run
        c.jump = 1 + (i - p + (R % p));              // the state iteration max_iter-1 lands on
        c.cyc_p = p; c.cyc_i = i;
        c.total += R;
        if (c.stage < 6) c.iters[c.stage] = cfg.max_iter;
        ++c.stage; c.it = 0; c.done = 1; c.exec = t;
        s_ctl = c;
      }
    }
    __syncthreads();
  }
  if (store && tid == 0) w.ctl[(t + 1) & 1] = s_ctl;
  return s_ctl;
}

__global__ __launch_bounds__(256) void gagm_large_init_kernel(const float* __restrict__ U0, ttdg_gagm_cfg_t cfg, GlWs w) {
  const size_t MU = (size_t)w.M * NU;
  for (size_t e = blockIdx.x * 256 + threadIdx.x; e < MU; e += (size_t)gridDim.x * 256) {
    w.ring[e] = U0[e];
    w.ring[MU + e] = 0.f;
    w.ring[2 * MU + e] = 0.f;          // lastU = zeros (:305)
  }
  if (blockIdx.x == 0 && threadIdx.x < 128) w.dn[threadIdx.x] = 0.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    GlCtl c;
    c.done = 0; c.stage = 0; c.hung = cfg.start_hungarian != 0; c.it = 0; c.total = 0; c.tau = cfg.tau0;
    for (int k = 0; k < 6; ++k) c.iters[k] = 0;
    c.jump = 0; c.cyc_p = 0; c.cyc_i = 0; c.exec = 0;
    w.ctl[0] = c;
    w.ctl[1] = c;
    for (int k = 0; k < 8; ++k) { w.lapstat[k] = 0; w.prof[k] = 0ull; w.bar[k] = 0u; }
  }
}

// [r4] every load is UNCONDITIONAL (clamped address, the zero selected after the wait): a load under a condition is waited for at
// the join, which serialised the 32 loads of a chunk; out-of-range rows / columns read a valid neighbour and are zeroed
__device__ __forceinline__ void gl_load_chunk_a(const float* __restrict__ Asrc, int lda, int nrows, int k0, int kend, int li, int kh, float (&ra)[16]) {
  const int kl = max(kend - 1, 0), k = k0 + li, kc = min(k, kl);
  const int rmax = max(nrows - 1, 0);
#pragma unroll
  for (int j = 0; j < 16; ++j) {        // 32 x 32 tile of A, coalesced 128-byte rows
    const int row = kh + 2 * j;
    ra[j] = Asrc[(size_t)min(row, rmax) * lda + kc];
  }
}
__device__ __forceinline__ void gl_load_chunk_u(const float* __restrict__ Ub, int k0, int kend, int li, int kh, float (&rb)[16]) {
  const int kl = max(kend - 1, 0);
#pragma unroll
  for (int s = 0; s < 16; ++s) {        // MFMA B operand: U[k][col], two k per step
    const int kk = k0 + 2 * s + kh;
    rb[s] = Ub[(size_t)min(kk, kl) * NU + li];
  }
}
__device__ __forceinline__ void gl_load_chunk(const float* __restrict__ Asrc, int lda, int nrows, const float* __restrict__ Ub,
                                              int k0, int kend, int li, int kh, float (&ra)[16], float (&rb)[16]) {
  gl_load_chunk_a(Asrc, lda, nrows, k0, kend, li, kh, ra);
  gl_load_chunk_u(Ub, k0, kend, li, kh, rb);
}
// the zeros of a chunk, applied where its registers are consumed (not behind the loads: that would wait for them at once)
__device__ __forceinline__ void gl_mask_chunk(int nrows, int k0, int kend, int li, int kh, float (&ra)[16], float (&rb)[16]) {
  const int k = k0 + li;
#pragma unroll
  for (int j = 0; j < 16; ++j) ra[j] = (kh + 2 * j < nrows && k < kend) ? ra[j] : 0.f;
#pragma unroll
  for (int s = 0; s < 16; ++s) rb[s] = (k0 + 2 * s + kh < kend) ? rb[s] : 0.f;
}

// One (row tile, K slice) item of the mul phase on a sub-group of 256 threads (4 wavefronts).  `sub` = index of the sub-group
// inside the workgroup (the two-launch kernel has one, the persistent kernel PT / 256), `tid` = thread index inside the
// sub-group, `active` = this sub-group has an item (an idle one still meets the workgroup barriers).  smem: GL_MUL_LDS floats.
#define GL_MUL_LDS (4 * GL_TILE * 33 + 2 * GL_TILE * 33)
// [r6] `ctlfn()` yields the control word and may synchronise the workgroup (the two-launch kernel evaluates the stage machine there: ~4 us of
// dependent L2 round trips on a cold cache); it is called AFTER the first chunk's 16 loads of W / A per lane are in flight - they depend on
// the item only, not on the iteration - so the stage machine runs under their latency.  A `done` word returns (two-launch kernel only).
template <class CtlFn>
__device__ __forceinline__ void gl_mul_item(const float* __restrict__ Apack, const float* __restrict__ W, const ttdg_graphs_t& gr,
                                            const GlWs& w, CtlFn&& ctlfn, int item_tile, int z, bool active, int tid, float* smem) {
  float* s_a = smem;                                  // per-wavefront A tiles, then the 4 accumulator planes
  float* s_bt = smem + 4 * GL_TILE * 33;
  float* s_ut = s_bt + GL_TILE * 33;
  const int wave = tid >> 6, lane = tid & 63, M = w.M;
  const size_t MU = (size_t)M * NU;

  int tile = active ? item_tile : 0, g = 0;
  size_t aoff = 0;
  for (;; ++g) {
    const int n = gr.off[g + 1] - gr.off[g], nt = (n + GL_TILE - 1) / GL_TILE;
    if (tile < nt) break;
    tile -= nt;
    aoff += (size_t)n * n;
  }
  const int o = gr.off[g], n = gr.off[g + 1] - o;
  const int row0 = o + tile * GL_TILE, nrows = active ? min(GL_TILE, o + n - row0) : 0;
  const bool wpart = z < w.ks;
  const float* Asrc;
  int lda, kbeg, kend;
  if (wpart) { Asrc = W + (size_t)row0 * M; lda = M; kbeg = z * w.Kc; kend = min(M, kbeg + w.Kc); }
  else       { Asrc = Apack + aoff + (size_t)tile * GL_TILE * n; lda = n; kbeg = 0; kend = n; }
  if (!active) kend = kbeg;
  const int li = lane & 31, kh = lane >> 5;
  float ra[16], rb[16], cb[16];
  int k0 = kbeg + wave * 32;
  gl_load_chunk_a(Asrc, lda, nrows, k0, kend, li, kh, ra);

  const GlCtl ctl = ctlfn();
  if (ctl.done) return;
  const float* U = w.ring + (size_t)(ctl.total % 3) * MU;
  const float* Ub = wpart ? U : U + (size_t)o * NU;
  gl_load_chunk_u(Ub, k0, kend, li, kh, rb);

  float* sa = s_a + wave * (GL_TILE * 33);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (; k0 < kend; k0 += 128) {
    gl_mask_chunk(nrows, k0, kend, li, kh, ra, rb);
#pragma unroll
    for (int j = 0; j < 16; ++j) sa[(kh + 2 * j) * 33 + li] = ra[j];
#pragma unroll
    for (int s = 0; s < 16; ++s) cb[s] = rb[s];
    wave_sync();
    gl_load_chunk(Asrc, lda, nrows, Ub, k0 + 128, kend, li, kh, ra, rb);   // in flight under the MFMAs (past the end: clamped, unused)
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[li * 33 + 2 * s + kh], cb[s], acc, 0, 0, 0);
    wave_sync();
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) s_a[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * kh) * NU + li] = acc[r];
  __syncthreads();
  float ut[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {            // (A slice) the tile's rows of U for its share of S: four loads in flight, not four round trips
    const int e = tid + 256 * j, row = e >> 5, col = e & 31;
    ut[j] = U[(size_t)min(row0 + row, M - 1) * NU + col];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tid + 256 * j, row = e >> 5, col = e & 31;
    const float v = (s_a[e] + s_a[1024 + e]) + (s_a[2048 + e] + s_a[3072 + e]);
    const bool ok = row < nrows;
    if (wpart) {
      if (ok) w.WUp[(size_t)z * MU + (size_t)(row0 + row) * NU + col] = v;
    } else {
      if (ok) w.B[(size_t)(row0 + row) * NU + col] = v;
      s_bt[row * 33 + col] = ok ? v : 0.f;
      s_ut[row * 33 + col] = ok ? ut[j] : 0.f;
    }
  }
  __syncthreads();
  if (wpart || !active) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {            // this tile's share of S = U^T B
    const int e = tid + 256 * j, u = e >> 5, v = e & 31;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < GL_TILE; ++r) s = fmaf(s_ut[r * 33 + u], s_bt[r * 33 + v], s);
    w.Sp[(size_t)item_tile * (NU * NU) + e] = s;
  }
}

__global__ __launch_bounds__(256) void gagm_large_mul_kernel(const float* __restrict__ Apack, const float* __restrict__ W,
                                                             ttdg_graphs_t gr, ttdg_gagm_cfg_t cfg, GlWs w, int t) {
  __shared__ __attribute__((aligned(16))) float s_mul[GL_MUL_LDS];
  gl_mul_item(Apack, W, gr, w, [&]() { return gl_control(w, gr.G, cfg, t, blockIdx.x == 0 && blockIdx.y == 0); }, blockIdx.x, blockIdx.y, true,
              threadIdx.x, s_mul);
}

#define GL_PTHREADS 1024
#define GL_PWAVES (GL_PTHREADS / 64)
#define GL_VL_OFF 1280   /* floats of dynamic LDS in front of the V_g tile (projector scratch: gl_project_blk needs 2 * 16 * 36 + 16) */

// ---- Sinkhorn projector of one graph block with n >= 32 nodes: rows = universe (32), columns = nodes -----------------
// One column per thread, held in registers (32 values): the column sweep is lane-local and exact (in-lane max); the row
// sweep reduces 33 lines (32 universe rows + the dummy row of multiplicity n - 32) over the workgroup: DPP wavefront
// reductions, one LDS hop across the 16 wavefronts.  Rows use the previous potential as the stabiliser (after any
// column sweep y = L - f - g <= 0), the exact row maximum on the first sweep and whenever a row sum leaves
// [2^-80, 2^80].  Same map as sk_forward (sinkhorn_device.h) on the oriented problem r = 32, c = n, mult = n - 32.
template <bool kMax>
__device__ __forceinline__ void gl_reduce33(const float (&val)[33], float* s_part, int wave, int lane) {
  float mine = 0.f;
#pragma unroll
  for (int p = 0; p < 33; ++p) {
    const float red = kMax ? wave_max_f32_dpp(val[p]) : wave_sum_f32_dpp(val[p]);
    mine = (lane == p) ? red : mine;
  }
  if (lane < 33) s_part[wave * 33 + lane] = mine;
}

__device__ __forceinline__ void gl_project_cols(const float* vl, int n, float scale, int iters, float* __restrict__ Unew,
                                                float* smem) {
  float* s_f = smem;                    // 33 potentials (index 32 = dummy row) + pad
  float* s_stab = smem + 36;            // 33 stabilisers of the sweep in flight
  float* s_part = smem + 72;            // GL_PWAVES x 33 partials
  int* s_flag = (int*)(s_part + GL_PWAVES * 33);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool live = tid < n;            // n <= 768 < GL_PTHREADS: one column per thread
  const int nw = (n + 63) >> 6;         // wavefronts that own columns; the others only keep the barriers company
  const bool wact = wave < nw;
  const int mult = n - NU;
  float L[NU];
  {
    const float* row = vl + (live ? tid : 0) * 33;       // V_g tile in LDS, row stride 33: conflict-free per-thread rows
#pragma unroll
    for (int k = 0; k < NU; ++k) L[k] = row[k] * scale;
  }
  float g = 0.f;
  if (tid < 36) s_f[tid] = 0.f;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    if ((it & 1) == 0) {
      float val[33];
      bool exact = (it == 0);
      for (;;) {
        if (exact) {   // stabiliser = exact line maximum
#pragma unroll
          for (int p = 0; p < NU; ++p) val[p] = live ? L[p] - g : -INFINITY;
          val[32] = (live && mult > 0) ? SK_DUMMY - g : -INFINITY;
          if (wact) gl_reduce33<true>(val, s_part, wave, lane);
          __syncthreads();
          if (tid < 33) {
            float m = -INFINITY;
            for (int k = 0; k < nw; ++k) m = fmaxf(m, s_part[k * 33 + tid]);
            s_stab[tid] = (m == -INFINITY) ? 0.f : m;
          }
          __syncthreads();
        } else if (tid < 33) {
          s_stab[tid] = s_f[tid];
        }
        if (!exact) __syncthreads();
        if (tid == 0) *s_flag = 0;
        if (wact) {
#pragma unroll
          for (int p = 0; p < NU; ++p) val[p] = live ? fast_exp2(L[p] - g - s_stab[p]) : 0.f;
          val[32] = (live && mult > 0) ? fast_exp2(SK_DUMMY - g - s_stab[32]) : 0.f;
          gl_reduce33<false>(val, s_part, wave, lane);
        }
        __syncthreads();
        if (tid < 33) {
          float sum = 0.f;
          for (int k = 0; k < nw; ++k) sum += s_part[k * 33 + tid];
          const bool used = tid < NU || mult > 0;
          if (used && !(sum > 8.3e-25f && sum < 1.2e24f)) *s_flag = 1;
          s_part[tid] = used ? s_stab[tid] + fast_log2(sum) : 0.f;    // candidate, committed below (wave 0 only touches row 0 of s_part)
        }
        __syncthreads();
        const bool redo = !exact && *s_flag != 0;
        if (!redo && tid < 33) s_f[tid] = s_part[tid];
        __syncthreads();
        if (!redo) break;
        exact = true;
      }
    } else if (wact) {
      float f[33];
#pragma unroll
      for (int p = 0; p < 33; ++p) f[p] = s_f[p];
      const float td0 = (mult > 0) ? SK_DUMMY - f[32] : -INFINITY;
      float m0 = td0, m1 = -INFINITY;
#pragma unroll
      for (int p = 0; p < NU; p += 2) { m0 = fmaxf(m0, L[p] - f[p]); m1 = fmaxf(m1, L[p + 1] - f[p + 1]); }
      const float m = fmaxf(m0, m1);
      float s0 = (mult > 0) ? (float)mult * fast_exp2(td0 - m) : 0.f, s1 = 0.f;
#pragma unroll
      for (int p = 0; p < NU; p += 2) { s0 += fast_exp2(L[p] - f[p] - m); s1 += fast_exp2(L[p + 1] - f[p + 1] - m); }
      g = m + fast_log2(s0 + s1);
    }
  }
  if (live) {
    float4* out = reinterpret_cast<float4*>(Unew + (size_t)tid * NU);
#pragma unroll
    for (int k = 0; k < NU / 4; ++k) {
      float4 v;
      v.x = fast_exp2(L[4 * k] - s_f[4 * k] - g); v.y = fast_exp2(L[4 * k + 1] - s_f[4 * k + 1] - g);
      v.z = fast_exp2(L[4 * k + 2] - s_f[4 * k + 2] - g); v.w = fast_exp2(L[4 * k + 3] - s_f[4 * k + 3] - g);
      out[k] = v;
    }
  }
}

// ---- [r4] block-layout Sinkhorn projector over ALL wavefronts of the workgroup (graphs of 33 .. 64 * 8 * PW / 8 nodes) --------
// gl_project_cols above keeps one column per THREAD: every row sweep is 33 wavefront reductions of a 33-register line plus two
// LDS hops and four barriers (3.5 us per sweep at 8 x 256).  Here the oriented problem (rows = 32 universe slots, columns =
// nodes) is dealt out as in the single-workgroup solver's projector (gagm.hip: sk_wave_project_blk): lane (bi = lane >> 3,
// bj = lane & 7) of wavefront w owns rows 4 bi .. 4 bi + 3 and columns 8 (w + PW b) + bj, b < CB = ceil(n / (8 PW)).  A sweep PAIR
// evaluates e = exp2(L - f - g) ONCE: the row sweep needs s_a = sum_q e (in-lane sum, 3-step DPP butterfly over bj, then the
// PW wavefront partials meet in LDS: 33 floats per wavefront, one barrier, summed in a fixed order - deterministic), the
// column sweep exp2(L - f_new - g) = e / s_a - a multiply - summed over the lane's rows and a butterfly over bi: columns never
// leave their wavefront.  The (n - 32) identical dummy rows are one replicated row.  As in every register projector the
// previous potentials stabilise the exponentials; the first pair, and any pair in which a sum leaves [2^-60, 1e6] (a second
// barrier settles that across the wavefronts), runs in the exact max-subtracted form.  Same map as sk_forward on r = 32,
// c = n, mult = n - 32.
template <int PW>
struct GlXchg {
  float* buf;      // 2 x PW x 36 floats (rows 0..31, dummy at 32), ping-pong
  int* flag;       // PW words
  int phase;
  __device__ __forceinline__ float* slot() { float* b = buf + (phase & 1) * (PW * 36); ++phase; return b; }
};

template <int PW, bool kMax>
__device__ __forceinline__ void gl_rows_meet(GlXchg<PW>& x, float (&v)[4], float& vd, int wave, int bi, int bj, int lane) {
  float* b = x.slot();
  if (bj == 0) *reinterpret_cast<float4*>(b + wave * 36 + 4 * bi) = make_float4(v[0], v[1], v[2], v[3]);
  if (lane == 0) b[wave * 36 + 32] = vd;
  __syncthreads();
  float4 acc = *reinterpret_cast<const float4*>(b + 4 * bi);
  float ad = b[32];
#pragma unroll
  for (int w = 1; w < PW; ++w) {
    const float4 t = *reinterpret_cast<const float4*>(b + w * 36 + 4 * bi);
    const float td = b[w * 36 + 32];
    if (kMax) { acc.x = fmaxf(acc.x, t.x); acc.y = fmaxf(acc.y, t.y); acc.z = fmaxf(acc.z, t.z); acc.w = fmaxf(acc.w, t.w); ad = fmaxf(ad, td); }
    else { acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; ad += td; }
  }
  v[0] = acc.x; v[1] = acc.y; v[2] = acc.z; v[3] = acc.w; vd = ad;
}

template <int PW, int CB>
__device__ __forceinline__ void gl_project_blk(const float* vl, int n, float scale, int iters, float* __restrict__ Unew, float* smem) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, bi = lane >> 3, bj = lane & 7;
  GlXchg<PW> x;
  x.buf = smem; x.flag = (int*)(smem + 2 * PW * 36); x.phase = 0;
  const float D = SK_DUMMY;
  const int mult = n - NU;                   // > 0 on this path
  const int cbu = (n + 8 * PW - 1) / (8 * PW);
  float L[4][CB], f[4], g[CB], fd = 0.f;
  bool cused[CB];
#pragma unroll
  for (int b = 0; b < CB; ++b) {
    const int q = 8 * (wave + PW * b) + bj;
    cused[b] = b < cbu && q < n;
    g[b] = 0.f;
    const float* col = vl + (cused[b] ? q : 0) * 33 + 4 * bi;      // V_g tile in LDS, row stride 33
#pragma unroll
    for (int a = 0; a < 4; ++a) L[a][b] = cused[b] ? col[a] * scale : -INFINITY;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) f[a] = 0.f;
  for (int it = 0; it < iters; it += 2) {
    const bool cols_too = it + 1 < iters;
    float fn[4], gn[CB], fdn = fd;
    bool exact = it == 0;
    if (!exact) {
      float e[4][CB], s[4], ed[CB], sd = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < CB; ++b)
          if (b < cbu) { e[a][b] = fast_exp2((L[a][b] - f[a]) - g[b]); acc += e[a][b]; }
        s[a] = bj_sum(acc);
      }
#pragma unroll
      for (int b = 0; b < CB; ++b)
        if (b < cbu) { ed[b] = cused[b] ? fast_exp2((D - fd) - g[b]) : 0.f; sd += ed[b]; }
      sd = bj_sum(sd);
      gl_rows_meet<PW, false>(x, s, sd, wave, bi, bj, lane);
      float lo = sd, hi = sd, r[4];
      fdn = fd + fast_log2(sd);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        lo = fminf(lo, s[a]); hi = fmaxf(hi, s[a]);
        fn[a] = f[a] + fast_log2(s[a]);
        r[a] = __builtin_amdgcn_rcpf(s[a]);
      }
      if (cols_too) {
        const float rd = (float)mult * __builtin_amdgcn_rcpf(sd);
#pragma unroll
        for (int b = 0; b < CB; ++b)
          if (b < cbu) {
            float acc = (e[0][b] * r[0] + e[1][b] * r[1]) + (e[2][b] * r[2] + e[3][b] * r[3]);
            acc = bi_sum(acc);
            acc += ed[b] * rd;
            const float cu = cused[b] ? acc : 1.f;
            lo = fminf(lo, cu); hi = fmaxf(hi, cu);
            gn[b] = cused[b] ? g[b] + fast_log2(acc) : 0.f;
          }
      }
      // the column sums are wavefront-local: the decision to redo the pair exactly must be the workgroup's
      const bool bad = __ballot(!(lo >= 8.6736174e-19f && hi <= 1.0e6f)) != 0ull;
      if (lane == 0) x.flag[wave] = bad ? 1 : 0;
      __syncthreads();
      int any = 0;
#pragma unroll
      for (int w = 0; w < PW; ++w) any |= x.flag[w];
      exact = any != 0;
    }
    if (exact) {
      // rows, max-subtracted: two meetings (maxima, then sums)
      float m[4], md = -INFINITY;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float t = -INFINITY;
#pragma unroll
        for (int b = 0; b < CB; ++b) if (b < cbu) t = fmaxf(t, L[a][b] - g[b]);
        m[a] = bj_max(t);
      }
#pragma unroll
      for (int b = 0; b < CB; ++b) if (b < cbu && cused[b]) md = fmaxf(md, -g[b]);
      md = bj_max(md);
      gl_rows_meet<PW, true>(x, m, md, wave, bi, bj, lane);
      float sa[4], sdd = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float ms = (m[a] == -INFINITY) ? 0.f : m[a];
        m[a] = ms;
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < CB; ++b) if (b < cbu) acc += fast_exp2((L[a][b] - g[b]) - ms);
        sa[a] = bj_sum(acc);
      }
#pragma unroll
      for (int b = 0; b < CB; ++b) if (b < cbu && cused[b]) sdd += fast_exp2(-g[b] - md);
      sdd = bj_sum(sdd);
      gl_rows_meet<PW, false>(x, sa, sdd, wave, bi, bj, lane);
#pragma unroll
      for (int a = 0; a < 4; ++a) fn[a] = m[a] + fast_log2(sa[a]);
      fdn = D + md + fast_log2(sdd);
      if (cols_too) {
        const float td = D - fdn;
#pragma unroll
        for (int b = 0; b < CB; ++b)
          if (b < cbu) {
            float mm = fmaxf(fmaxf(L[0][b] - fn[0], L[1][b] - fn[1]), fmaxf(L[2][b] - fn[2], L[3][b] - fn[3]));
            mm = fmaxf(bi_max(mm), td);
            float acc = (fast_exp2((L[0][b] - fn[0]) - mm) + fast_exp2((L[1][b] - fn[1]) - mm)) +
                        (fast_exp2((L[2][b] - fn[2]) - mm) + fast_exp2((L[3][b] - fn[3]) - mm));
            acc = bi_sum(acc);
            acc += (float)mult * fast_exp2(td - mm);
            gn[b] = cused[b] ? mm + fast_log2(acc) : 0.f;
          }
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) f[a] = fn[a];
    fd = fdn;
    if (cols_too) {
#pragma unroll
      for (int b = 0; b < CB; ++b) if (b < cbu) g[b] = gn[b];
    }
  }
  // U[q][p] = exp2(L - f_p - g_q): node q's four universe slots 4 bi .. 4 bi + 3 are one 16-byte store
#pragma unroll
  for (int b = 0; b < CB; ++b)
    if (cused[b]) {
      float4 o;
      o.x = fast_exp2((L[0][b] - f[0]) - g[b]); o.y = fast_exp2((L[1][b] - f[1]) - g[b]);
      o.z = fast_exp2((L[2][b] - f[2]) - g[b]); o.w = fast_exp2((L[3][b] - f[3]) - g[b]);
      *reinterpret_cast<float4*>(Unew + (size_t)(8 * (wave + PW * b) + bj) * NU + 4 * bi) = o;
    }
}

// After the projection of graph g: the G == 2 identity pin (:358-359), the graph's two squared norms for the stage
// machine (:361) and - Hungarian stage - its state code + hash for the cycle shortcut (gl_control).
template <int THREADS>
__device__ __forceinline__ void gl_finish_projection(const ttdg_graphs_t& gr, const ttdg_gagm_cfg_t& cfg, const GlWs& w, const GlCtl& s_c,
                                                     int g, float* Unew, const float* Ucur, const float* Uprev, bool hung) {
  __shared__ float s_red[2 * (THREADS / 64)];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, G = gr.G, M = w.M;
  const int o = gr.off[g], n = gr.off[g + 1] - o;
  const int total = s_c.total;
  if (G == 2 && g == 0) {   // :358-359
    for (int e = tid; e < n * NU; e += THREADS) Unew[e] = ((e >> 5) == (e & 31)) ? 1.f : 0.f;
    __syncthreads();
  }
  float d1 = 0.f, d2 = 0.f;
  for (int e = tid; e < n * NU; e += THREADS) {
    const float un = Unew[e], a = un - Ucur[e], b = un - Uprev[e];
    d1 = fmaf(a, a, d1);
    d2 = fmaf(b, b, d2);
    if (total == 0) w.U1[(size_t)o * NU + e] = un;
  }
  d1 = wave_sum(d1); d2 = wave_sum(d2);
  if (lane == 0) { s_red[2 * wave] = d1; s_red[2 * wave + 1] = d2; }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < THREADS / 64; ++k) { a += s_red[2 * k]; b += s_red[2 * k + 1]; }
    w.dn[2 * g] = a;
    w.dn[2 * g + 1] = b;
  }
  if (hung && !cfg.no_cycle_skip && s_c.it < GL_HIST) {   // state code + hash for the cycle shortcut (gl_control)
    __shared__ unsigned long long s_h[THREADS / 64];
    unsigned long long hx = 0ull;
    for (int r = tid; r < n; r += THREADS) {
      int code = 255;
#pragma unroll
      for (int u = 0; u < NU; ++u) if (Unew[r * NU + u] != 0.f) code = u;
      w.hist[(size_t)s_c.it * M + o + r] = (unsigned char)code;
      unsigned long long z = (unsigned long long)((o + r) * 256 + code) + 0x9E3779B97F4A7C15ull;     // splitmix64
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      hx ^= z ^ (z >> 31);
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
      const unsigned lo = __shfl_xor((unsigned)hx, sft, 64), hi = __shfl_xor((unsigned)(hx >> 32), sft, 64);
      hx ^= ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 0) s_h[wave] = hx;
    __syncthreads();
    if (tid == 0) {
      unsigned long long h = 0ull;
      for (int k = 0; k < THREADS / 64; ++k) h ^= s_h[k];
      w.hg[g] = h;
    }
  }
}

// PT threads: 512 when every graph has <= 512 nodes (2 wavefronts per SIMD: a 256-VGPR budget - with 1024 threads the
// 128-VGPR cap made the register-resident Sinkhorn column + its 33 partial lines spill into scratch inside the sweep
// loop), 1024 for graphs of 513..768 nodes (one column per thread).
// [r6] `ctl_src` != nullptr (two-launch kernel): the control word is still in global memory - thread 0 requests it first, the operand loads
// of the S and V phases (which depend on the graph only, not on the iteration) follow at once, and the word reaches `s_c` (LDS) in front of
// the S phase's barrier: the stage machine's round trip hides under the operands'.  A `done` word returns after that barrier.
template <int PT>
__device__ __forceinline__ void gl_project_graph(const ttdg_graphs_t& gr, const ttdg_gagm_cfg_t& cfg, const GlWs& w, GlCtl& s_c,
                                                 const int g, float* gl_smem, const GlCtl* ctl_src = nullptr) {
  __shared__ __attribute__((aligned(16))) float s_S[NU * NU];
  __shared__ __attribute__((aligned(16))) float s_brow[(PT / 64) * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, G = gr.G, M = w.M;
  const size_t MU = (size_t)M * NU;
  int4 early[4];                          // the 64-byte word as four 16-byte registers (a GlCtl local would live in scratch)
  if (ctl_src && tid == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) early[k] = reinterpret_cast<const int4*>(ctl_src)[k];
  }
  const int o = gr.off[g], n = gr.off[g + 1] - o;
  long long tph = cfg.profile ? (long long)__builtin_readcyclecounter() : 0;      // phase clock (thread 0 adds to w.prof)
#define GL_PHASE(k)                                                                       \
  if (cfg.profile && tid == 0) {                                                          \
    const long long now = (long long)__builtin_readcyclecounter();                        \
    atomicAdd(&w.prof[k], (unsigned long long)(now - tph));                               \
    if ((k) == 3) atomicMax(&w.prof[6], (unsigned long long)(now - tph));                 \
    if ((k) == 4) atomicMax(&w.prof[7], (unsigned long long)(now - tph));                 \
    tph = now;                                                                            \
  }
  const int li = lane & 31, kh = lane >> 5;
  constexpr int PW = PT / 64;
  constexpr int CHUNK = 2 * PW * 8;          // rows per pass of the V loop (256): a wavefront takes two rows at a time
  float bv[8], wu[8];
  // operands of the first V chunk: issued before the S reduction so that both sets of L2 round trips overlap
#define GL_LOAD_V_OPERANDS(base)     /* unconditional loads (clamped row / plane), zeros selected afterwards */ \
  _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                       \
    const int i = (base) + j * 2 * PW + wave * 2 + kh;                                            \
    const bool ok = i < n;                                                                               \
    const size_t idx = (size_t)(o + (ok ? i : 0)) * NU + li;                                             \
    const float b0 = w.B[idx];                                                                           \
    float p[GL_MAXKS];                                                                                   \
    _Pragma("unroll") for (int z = 0; z < GL_MAXKS; ++z) p[z] = w.WUp[(size_t)min(z, w.ks - 1) * MU + idx]; \
    _Pragma("unroll") for (int z = 0; z < GL_MAXKS; ++z) p[z] = (ok && z < w.ks) ? p[z] : 0.f;           \
    bv[j] = ok ? b0 : 0.f;                                                                               \
    wu[j] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));                           \
  }
  GL_LOAD_V_OPERANDS(0)
  // S = sum of the tile shares: one element per thread, 16 independent loads in flight per round (the plain
  // accumulate-as-you-go loop pays one L2 round trip per tile)
  {
    // [r4] every element's tile shares are requested 16 at a time for BOTH elements of a thread (unconditional loads: a clamped
    // index, the select after the wait), i.e. four L2 round trips for the 64 tiles of cfg-3 instead of eight; the order of the
    // additions is unchanged (share t goes to accumulator t mod 16, the accumulators meet in the same tree)
    constexpr int EPT = (NU * NU) / PT;              // 2 (512 threads) or 1
    float acc[EPT][16];
#pragma unroll
    for (int x = 0; x < EPT; ++x)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[x][k] = 0.f;
    for (int t0 = 0; t0 < w.ntiles; t0 += 16) {
      float sv[EPT][16];
#pragma unroll
      for (int x = 0; x < EPT; ++x)
#pragma unroll
        for (int k = 0; k < 16; ++k) sv[x][k] = w.Sp[(size_t)min(t0 + k, w.ntiles - 1) * (NU * NU) + tid + x * PT];
#pragma unroll
      for (int x = 0; x < EPT; ++x)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[x][k] += (t0 + k < w.ntiles) ? sv[x][k] : 0.f;
    }
#pragma unroll
    for (int x = 0; x < EPT; ++x)
      s_S[tid + x * PT] = (((acc[x][0] + acc[x][1]) + (acc[x][2] + acc[x][3])) + ((acc[x][4] + acc[x][5]) + (acc[x][6] + acc[x][7]))) +
                          (((acc[x][8] + acc[x][9]) + (acc[x][10] + acc[x][11])) + ((acc[x][12] + acc[x][13]) + (acc[x][14] + acc[x][15])));
  }
  if (ctl_src && tid == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<int4*>(&s_c)[k] = early[k];
  }
  __syncthreads();
  if (s_c.done) return;
  const int total = s_c.total;
  const bool hung = s_c.hung != 0;
  const float tau = s_c.tau;
  const float* Ucur = w.ring + (size_t)(total % 3) * MU + (size_t)o * NU;
  float* Unew = w.ring + (size_t)((total + 1) % 3) * MU + (size_t)o * NU;
  const float* Uprev = w.ring + (size_t)((total + 2) % 3) * MU + (size_t)o * NU;
  GL_PHASE(0)
  // V_g = (2q B_g S + W U) / G: B rows broadcast from LDS, S column in registers.  V_g goes to the workspace (trace /
  // next launch) AND to an LDS tile with row stride 33 that both projectors read (no global round trip in between)
  float* vl = gl_smem + GL_VL_OFF;
  {
    float sc[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) sc[k] = s_S[k * NU + li];
    const float qw2 = 2.f * cfg.quad_weight, invG = 1.f / (float)G;
    float* br = s_brow + wave * 64;
    for (int base = 0; base < n; base += CHUNK) {
      if (base > 0) { GL_LOAD_V_OPERANDS(base) }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = base + j * 2 * PW + wave * 2 + kh;
        br[lane] = bv[j];
        wave_sync();
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < NU; k += 4) {
          const float4 b4 = *reinterpret_cast<const float4*>(br + kh * 32 + k);
          a0 = fmaf(b4.x, sc[k], a0); a1 = fmaf(b4.y, sc[k + 1], a1);
          a0 = fmaf(b4.z, sc[k + 2], a0); a1 = fmaf(b4.w, sc[k + 3], a1);
        }
        const float v = (qw2 * (a0 + a1) + wu[j]) * invG;
        if (i < n) {
          const size_t idx = (size_t)(o + i) * NU + li;
          w.V[idx] = v;
          vl[i * 33 + li] = v;
          if (total == 0) w.V0[idx] = v;
        }
        wave_sync();
      }
    }
  }
#undef GL_LOAD_V_OPERANDS
  __syncthreads();
  GL_PHASE(1)
  const float* Vg = w.V + (size_t)o * NU;

  if (!hung) {
    // Sinkhorn orientation (Appendix B steps 1-3) in a batch whose largest graph exceeds the universe (always true on this
    // path): rows = universe unless n_g < 32; with equal sizes the caller transposes n > 32 itself -- the same rule
    SkProb pb;
    pb.src = Vg; pb.splane = 0; pb.nplanes = 1; pb.bias = 0.f; pb.scale = TTDG_LOG2E / tau;
    const bool rows_nodes = n < NU;
    if (rows_nodes) {
      pb.r = n; pb.c = NU; pb.sp = NU; pb.sq = 1; pb.op = NU; pb.oq = 1;
      pb.out = Unew; pb.mir = nullptr; pb.mp = pb.mq = 0;
      pb.mult = pb.c - pb.r;
      pb.pot = nullptr; pb.potld = 0;
      sk_forward<true>(pb, gl_smem, cfg.sk_iter);
    } else if (n > 8 * PW * 8 || (cfg.variant & TTDG_GAGM_COLUMN_PROJECTOR)) {      // wider than 8 column blocks per lane (cannot happen below 513 / 1025 nodes), or the A/B switch
      gl_project_cols(vl, n, pb.scale, cfg.sk_iter, Unew, gl_smem);
    } else if (n <= 8 * PW * 2) {
      gl_project_blk<PW, 2>(vl, n, pb.scale, cfg.sk_iter, Unew, gl_smem);
    } else if (n <= 8 * PW * 4) {
      gl_project_blk<PW, 4>(vl, n, pb.scale, cfg.sk_iter, Unew, gl_smem);
    } else {
      gl_project_blk<PW, 8>(vl, n, pb.scale, cfg.sk_iter, Unew, gl_smem);
    }
  } else {
    // Hungarian stage: one wavefront runs the scipy-exact LAP on the LDS tile of V_g.  (With 1024 threads per workgroup
    // the 128-VGPR cap made this code spill 44 VGPRs to scratch; the 512-thread build has the registers.  Tried and
    // dropped in round 2: the LAP as its own 64-thread launch - spill-free too, but the third launch per iteration and
    // the V_g reload cost more than they saved: 100.8 vs 91.5 us per iteration at 8 x 256.)
    const bool tr = n > NU;
    const int nr = tr ? NU : n, nc = tr ? n : NU;
    for (int e = tid; e < n * NU; e += PT) Unew[e] = 0.f;       // (V_g is already in the LDS tile)
    void* lscr = vl + ((n * 33 + 3) & ~3);                      // behind the V_g tile
    bool certified = false;
    if (tr && n <= 512 && !(cfg.variant & TTDG_GAGM_SCIPY_ORDER_LAP)) {
      // lap_certified.h: warm-started workgroup LAP + uniqueness certificate; the scipy-order solver below only without one
      const LapCertScratch cs = lap_cert_carve(lscr, n);
      const double* warm = (s_c.it > 0 && w.lapok[g]) ? w.lapv + o : nullptr;
      int* stat = cfg.profile ? w.lapstat : nullptr;
      certified = n <= 64 ? lap_certified_solve<PT, 1>(n, vl, cs, warm, stat) : n <= 128 ? lap_certified_solve<PT, 2>(n, vl, cs, warm, stat)
                : n <= 256 ? lap_certified_solve<PT, 4>(n, vl, cs, warm, stat) : lap_certified_solve<PT, 8>(n, vl, cs, warm, stat);
      if (certified) {
        if (tid < NU) Unew[cs.col4row[tid] * NU + tid] = 1.f;
        for (int j = tid; j < n; j += PT) w.lapv[o + j] = cs.v[j];
      }
      if (tid == 0) { w.lapok[g] = certified ? 1 : 0; if (certified) atomicAdd(&w.lapstat[0], 1); }
      GL_PHASE(3)
    }
    __syncthreads();                                                     // zeros land before the ones are written
    // [r5] blocks of a narrow value range (what follows a collapsed Sinkhorn stage: scipy's tie rules decide) take the INTEGER
    // statement of the scipy-order solver on one wavefront (lap_device.h: lap_wave_solve_int; admission + conversion of the tile by
    // the whole workgroup, lap_certified.h: lap_int_admit) - the same decisions at a quarter of the cycles per step
    bool intlap = false;
    if (!certified && tr && n <= 512 && !(cfg.variant & TTDG_GAGM_SCIPY_ORDER_LAP)) {
      intlap = !(cfg.variant & TTDG_GAGM_NO_INT_LAP) && lap_int_admit<PT>(n, vl, (int*)lscr);
      if (intlap && wave == 0) {
        const int b = n <= 256 ? lap_wave_solve_int<4>(n, (const int*)vl, 33) : lap_wave_solve_int<8>(n, (const int*)vl, 33);
        wave_sync();
        if (lane < NU) Unew[b * NU + lane] = 1.f;
      }
      if (tid == 0) atomicAdd(&w.lapstat[intlap ? 5 : 1], 1);      // integer scipy-order solve / fp64 step-by-step fallback
    }
    if (intlap) {
    } else if (!certified && tr && n <= PT && !(cfg.variant & TTDG_GAGM_SCIPY_ORDER_LAP)) {
      // scipy-order LAP over all wavefronts of the workgroup (lap_certified.h: lap_block_solve_exact), one column per thread
      const LapBlockScratch bs = lap_block_carve(lscr, n);
      lap_block_solve_exact<PT>(n, vl, bs);
      __syncthreads();
      if (tid < NU) Unew[bs.col4row[tid] * NU + tid] = 1.f;
    } else if (!certified && wave == 0) {
      if (nc <= 64) {
        const int b = lap_wave_solve_reg<0, true>(nr, nc, vl, tr ? 1 : 33, tr ? 33 : 1);
        wave_sync();
        if (lane < nr) { if (tr) Unew[b * NU + lane] = 1.f; else Unew[lane * NU + b] = 1.f; }
      } else if (nc <= 256) {   // 2 or 4 columns per lane, still register-resident
        const int b = nc <= 128 ? lap_wave_solve_regw<2>(nr, nc, vl, tr ? 1 : 33, tr ? 33 : 1) : lap_wave_solve_regw<4>(nr, nc, vl, tr ? 1 : 33, tr ? 33 : 1);
        wave_sync();
        if (lane < nr) { if (tr) Unew[b * NU + lane] = 1.f; else Unew[lane * NU + b] = 1.f; }
      } else {
        LapScratch sc = lap_carve(lscr, nr, nc);
        lap_wave_solve(nr, nc, vl, tr ? 1 : 33, tr ? 33 : 1, sc);
        wave_sync();
        for (int a = lane; a < nr; a += 64) {
          const int b = sc.col4row[a];
          if (tr) Unew[b * NU + a] = 1.f; else Unew[a * NU + b] = 1.f;
        }
      }
    }
    __threadfence_block();
  }
  __syncthreads();
  GL_PHASE(hung ? 4 : 2)
  gl_finish_projection<PT>(gr, cfg, w, s_c, g, Unew, Ucur, Uprev, hung);
  __syncthreads();
  GL_PHASE(5)
#undef GL_PHASE
}

template <int PT>
__global__ __launch_bounds__(PT) void gagm_large_project_kernel(ttdg_graphs_t gr, ttdg_gagm_cfg_t cfg, GlWs w, int t) {
  extern __shared__ __attribute__((aligned(16))) float gl_smem[];
  __shared__ __attribute__((aligned(16))) GlCtl s_c;
  gl_project_graph<PT>(gr, cfg, w, s_c, blockIdx.x, gl_smem, &w.ctl[(t + 1) & 1]);
}

// control word after `t` enqueued iterations -> w.res (read by the finish kernel) and, when the host gave one, its page-locked flag
// words {done, total}: the host reads them straight out of host memory after the chunk's stream synchronisation - no copy
__global__ __launch_bounds__(256) void gagm_large_peek_kernel(ttdg_graphs_t gr, ttdg_gagm_cfg_t cfg, GlWs w, int t, int32_t* hostflag) {
  const GlCtl c = gl_control(w, gr.G, cfg, t, false);
  if (threadIdx.x == 0) {
    const int32_t* p = (const int32_t*)&c;
    for (int k = 0; k < 16; ++k) w.res[k] = p[k];
    if (hostflag) {
      hostflag[1] = c.total;
      hostflag[2] = c.exec;
      __hip_atomic_store(&hostflag[0], c.done ? 1 : 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// U and info[] from the final control word (grid-stride over `nthreads` threads, `gtid` = this thread's index among them)
__device__ __forceinline__ void gl_write_result(const GlWs& w, const GlCtl* c, float* __restrict__ Uout, int32_t* __restrict__ info,
                                                int profile, size_t gtid, size_t nthreads) {
  const size_t MU = (size_t)w.M * NU;
  if (c->jump > 0) {   // cycle shortcut: the final state is a remembered one
    const unsigned char* code = w.hist + (size_t)(c->jump - 1) * w.M;
    for (size_t e = gtid; e < MU; e += nthreads) Uout[e] = (code[e >> 5] == (e & 31)) ? 1.f : 0.f;
  } else {
    const float* U = w.ring + (size_t)(c->total % 3) * MU;
    for (size_t e = gtid; e < MU; e += nthreads) Uout[e] = U[e];
  }
  if (gtid == 0) {
    for (int k = 0; k < 6; ++k) info[k] = c->iters[k];
    info[6] = c->total; info[7] = c->stage;
    info[14] = c->cyc_p; info[15] = c->cyc_i;
    info[22] = c->exec;                                    // iterations executed (launch pairs that did work)
    info[12] = w.lapstat[0]; info[13] = w.lapstat[1];      // Hungarian stage: certified workgroup LAPs / scipy-order fallbacks
    info[21] = w.lapstat[5];                               // ... / narrow-range blocks solved by the integer scipy-order solver
    // info[8] is the STATUS word and nothing else (written by the cooperative kernel on a barrier failure, 0 otherwise - the caller
    // zero-initialises info).  cfg.profile != 0: info[16..20], cycles / 1024 summed over graphs and iterations:
    //   profile 1: [16] operands + S, [17] V, [18] projector (Sinkhorn + both LAPs), [19] norms / hash
    //   profile 2: the projector split - [16] Sinkhorn, [17] certified LAP, [18] scipy-order LAP, [19] norms / hash
    //   profile 3: [16] pricing rounds, [17] rows augmented, [18] Dijkstra steps, [19] / [20] longest certified / scipy-order LAP
    if (profile == 3) { info[16] = w.lapstat[2]; info[17] = w.lapstat[3]; info[18] = w.lapstat[4]; info[19] = (int32_t)(w.prof[6] >> 10); info[20] = (int32_t)(w.prof[7] >> 10); return; }
    if (profile == 2) { info[16] = (int32_t)(w.prof[2] >> 10); info[17] = (int32_t)(w.prof[3] >> 10); info[18] = (int32_t)(w.prof[4] >> 10); info[19] = (int32_t)(w.prof[5] >> 10); }
    else if (profile) { info[16] = (int32_t)(w.prof[0] >> 10); info[17] = (int32_t)(w.prof[1] >> 10); info[18] = (int32_t)((w.prof[2] + w.prof[3] + w.prof[4]) >> 10); info[19] = (int32_t)(w.prof[5] >> 10); }
  }
}

__global__ __launch_bounds__(256) void gagm_large_finish_kernel(GlWs w, float* __restrict__ Uout, int32_t* __restrict__ info, int profile) {
  gl_write_result(w, (const GlCtl*)w.res, Uout, info, profile, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

// ---- [r4] the whole solve in ONE cooperative launch ------------------------------------------------------------------
// The two-launch form pays, per iteration, two kernel boundaries (launch gaps + an L2 flush / invalidate each: 55-65 us per
// iteration at cfg-3 for ~45 us of kernel time) and the host enqueues iterations in chunks with one stream synchronisation per
// chunk.  Here the iteration loop runs on the device: every workgroup evaluates the stage machine (gl_control), takes its share of
// the (row tile, K slice) items of the mul phase - a workgroup of PT threads is PT / 256 sub-groups, each working exactly like a
// workgroup of gagm_large_mul_kernel - meets the others in a grid barrier, workgroups 0..G-1 project their graph
// (gl_project_graph, the body of gagm_large_project_kernel), second barrier.  Same arithmetic in the same order: the result is
// bit-identical to the two-launch form (tests compare the two).  The barrier is a monotonic arrival counter in the workspace
// (agent-scope release before, acquire after; a workgroup that waits longer than ~2^26 cycles raises bar[1] and everybody leaves:
// the host then reports an error instead of hanging).  Launched with hipLaunchCooperativeKernel: co-residency is checked by the
// runtime; if the launch is refused the two-launch form runs.
// MEASURED (MI355X, 8 x 256 nodes, tools/bench_cfg3_solver.py, profiles/r04_cfg3_solver_one_launch.json): 80 us per Sinkhorn-stage
// iteration and 76 us per Hungarian-stage iteration against 62 and 55 us for the two-launch form.  A grid barrier across 256
// workgroups on 8 XCDs costs what a kernel boundary costs: the agent-scope acquire behind it invalidates the XCD's whole L2, so W
// (16 MB, read-only) is fetched from the memory side again in every iteration exactly as after a launch, the barrier itself is an
// atomic round trip to memory plus a polling loop, and the 576 items of the mul phase take two rounds on 512 sub-groups.  The
// single launch therefore stays an OPTION (cfg.variant = TTDG_GAGM_ONE_LAUNCH: no host synchronisation at all - the two-launch
// form reads one flag per chunk of iterations); the default is the faster two-launch form.
// `naps` = s_sleep(16) periods (~0.45 us each) between two looks at the counter.  Every look is an agent-scope load that travels to
// the memory side; 248 idle workgroups looking every 128 cycles (the first build) saturated the counter's channel and slowed the
// eight projecting workgroups' own loads fivefold (operands + S: 40 us instead of 8).
// `wrote` / `reads`: this workgroup stored data other workgroups will load / will load data other workgroups stored.  The
// agent-scope release (an L2 write-back) and acquire (an L2 + L1 invalidate) are issued by ONE wavefront per workgroup, and only
// where they are needed: with every wavefront of all 256 workgroups fencing on both sides (the first build: ~4000 write-backs and
// ~4000 invalidates per barrier) the L2s were busy with cache maintenance while the eight projecting workgroups tried to load.
__device__ __forceinline__ bool gl_grid_barrier(unsigned* bar, unsigned target, int naps, bool wrote, bool reads) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // this wavefront's stores have been acknowledged by the L2
  __syncthreads();
  __shared__ int s_bad;
  if (threadIdx.x == 0) {
    if (wrote) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int bad = 0;
    const long long t0 = (long long)__builtin_readcyclecounter();
    while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      for (int k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(16);
      if (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { bad = 1; break; }
      if ((long long)__builtin_readcyclecounter() - t0 > (1ll << 26)) { __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); bad = 1; break; }
    }
    if (reads) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_bad = bad;
  }
  __syncthreads();
  return s_bad == 0;
}

template <int PT>
__global__ __launch_bounds__(PT) void gagm_large_persistent_kernel(const float* __restrict__ Apack, const float* __restrict__ W,
                                                                   ttdg_graphs_t gr, ttdg_gagm_cfg_t cfg, GlWs w, float* __restrict__ Uout,
                                                                   int32_t* __restrict__ info, int cap) {
  extern __shared__ __attribute__((aligned(16))) float gl_smem[];
  __shared__ GlCtl s_c;
  constexpr int NSUB = PT / 256;
  const int tid = threadIdx.x, sub = tid >> 8, stid = tid & 255;
  const int nitems = w.ntiles * (w.ks + 1);
  const unsigned nb = gridDim.x;
  unsigned arrivals = 0;
  bool good = true;
  for (int t = 0; t < cap; ++t) {
    {
      const GlCtl ctl = gl_control(w, gr.G, cfg, t, blockIdx.x == 0);
      __syncthreads();                                  // (gl_control's LDS words are free again)
      if (tid == 0) s_c = ctl;                          // the word lives in LDS from here on, not in 16 registers per lane
      __syncthreads();
    }
    if (s_c.done) break;
    for (int base = blockIdx.x * NSUB; base < nitems; base += (int)nb * NSUB) {
      const int item = base + sub;
      const bool active = item < nitems;
      gl_mul_item(Apack, W, gr, w, [&]() -> GlCtl { return s_c; }, active ? item % w.ntiles : 0, active ? item / w.ntiles : 0, active, stid, gl_smem + sub * GL_MUL_LDS);
      __syncthreads();
    }
    arrivals += nb;
    const bool projects = (int)blockIdx.x < gr.G;
    if (!gl_grid_barrier(w.bar, arrivals, 1, true, projects)) { good = false; break; }          // everybody arrives within a few us
    if (projects) gl_project_graph<PT>(gr, cfg, w, s_c, blockIdx.x, gl_smem);
    arrivals += nb;
    if (!gl_grid_barrier(w.bar, arrivals, projects ? 1 : 4, projects, true)) { good = false; break; }      // the idle majority looks every ~2 us
  }
  __syncthreads();
  if (!good || !s_c.done) {      // a barrier timed out (1) or the stage machine did not stop within `cap` iterations (2): NaN out, loudly
    const size_t MU = (size_t)w.M * NU;
    for (size_t e = (size_t)blockIdx.x * PT + tid; e < MU; e += (size_t)nb * PT) Uout[e] = __int_as_float(0x7fc00000);
    if (blockIdx.x == 0 && tid == 0) info[8] = good ? 2 : 1;
    return;
  }
  gl_write_result(w, &s_c, Uout, info, (int)cfg.profile, (size_t)blockIdx.x * PT + tid, (size_t)nb * PT);
}

// entry point used by ttdg_gagm_solve (gagm.hip) when a graph has more than 128 nodes
int ttdg_gagm_large_solve(const float* Apack, const float* W, const float* U0, ttdg_graphs_t gr, ttdg_gagm_cfg_t cfg,
                          float* U, int32_t* info, void* ws, hipStream_t st) {
  int cmax = 0;
  for (int g = 0; g < gr.G; ++g) cmax = gr.off[g + 1] - gr.off[g] > cmax ? gr.off[g + 1] - gr.off[g] : cmax;
  TTDG_LIMIT(cmax <= 768, "gagm: graphs with more than 768 nodes are not supported");
  TTDG_LIMIT(gr.off[gr.G] <= 16384, "gagm: more than 16384 nodes in total");
  GlWs w = gl_carve((float*)ws, gr);
  // dynamic LDS of the projection launch: Sinkhorn (f, g, oriented matrix) or LAP (V_g copy + scratch), whichever is larger
  const int r = cmax < NU ? cmax : NU, c = cmax < NU ? NU : cmax;
  // [projector scratch: GL_VL_OFF floats][V_g tile: n x 33][LAP scratch]; the generic Sinkhorn of a < 32-node graph in
  // a large batch (rows = nodes) needs 2*32+1 + 31*33 floats and runs after the tile was consumed (it re-reads V from L2)
  size_t lapb = lap_scratch_bytes(r, c);
  if (lap_cert_scratch_bytes(c) > lapb) lapb = lap_cert_scratch_bytes(c);
  if (lap_block_scratch_bytes(c) > lapb) lapb = lap_block_scratch_bytes(c);
  const size_t bytes = (size_t)(GL_VL_OFF + ((cmax * 33 + 3) & ~3)) * sizeof(float) + lapb + 16;
  TTDG_ALLOW_LDS(gagm_large_project_kernel<512>, bytes);
  TTDG_ALLOW_LDS(gagm_large_project_kernel<1024>, bytes);
  const int M = gr.off[gr.G];
  const int cblocks = (M * NU + 255) / 256 < 256 ? (M * NU + 255) / 256 : 256;
  hipLaunchKernelGGL(gagm_large_init_kernel, dim3(cblocks), dim3(256), 0, st, U0, cfg, w);
  const long long cap = 8LL * cfg.max_iter + 8;
  if (cfg.variant & TTDG_GAGM_ONE_LAUNCH) {
    // the whole solve in one cooperative launch (gagm_large_persistent_kernel) - opt-in: measured SLOWER than the two-launch form on
    // MI355X (header of the kernel); refused launches fall through to the two-launch form
    const int pt = cmax <= 512 ? 512 : 1024, nsub = pt / 256;
    const size_t mul_bytes = (size_t)nsub * GL_MUL_LDS * sizeof(float);
    const size_t pbytes = bytes > mul_bytes ? bytes : mul_bytes;
    const void* fn = pt == 512 ? (const void*)gagm_large_persistent_kernel<512> : (const void*)gagm_large_persistent_kernel<1024>;
    if (pt == 512) TTDG_ALLOW_LDS(gagm_large_persistent_kernel<512>, pbytes); else TTDG_ALLOW_LDS(gagm_large_persistent_kernel<1024>, pbytes);
    int dev = 0, ncu = 0, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, pt, pbytes) == hipSuccess && ncu > 0 && per_cu > 0) {
      const int nitems = w.ntiles * (w.ks + 1);
      int nb = (nitems + nsub - 1) / nsub;
      if (nb < gr.G) nb = gr.G;
      if (nb > ncu) nb = ncu;              // one workgroup per CU; a longer item list is taken in rounds
      if (nb >= gr.G) {
        int cap_i = (int)cap;
        const float* a0 = Apack; const float* a1 = W;
        void* args[] = {(void*)&a0, (void*)&a1, (void*)&gr, (void*)&cfg, (void*)&w, (void*)&U, (void*)&info, (void*)&cap_i};
        if (hipLaunchCooperativeKernel(fn, dim3(nb), dim3(pt), args, pbytes, st) == hipSuccess) return ttdg_launch_status("gagm_large_persistent");
        (void)hipGetLastError();           // refused (co-residency, LDS): clear and fall through
      }
    }
  }
  // Iterations are enqueued in chunks; a converged solve turns the rest of its chunk into no-op launches (6 us each, 12 us per iteration:
  // a dependent kernel's launch latency), and every chunk ends in ONE host read of {done, total, executed} (~18 us of idle device).
  // [r6] The first chunk is sized from the previous solve of this thread: the iterations it EXECUTED, + 1 (the peek kernel evaluates the
  // stage machine itself, so exactly `executed` launch pairs already suffice; one spare costs less than a read).  Until this round the
  // hint was the reference's iteration COUNT + 2, which a Hungarian-stage cycle jump inflates to ~200: two of the five cfg-3 bench inputs
  // launched 40 pairs for 28 and 35 executed (profiles/r06_cfg3_launches.txt).  A solve that needs more continues in chunks of 4, 8, 16,
  // 32 (rounds 4-5 started every solve that way: three reads for a 20-iteration solve).  The flag lives in page-locked host memory the
  // peek kernel writes directly.  The hint and the flag buffer are the library's only state besides the error string: per host thread,
  // never read by a kernel, results do not depend on them.
  static thread_local int32_t* hflag = nullptr;
  static thread_local int last_exec = 0;
  if (!hflag) {
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped) == hipSuccess) hflag = (int32_t*)p;
    else (void)hipGetLastError();
  }
  int32_t* dflag = nullptr;
  if (hflag && hipHostGetDevicePointer((void**)&dflag, hflag, 0) != hipSuccess) { (void)hipGetLastError(); dflag = nullptr; }
  int t = 0, chunk = last_exec + 1;
  chunk = chunk < 8 ? 8 : (chunk > 64 ? 64 : chunk);
  int32_t h[16];
  for (int round = 0;; ++round) {
    for (int k = 0; k < chunk; ++k, ++t) {
      hipLaunchKernelGGL(gagm_large_mul_kernel, dim3(w.ntiles, w.ks + 1), dim3(256), 0, st, Apack, W, gr, cfg, w, t);
      if (cmax <= 512) hipLaunchKernelGGL(gagm_large_project_kernel<512>, dim3(gr.G), dim3(512), bytes, st, gr, cfg, w, t);
      else hipLaunchKernelGGL(gagm_large_project_kernel<1024>, dim3(gr.G), dim3(1024), bytes, st, gr, cfg, w, t);
    }
    if (dflag) hflag[0] = -1;
    hipLaunchKernelGGL(gagm_large_peek_kernel, dim3(1), dim3(256), 0, st, gr, cfg, w, t, dflag);
    if (int e = ttdg_launch_status("gagm_large")) return e;
    if (dflag) {
      TTDG_HIP(hipStreamSynchronize(st));     // the one convergence read per chunk (the reference reads two norms per iteration)
      h[0] = __atomic_load_n(&hflag[0], __ATOMIC_ACQUIRE);
      h[4] = hflag[1];
      h[15] = hflag[2];
      TTDG_REQUIRE(h[0] >= 0, "gagm: the convergence flag was not written");
    } else {
      TTDG_HIP(hipMemcpyAsync(h, w.res, sizeof(h), hipMemcpyDeviceToHost, st));
      TTDG_HIP(hipStreamSynchronize(st));
    }
    if (h[0]) { last_exec = h[15]; break; }
    TTDG_REQUIRE(t < cap, "gagm: the stage machine did not terminate");
    chunk = round >= 3 ? 32 : 4 << round;
  }
  hipLaunchKernelGGL(gagm_large_finish_kernel, dim3(cblocks), dim3(256), 0, st, w, U, info, (int)cfg.profile);
  return ttdg_launch_status("gagm_large_finish");
}
