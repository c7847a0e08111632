/* ttdg_mgm.h — C ABI of libttdg_mgm.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the multi-graph-matching test-time-adaptation hot path
 * of Yore0/TTDG-MGM.  The reference has no FFI of its own: the
 * path is a sequence of PyTorch ops inside adapteacher/modeling/GModule/.  Each
 * entry point below replaces one reference operator; the citation names the
 * reference lines whose arithmetic it implements (paths relative to
 * /root/reference/adapteacher/modeling/GModule/).  INTEGRATION.md shows the
 * ctypes binding a reference maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 / int32 unless stated; row-major;
 *  - the caller (PyTorch caching allocator) owns every buffer incl. workspaces,
 *    the library allocates nothing and keeps no state but a thread-local
 *    error string.  Two exceptions, both outside the product path (no caller in
 *    ttdg-mgm_amd/, used by tests/ and tools/ only) and both process-global:
 *    ttdg_debug_set_lap_variant (which of two arg-min lowerings ttdg_lap_batched
 *    uses) and ttdg_debug_set_roi_align_sliced (which of the ROIPooler kernels
 *    serves a call).  Every other A/B choice travels per call
 *    (ttdg_gagm_cfg_t.variant).  A third, on the product path and per HOST THREAD:
 *    ttdg_gagm_solve's multi-workgroup solver keeps a 64-byte page-locked buffer
 *    (its convergence flag, written by the device, read by the host without a copy)
 *    and the iteration count of the thread's previous solve (sizes the first chunk
 *    of enqueued iterations); no kernel reads either, results do not depend on them;
 *  - all work is enqueued on `stream` (a hipStream_t); no implicit device sync;
 *  - return value 0 = ok, otherwise a negative TTDG_E* code or a positive
 *    hipError_t; ttdg_last_error() describes the last failure on this thread.
 */
#ifndef TTDG_MGM_H
#define TTDG_MGM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: what this header declares is exactly what libttdg_mgm.so exports
 * (tests/test_abi.py compares the dynamic symbol table with this file). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define TTDG_VERSION 111 /* 0.1.2: ttdg_mm_t gained C2 / ldc2 / nsplit at its end (callers zero-initialise the struct and rebuild); 0.1.1: ttdg_gagm_solve writes TTDG_GAGM_INFO_WORDS = 24 int32 into `info` (0.1.0: 16, profile clocks at [8..11]);
                          * a caller built against 0.1.0 that passes a 16-word buffer must be rebuilt (INTEGRATION.md "ABI changes");
                          * new entry points ttdg_mm_f32 / ttdg_mm_workspace_bytes */
#define TTDG_GAGM_INFO_WORDS 24 /* int32 words of the `info` buffer of ttdg_gagm_solve */
#define TTDG_MAX_GRAPHS 64
#define TTDG_UNIV 32            /* universe size (rcnn.py:116) */
#define TTDG_EINVAL (-1)
#define TTDG_ELIMIT (-2)        /* shape outside what the kernels support */

typedef void* ttdg_stream_t;

int ttdg_version(void);
const char* ttdg_last_error(void);

/* Graph partition of the M = sum(n_g) stacked nodes: off[0]=0 .. off[G]=M.
 * Passed by value (kernel argument), never dereferenced on the host side of a stream. */
typedef struct {
  int32_t G;
  int32_t off[TTDG_MAX_GRAPHS + 1];
} ttdg_graphs_t;

/* ---- dense fp32 GEMM on MFMA (v_mfma_f32_32x32x2_f32, exact fp32) ----------------
 * C[m,n] = alpha * sum_k A(m,k) * B(n,k) + bias[n] + beta * C[m,n]
 * A(m,k) = A[m*sam + k*sak], B(n,k) = B[n*sbn + k*sbk], C[m*scm + n*scn]; bias may be NULL.
 * Covers nn.Linear forward (utils/affinity.py:46-47,55; utils/attentions.py:72-74),
 * its two backward products and x @ U^T (multi_graph_matching.py:531). */
int ttdg_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                  float* C, int64_t scm, int64_t scn, const float* bias, int M, int N, int K,
                  float alpha, float beta, ttdg_stream_t stream);

/* Same product with K split over `kslices` workgroup planes (weight-gradient shapes: few output tiles, K = sum n_g up to
 * thousands -- the backward of utils/affinity.py:46-47,55).  Deterministic: every slice writes its own M x N plane of
 * `ws` (ttdg_gemm_splitk_workspace_bytes), a second kernel adds the planes in a fixed order, then bias and beta * C. */
size_t ttdg_gemm_splitk_workspace_bytes(int M, int N, int kslices);
int ttdg_gemm_f32_splitk(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                         float* C, int64_t scm, int64_t scn, const float* bias, int M, int N, int K,
                         float alpha, float beta, int kslices, void* ws, ttdg_stream_t stream);

/* Grouped form: up to TTDG_GEMM_GROUP_MAX independent products in ONE launch (32 x 32 output tiles, the four wavefronts of
 * a workgroup split K).  `descs` is a HOST array; every product is
 *     C[m,n] = alpha * (sum_{k<K} A(m,k) B(n,k) + sum_{k<K2} A2(m,k) B2(n,k)) + bias[n] + beta * C[m,n]
 * (A2 / B2 / K2 = 0: no second segment).  Replaces the per-product launches behind nn.Linear forward / backward of
 * utils/affinity.py:46-47,55, utils/attentions.py:72-74 and multi_graph_matching.py:531 on the matching path. */
#define TTDG_GEMM_GROUP_MAX 8
typedef struct {
  const float *A, *B, *bias, *A2, *B2;
  float* C;
  int64_t sam, sak, sbn, sbk, scm, scn, sam2, sak2, sbn2, sbk2;
  int M, N, K, K2;
  float alpha, beta;
} ttdg_gemm_desc_t;
int ttdg_gemm_f32_grouped(const ttdg_gemm_desc_t* descs, int n, ttdg_stream_t stream);

/* column sums: out[n] = sum_m X[m*ld + n]  (bias gradients) */
int ttdg_colsum_f32(const float* X, int64_t ld, float* out, int M, int N, ttdg_stream_t stream);

/* ---- A4 affinity, decomposed form (utils/affinity.py:44-57) ----------------------
 * P = (X Psr^T) W1[:, :d]^T, Q = (X Ptg^T) W1[:, d:]^T + b1 are produced with ttdg_gemm_f32.
 * fwd: part[s][i][j] = sum_{k in slice s} w2[k] * relu(P[i,k] + Q[j,k]) for every (i,j) whose
 *      graphs satisfy g(i) >= g(j) (the reference's src_idx >= tgt_idx pairs,
 *      multi_graph_matching.py:507-513); M_ij = sum_s part[s][i][j] + b2 is folded into the
 *      Sinkhorn load.  part is (ksplit, M, M); entries with g(i) < g(j) are unspecified. */
int ttdg_affinity_pairwise_fwd(const float* P, const float* Q, const float* w2, int H, ttdg_graphs_t gr,
                               int ksplit, float* part, ttdg_stream_t stream);
/* bwd: given dM (M x M, read only where g(i) > g(j)):
 *      dP[i,k] = w2[k] * sum_j dM[i,j] [P[i,k]+Q[j,k] > 0],  dQ[j,k] likewise over i,
 *      dw2[k]  = sum_i P[i,k] S[i,k] + sum_j Q[j,k] R[j,k]  (S,R = the unscaled sums), db2 = sum dM.
 *      ws: ttdg_affinity_bwd_workspace_bytes(M,H) bytes (partial planes of the range-split reduction). */
size_t ttdg_affinity_bwd_workspace_bytes(int M, int H);
int ttdg_affinity_pairwise_bwd(const float* P, const float* Q, const float* w2, const float* dM, int H,
                               ttdg_graphs_t gr, float* dP, float* dQ, float* dw2, float* db2, void* ws,
                               ttdg_stream_t stream);

/* ---- A5 log-space Sinkhorn, pair stage (utils/sinkhorn.py:85-87 -> pygmtools [3P];
 *      call sites multi_graph_matching.py:518-525) --------------------------------
 * For every ordered pair a >= b: block = (sum_s part[s] + b2)[a-rows, b-cols], oriented rows<=cols,
 * dummy rows (-100), `iters` alternating row/col normalisations at temperature tau, exp; the
 * result is written to Wds[a,b] and (a != b) transposed to Wds[b,a]  (Wds is M x M, fully written).
 * pot receives the per-sweep potentials needed by the backward ((npairs, iters, cmax+1) floats,
 * cmax = max n_g; log2 domain; a row sweep's entry r is the dummy row's potential WITHOUT the constant fill, i.e.
 * lse_q(-g_q) [0.1.2: rounds 1-5 logged it with the fill, -100 log2(e) + that]; the buffer is opaque to callers: it only
 * travels from a forward to the matching backward of the same library version); pass NULL when no backward will follow. */
int ttdg_sinkhorn_pairs_fwd(const float* part, int ksplit, const float* b2, ttdg_graphs_t gr, float tau, int iters,
                            float* Wds, float* pot, ttdg_stream_t stream);
/* bwd: dWds (M x M; only blocks a<b are read, as the loss only touches those,
 * multi_graph_matching.py:615-631) -> dM (M x M, written for g(i) > g(j)). */
int ttdg_sinkhorn_pairs_bwd(const float* part, int ksplit, const float* b2, const float* pot, const float* dWds,
                            ttdg_graphs_t gr, float tau, int iters, float* dM, ttdg_stream_t stream);

/* ---- A4 + A5 fused: the whole pair stage of MGM3_unsup.forward in ONE launch (multi_graph_matching.py:504-525:
 *      _forward_aff -> Affinity.forward utils/affinity.py:44-57, then self.sinkhorn(..., dummy_row=True)
 *      utils/sinkhorn.py:85-87, then the symmetric fill) for graphs of at most 64 nodes each.
 * One workgroup per ordered pair a >= b: affinity block over the whole hidden dimension H, oriented rows <= cols, dummy
 * rows, `iters` sweeps in registers, exp.  Wds (M x M) is fully written; aff (M x M, may be NULL) receives the affinity
 * WITHOUT b2 for g(i) >= g(j) (= one plane of ttdg_affinity_pairwise_fwd with ksplit 1: what the backward reads);
 * pot as ttdg_sinkhorn_pairs_fwd ((npairs, iters, cmax+1) floats, NULL when no backward follows).
 * Returns TTDG_EINVAL when a graph has more than 64 nodes (use the two-launch form). */
int ttdg_pair_stage_fwd(const float* P, const float* Q, const float* w2, const float* b2, int H, ttdg_graphs_t gr, float tau,
                        int iters, float* aff, float* Wds, float* pot, ttdg_stream_t stream);
/* bwd of the Sinkhorn half (same contract as ttdg_sinkhorn_pairs_bwd with part = aff, ksplit = 1; graphs <= 64 nodes):
 * dWds (blocks a < b are read) -> dM (written for g(i) > g(j)), to be followed by ttdg_affinity_pairwise_bwd. */
int ttdg_pair_stage_bwd(const float* aff, const float* b2, const float* pot, const float* dWds, ttdg_graphs_t gr, float tau,
                        int iters, float* dM, ttdg_stream_t stream);

/* stand-alone batched Sinkhorn (the operator behind GModule.utils.sinkhorn.Sinkhorn.forward):
 * s is (b, r, c) with strides (sb, sr, sc); n1/n2 optional per-matrix valid sizes (device int32);
 * out is (b, r, c) contiguous.  Orientation, dummy rows and padding as SURVEY.md Appendix B. */
int ttdg_sinkhorn_batched_fwd(const float* s, int64_t sb, int64_t sr, int64_t sc, int b, int r, int c,
                              const int32_t* n1, const int32_t* n2, int dummy_row, float tau, int iters,
                              float* out, float* pot, ttdg_stream_t stream);
/* pot (optional, forward): (b, iters, max(r,c)+1) floats of per-sweep potentials, iters <= 64; required by the backward:
 * dout = d loss / d out (b, r, c) contiguous -> ds = d loss / d s (b, r, c) contiguous, zero on the padding. */
int ttdg_sinkhorn_batched_bwd(const float* s, int64_t sb, int64_t sr, int64_t sc, int b, int r, int c,
                              const int32_t* n1, const int32_t* n2, int dummy_row, float tau, int iters,
                              const float* pot, const float* dout, float* ds, ttdg_stream_t stream);

/* ---- A3 intra-graph attention adjacency (utils/attentions.py:60-86, v2, 1 head) --
 * q, k: (M, d) projections (ttdg_gemm_f32).  Apack receives, per graph, softmax(q k^T * scale)
 * with the diagonal zeroed when zero_diag != 0 (multi_graph_matching.py:496-502), packed block after block
 * (block g at offset sum_{h<g} n_h^2).  drop_p > 0 applies train-mode dropout on the attention
 * (attentions.py:40) from a Philox stream keyed by (seed, graph, row, col). */
int ttdg_mha_adjacency(const float* q, const float* k, int d, ttdg_graphs_t gr, float scale, float drop_p,
                       uint64_t seed, int zero_diag, float* Apack, ttdg_stream_t stream);

/* ---- A6+A7 graduated-assignment multi-graph matching, whole solve on device ------
 * (multi_graph_matching.py:223-244, 300-389 with num_clusters == 1; utils/hungarian.py:8-66 ->
 * scipy.optimize.linear_sum_assignment [3P] re-implemented on device, one wavefront per LAP.)
 * Apack as above, W = Wds (M x M), U0 (M x 32).  U (M x 32) receives the 0/1 matching.
 * info (int32[TTDG_GAGM_INFO_WORDS], device, zero-initialised by the caller): [0..5] iterations per stage, [6] total, [7] stages run,
 * [8] STATUS and nothing else (0 = ok; the cooperative multi-workgroup kernel: 1 = a grid barrier timed out, 2 = the stage
 * machine did not stop - U is then NaN); [12] Hungarian-stage LAPs solved with a uniqueness certificate, [13] LAPs that fell back to
 * the scipy-order solver (multi-workgroup solver), [21] narrow-range LAPs solved by the integer scipy-order solver; [14], [15] period and detection iteration of a Hungarian-stage cycle; [22] iterations the multi-workgroup solver executed (launch pairs that did work: [6] also counts the iterations a cycle jump skipped);
 * cfg.profile != 0: single-workgroup kernel [9..13] = cycle-counter ticks / 64 spent in B, S, V, projection, convergence;
 * multi-workgroup solver [16..20] = cycles / 1024 per phase (csrc/gagm_large.hip: gl_write_result).
 * ws: workspace of ttdg_gagm_workspace_bytes(M) bytes; its first 2*M*32 floats receive the
 * first-iteration V and the first projected U (parity tests). */
typedef struct {
  float tau0, gamma, min_tau, tol, quad_weight;
  int32_t max_iter, sk_iter;
  int32_t max_stages;        /* 0 = run the full schedule; k > 0 stops after k stages (parity tests) */
  int32_t start_hungarian;   /* non-zero: the first stage already uses the Hungarian projector (parity tests) */
  int32_t no_cycle_skip;     /* non-zero: disable the exact Hungarian-stage cycle shortcut (parity tests) */
  int32_t profile;           /* non-zero: phase clocks in info[9..13] (single-workgroup kernel) / info[16..20] (multi-workgroup solver) */
  int32_t variant;           /* 0 = the product path.  A/B and parity-test selectors, per call (the library keeps no switches):
                              *   TTDG_GAGM_LDS_PROJECTORS    round 1's LDS-exchange Sinkhorn projectors for graphs of <= 64 nodes
                              *   TTDG_GAGM_FORCE_LARGE       the multi-workgroup solver even where one workgroup would do
                              *   TTDG_GAGM_FORCE_SINGLE      the single-workgroup kernel wherever it can run (every graph <= 128 nodes)
                              *   TTDG_GAGM_256_THREADS       single-workgroup kernel built for 256 threads (graphs <= 64 nodes)
                              *   TTDG_GAGM_COLUMN_PROJECTOR  multi-workgroup solver: round 3's column-per-thread Sinkhorn projector
                              *   TTDG_GAGM_ONE_LAUNCH        multi-workgroup solver: the whole solve in one cooperative launch (device-side iteration
                              *                               loop, grid barriers, no host synchronisation; bit-identical, measured slower -
                              *                               default: two launches per iteration enqueued by the host in chunks)
                              *   TTDG_GAGM_SCIPY_ORDER_LAP   multi-workgroup solver: every Hungarian-stage LAP by the one-wavefront scipy-order
                              *                               solver (default: warm-started workgroup LAP + uniqueness certificate,
                              *                               csrc/lap_certified.h, scipy-order only when the certificate fails)
                              *   TTDG_GAGM_NO_INT_LAP        multi-workgroup solver: uncertifiable blocks of a narrow value range (what follows a
                              *                               collapsed Sinkhorn stage) also go through the fp64 step-by-step scipy-order solver
                              *                               (default: its integer statement, lap_wave_solve_int - same decisions) */
} ttdg_gagm_cfg_t;
#define TTDG_GAGM_LDS_PROJECTORS 1
#define TTDG_GAGM_FORCE_LARGE 2
#define TTDG_GAGM_FORCE_SINGLE 4
#define TTDG_GAGM_256_THREADS 8
#define TTDG_GAGM_COLUMN_PROJECTOR 16
#define TTDG_GAGM_SCIPY_ORDER_LAP 32
#define TTDG_GAGM_ONE_LAUNCH 64
#define TTDG_GAGM_NO_INT_LAP 128
size_t ttdg_gagm_workspace_bytes(int M);
int ttdg_gagm_solve(const float* Apack, const float* W, const float* U0, ttdg_graphs_t gr, ttdg_gagm_cfg_t cfg,
                    float* U, int32_t* info, void* ws, ttdg_stream_t stream);

/* batched LAP (maximise), the operator behind GModule.utils.hungarian.hungarian:
 * s (b, r, c) contiguous -> x (b, r, c) 0/1.  One wavefront per matrix, fp64 duals,
 * tie-breaking identical to scipy's rectangular LSAP. min(r,c) <= 64, max(r,c) <= 256. */
int ttdg_lap_batched(const float* s, int b, int r, int c, float* x, ttdg_stream_t stream);
/* benchmarking aid: 0 = compiler-lowered fp64 arg-min (default), 1 = hand-scheduled inline asm; affects ttdg_lap_batched only */
int ttdg_debug_set_lap_variant(int v);
/* micro-benchmark hook: `reps` projections (mode 0 Sinkhorn with the product's block-layout projector, 1 LAP, 2 the
 * LDS-exchange Sinkhorn projectors of round 1) of G graphs x n nodes from LDS, one wavefront per
 * graph, as inside ttdg_gagm_solve; ticks[0] receives the shader-cycle count. */
int ttdg_debug_project(const float* V, int n, int G, float tau, int iters, int reps, int mode, float* U,
                       long long* ticks, ttdg_stream_t stream);

/* ---- A8+A9 pseudo-label permutation loss (multi_graph_matching.py:535-564,
 *      utils/losses.py:83-103,419-455) ---------------------------------------------
 * loss = mean over pairs a<b of mean over elements of focal-BCE(clamp(Wds[a,b]), U_a U_b^T).
 * dWds (M x M) is fully written (zero outside the a<b blocks) with d loss / d Wds.
 * flag[0] is set to 1 if any Wds entry of an a<b block is outside [0,1] (losses.py:437-439).
 * pair_ws: ttdg_perm_loss_workspace_bytes(gr) of scratch (one partial sum per 64 x 64 tile of every pair block,
 * added in a fixed order). */
size_t ttdg_perm_loss_workspace_bytes(ttdg_graphs_t gr);
int ttdg_perm_loss_fwd_bwd(const float* Wds, const float* U, ttdg_graphs_t gr, float alpha, float eps,
                           float* loss, float* dWds, int32_t* flag, float* pair_ws, ttdg_stream_t stream);

/* ---- A2 node sampler (build_graph.py:27-115,133-250) -----------------------------
 * Step 1: per image, per FPN location: label of the minimum-area box that contains the point and
 * cares about the level (0 = none).  boxes (B, kmax, 4) xyxy, classes (B, kmax) int32, nbox (B).
 * Level l has h[l] x w[l] points at stride[l]; labels is (B, npts_total) int32.  B here counts the
 * images that HAVE boxes: the reference skips box-less images when it builds labels but indexes
 * features by list position (build_graph.py:79 vs :173-181); the host wrapper reproduces that pairing.
 * Step 2: per (image, level): positives in raster order, step = cnt // sample_dist, keep [::step]
 * when step > 1; sel_idx (B, cap) receives level-local point ids tagged with the level
 * (id | level << 28), sel_lab the labels, count[B] the node counts (cap >= 5 * (2*sample_dist - 1)).
 * Step 3/4: gather / scatter-add of the selected 256-d feature columns from/to NCHW maps. */
#define TTDG_MAX_LEVELS 8
typedef struct {
  int32_t n;                       /* number of FPN levels (5: p2..p6) */
  int32_t h[TTDG_MAX_LEVELS], w[TTDG_MAX_LEVELS], stride[TTDG_MAX_LEVELS];
  float lo[TTDG_MAX_LEVELS], hi[TTDG_MAX_LEVELS];   /* object_sizes_of_interest (build_graph.py:28-33) */
} ttdg_levels_t;
typedef struct {
  int32_t n, C;
  int32_t h[TTDG_MAX_LEVELS], w[TTDG_MAX_LEVELS];
  float* feat[TTDG_MAX_LEVELS];    /* NCHW maps (B, C, h, w), one per level */
} ttdg_fpn_t;
int ttdg_node_labels(const float* boxes, const int32_t* classes, const int32_t* nbox, int B, int kmax,
                     ttdg_levels_t lv, int32_t* labels, ttdg_stream_t stream);
int ttdg_node_select(const int32_t* labels, int B, ttdg_levels_t lv, int sample_dist, int cap,
                     int32_t* sel_idx, int32_t* sel_lab, int32_t* count, ttdg_stream_t stream);
/* img[i], pid[i] (= point | level << 28) name node i; out/dout are (n, C) row-major.
 * The backward ADDS into the (caller-zeroed) gradient maps. */
int ttdg_node_gather_fwd(ttdg_fpn_t fp, const int32_t* img, const int32_t* pid, int n, float* out,
                         ttdg_stream_t stream);
int ttdg_node_gather_bwd(ttdg_fpn_t dfp, const int32_t* img, const int32_t* pid, int n, const float* dout,
                         ttdg_stream_t stream);
/* the same on channels-last maps: fp.feat[l] is (B, H_l, W_l, C) contiguous (the layout the backbone runs in) */
int ttdg_node_gather_fwd_nhwc(ttdg_fpn_t fp, const int32_t* img, const int32_t* pid, int n, float* out, ttdg_stream_t stream);
int ttdg_node_gather_bwd_nhwc(ttdg_fpn_t dfp, const int32_t* img, const int32_t* pid, int n, const float* dout,
                              ttdg_stream_t stream);

/* ---- A11 fused multi-tensor SGD (torch.optim.SGD as built by detectron2 [3P];
 *      engine/trainer.py:480-482) ---------------------------------------------------
 * For every tensor t: d = g + wd[t]*p; buf = first[t] ? d : momentum*buf + d; p -= lr*buf.
 * table: device array of ntensors descriptors; chunk tables map thread blocks to tensors. */
typedef struct {
  float* p;
  const float* g;
  float* buf;
  int64_t n;
  float wd;
  int32_t first;
} ttdg_sgd_tensor_t;
int ttdg_sgd_multi_tensor(const ttdg_sgd_tensor_t* table, const int32_t* chunk_tensor, const int64_t* chunk_off,
                          int nchunks, int chunk, float lr, float momentum, ttdg_stream_t stream);

/* ---- FrozenBN scale folded into many convolution filters at once (detectron2 FrozenBatchNorm2d [3P] behind every ResNet
 *      convolution of the backbone called at rcnn.py:219 / :181): out_t[r][j] = in_t[r][j] * scale_t[r], row = output channel;
 *      also its backward (grad_w = grad_wf * scale).  Up to TTDG_ROW_SCALE_MAX tensors per launch, table passed by value. */
#define TTDG_ROW_SCALE_MAX 64
typedef struct {
  const float* in;
  const float* scale;
  float* out;
  int rows, rowlen;
} ttdg_row_scale_t;
int ttdg_row_scale_multi(const ttdg_row_scale_t* items, int n, ttdg_stream_t stream);

/* ---- detection helpers of the torch-native Mask R-CNN stand-in (stands in for the
 *      un-vendored detectron2 ROIAlignV2 / torchvision nms [3P]); forward-only on the TTA path -------------
 * rois (R,5) = (batch idx, x1,y1,x2,y2) image coords; feat (B,C,H,W); out (R,C,P,P); aligned, adaptive sampling. */
int ttdg_roi_align_fwd(const float* feat, int B, int C, int H, int W, const float* rois, int R, float scale, int P,
                       float* out, ttdg_stream_t stream);
/* greedy NMS over boxes (N,4) pre-sorted by descending score; boxes with different group ids never interact.
 * mask_ws: N*ceil(N/64) uint64 of scratch; keep (N) receives kept indices, nkeep their count. */
int ttdg_nms(const float* boxes, const int32_t* group, int N, float thr, void* mask_ws, int32_t* keep,
             int32_t* nkeep, ttdg_stream_t stream);

/* grouped variant: boxes sorted by (group, descending score); seg (ngroups+1 device ints) delimits the groups;
 * max_group = an upper bound of the largest group (sizes the launch and the N x ceil(max_group/64) uint64 scratch);
 * flags (N bytes) receives 1 for kept boxes.  One wavefront per group: the groups are swept concurrently. */
int ttdg_nms_grouped(const float* boxes, const int32_t* seg, int ngroups, int N, int max_group, float thr,
                     void* mask_ws, unsigned char* flags, ttdg_stream_t stream);

/* Fused box pipelines of the stand-in detector (detectron2 Box2BoxTransform.apply_deltas + Boxes.clip + the validity
 * filters of find_top_rpn_proposals / fast_rcnn_inference [3P]); candidates that fail a filter get score -inf instead of
 * being compacted away, so every later step keeps static shapes.  sizes: (B, 2) floats = (height, width) per image.
 * rpn_decode: one FPN level; deltas (B, A*4, H, W) as produced by the head, anchors (H*W*A, 4), idx/score (B, k) = the
 *   level's top-k over the (h, w, a) raster; writes boxes (B, K, 4) / scores (B, K) at columns [col0, col0 + k).
 * box_inference: logits (N, C+1), deltas (N, 4C), rois (N, 5) = (image, x1, y1, x2, y2); softmax, per-class decode with
 *   weights (wx, wy, ww, wh), clip, score > score_thresh; boxes (N, C, 4), scores (N, C).
 * paste_masks: soft masks (R, S, S) -> (R, H, W) bytes (0/1): bilinear resampling inside boxes (R, 4), >= threshold. */
int ttdg_rpn_decode(const float* deltas, const float* anchors, const int64_t* idx, const float* score, const float* sizes,
                    int B, int k, int A, int H, int W, int K, int col0, float* boxes, float* scores, ttdg_stream_t stream);
/* rpn_select: find_top_rpn_proposals [3P] up to the NMS for ALL levels and images in one launch - per (image, level) the k best
 *   logits (B, A, H, W as produced by the head) in descending order (ties: ascending (h, w, a) raster index), decoded as rpn_decode
 *   does, into columns [col0, col0 + k) of boxes (B, K, 4) / scores (B, K).  k <= min(2048, A*H*W).  Replaces per level a
 *   permute + torch.topk + rpn_decode. */
#define TTDG_RPN_LEVELS_MAX 8
typedef struct {
  const float* logits;
  const float* deltas;
  const float* anchors;
  int H, W, k, col0;
} ttdg_rpn_level_t;
int ttdg_rpn_select(const ttdg_rpn_level_t* levels, int nlevels, int B, int A, const float* sizes, int K, float* boxes,
                    float* scores, ttdg_stream_t stream);
/* the same with channels-last head outputs: logits (B, H, W, A), deltas (B, H, W, 4A) in storage */
int ttdg_rpn_select_nhwc(const ttdg_rpn_level_t* levels, int nlevels, int B, int A, const float* sizes, int K, float* boxes,
                         float* scores, ttdg_stream_t stream);
int ttdg_box_inference(const float* logits, const float* deltas, const float* rois, const float* sizes, int N, int C,
                       float wx, float wy, float ww, float wh, float score_thresh, float* boxes, float* scores,
                       ttdg_stream_t stream);
int ttdg_paste_masks(const float* masks, const float* boxes, int R, int S, int H, int W, float threshold,
                     unsigned char* out, ttdg_stream_t stream);

/* y (N, C, H*W) <- act(y + bias[c] (+ residual) (+ bias2[c])) in place; bias / residual / bias2 may be NULL; relu != 0
 * applies max(., 0), evaluated as (y + bias) + (residual + bias2).  The shift of a folded FrozenBN, the residual add and the
 * ReLU of a bottleneck in one pass. */
int ttdg_bias_act(float* y, const float* bias, const float* residual, const float* bias2, int N, int C, int HW,
                  int relu, ttdg_stream_t stream);
/* the same epilogue on a channels-last activation: y (rows, C) with rows = N*H*W, C % 4 == 0, 16-byte aligned pointers */
int ttdg_bias_act_nhwc(float* y, const float* bias, const float* residual, const float* bias2, int64_t rows, int C, int relu,
                       ttdg_stream_t stream);
/* backward of that epilogue with relu != 0: gin[e] = out[e] > 0 ? gout[e] : 0 over `total` floats (torch threshold_backward
 * behind F.relu_, one pass; the result is the gradient of both the convolution output and the residual branch). */
int ttdg_relu_bwd(const float* gout, const float* out, float* gin, size_t total, ttdg_stream_t stream);

/* The backbone's pointwise (1 x 1) convolutions in channels-last memory as ONE streaming fp32 MFMA product with the epilogue
 * above applied where the accumulators leave the matrix cores (the detectron2 Conv2d + FrozenBatchNorm2d + F.relu_ blocks [3P]
 * that rcnn.py:331-345 runs; twice per adapted batch, trainer.py:469-485), and their two backward products:
 *     C[m, n] = act( sum_k A'(m, k) B(n, k) + bias[n] + (res[r(m), n] + bias2[n]) ),   A' = A or relu(A + pbias[k])
 * a_layout / b_layout: 0 = the reduction index k is the contiguous one (A[m * lda + k]), 1 = the output index is
 * (A[k * lda + m]); (1, 0) is not built.  a_stride > 1: A row m = pixel (img, ho, wo) of the strided output map reads input pixel
 * (img, a_stride * ho, a_stride * wo) of an (a_h, a_w) map (a stride-2 pointwise convolution).  res_up: the residual row of
 * pixel (img, h, w) of a (res_h, res_w) map is pixel (img, h / 2, w / 2) of the half-size map (the FPN's top-down sum).
 * kslices > 1: the reduction is split over workgroup planes in `ws` (ttdg_mm_workspace_bytes) and added in a fixed order by a
 * second kernel (weight gradients: dW = dY^T X over hundreds of thousands of pixels); no residual / ReLU then.
 * tile: 0 = chosen by the library, 1 + code (code bit 1: 64 instead of 128 rows, bit 0: 64 instead of 128 columns) = forced.
 * N, K (k-contiguous operands), leading dimensions: multiples of 4; every pointer 16-byte aligned.  The order of the additions
 * inside an accumulator is fixed per shape (deterministic), not ascending in k (csrc/pointwise.hip). */
typedef struct {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* res;
  const float* bias2;
  const float* pbias;
  void* ws;
  int64_t lda, ldb, ldc, ldres;
  int32_t M, N, K;
  int32_t a_layout, b_layout;
  int32_t a_stride, a_h, a_w;
  int32_t res_up, res_h, res_w;
  int32_t relu, prelu;
  int32_t kslices;
  int32_t tile;
  /* optional second reduction segment (A2 != NULL): + sum_{k < K2} A2(m, k) B2(n, k), both k-contiguous, A2 with its own strided
   * row map (a2_stride / a2_h / a2_w as a_stride / a_h / a_w) - the projection shortcut of a bottleneck's first block accumulated
   * into conv3's product: bias = conv3's shift, bias2 = the shortcut's, no residual.  K and K2 multiples of 32; 64 x 64 tiles. */
  const float* A2;
  const float* B2;
  int64_t lda2, ldb2;
  int32_t K2;
  int32_t a2_stride, a2_h, a2_w;
  /* [0.1.2] optional split output (C2 != NULL): columns [nsplit, N) are written to C2[m * ldc2 + (n - nsplit)] instead of C - two heads
   * that read one input (the RPN's objectness and anchor-delta filters) in one pass.  nsplit, ldc2 multiples of 4; no split reduction. */
  float* C2;
  int64_t ldc2;
  int32_t nsplit;
} ttdg_mm_t;
size_t ttdg_mm_workspace_bytes(int M, int N, int kslices);
int ttdg_mm_f32(const ttdg_mm_t* desc, ttdg_stream_t stream);
/* up to 8 plain products (no input activation / second segment / row maps / forced tile) that share their operand layouts in ONE launch
 * (+ one reduce launch for the split ones): the nn.Linear-shaped products of a matching step beyond 512 stacked nodes, 0.3 - 0.5 GFLOP each
 * (utils/affinity.py:46-47,55, utils/attentions.py:72-74, multi_graph_matching.py:531 and their gradients).  `descs` is a HOST array. */
int ttdg_mm_f32_grouped(const ttdg_mm_t* descs, int n, ttdg_stream_t stream);

/* detectron2 ROIPooler [3P] in one launch: every ROI (image, x1, y1, x2, y2) picks its FPN level
 * clamp(floor(canonical_level + log2(sqrt(area) / canonical_size + 1e-8)), min_level, min_level + fp.n - 1) inside the
 * kernel and is ROIAlign-ed (aligned, adaptive sampling) from that level's map; lv.stride[l] gives the level scales.
 * out (R, C, P, P).  No per-level compaction, hence no host read. */
int ttdg_roi_align_multilevel(ttdg_fpn_t fp, ttdg_levels_t lv, const float* rois, int R, int P, float canonical_size,
                              int canonical_level, int min_level, float* out, ttdg_stream_t stream);
/* The same pooler on channels-last copies of the FPN maps (fp.feat[l] = (B, H_l, W_l, C)): a lane is a channel, every tap one
 * coalesced read per wavefront; ttdg_nchw_to_nhwc makes the copies ((B, C, H, W) -> (B, H, W, C), tiled through LDS), once per
 * forward for every pooler call that shares the maps.  out is (R, C, P, P) as above; P <= 14. */
int ttdg_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, ttdg_stream_t stream);
int ttdg_roi_align_multilevel_nhwc(ttdg_fpn_t fp, ttdg_levels_t lv, const float* rois, int R, int P, float canonical_size,
                                   int canonical_level, int min_level, float* out, ttdg_stream_t stream);
/* A/B hook for ttdg_roi_align_multilevel: 2 (default) = separable table kernel, one workgroup per (ROI, channel slice = XCD);
 * 1 = direct per-output kernel with the XCD-sliced work mapping; 0 = direct kernel, flat mapping (round 1). */
int ttdg_debug_set_roi_align_sliced(int on);
/* ---- N4 loader: the test mapper's resize on the device (DatasetMapper(is_train=False) -> ResizeShortestEdge [3P],
 *      reference data/build.py:122-154, iterated at engine/trainer.py:470,485) ----------------------------------------
 * src: `planes` contiguous uint8 images of H x W (planes = batch x channels), dst: planes x OH x OW uint8.  Bilinear,
 * align_corners = False, antialiased when OH < H or OW < W (then `ws` must hold ttdg_resize_u8_workspace_bytes bytes),
 * round-half-even, clamp: the arithmetic of torch's F.interpolate on float32 as ttdg_mgm_amd.data.map_for_test runs it. */
size_t ttdg_resize_u8_workspace_bytes(int planes, int H, int W, int OH, int OW);
int ttdg_resize_bilinear_u8(const unsigned char* src, unsigned char* dst, int planes, int H, int W, int OH, int OW, void* ws,
                            ttdg_stream_t stream);


/* DiceEvaluator reductions (reference evaluation/dice_metric.py:25-92 with enhanced_align :110-143 and
 * Structure_measure :147-240): for `npairs` (predicted mask, same-class ground-truth mask) pairs of H x W byte maps (0/1,
 * rows contiguous; pred[i] / gt[i] are DEVICE ADDRESSES), the counts n(p AND g), n(p), n(g) in the four quadrants of
 * the image split at row cy[i] / column cx[i] (the S-measure's centroid split, :196-214):
 *   counts[i][(2*(row >= cy) + (col >= cx)) * 3 + {0, 1, 2}].
 * Dice, E-measure and S-measure of boolean maps are closed forms of these twelve integers (the host mirror evaluates them
 * in float64); one launch covers every kept prediction of an eval batch.  Zeroes `counts` itself. */
int ttdg_mask_pair_counts(const unsigned long long* pred, const unsigned long long* gt, const int32_t* cy, const int32_t* cx,
                          int npairs, int H, int W, int32_t* counts, ttdg_stream_t stream);
/* The closed forms themselves (dice_metric.py:54-66, :110-143, :147-240 for boolean maps), float64, one thread per pair, and the
 * maximum over the same-class ground truths of a prediction (:76-92): best[owner[i]][{0,1,2}] = max(best, 100 * {Dice, E, S}_i).
 * best (npred, 3) doubles must be zero-initialised (a prediction without a same-class ground truth scores 0). */
int ttdg_mask_measures(const int32_t* counts, const int32_t* cy, const int32_t* cx, const int32_t* owner, int npairs, int H, int W,
                       double alpha, double* best, ttdg_stream_t stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* TTDG_MGM_H */
