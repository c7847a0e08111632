"""Operators over the C ABI (include/ttdg_mgm.h) + their autograd wiring.

Every function here launches hand-written HIP kernels through ``_lib.call``;
PyTorch only provides device memory, the current stream and the autograd tape.
"""
import ctypes as C

import os

import torch

from . import _lib
from ._lib import call, check_f32, graphs, ptr, stream

DIM, HID, UNIV = 256, 512, 32

# bench.py sets this to a list to have HIP events recorded (on the launch stream) around the dominant kernel
KERNEL_TIMERS = None
# An event pair is two marker packets in the launch queue: around EVERY hand-written launch (~250 pairs per adapted batch) they cost
# 1.5 ms of a 38.8 ms batch (bench.py --no-kernel-timers, profiles/r06_event_overhead.txt).  The kernels launched dozens of times per
# batch are therefore stamped every KERNEL_TIMER_EVERY-th launch (7: coprime with their per-batch launch counts, so the sampled
# positions rotate through all layers over the timed batches); kernels launched once or twice per batch are always stamped.
KERNEL_TIMER_EVERY = 1
KERNEL_TIMER_SAMPLED = ("bias_act", "relu_bwd", "pointwise_fwd", "pointwise_dx", "pointwise_dw")
KERNEL_TIMER_ONLY = None          # a set of stamp names: only these are recorded (the headline pass stamps its dominant kernel only)
_timer_counts = {}


class _timed:
    """Records (name, start_event, end_event, meta, None) into KERNEL_TIMERS around a launch when bench.py asks for it."""

    def __init__(self, name, meta):
        self.name, self.meta, self.on = name, meta, KERNEL_TIMERS is not None and (KERNEL_TIMER_ONLY is None or name in KERNEL_TIMER_ONLY)
        if self.on and KERNEL_TIMER_EVERY > 1 and name in KERNEL_TIMER_SAMPLED:
            n = _timer_counts.get(name, 0)
            _timer_counts[name] = n + 1
            self.on = n % KERNEL_TIMER_EVERY == 0

    def __enter__(self):
        if self.on:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            KERNEL_TIMERS.append((self.name, self.e0, self.e1, self.meta, None))
        return False


def is_channels_last(t):
    """A 4-D tensor stored (N, H, W, C) and NOT also NCHW-contiguous (1 x 1 maps / one channel are both: NCHW code paths apply)."""
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)


def like_layout(t, ref):
    """``t`` as a dense tensor with the memory layout of ``ref`` (contiguous or channels-last)."""
    if is_channels_last(ref):
        return t if is_channels_last(t) or (t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)) else t.contiguous(memory_format=torch.channels_last)
    return t.contiguous()


# ------------------------------------------------------------------------------------------- GEMM
def gemm(A, sam, sak, B, sbn, sbk, Cout, scm, scn, M, N, K, bias=None, alpha=1.0, beta=0.0,
         a_off=0, b_off=0, c_off=0):
    """C[m,n] = alpha * sum_k A(m,k) B(n,k) + bias[n] + beta*C[m,n] (element strides; offsets in elements).
    Few output tiles with a long K (weight gradients at cfg-3: 256x256 outputs, K = 2048) go through the split-K entry."""
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    if tiles <= 64 and K >= 1024:
        ks = max(2, min(16, 256 // tiles, K // 128))
        ws = torch.empty(ks * M * N, device=Cout.device, dtype=torch.float32)
        call("ttdg_gemm_f32_splitk", ptr(A) + 4 * a_off, sam, sak, ptr(B) + 4 * b_off, sbn, sbk, ptr(Cout) + 4 * c_off, scm, scn,
             ptr(bias), M, N, K, float(alpha), float(beta), ks, ptr(ws), stream())
        return Cout
    call("ttdg_gemm_f32", ptr(A) + 4 * a_off, sam, sak, ptr(B) + 4 * b_off, sbn, sbk, ptr(Cout) + 4 * c_off, scm, scn,
         ptr(bias), M, N, K, float(alpha), float(beta), stream())
    return Cout


def gdesc(A, sam, sak, B, sbn, sbk, Cout, scm, scn, M, N, K, bias=None, alpha=1.0, beta=0.0, a_off=0, b_off=0, c_off=0, second=None):
    """One product of a grouped launch (same argument meaning as ``gemm``); ``second`` = (A2, sam2, sak2, B2, sbn2, sbk2, K2
    [, a2_off, b2_off]) adds a second K segment into the same accumulators."""
    d = _lib.GemmDesc()
    d.A, d.B, d.C, d.bias = ptr(A) + 4 * a_off, ptr(B) + 4 * b_off, ptr(Cout) + 4 * c_off, ptr(bias)
    d.sam, d.sak, d.sbn, d.sbk, d.scm, d.scn = sam, sak, sbn, sbk, scm, scn
    d.M, d.N, d.K, d.K2, d.alpha, d.beta = M, N, K, 0, float(alpha), float(beta)
    if second is not None:
        A2, sam2, sak2, B2, sbn2, sbk2, K2 = second[:7]
        o2a, o2b = (second[7], second[8]) if len(second) > 7 else (0, 0)
        d.A2, d.B2, d.sam2, d.sak2, d.sbn2, d.sbk2, d.K2 = ptr(A2) + 4 * o2a, ptr(B2) + 4 * o2b, sam2, sak2, sbn2, sbk2, K2
    return d


def gemm_grouped(descs):
    """Up to 8 independent products in ONE launch (csrc/gemm_grouped.hip).  The tensors behind the descriptors must stay alive
    until the launch is enqueued (they do: the callers hold them)."""
    for i in range(0, len(descs), _lib.GEMM_GROUP_MAX):
        chunk = descs[i:i + _lib.GEMM_GROUP_MAX]
        arr = (_lib.GemmDesc * len(chunk))(*chunk)
        call("ttdg_gemm_f32_grouped", arr, len(chunk), stream())


GROUPED_GEMM = True      # False = one launch per product (parity tests compare the two)
GROUPED_GEMM_MAX_ROWS = 512      # stacked nodes up to which the 32 x 32-tile grouped kernel is used: it is built for M ~ 120 (launch-bound
                                 # products); at cfg-3 (M = 2048) the 64 x 64-tile kernel is twice as fast per product (84 vs 2 x 22 us)


def linear_raw(x, W, b=None, w_col_off=0, w_ld=None, out=None):
    """y = x @ W[:, off:off+K]^T + b  (x: (M,K) contiguous, W row-major with leading dimension w_ld)."""
    M, K = x.shape
    N = W.shape[0]
    ld = W.shape[1] if w_ld is None else w_ld
    y = out if out is not None else torch.empty(M, N, device=x.device, dtype=torch.float32)
    gemm(x, K, 1, W, ld, 1, y, N, 1, M, N, K, bias=b, b_off=w_col_off)
    return y


def _mm_desc(A, B, out, M, N, K, lda, ldb, ldc, bias=None, res=None, ldres=0, bias2=None, pbias=None, relu=False, prelu=False,
             a_layout=0, b_layout=0, a_stride=0, a_hw=(0, 0), res_up=False, res_hw=(0, 0), kslices=0, tile=0, second=None, split=None):
    """(ttdg_mm_t, workspace tensor to keep alive until the launch is enqueued)"""
    ws = torch.empty(kslices * M * N, device=out.device, dtype=torch.float32) if kslices > 1 else None
    # (positional: the field order of ttdg_mm_t; one constructor call instead of 27 attribute stores on the launch path)
    d = _lib.Mm(ptr(A), ptr(B), ptr(out), ptr(bias), ptr(res), ptr(bias2), ptr(pbias), ptr(ws), int(lda), int(ldb), int(ldc), int(ldres),
                int(M), int(N), int(K), int(a_layout), int(b_layout), int(a_stride), int(a_hw[0]), int(a_hw[1]),
                int(bool(res_up)), int(res_hw[0]), int(res_hw[1]), int(bool(relu)), int(bool(prelu)), int(kslices), int(tile))
    if second is not None:          # (A2, B2, lda2, ldb2, K2, stride, (H, W)): a second reduction segment into the same accumulators
        A2, B2, lda2, ldb2, K2, st2, hw2 = second
        d.A2, d.B2, d.lda2, d.ldb2, d.K2, d.a2_stride, d.a2_h, d.a2_w = ptr(A2), ptr(B2), int(lda2), int(ldb2), int(K2), int(st2), int(hw2[0]), int(hw2[1])
    if split is not None:           # (out2, ldc2, nsplit): columns [nsplit, N) go to out2
        d.C2, d.ldc2, d.nsplit = ptr(split[0]), int(split[1]), int(split[2])
    return d, ws


def mm(A, B, out, M, N, K, lda, ldb, ldc, **kw):
    """ttdg_mm_f32 (csrc/pointwise.hip): out[m, n] = act(sum_k A'(m, k) B(n, k) + bias[n] + (res[r(m), n] + bias2[n])).  Tensors are
    passed as storage (pointer + leading dimensions); see include/ttdg_mgm.h for the row maps and layouts (keywords: _mm_desc)."""
    d, ws = _mm_desc(A, B, out, M, N, K, lda, ldb, ldc, **kw)
    call("ttdg_mm_f32", C.byref(d), stream())
    return out


def mm_grouped(products):
    """ttdg_mm_f32_grouped: up to 8 plain products of one operand-layout class in ONE launch.  ``products``: list of (args, kwargs) of mm()."""
    for i in range(0, len(products), 8):
        built = [_mm_desc(*a, **k) for a, k in products[i:i + 8]]
        arr = (_lib.Mm * len(built))(*[d for d, _ in built])
        call("ttdg_mm_f32_grouped", arr, len(built), stream())


# The nn.Linear-shaped products of a matching step with more than GROUPED_GEMM_MAX_ROWS stacked nodes (cfg-3: 2048) on the streaming
# product of csrc/pointwise.hip (64 x 64 tiles, five workgroups per CU, LDS-DMA: 8.6 - 9.2 us per 2048-row projection against 10.2 -
# 11.6 us for gemm_f32's one workgroup per CU; weight gradients over 8 row slices 15.0 against 16.4 us) - "gemm" = the round-4 kernel
# (ascending-k accumulation; the parity tests compare the two).  "mm_grouped" (default): the same products, those that do not depend on
# each other in ONE launch per operand-layout class (ttdg_mm_f32_grouped: 21 launches per forward + backward -> 9).
LARGE_GEMM_ENGINE = "mm_grouped"


def _lin_spec(x, W, y, b=None, col_off=0, ld=None):
    M, K = x.shape
    Wv = W if col_off == 0 else W[:, col_off:]
    return (x, Wv, y, M, W.shape[0], K, K, W.shape[1] if ld is None else ld, y.shape[1]), dict(bias=b)


def _dx_spec(dy, W, dx, col_off=0):
    M, N = dy.shape
    Wv = W if col_off == 0 else W[:, col_off:]
    return (dy, Wv, dx, M, dx.shape[1], N, N, W.shape[1], dx.shape[1]), dict(b_layout=1)


def _dw_spec(dy, x, out, col_off=0):
    M, N = dy.shape
    K = x.shape[1]
    ov = out if col_off == 0 else out[:, col_off:]
    return (dy, x, ov, N, K, M, N, K, out.shape[1]), dict(a_layout=1, b_layout=1, kslices=max(0, min(8, M // 256)))


def _big_linear(x, W, b=None, col_off=0, ld=None):
    """y = x @ W[:, off:off + K]^T + b for a tall x (the large-graph path of MatchingLossFn)."""
    if LARGE_GEMM_ENGINE == "gemm":
        return linear_raw(x, W, b, col_off, ld)
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    Wv = W if col_off == 0 else W[:, col_off:]
    return mm(x, Wv, y, M, N, K, K, W.shape[1] if ld is None else ld, N, bias=b)


def _big_dx(dy, W, K, col_off=0, out=None, accumulate=False):
    """dx (M, K) = dy (M, N) @ W[:, off:off + K] (W row-major (N, ldw)); ``accumulate``: += into ``out``."""
    M, N = dy.shape
    dx = out if out is not None else torch.empty(M, K, device=dy.device, dtype=torch.float32)
    if LARGE_GEMM_ENGINE == "gemm":
        return gemm(dy, N, 1, W, 1, W.shape[1], dx, K, 1, M, K, N, b_off=col_off, beta=1.0 if accumulate else 0.0)
    Wv = W if col_off == 0 else W[:, col_off:]
    return mm(dy, Wv, dx, M, K, N, N, W.shape[1], K, b_layout=1, res=dx if accumulate else None, ldres=K)


def _big_dw(dy, x, out, col_off=0):
    """out[:, off:off + K] (N, ldo) = dy (M, N)^T @ x (M, K): the reduction over the M stacked nodes split over 8 workgroup planes,
    added in a fixed order."""
    M, N = dy.shape
    K = x.shape[1]
    if LARGE_GEMM_ENGINE == "gemm":
        return gemm(dy, 1, N, x, 1, K, out, out.shape[1], 1, N, K, M, c_off=col_off)
    ov = out if col_off == 0 else out[:, col_off:]
    return mm(dy, x, ov, N, K, M, N, K, out.shape[1], a_layout=1, b_layout=1, kslices=max(0, min(8, M // 256)))


def colsum(X):
    out = torch.empty(X.shape[1], device=X.device, dtype=torch.float32)
    call("ttdg_colsum_f32", ptr(X), X.shape[1], ptr(out), X.shape[0], X.shape[1], stream())
    return out


def row_scale_multi(tensors, scales):
    """[t * s.view(-1, 1, ...) for t, s in zip(tensors, scales)] in ONE launch per 64 tensors (csrc/fold.hip): every tensor is
    dense fp32 with its rows (dim 0) scaled by the matching entry of ``s``.  The results are views of one flat buffer."""
    if len(tensors) != len(scales):
        raise ValueError("row_scale_multi: %d tensors, %d scale vectors" % (len(tensors), len(scales)))
    if not tensors:
        return []
    ins, offs, total, cl = [], [], 0, []
    for t, sc in zip(tensors, scales):
        # rows = dim 0 is the outermost index of a contiguous AND of a channels-last filter (O, kh, kw, I): the kernel sees storage order
        cl.append(is_channels_last(t))
        check_f32(t.permute(0, 2, 3, 1) if cl[-1] else t, sc)
        if sc.numel() != t.shape[0] or sc.device != t.device:
            raise ValueError("row_scale_multi: scale of %d entries for a tensor of %d rows" % (sc.numel(), t.shape[0]))
        ins.append(t)
        offs.append(total)
        total += (t.numel() + 63) // 64 * 64               # every result starts on a 256-byte boundary, like an allocation of its own
    flat = torch.empty(total, device=tensors[0].device, dtype=torch.float32)
    outs = [flat[o:o + t.numel()].view(t.shape[0], t.shape[2], t.shape[3], t.shape[1]).permute(0, 3, 1, 2) if c else flat[o:o + t.numel()].view(t.shape)
            for o, t, c in zip(offs, ins, cl)]                # results keep the layout of their inputs
    items = []
    for t, sc, o in zip(ins, scales, outs):
        it = _lib.RowScale()
        it.inp, it.scale, it.out = ptr(t), ptr(sc), ptr(o)
        it.rows, it.rowlen = t.shape[0], t.numel() // max(1, t.shape[0])
        items.append(it)
    with _timed("row_scale_multi", 8 * sum(t.numel() for t in ins)):          # algorithmic bytes: read + write every value
        for i in range(0, len(items), _lib.ROW_SCALE_MAX):
            chunk = items[i:i + _lib.ROW_SCALE_MAX]
            call("ttdg_row_scale_multi", (_lib.RowScale * len(chunk))(*chunk), len(chunk), stream())
    return outs


class FoldFiltersFn(torch.autograd.Function):
    """wf_i = w_i * scale_i (per output channel) for a LIST of convolution filters: one launch forward, one backward
    (grad_w_i = grad_wf_i * scale_i - the same multiplications autograd's MulBackward0 performs filter by filter)."""

    @staticmethod
    def forward(ctx, scales, *weights):
        ctx.scales = scales
        ctx.set_materialize_grads(False)          # an unused folded filter hands back None, not a tensor of zeros
        return tuple(row_scale_multi(list(weights), scales))

    @staticmethod
    def backward(ctx, *grads):
        live = [i for i, g in enumerate(grads) if g is not None]
        out = [None] * len(grads)
        dense = [g if (g.is_contiguous() or is_channels_last(g)) else g.contiguous() for g in (grads[i] for i in live)]
        for i, r in zip(live, row_scale_multi(dense, [ctx.scales[i] for i in live])):
            out[i] = r
        return (None,) + tuple(out)


class LinearFn(torch.autograd.Function):
    """nn.Linear on the MFMA GEMM (forward + both backward products)."""

    @staticmethod
    def forward(ctx, x, W, b):
        check_f32(x, W)
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return linear_raw(x, W, b)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = x.shape
        N = W.shape[0]
        dx = torch.empty_like(x)
        gemm(dy, N, 1, W, 1, K, dx, K, 1, M, K, N)                 # dx = dy W
        dW = torch.empty_like(W)
        gemm(dy, 1, N, x, 1, K, dW, K, 1, N, K, M)                 # dW = dy^T x
        db = colsum(dy) if ctx.has_bias else None
        return dx, dW, db


# ------------------------------------------------------------------------------------------- pieces
def pick_ksplit(M, H=HID, nmax=0):
    """Split K so that the pairwise kernel launches ~4 workgroups per CU on small batches.  Above 128 nodes per graph
    the pair Sinkhorn holds the matrix in registers and reads a single plane: keep 1 (the tile count is large anyway)."""
    if nmax > 128:
        return 1
    nt = (M + 63) // 64
    tiles = nt * (nt + 1) // 2
    ks = 1
    while ks < 16 and tiles * ks < 1024 and H % (ks * 2 * 32) == 0:      # ~4 workgroups per CU
        ks *= 2
    return ks


def affinity_pairwise_fwd(P, Q, w2, gr, ksplit):
    M = P.shape[0]
    part = torch.empty(ksplit, M, M, device=P.device, dtype=torch.float32)
    call("ttdg_affinity_pairwise_fwd", ptr(P), ptr(Q), ptr(w2), P.shape[1], gr, ksplit, ptr(part), stream())
    return part


def affinity_pairwise_bwd(P, Q, w2, dM, gr):
    dP, dQ = torch.empty_like(P), torch.empty_like(Q)
    dw2 = torch.empty(P.shape[1], device=P.device, dtype=torch.float32)
    db2 = torch.empty(1, device=P.device, dtype=torch.float32)
    ws = torch.empty(_lib.load().ttdg_affinity_bwd_workspace_bytes(P.shape[0], P.shape[1]) // 4 + 1, device=P.device, dtype=torch.float32)
    call("ttdg_affinity_pairwise_bwd", ptr(P), ptr(Q), ptr(w2), ptr(dM), P.shape[1], gr, ptr(dP), ptr(dQ), ptr(dw2),
         ptr(db2), ptr(ws), stream())
    return dP, dQ, dw2, db2


def sinkhorn_pairs_fwd(part, b2, gr, sizes, tau, iters, want_pot=True):
    ks, M, _ = part.shape
    Wds = torch.empty(M, M, device=part.device, dtype=torch.float32)
    G = len(sizes)
    pot = None
    if want_pot:
        pot = torch.empty(G * (G + 1) // 2, iters, max(sizes) + 1, device=part.device, dtype=torch.float32)
    call("ttdg_sinkhorn_pairs_fwd", ptr(part), ks, ptr(b2), gr, float(tau), int(iters), ptr(Wds), ptr(pot), stream())
    return Wds, pot


FUSED_PAIR_STAGE = True     # graphs of <= 64 nodes: affinity + pair Sinkhorn in one launch (csrc/pair_stage.hip); False = the two-launch form larger graphs take (parity tests)
PAIR_STAGE_MAX = 64


def pair_stage_fwd(P, Q, w2, b2, gr, sizes, tau, iters, want_pot=True):
    """-> (aff (1, M, M): the affinity without b2, Wds (M, M), pot) in ONE launch; graphs of at most 64 nodes."""
    M = P.shape[0]
    G = len(sizes)
    aff = torch.empty(1, M, M, device=P.device, dtype=torch.float32)
    Wds = torch.empty(M, M, device=P.device, dtype=torch.float32)
    pot = torch.empty(G * (G + 1) // 2, iters, max(sizes) + 1, device=P.device, dtype=torch.float32) if want_pot else None
    call("ttdg_pair_stage_fwd", ptr(P), ptr(Q), ptr(w2), ptr(b2), P.shape[1], gr, float(tau), int(iters), ptr(aff), ptr(Wds), ptr(pot), stream())
    return aff, Wds, pot


def pair_stage_bwd(aff, b2, pot, dWds, gr, tau, iters):
    M = aff.shape[-1]
    dM = torch.empty(M, M, device=aff.device, dtype=torch.float32)
    call("ttdg_pair_stage_bwd", ptr(aff), ptr(b2), ptr(pot), ptr(dWds), gr, float(tau), int(iters), ptr(dM), stream())
    return dM


def sinkhorn_pairs_bwd(part, b2, pot, dWds, gr, tau, iters):
    ks, M, _ = part.shape
    dM = torch.empty(M, M, device=part.device, dtype=torch.float32)
    call("ttdg_sinkhorn_pairs_bwd", ptr(part), ks, ptr(b2), ptr(pot), ptr(dWds), gr, float(tau), int(iters), ptr(dM), stream())
    return dM


def sinkhorn_batched(s, n1=None, n2=None, dummy_row=False, tau=1.0, iters=10, want_pot=False):
    """(b, r, c) -> (b, r, c); any strides on the input.  ``want_pot`` also returns the per-sweep potentials the backward
    needs ((b, iters, max(r,c)+1), iters <= 64)."""
    assert s.dim() == 3 and s.dtype == torch.float32
    b, r, c = s.shape
    out = torch.empty(b, r, c, device=s.device, dtype=torch.float32)
    n1 = None if n1 is None else n1.to(device=s.device, dtype=torch.int32).contiguous()
    n2 = None if n2 is None else n2.to(device=s.device, dtype=torch.int32).contiguous()
    pot = torch.zeros(b, int(iters), max(r, c) + 1, device=s.device, dtype=torch.float32) if want_pot else None
    call("ttdg_sinkhorn_batched_fwd", ptr(s), s.stride(0), s.stride(1), s.stride(2), b, r, c, ptr(n1), ptr(n2),
         int(bool(dummy_row)), float(tau), int(iters), ptr(out), ptr(pot), stream())
    return (out, pot) if want_pot else out


def sinkhorn_batched_bwd(s, pot, dout, n1=None, n2=None, dummy_row=False, tau=1.0, iters=10):
    b, r, c = s.shape
    n1 = None if n1 is None else n1.to(device=s.device, dtype=torch.int32).contiguous()
    n2 = None if n2 is None else n2.to(device=s.device, dtype=torch.int32).contiguous()
    ds = torch.empty(b, r, c, device=s.device, dtype=torch.float32)
    call("ttdg_sinkhorn_batched_bwd", ptr(s), s.stride(0), s.stride(1), s.stride(2), b, r, c, ptr(n1), ptr(n2),
         int(bool(dummy_row)), float(tau), int(iters), ptr(pot), ptr(dout.float().contiguous()), ptr(ds), stream())
    return ds


class SinkhornFn(torch.autograd.Function):
    """Differentiable stand-alone Sinkhorn (utils/sinkhorn.py:58-87): forward logs 20-odd small potential vectors, the
    backward rebuilds every sweep from them (no K stored matrices, no autograd graph through the sweeps)."""

    @staticmethod
    def forward(ctx, s, n1, n2, dummy_row, tau, iters):
        out, pot = sinkhorn_batched(s, n1, n2, dummy_row, tau, iters, want_pot=True)
        ctx.save_for_backward(s, pot)
        ctx.args = (n1, n2, dummy_row, tau, iters)
        return out

    @staticmethod
    def backward(ctx, dout):
        s, pot = ctx.saved_tensors
        n1, n2, dummy_row, tau, iters = ctx.args
        return sinkhorn_batched_bwd(s, pot, dout, n1, n2, dummy_row, tau, iters), None, None, None, None, None


def mha_adjacency(q, k, gr, sizes, scale, drop_p=0.0, seed=0, zero_diag=True):
    apack = torch.empty(sum(n * n for n in sizes), device=q.device, dtype=torch.float32)
    call("ttdg_mha_adjacency", ptr(q), ptr(k), q.shape[1], gr, float(scale), float(drop_p), C.c_uint64(int(seed)),
         int(bool(zero_diag)), ptr(apack), stream())
    return apack


def gagm_cfg(tau0=0.1, gamma=0.5, min_tau=1e-2, tol=1e-3, quad_weight=0.5, max_iter=200, sk_iter=20,
             max_stages=0, start_hungarian=False, profile=False, no_cycle_skip=False, variant=0):
    c = _lib.GagmCfg()
    c.tau0, c.gamma, c.min_tau, c.tol, c.quad_weight = tau0, gamma, min_tau, tol, quad_weight
    c.max_iter, c.sk_iter = int(max_iter), int(sk_iter)
    c.max_stages, c.start_hungarian, c.profile = int(max_stages), int(bool(start_hungarian)), int(profile)
    c.no_cycle_skip = int(bool(no_cycle_skip))
    c.variant = int(variant) | GAGM_VARIANT       # per-call A/B selector (_lib.GAGM_*); GAGM_VARIANT = what bench.py's A/B flags set
    return c


GAGM_VARIANT = 0


def gagm_one_step(apack, W, Ucur, gr, sizes, tau=None, quad_weight=0.5, sk_iter=20, variant=0):
    """One application of the solver's map U -> project(V(U)) (parity tests): Sinkhorn projector at ``tau``,
    Hungarian projector when ``tau`` is None.  Returns (U_next, V)."""
    cfg = gagm_cfg(tau0=(tau or 1.0), quad_weight=quad_weight, max_iter=1, sk_iter=sk_iter, max_stages=1,
                   start_hungarian=tau is None, variant=variant)
    U, info, V0 = gagm_solve(apack, W, Ucur, gr, sizes, cfg)
    return U, V0


GAGM_MAX_NODES = 128      # per-graph limit of the single-workgroup kernel; larger graphs take csrc/gagm_large.hip


def gagm_solve_hostloop(apack, W, U0, sizes, cfg, states=None):
    """Host-driven statement of the solver (reference multi_graph_matching.py:300-389) on the stand-alone device
    operators: one iteration = a handful of launches (MFMA GEMMs for B = A U, S = U^T B, V = (2q B S + W U)/G; batched
    Sinkhorn / LAP projectors) and one host read of the two convergence norms, like the reference.  NOT on the product
    path: it is the independent cross-check of the native multi-workgroup solver in the GPU tests.
    Returns (U, info, V0)."""
    dev = W.device
    G, M = len(sizes), sum(sizes)
    off = [0]
    for n in sizes:
        off.append(off[-1] + n)
    aoff = [0]
    for n in sizes:
        aoff.append(aoff[-1] + n * n)
    U = U0.detach().clone().contiguous()
    lastU = torch.zeros_like(U)
    tau, hung, stage, total = float(cfg.tau0), bool(cfg.start_hungarian), 0, 0
    iters, V0 = [], None
    qw2, invG = 2.0 * float(cfg.quad_weight), 1.0 / G
    B = torch.empty(M, UNIV, device=dev)
    S = torch.empty(UNIV, UNIV, device=dev)
    V = torch.empty(M, UNIV, device=dev)
    equal = all(n == sizes[0] for n in sizes)
    while True:
        i = 0
        for i in range(int(cfg.max_iter)):
            lastU2, lastU = lastU, U
            for g, n in enumerate(sizes):                                    # B_g = A_g U_g
                gemm(apack, n, 1, U, 1, UNIV, B, UNIV, 1, n, UNIV, n, a_off=aoff[g], b_off=off[g] * UNIV, c_off=off[g] * UNIV)
            gemm(U, 1, UNIV, B, 1, UNIV, S, UNIV, 1, UNIV, UNIV, M)          # S = U^T B
            gemm(B, UNIV, 1, S, 1, UNIV, V, UNIV, 1, M, UNIV, UNIV, alpha=qw2 * invG)          # V = 2q/G * B S
            gemm(W, M, 1, U, 1, UNIV, V, UNIV, 1, M, UNIV, M, alpha=invG, beta=1.0)            #   + (W U)/G
            if V0 is None:
                V0 = V.clone()
            if hung:
                U = torch.cat([lap_batched(V[off[g]:off[g + 1]].unsqueeze(0))[0] for g in range(G)], 0)
            elif equal:
                n = sizes[0]
                v3 = V.view(G, n, UNIV)
                U = (sinkhorn_batched(v3, None, None, True, tau, cfg.sk_iter) if n <= UNIV else
                     sinkhorn_batched(v3.transpose(1, 2), None, None, True, tau, cfg.sk_iter).transpose(1, 2)).reshape(M, UNIV)
            else:
                nmax = max(sizes)
                pad = torch.zeros(G, nmax, UNIV, device=dev)
                for g, n in enumerate(sizes):
                    pad[g, :n] = V[off[g]:off[g + 1]]
                Up = sinkhorn_batched(pad, torch.tensor(sizes), None, True, tau, cfg.sk_iter)
                U = torch.cat([Up[g, :n] for g, n in enumerate(sizes)], 0)
            U = U.contiguous()
            if G == 2:
                U[:sizes[0]] = torch.eye(sizes[0], UNIV, device=dev)
            total += 1
            if states is not None:
                states.append((hung, tau, lastU, U, V.clone()))
            d = torch.stack((torch.norm(U - lastU), torch.norm(U - lastU2))).tolist()        # the reference's two syncs
            if d[0] < float(cfg.tol) or d[1] == 0:
                break
        iters.append(i + 1)
        stage += 1
        if hung or (cfg.max_stages > 0 and stage >= cfg.max_stages):
            break
        if tau > float(cfg.min_tau):
            tau *= float(cfg.gamma)
        else:
            hung = True
    info = torch.zeros(_lib.GAGM_INFO_WORDS, dtype=torch.int32)
    info[:len(iters)] = torch.tensor(iters, dtype=torch.int32)
    info[6], info[7] = total, stage
    return U, info.to(dev), V0


def gagm_solve(apack, W, U0, gr, sizes, cfg=None, check_status=True):
    """Returns (U (M,32) 0/1, info int32[TTDG_GAGM_INFO_WORDS] on device (include/ttdg_mgm.h), V0 (M,32) first-iteration V).  With the
    cooperative one-launch form (TTDG_GAGM_ONE_LAUNCH) the status word info[8] is read back (a host synchronisation; pass
    ``check_status=False`` to leave it to the caller, who reads ``info`` anyway) and a failed grid barrier raises."""
    M = sum(sizes)
    cfg = cfg or gagm_cfg()
    nbytes = _lib.load().ttdg_gagm_workspace_bytes(M)
    ws = torch.empty(nbytes // 4, device=W.device, dtype=torch.float32)
    U = torch.empty(M, UNIV, device=W.device, dtype=torch.float32)
    info = torch.zeros(_lib.GAGM_INFO_WORDS, device=W.device, dtype=torch.int32)
    timers = KERNEL_TIMERS if (KERNEL_TIMER_ONLY is None or "gagm" in KERNEL_TIMER_ONLY) else None
    if timers is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("ttdg_gagm_solve", ptr(apack), ptr(W), ptr(U0), gr, cfg, ptr(U), ptr(info), ptr(ws), stream())
    if timers is not None:
        e1.record()
        timers.append(("gagm", e0, e1, list(sizes), info))
    if check_status and int(getattr(cfg, "variant", 0)) & _lib.GAGM_ONE_LAUNCH:          # a barrier failure would otherwise only show as NaN in U
        status = int(info[8])
        if status:
            raise RuntimeError("ttdg_gagm_solve (one cooperative launch): status %d (%s)" % (status, "a grid barrier timed out" if status == 1 else "the stage machine did not stop"))
    gagm_solve.last_U1 = ws[M * UNIV:2 * M * UNIV].view(M, UNIV)   # first projected U (debug / parity tests)
    return U, info, ws[:M * UNIV].view(M, UNIV)


def lap_batched(s):
    assert s.dim() == 3
    s = s.contiguous().float()
    x = torch.empty_like(s)
    call("ttdg_lap_batched", ptr(s), s.shape[0], s.shape[1], s.shape[2], ptr(x), stream())
    return x


def perm_loss_fwd_bwd(Wds, U, gr, G, alpha=0.25, eps=1e-6):
    M = Wds.shape[0]
    loss = torch.empty((), device=Wds.device, dtype=torch.float32)
    dWds = torch.empty(M, M, device=Wds.device, dtype=torch.float32)
    flag = torch.empty(1, device=Wds.device, dtype=torch.int32)
    pws = torch.empty(max(1, _lib.load().ttdg_perm_loss_workspace_bytes(gr) // 4), device=Wds.device, dtype=torch.float32)
    call("ttdg_perm_loss_fwd_bwd", ptr(Wds), ptr(U), gr, float(alpha), float(eps), ptr(loss), ptr(dWds), ptr(flag),
         ptr(pws), stream())
    return loss, dWds, flag


# ------------------------------------------------------------------------------------------- fused matching loss
class MatchingLossFn(torch.autograd.Function):
    """MGM3_unsup.forward (reference multi_graph_matching.py:487-569) as one tape node.

    forward : P/Q projections (MFMA GEMM) -> pairwise affinity -> pair Sinkhorn (Wds) ;
              q/k projections -> attention adjacency (A) ; U0 = X U^T ; GA-MGM solve (no grad) ;
              focal permutation loss, whose kernel also emits d loss / d Wds.
    backward: Sinkhorn bwd -> affinity bwd -> GEMM bwd, into the stacked node features and the six
              node_affinity tensors.  intra_domain_graph.* and U receive no gradient, as in the
              reference (A and U0 only feed the gradient-free solver).
    """

    @staticmethod
    def forward(ctx, X, W1, b1, w2, b2, Psr, Ptg, Wq, bq, Wk, bk, U, sizes, opts):
        check_f32(X, W1, b1, w2, b2, Psr, Ptg, Wq, bq, Wk, bk, U)
        sizes = [int(s) for s in sizes]
        gr = graphs(sizes)
        G, M = len(sizes), sum(sizes)
        grouped = GROUPED_GEMM and M <= GROUPED_GEMM_MAX_ROWS
        if grouped:
            # seven projections in two launches: {Xs, Xt, q, k, U0} <- X, then {P, Q} <- {Xs, Xt}
            dev = X.device
            Xs, Xt, q, k = (torch.empty(M, DIM, device=dev, dtype=torch.float32) for _ in range(4))
            U0 = torch.empty(M, U.shape[0], device=dev, dtype=torch.float32)
            P, Q = (torch.empty(M, HID, device=dev, dtype=torch.float32) for _ in range(2))
            gemm_grouped([gdesc(X, DIM, 1, Psr, DIM, 1, Xs, DIM, 1, M, DIM, DIM), gdesc(X, DIM, 1, Ptg, DIM, 1, Xt, DIM, 1, M, DIM, DIM),
                          gdesc(X, DIM, 1, Wq, DIM, 1, q, DIM, 1, M, DIM, DIM, bias=bq), gdesc(X, DIM, 1, Wk, DIM, 1, k, DIM, 1, M, DIM, DIM, bias=bk),
                          gdesc(X, DIM, 1, U, DIM, 1, U0, U.shape[0], 1, M, U.shape[0], DIM)])
            gemm_grouped([gdesc(Xs, DIM, 1, W1, HID, 1, P, HID, 1, M, HID, DIM),
                          gdesc(Xt, DIM, 1, W1, HID, 1, Q, HID, 1, M, HID, DIM, bias=b1, b_off=DIM)])
        elif LARGE_GEMM_ENGINE == "mm_grouped":
            dev = X.device
            Xs, Xt, q, k = (torch.empty(M, DIM, device=dev, dtype=torch.float32) for _ in range(4))
            U0 = torch.empty(M, U.shape[0], device=dev, dtype=torch.float32)
            P, Q = (torch.empty(M, HID, device=dev, dtype=torch.float32) for _ in range(2))
            mm_grouped([_lin_spec(X, Psr, Xs), _lin_spec(X, Ptg, Xt), _lin_spec(X, Wq, q, bq), _lin_spec(X, Wk, k, bk), _lin_spec(X, U, U0)])
            mm_grouped([_lin_spec(Xs, W1, P, None, 0, HID), _lin_spec(Xt, W1, Q, b1, DIM, HID)])
        else:
            Xs = _big_linear(X, Psr)
            Xt = _big_linear(X, Ptg)
            P = _big_linear(Xs, W1, None, 0, HID)
            Q = _big_linear(Xt, W1, b1, DIM, HID)
        w2f = w2.reshape(-1)
        tau, iters = opts.get("pair_tau", 0.05), opts.get("pair_iters", 20)
        fused = FUSED_PAIR_STAGE and max(sizes) <= PAIR_STAGE_MAX and not opts.get("ksplit")
        if fused:
            with _timed("pair_stage_fwd", sizes):
                part, Wds, pot = pair_stage_fwd(P, Q, w2f, b2, gr, sizes, tau, iters)
        else:
            ks = opts.get("ksplit") or pick_ksplit(M, HID, max(sizes))
            with _timed("affinity_fwd", sizes):
                part = affinity_pairwise_fwd(P, Q, w2f, gr, ks)
            with _timed("sinkhorn_pairs_fwd", sizes):
                Wds, pot = sinkhorn_pairs_fwd(part, b2, gr, sizes, tau, iters)
        if not grouped and LARGE_GEMM_ENGINE != "mm_grouped":
            q = _big_linear(X, Wq, bq)
            k = _big_linear(X, Wk, bk)
            U0 = _big_linear(X, U)
        apack = mha_adjacency(q, k, gr, sizes, DIM ** -0.5, opts.get("drop_p", 0.0), opts.get("seed", 0))
        if opts.get("forced_U") is not None:      # parity tests: pseudo-labels supplied by the caller
            Ub, info, V0 = opts["forced_U"].contiguous(), None, None
        else:
            Ub, info, V0 = gagm_solve(apack, Wds, U0, gr, sizes, opts.get("gagm_cfg"))
        loss, dWds, flag = perm_loss_fwd_bwd(Wds, Ub, gr, G)
        ctx.save_for_backward(X, Xs, Xt, P, Q, W1, w2f, b2, Psr, Ptg, part, pot, dWds)
        ctx.meta = (sizes, tau, iters, fused)
        trace = opts.get("trace")
        if trace is not None:
            trace.update(Wds=Wds, apack=apack, U0=U0, Ub=Ub, info=info, V0=V0, P=P, Q=Q, part=part)
        ctx.mark_non_differentiable(flag)
        return loss, flag

    @staticmethod
    def backward(ctx, gloss, _gflag):
        X, Xs, Xt, P, Q, W1, w2f, b2, Psr, Ptg, part, pot, dWds = ctx.saved_tensors
        sizes, tau, iters, fused = ctx.meta
        gr = graphs(sizes)
        M = X.shape[0]
        # everything downstream is linear in dWds: apply the incoming loss scale (normally 1.0) once, here
        if fused:
            with _timed("pair_stage_bwd", sizes):
                dM = pair_stage_bwd(part, b2, pot, dWds * gloss, gr, tau, iters)
        else:
            with _timed("sinkhorn_pairs_bwd", sizes):
                dM = sinkhorn_pairs_bwd(part, b2, pot, dWds * gloss, gr, tau, iters)
        with _timed("affinity_bwd", sizes):
            dP, dQ, dw2, db2 = affinity_pairwise_bwd(P, Q, w2f, dM, gr)
        if GROUPED_GEMM and M <= GROUPED_GEMM_MAX_ROWS:
            # eight gradient products in two launches
            dW1, dXs, dXt = torch.empty_like(W1), torch.empty_like(Xs), torch.empty_like(Xt)
            gemm_grouped([gdesc(dP, 1, HID, Xs, 1, DIM, dW1, HID, 1, HID, DIM, M),                     # dW1[:, :256] = dP^T Xs
                          gdesc(dQ, 1, HID, Xt, 1, DIM, dW1, HID, 1, HID, DIM, M, c_off=DIM),          # dW1[:, 256:] = dQ^T Xt
                          gdesc(dP, HID, 1, W1, 1, HID, dXs, DIM, 1, M, DIM, HID),                     # dXs = dP W1[:, :256]
                          gdesc(dQ, HID, 1, W1, 1, HID, dXt, DIM, 1, M, DIM, HID, b_off=DIM)])         # dXt = dQ W1[:, 256:]
            db1 = colsum(dQ)
            dPsr, dPtg, dX = torch.empty_like(Psr), torch.empty_like(Ptg), torch.empty_like(X)
            gemm_grouped([gdesc(dXs, 1, DIM, X, 1, DIM, dPsr, DIM, 1, DIM, DIM, M),                    # dPsr = dXs^T X
                          gdesc(dXt, 1, DIM, X, 1, DIM, dPtg, DIM, 1, DIM, DIM, M),
                          gdesc(dXs, DIM, 1, Psr, 1, DIM, dX, DIM, 1, M, DIM, DIM,                     # dX = dXs Psr + dXt Ptg (two K segments)
                                second=(dXt, DIM, 1, Ptg, 1, DIM, DIM))])
            return (dX, dW1, db1, dw2.view(1, HID), db2, dPsr, dPtg, None, None, None, None, None, None, None)
        if LARGE_GEMM_ENGINE == "mm_grouped":
            dW1, dPsr, dPtg = torch.empty_like(W1), torch.empty_like(Psr), torch.empty_like(Ptg)
            dXs, dXt, dX = torch.empty_like(Xs), torch.empty_like(Xt), torch.empty_like(X)
            mm_grouped([_dw_spec(dP, Xs, dW1), _dw_spec(dQ, Xt, dW1, DIM)])           # dW1 = [dP^T Xs | dQ^T Xt]
            db1 = colsum(dQ)
            mm_grouped([_dx_spec(dP, W1, dXs), _dx_spec(dQ, W1, dXt, DIM)])           # dXs = dP W1[:, :256], dXt = dQ W1[:, 256:]
            mm_grouped([_dw_spec(dXs, X, dPsr), _dw_spec(dXt, X, dPtg)])
            _big_dx(dXs, Psr, DIM, out=dX)                                            # dX = dXs Psr + dXt Ptg (the second reads the first)
            _big_dx(dXt, Ptg, DIM, out=dX, accumulate=True)
            return (dX, dW1, db1, dw2.view(1, HID), db2, dPsr, dPtg, None, None, None, None, None, None, None)
        dW1 = torch.empty_like(W1)
        _big_dw(dP, Xs, dW1)                                                      # dW1[:, :256] = dP^T Xs
        _big_dw(dQ, Xt, dW1, DIM)                                                 # dW1[:, 256:] = dQ^T Xt
        db1 = colsum(dQ)
        dXs = _big_dx(dP, W1, DIM)                                                # dXs = dP W1[:, :256]
        dXt = _big_dx(dQ, W1, DIM, DIM)                                           # dXt = dQ W1[:, 256:]
        dPsr = torch.empty_like(Psr)
        dPtg = torch.empty_like(Ptg)
        _big_dw(dXs, X, dPsr)                                                     # dPsr = dXs^T X
        _big_dw(dXt, X, dPtg)
        dX = _big_dx(dXs, Psr, DIM)                                               # dX = dXs Psr + dXt Ptg
        _big_dx(dXt, Ptg, DIM, out=dX, accumulate=True)
        return (dX, dW1, db1, dw2.view(1, HID), db2, dPsr, dPtg, None, None, None, None, None, None, None)


# ------------------------------------------------------------------------------------------- node gather
class NodeGatherFn(torch.autograd.Function):
    """Gather selected FPN points (build_graph.py:181-195): rows[i] = feat[level][img[i], :, point]."""

    @staticmethod
    def forward(ctx, img, pid, *feats):
        n = img.numel()
        Cc = feats[0].shape[1]
        fp = _lib.Fpn()
        fp.n, fp.C = len(feats), Cc
        nhwc = all(f.dim() == 4 and f.is_contiguous(memory_format=torch.channels_last) for f in feats) and any(is_channels_last(f) for f in feats)
        for l, f in enumerate(feats):
            if f.dtype != torch.float32 or not (f.is_contiguous(memory_format=torch.channels_last) if nhwc else f.is_contiguous()):
                raise TypeError("FPN maps must be dense float32, all NCHW or all channels-last")
            fp.h[l], fp.w[l], fp.feat[l] = f.shape[2], f.shape[3], ptr(f)
        out = torch.empty(n, Cc, device=feats[0].device, dtype=torch.float32)
        call("ttdg_node_gather_fwd_nhwc" if nhwc else "ttdg_node_gather_fwd", fp, ptr(img), ptr(pid), n, ptr(out), stream())
        ctx.save_for_backward(img, pid)
        ctx.shapes = [tuple(f.shape) for f in feats]
        ctx.nhwc = nhwc
        return out

    @staticmethod
    def backward(ctx, dout):
        img, pid = ctx.saved_tensors
        dout = dout.contiguous()
        fmt = torch.channels_last if ctx.nhwc else torch.contiguous_format
        grads = [torch.empty(s, device=dout.device, dtype=torch.float32, memory_format=fmt).zero_() for s in ctx.shapes]
        fp = _lib.Fpn()
        fp.n, fp.C = len(grads), ctx.shapes[0][1]
        for l, g in enumerate(grads):
            fp.h[l], fp.w[l], fp.feat[l] = g.shape[2], g.shape[3], ptr(g)
        call("ttdg_node_gather_bwd_nhwc" if ctx.nhwc else "ttdg_node_gather_bwd", fp, ptr(img), ptr(pid), img.numel(), ptr(dout), stream())
        return (None, None, *grads)


def levels_desc(shapes, strides=(4, 8, 16, 32, 64),
                ranges=((-1, 64), (64, 128), (128, 256), (256, 512), (512, 100000000))):
    lv = _lib.Levels()
    lv.n = len(shapes)
    for l, (h, w) in enumerate(shapes):
        lv.h[l], lv.w[l], lv.stride[l] = int(h), int(w), int(strides[l])
        lv.lo[l], lv.hi[l] = float(ranges[l][0]), float(ranges[l][1])
    return lv


def node_labels(boxes, classes, nbox, lv, npts):
    B, kmax = boxes.shape[0], boxes.shape[1]
    labels = torch.empty(B, npts, device=boxes.device, dtype=torch.int32)
    call("ttdg_node_labels", ptr(boxes), ptr(classes), ptr(nbox), B, kmax, lv, ptr(labels), stream())
    return labels


def node_select(labels, lv, sample_dist, cap):
    B = labels.shape[0]
    sel_idx = torch.empty(B, cap, device=labels.device, dtype=torch.int32)
    sel_lab = torch.empty(B, cap, device=labels.device, dtype=torch.int32)
    count = torch.empty(B, device=labels.device, dtype=torch.int32)
    call("ttdg_node_select", ptr(labels), B, lv, int(sample_dist), int(cap), ptr(sel_idx), ptr(sel_lab), ptr(count), stream())
    return sel_idx, sel_lab, count


# ------------------------------------------------------------------------------------------- detection helpers
def roi_align(feat, rois, scale, P):
    """feat (B,C,H,W) fp32 contiguous, rois (R,5) -> (R,C,P,P); forward only."""
    feat = feat.detach()
    if feat.dtype != torch.float32 or not feat.is_contiguous():
        feat = feat.float().contiguous()
    rois = rois.detach().float().contiguous()
    R = rois.shape[0]
    B, Cc, H, W = feat.shape
    out = torch.empty(R, Cc, P, P, device=feat.device, dtype=torch.float32)
    call("ttdg_roi_align_fwd", ptr(feat), B, Cc, H, W, ptr(rois), R, float(scale), int(P), ptr(out), stream())
    return out


ROI_ALIGN_NHWC = True     # the ROI heads pool from channels-last copies of the FPN maps (one transposition per forward)


def to_nhwc(feats):
    """Channels-last copies (B, H, W, C) of NCHW fp32 maps for roi_align_multilevel(..., nhwc=...): one LDS-tiled
    transposition per level, shared by every pooler call of a forward."""
    out = []
    for f in feats:
        f = f.detach()
        if f.dtype == torch.float32 and f.dim() == 4 and f.is_contiguous(memory_format=torch.channels_last) and (is_channels_last(f) or f.shape[1] == 1):
            out.append(f.permute(0, 2, 3, 1))              # a channels-last map IS the (B, H, W, C) tensor: a view, no transposition
            continue
        if f.dtype != torch.float32 or not f.is_contiguous():
            f = f.float().contiguous()
        B, Cc, H, W = f.shape
        t = torch.empty(B, H, W, Cc, device=f.device, dtype=torch.float32)
        call("ttdg_nchw_to_nhwc", ptr(f), ptr(t), B, Cc, H, W, stream())
        out.append(t)
    return out


def roi_align_multilevel(feats, rois, strides, P, canonical_size=224, canonical_level=4, min_level=2, nhwc=None):
    """ROIPooler [3P] in one launch: feats = the FPN maps of levels min_level.. (fp32 NCHW), rois (R, 5); every ROI picks
    its level inside the kernel.  Returns (R, C, P, P).  No host read.  ``nhwc`` (from ``to_nhwc(feats)``) selects the
    channels-last kernel (lane = channel, coalesced taps)."""
    rois = rois.detach().float().contiguous()
    R = rois.shape[0]
    if nhwc is not None and P <= 14:
        Cc = nhwc[0].shape[3]
        fp = _lib.Fpn()
        fp.n, fp.C = len(nhwc), Cc
        for l, t in enumerate(nhwc):
            fp.h[l], fp.w[l], fp.feat[l] = t.shape[1], t.shape[2], ptr(t)
        lv = levels_desc([t.shape[1:3] for t in nhwc], strides=tuple(strides) + (0,) * (8 - len(strides)), ranges=((0, 0),) * len(nhwc))
        out = torch.empty(R, Cc, P, P, device=rois.device, dtype=torch.float32)
        # algorithmic bytes: every map once (the ROIs of a trained RPN cover the objects many times over: what a perfect cache
        # would fetch) + the pooled output
        with _timed("roi_align_nhwc", sum(t.numel() for t in nhwc) * 4 + out.numel() * 4):
            call("ttdg_roi_align_multilevel_nhwc", fp, lv, ptr(rois), R, int(P), float(canonical_size), int(canonical_level), int(min_level),
                 ptr(out), stream())
        return out
    Cc = feats[0].shape[1]
    fp = _lib.Fpn()
    fp.n, fp.C = len(feats), Cc
    keep = []
    for l, f in enumerate(feats):
        f = f.detach()
        if f.dtype != torch.float32 or not f.is_contiguous():
            f = f.float().contiguous()
        keep.append(f)
        fp.h[l], fp.w[l], fp.feat[l] = f.shape[2], f.shape[3], ptr(f)
    lv = levels_desc([f.shape[2:] for f in keep], strides=tuple(strides) + (0,) * (8 - len(strides)), ranges=((0, 0),) * len(keep))
    out = torch.empty(R, Cc, P, P, device=rois.device, dtype=torch.float32)
    call("ttdg_roi_align_multilevel", fp, lv, ptr(rois), R, int(P), float(canonical_size), int(canonical_level), int(min_level),
         ptr(out), stream())
    return out


def nms_launch(boxes, scores, thr, group=None, ngroups=None, max_group=None, topk=None):
    """Enqueue greedy NMS without synchronising.  With ``group`` (ids in [0, ngroups)) the groups are independent
    problems swept concurrently (one wavefront each); ``max_group`` bounds the largest group (default N).
    Returns a handle for nms_collect: kept indices come back sorted by descending score, at most ``topk`` of them."""
    N = boxes.shape[0]
    dev = boxes.device
    sc = scores.detach().float()
    if N == 0:
        return ("empty", torch.empty(0, dtype=torch.int64, device=dev))
    order = torch.argsort(sc, descending=True)
    if group is None:
        ngroups, seg = 1, torch.tensor([0, N], dtype=torch.int32, device=dev)
        final = order
    else:
        gs = group[order].to(torch.int64)
        o2 = torch.argsort(gs, stable=True)                      # (group, descending score)
        final = order[o2]
        seg = torch.searchsorted(gs[o2].contiguous(), torch.arange(ngroups + 1, device=dev, dtype=torch.int64)).to(torch.int32)
    mg = N if max_group is None else min(int(max_group), N)
    b = boxes.detach().float()[final].contiguous()
    words = (mg + 63) // 64
    ws = torch.empty(N * words, dtype=torch.int64, device=dev)
    flags = torch.zeros(N, dtype=torch.uint8, device=dev)
    call("ttdg_nms_grouped", ptr(b), ptr(seg), int(ngroups), N, mg, float(thr), ptr(ws), ptr(flags), stream())
    keepf = flags.bool()
    masked = torch.where(keepf, sc[final], sc.new_full((), float("-inf")))
    k = N if topk is None else min(int(topk), N)
    top = torch.topk(masked, k).indices                          # kept boxes first, by descending score
    return ("grouped", final[top], keepf.sum().clamp(max=k).reshape(1).to(torch.int32))


def nms_collect(launched):
    """Resolve a list of nms_launch() handles with ONE host synchronisation; returns the kept-index tensors."""
    if not launched:
        return []
    live = [h for h in launched if h[0] != "empty"]
    counts = iter(torch.cat([h[2] for h in live]).tolist() if live else [])
    return [h[1] if h[0] == "empty" else h[1][:next(counts)] for h in launched]


def bias_act_(y, bias=None, residual=None, bias2=None, relu=True):
    """In place on a contiguous fp32 NCHW tensor: y <- act((y + bias[c]) + (residual + bias2[c])).  No autograd: where
    gradients flow use BiasActFn."""
    if is_channels_last(y):          # (N, H, W, C) storage: the channel is the fastest index
        if residual is not None and (residual.shape != y.shape or residual.stride() != y.stride()):
            raise ValueError("bias_act_: residual must have the shape and the channels-last layout of y")
        for t in (y, bias, residual, bias2):
            if t is not None and t.dtype != torch.float32:
                raise TypeError("bias_act_: float32 tensors only (got %s)" % t.dtype)
        for t in (bias, bias2):
            if t is not None and not t.is_contiguous():
                raise TypeError("bias_act_: contiguous bias vectors only")
        Cc = y.shape[1]
        with _timed("bias_act", y.numel() * 4 * (3 if residual is not None else 2)):
            call("ttdg_bias_act_nhwc", ptr(y), ptr(bias), ptr(residual), ptr(bias2), y.numel() // Cc, Cc, int(bool(relu)), stream())
        return y
    for t in (y, bias, residual, bias2):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise TypeError("bias_act_: contiguous float32 tensors only (got %s)" % t.dtype)
    if residual is not None and residual.shape != y.shape:
        raise ValueError("bias_act_: residual shape %s != %s" % (tuple(residual.shape), tuple(y.shape)))
    N, C = y.shape[0], y.shape[1]
    HW = y.numel() // max(1, N * C)
    with _timed("bias_act", y.numel() * 4 * (3 if residual is not None else 2)):       # algorithmic bytes: read y (+ residual), write y
        call("ttdg_bias_act", ptr(y), ptr(bias), ptr(residual), ptr(bias2), N, C, HW, int(bool(relu)), stream())
    return y


class BiasActFn(torch.autograd.Function):
    """relu((y + bias[c]) + (residual + bias2[c])) IN PLACE on the convolution output y, with gradients: the fused epilogue of
    the adapted res3 - res5 bottlenecks (FrozenBN shift, residual add, ReLU: one pass forward, one pass backward, instead of
    torch's bias add_, add, clamp_min_ and threshold_backward kernels).  bias / bias2 are FrozenBN shifts (buffers, no gradient)."""

    @staticmethod
    def forward(ctx, y, bias, residual, bias2):
        bias_act_(y, bias, residual, bias2, relu=True)
        ctx.mark_dirty(y)
        ctx.save_for_backward(y)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, gout):
        (out,) = ctx.saved_tensors
        gout = like_layout(gout, out)                   # element-wise over storage: both operands in the layout of the activation
        if gout.dtype != torch.float32:
            raise TypeError("BiasActFn: float32 gradients only")
        gin = torch.empty_like(out)
        with _timed("relu_bwd", out.numel() * 12):                                          # read gout, read out, write gin
            call("ttdg_relu_bwd", ptr(gout), ptr(out), ptr(gin), C.c_size_t(out.numel()), stream())
        return gin, None, (gin if ctx.has_res else None), None


def pointwise_ok(x, w):
    """Does the streaming product (csrc/pointwise.hip) take this 1 x 1 convolution?  fp32, channels-last activation with at least two
    channels-last dimensions' worth of pixels, a dense (Cout, Cin) filter, channel counts that are multiples of 4."""
    return (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 4 and w.dim() == 4
            and w.shape[2] == 1 and w.shape[3] == 1 and x.shape[1] % 4 == 0 and w.shape[0] % 4 == 0 and x.shape[1] >= 8
            and x.is_contiguous(memory_format=torch.channels_last) and w.stride(0) == w.shape[1] and w.stride(1) == 1)


def _cl_empty(B, Cc, H, W, device):
    return torch.empty(B, H, W, Cc, device=device, dtype=torch.float32).permute(0, 3, 1, 2)


def pointwise_conv(x, w, bias=None, residual=None, bias2=None, relu=False, stride=1, res_up=False, pbias=None, prelu=False, second=None):
    """1 x 1 convolution of a channels-last activation as ONE fused product: act(conv(x', w) + bias + (residual + bias2)) with
    x' = x or relu(x + pbias[c]) (the previous layer's shift + ReLU applied on the fly).  ``stride``: every stride-th pixel of every
    stride-th row (no padding); ``res_up``: the residual is the half-size map, nearest-neighbour up-sampled (FPN top-down sum).
    ``second`` = (x2, w2, stride2): + conv(x2, w2) with its own stride accumulated in the same product (the projection shortcut of a
    bottleneck's first block next to conv3; bias2 = its shift; no residual then).  No autograd: where gradients flow use PointwiseConvFn."""
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    if not pointwise_ok(x, w):
        raise TypeError("pointwise_conv: fp32 channels-last activation and a dense (Cout, Cin, 1, 1) filter expected")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = _cl_empty(B, Cout, Ho, Wo, x.device)
    if residual is not None:
        want = (B, Cout, Ho // 2, Wo // 2) if res_up else (B, Cout, Ho, Wo)
        if tuple(residual.shape) != want or residual.dtype != torch.float32 or not residual.is_contiguous(memory_format=torch.channels_last):
            raise ValueError("pointwise_conv: residual of shape %s (channels-last fp32) expected, got %s" % (want, tuple(residual.shape)))
        if res_up and (Ho % 2 or Wo % 2):
            raise ValueError("pointwise_conv: the up-sampled residual needs an even output map")
    for t in (bias, bias2, pbias):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise TypeError("pointwise_conv: contiguous float32 shift vectors only")
    M = B * Ho * Wo
    nbytes = 4 * (M * Cin + M * Cout * (2 if residual is not None and not res_up else 1) + Cin * Cout)
    flops = 2 * M * Cin * Cout
    seg = None
    if second is not None:
        x2, w2, st2 = second
        if residual is not None or not pointwise_ok(x2, w2) or w2.shape[0] != Cout or stride != 1 or Cin % 32 or x2.shape[1] % 32:
            raise ValueError("pointwise_conv: the second segment needs no residual, channel counts that are multiples of 32 and a stride-1 first segment")
        B2, C2, H2, W2 = x2.shape
        if B2 != B or (H2 - 1) // st2 + 1 != Ho or (W2 - 1) // st2 + 1 != Wo:
            raise ValueError("pointwise_conv: the second segment's strided map %s does not match the output %s" % (tuple(x2.shape), (B, Cout, Ho, Wo)))
        seg = (x2, w2, C2, C2, C2, st2 if st2 > 1 else 0, (H2, W2))
        nbytes += 4 * (M * C2 + C2 * Cout)
        flops += 2 * M * C2 * Cout
    with _timed("pointwise_fwd", (nbytes, flops)):
        mm(x, w, out, M, Cout, Cin, Cin, Cin, Cout, bias=bias, res=residual, ldres=Cout, bias2=bias2, pbias=pbias, relu=relu, prelu=prelu,
           a_stride=stride if stride > 1 else 0, a_hw=(H, W), res_up=res_up, res_hw=(Ho, Wo), second=seg)
    return out


def _dw_slices(Cout, Cin, M):
    """pixel slices of the weight-gradient product: ~512 workgroups of 64 x 64 tiles, at least 256 pixels per slice"""
    tiles = ((Cout + 63) // 64) * ((Cin + 63) // 64)
    ks = min(M // 256, (512 + tiles - 1) // tiles)
    return ks if ks >= 2 else 0


def rpn_heads_product(t, w, b, pbias, logits, deltas):
    """Both 1 x 1 heads of one RPN level in one launch: t (B, C, H, W) channels-last = the 3 x 3 filter's output WITHOUT its bias,
    w (Ap + 4 Ap, C) / b the packed head filters (modeling/detector.py: RPNHead._packed_heads), pbias the 3 x 3 filter's bias (applied with the
    ReLU where the product fetches its operand); logits (B, H, W, Ap) and deltas (B, H, W, 4 Ap) receive the two column groups."""
    B, Cc, H, W = t.shape
    M, Ap = B * H * W, logits.shape[-1]
    N = w.shape[0]
    with _timed("rpn_heads", 4 * (M * Cc + M * N + N * Cc)):          # 20 output columns: a streaming read of the map, HBM-bound (its own stamp)
        mm(t, w, logits, M, N, Cc, Cc, Cc, Ap, bias=b, pbias=pbias, prelu=True, split=(deltas, N - Ap, Ap))


class StridedSliceFn(torch.autograd.Function):
    """x[:, :, ::s, ::s] as a dense channels-last tensor, with the scatter into zeros as its backward: the input of a stride-s
    pointwise convolution, compacted ONCE for the bottleneck's conv1 and its projection shortcut (their two input gradients are
    added on the compact map, then scattered once)."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s, ctx.shape = s, tuple(x.shape)
        return x[:, :, ::s, ::s].contiguous(memory_format=torch.channels_last)

    @staticmethod
    def backward(ctx, g):
        B, Cc, H, W = ctx.shape
        dx = torch.zeros(B, H, W, Cc, device=g.device, dtype=g.dtype).permute(0, 3, 1, 2)
        dx[:, :, ::ctx.s, ::ctx.s] = g
        return dx, None


# which kernels compute dX / dW behind PointwiseConvFn: "vendor" (MIOpen through aten::convolution_backward) or "own" (the streaming
# product's backward layouts: deterministic split over pixels, no atomics / zero fills).  Module attribute, flipped by the parity tests
# and the A/B tools; nothing reads the environment.
POINTWISE_BACKWARD = "vendor"


class PointwiseConvFn(torch.autograd.Function):
    """pointwise_conv with gradients (the adapted res3 - res5 bottlenecks and the FPN laterals inside the TTA step): forward one
    fused product; backward g = gout * (out > 0) (one pass, only behind a ReLU), dX = g W, dW = g^T X over pixel slices added in
    a fixed order (no atomics, no zero fill), d bias = column sums, d residual = g (summed over 2 x 2 blocks behind ``res_up``).
    bias2 is a FrozenBN shift (no gradient).  A strided convolution runs on the compacted input, which is also what dW reads."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, bias2, relu, stride, res_up):
        xs = x if stride == 1 else x[:, :, ::stride, ::stride].contiguous(memory_format=torch.channels_last)
        out = pointwise_conv(xs, w, bias, residual, bias2, relu=relu, res_up=res_up)
        ctx.save_for_backward(xs, w, out if relu else None)
        ctx.cfg = (relu, stride, res_up, tuple(x.shape), residual is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        xs, w, out = ctx.saved_tensors
        relu, stride, res_up, xshape, has_res = ctx.cfg
        B, Cin, Ho, Wo = xs.shape
        Cout = w.shape[0]
        M = B * Ho * Wo
        if gout.dtype != torch.float32:
            raise TypeError("PointwiseConvFn: float32 gradients only")
        g = gout if gout.is_contiguous(memory_format=torch.channels_last) else gout.contiguous(memory_format=torch.channels_last)
        if relu:
            gin = torch.empty_like(out)
            with _timed("relu_bwd", out.numel() * 12):
                call("ttdg_relu_bwd", ptr(g), ptr(out), ptr(gin), C.c_size_t(out.numel()), stream())
            g = gin
        dx = dw = db = dres = None
        if POINTWISE_BACKWARD == "vendor":
            # in the network MIOpen's tuned data- / weight-gradient kernels for these shapes run 48 - 52 us where the streaming
            # product's two backward layouts take 53 - 81 us (profiles/r06_pointwise_ab_in_situ.txt): the backward products stay with
            # the vendor; the forward product keeps its fused epilogue.
            mask = (bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1]), False)
            dxs = None
            if mask[0] or mask[1]:
                dxs, dw, _ = torch.ops.aten.convolution_backward(g, xs, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, mask)
        else:
            dxs = None
            if ctx.needs_input_grad[0]:
                dxs = _cl_empty(B, Cin, Ho, Wo, xs.device)
                with _timed("pointwise_dx", (4 * (M * Cin + M * Cout + Cin * Cout), 2 * M * Cin * Cout)):
                    mm(g, w, dxs, M, Cin, Cout, Cout, Cin, Cin, b_layout=1)                     # B(cin, cout) = w[cout, cin]
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(w)
                if dw.stride(0) != Cin or dw.stride(1) != 1:
                    dw = torch.empty(Cout, Cin, 1, 1, device=w.device, dtype=torch.float32)
                with _timed("pointwise_dw", (4 * (M * Cin + M * Cout + Cin * Cout), 2 * M * Cin * Cout)):
                    mm(g, xs, dw, Cout, Cin, M, Cout, Cin, Cin, a_layout=1, b_layout=1, kslices=_dw_slices(Cout, Cin, M))
                if dw.stride() != w.stride():
                    dw = dw.as_strided(w.shape, w.stride())                                     # same storage order for a 1 x 1 filter
        if dxs is not None:
            if stride == 1:
                dx = dxs
            else:
                dx = torch.zeros(xshape[0], xshape[2], xshape[3], xshape[1], device=xs.device, dtype=torch.float32).permute(0, 3, 1, 2)
                dx[:, :, ::stride, ::stride] = dxs
        if ctx.needs_input_grad[2]:
            db = g.sum((0, 2, 3))
        if has_res and ctx.needs_input_grad[3]:
            dres = g if not res_up else torch.nn.functional.avg_pool2d(g, 2, divisor_override=1)
        return dx, dw, db, dres, None, None, None, None


class BiasAddFn(torch.autograd.Function):
    """(y + bias[c]) + residual IN PLACE on a convolution output, with gradients: the bias of the FPN's lateral / output
    convolutions (and the top-down sum) in one vectorised pass instead of torch's broadcast add_ (+ add).  Backward: the
    gradient passes through to y and the residual unchanged, the bias gets its sum over (N, H, W)."""

    @staticmethod
    def forward(ctx, y, bias, residual):
        bias_act_(y, bias, residual, None, relu=False)
        ctx.mark_dirty(y)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, gout):
        gb = gout.sum((0, 2, 3)) if ctx.needs_input_grad[1] else None
        return gout, gb, (gout if ctx.has_res else None)


def resize_u8(img, oh, ow):
    """(..., H, W) uint8 on the device -> (..., oh, ow) uint8: the test mapper's bilinear resize (antialiased when shrinking),
    bit-compatible with data.map_for_test to <= 1 LSB (csrc/resize.hip)."""
    if img.dtype != torch.uint8 or not img.is_contiguous():
        raise TypeError("resize_u8: contiguous uint8 tensor expected")
    H, W = int(img.shape[-2]), int(img.shape[-1])
    planes = img.numel() // max(1, H * W)
    out = torch.empty(tuple(img.shape[:-2]) + (int(oh), int(ow)), device=img.device, dtype=torch.uint8)
    nws = _lib.load().ttdg_resize_u8_workspace_bytes(planes, H, W, int(oh), int(ow))
    ws = torch.empty(nws, device=img.device, dtype=torch.uint8) if nws else None
    call("ttdg_resize_bilinear_u8", ptr(img), ptr(out), planes, H, W, int(oh), int(ow), ptr(ws), stream())
    return out


_SIZES_CACHE = {}


def image_sizes_tensor(sizes, device):
    """(B, 2) float tensor of (height, width) per image, cached per (sizes, device) - the fused box kernels clip with it."""
    key = (tuple((int(h), int(w)) for h, w in sizes), str(device))
    t = _SIZES_CACHE.get(key)
    if t is None:
        if len(_SIZES_CACHE) > 64:
            _SIZES_CACHE.clear()
        t = torch.tensor(key[0], dtype=torch.float32).reshape(-1, 2).to(device)
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()       # cached constants may be read from other streams
        _SIZES_CACHE[key] = t
    return t


def rpn_decode(deltas, anchors, idx, score, sizes_t, boxes, scores, col0):
    """One FPN level of find_top_rpn_proposals [3P]: decode the top-k anchors from the NCHW head output, clip, validity
    -> columns [col0, col0+k) of boxes (B, K, 4) / scores (B, K); invalid candidates get score -inf."""
    B, A4, H, W = deltas.shape
    k = idx.shape[1]
    call("ttdg_rpn_decode", ptr(deltas.contiguous()), ptr(anchors), ptr(idx.contiguous()), ptr(score.contiguous()), ptr(sizes_t),
         B, k, A4 // 4, H, W, boxes.shape[1], int(col0), ptr(boxes), ptr(scores), stream())


RPN_SELECT = True      # False = per level permute + torch.topk + rpn_decode: the path beyond the selection kernel's limits (parity test)


def rpn_select(logits, deltas, anchors, ks, sizes_t, boxes, scores):
    """find_top_rpn_proposals [3P] up to the NMS for every level and image in ONE launch (csrc/detection.hip rpn_select_kernel):
    logits[l] (B, A, H, W) / deltas[l] (B, 4A, H, W) as the head produces them, anchors[l] (H*W*A, 4); the ks[l] best logits of
    every (image, level) in descending order, decoded + clipped + tested into their column block of boxes (B, K, 4) / scores
    (B, K) (-inf = rejected, box zeroed).  Falls back to the per-level path for inputs the kernel does not take."""
    B, A = logits[0].shape[0], logits[0].shape[1]
    heads = list(logits) + list(deltas)
    nhwc = all(t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) for t in heads) and any(is_channels_last(t) for t in heads)
    if not nhwc:
        logits, deltas = [t.contiguous() for t in logits], [t.contiguous() for t in deltas]
    fused = RPN_SELECT and len(logits) <= _lib.RPN_LEVELS_MAX and max(ks) <= _lib.RPN_SELECT_MAX_K and \
        all(t.dtype == torch.float32 for t in heads) and all(t.dtype == torch.float32 and t.is_contiguous() for t in anchors) and all(lg.shape[1] == A for lg in logits)
    col = 0
    if not fused:
        for lg, dl, an, k in zip(logits, deltas, anchors, ks):
            sc, idx = lg.permute(0, 2, 3, 1).reshape(B, -1).topk(k, dim=1)
            rpn_decode(dl, an, idx, sc.float(), sizes_t, boxes, scores, col)
            col += k
        return
    items = []
    for lg, dl, an, k in zip(logits, deltas, anchors, ks):
        it = _lib.RpnLevel()
        it.logits, it.deltas, it.anchors = ptr(lg), ptr(dl), ptr(an)
        it.H, it.W, it.k, it.col0 = lg.shape[2], lg.shape[3], int(k), col
        items.append(it)
        col += k
    call("ttdg_rpn_select_nhwc" if nhwc else "ttdg_rpn_select", (_lib.RpnLevel * len(items))(*items), len(items), B, A, ptr(sizes_t), boxes.shape[1],
         ptr(boxes), ptr(scores), stream())


def _grouped_flags(bx, sc, gf, ngroups, max_group, thr):
    """keep mask (N,) of greedy NMS inside each group; group id == ngroups marks candidates that take no part."""
    N = sc.numel()
    order = torch.argsort(sc, descending=True)
    gs = gf[order]
    o2 = torch.argsort(gs, stable=True)                          # (group, descending score)
    final = order[o2]
    seg = torch.searchsorted(gs[o2].contiguous(), torch.arange(ngroups + 1, device=sc.device, dtype=gs.dtype)).to(torch.int32)
    b = bx[final].contiguous()
    mg = max(1, min(int(max_group), N))
    ws = torch.empty(N * ((mg + 63) // 64), dtype=torch.int64, device=sc.device)
    flags = torch.zeros(N, dtype=torch.uint8, device=sc.device)
    call("ttdg_nms_grouped", ptr(b), ptr(seg), int(ngroups), N, mg, float(thr), ptr(ws), ptr(flags), stream())
    keep = torch.zeros(N, dtype=torch.bool, device=sc.device)
    keep[final] = flags.bool()
    return keep


_SEG_CACHE = {}


def _presorted_flags(boxes, level_sizes, thr, max_group):
    """keep mask (B, K) of greedy NMS inside every (image, level) group when the columns of ``boxes`` (B, K, 4) are the
    blocks [level 0 | level 1 | ...] of ``level_sizes`` columns, each block already in descending score order (the per-level
    top-k output): the (group, descending score) order the sweep needs is the memory order - no sort, no gather, no
    scatter (two argsorts + searchsorted + three gathers = ~30 launches per RPN call otherwise).  Dead candidates (score
    -inf: non-finite or empty after clipping) stay where they are: a box without area overlaps nothing, so it suppresses
    nothing, and its own flag is masked by its score afterwards."""
    B, K = boxes.shape[:2]
    dev = boxes.device
    key = (B, tuple(level_sizes), str(dev))
    seg = _SEG_CACHE.get(key)
    if seg is None:
        ends, acc = [0], 0
        for k in level_sizes:
            acc += k
            ends.append(acc)
        seg = torch.tensor([b * K + e for b in range(B) for e in ends[:-1]] + [B * K], dtype=torch.int32).to(dev)
        torch.cuda.current_stream(dev).synchronize()        # cached constants may be read from other streams
        _SEG_CACHE[key] = seg
    N = B * K
    b = boxes.reshape(-1, 4)
    check_f32(b)
    mg = max(1, min(int(max_group), N))
    ws = torch.empty(N * ((mg + 63) // 64), dtype=torch.int64, device=dev)
    flags = torch.zeros(N, dtype=torch.uint8, device=dev)
    call("ttdg_nms_grouped", ptr(b), ptr(seg), B * len(level_sizes), N, mg, float(thr), ptr(ws), ptr(flags), stream())
    return flags.view(B, K).bool()


def nms_batched(boxes, scores, lvl, nlvl, thr, max_group, topk, device_counts=False, level_sizes=None):
    """RPN selection for a whole batch in one pass: boxes (B, K, 4), scores (B, K) with -inf for dead candidates, lvl (K,)
    level of every column.  Greedy NMS inside every (image, level) group - all groups swept concurrently - then the
    `topk` best survivors per image.  Returns (idx (B, topk) sorted by descending score, counts list[int]); ONE host
    synchronisation - or none with ``device_counts`` (counts stay a device tensor: the whole call is capturable).
    ``level_sizes``: the columns are per-level blocks of these sizes, each sorted by descending score (see _presorted_flags)."""
    B, K = scores.shape
    dev = scores.device
    if level_sizes is not None and sum(level_sizes) == K and len(level_sizes) == nlvl and boxes.dtype == torch.float32 and boxes.is_contiguous():
        keep = _presorted_flags(boxes, level_sizes, thr, max_group)
    else:
        sc = scores.reshape(-1)
        valid = sc > float("-inf")
        g = (torch.arange(B, device=dev, dtype=torch.int64)[:, None] * nlvl + lvl.to(torch.int64)[None, :]).reshape(-1)
        g = torch.where(valid, g, torch.full_like(g, B * nlvl))
        keep = _grouped_flags(boxes.reshape(-1, 4).float(), sc.float(), g, B * nlvl, max_group, thr)
    masked = torch.where(keep.view(B, K), scores, scores.new_full((), float("-inf")))
    top = masked.topk(min(int(topk), K), dim=1)
    counts = (top.values > float("-inf")).sum(1)
    return top.indices, (counts if device_counts else counts.tolist())


def box_inference(logits, deltas, rois, sizes_t, num_classes, weights, score_thresh):
    """fast_rcnn_inference [3P] up to the NMS, fused: softmax, per-class decode, clip, validity, score threshold.
    Returns boxes (N, C, 4) and scores (N, C) with -inf for rejected candidates."""
    N = logits.shape[0]
    boxes = torch.empty(N, num_classes, 4, device=logits.device, dtype=torch.float32)
    scores = torch.empty(N, num_classes, device=logits.device, dtype=torch.float32)
    call("ttdg_box_inference", ptr(logits.float().contiguous()), ptr(deltas.float().contiguous()), ptr(rois.float().contiguous()),
         ptr(sizes_t), N, int(num_classes), float(weights[0]), float(weights[1]), float(weights[2]), float(weights[3]),
         float(score_thresh), ptr(boxes), ptr(scores), stream())
    return boxes, scores


def nms_ragged(boxes, scores, rois_per_image, num_classes, thr, topk):
    """Per-class NMS + top-k per image over the candidates of a whole batch: boxes (N, C, 4), scores (N, C) (-inf = dead);
    image b owns rows [sum(rois_per_image[:b]), ...).  Returns, per image, the flat (row * C + class) indices of its
    detections by descending score; ONE host synchronisation."""
    N, C = scores.shape
    dev = scores.device
    B = len(rois_per_image)
    if N == 0:
        return [torch.empty(0, dtype=torch.int64, device=dev) for _ in range(B)]
    sc = scores.reshape(-1)
    img = torch.repeat_interleave(torch.arange(B, device=dev, dtype=torch.int64),
                                  torch.tensor(rois_per_image, device=dev, dtype=torch.int64), output_size=N)
    g = (img[:, None] * C + torch.arange(C, device=dev, dtype=torch.int64)[None, :]).reshape(-1)
    g = torch.where(sc > float("-inf"), g, torch.full_like(g, B * C))
    keep = _grouped_flags(boxes.reshape(-1, 4), sc, g, B * C, max(rois_per_image), thr)
    masked = torch.where(keep, sc, sc.new_full((), float("-inf")))
    # dense (B, maxlen) view of the ragged per-image candidate ranges
    lens = [n * C for n in rois_per_image]
    maxlen = max(lens)
    starts = [0]
    for n in lens[:-1]:
        starts.append(starts[-1] + n)
    ar = torch.arange(maxlen, device=dev, dtype=torch.int64)
    st = torch.tensor(starts, device=dev, dtype=torch.int64)[:, None]
    ln = torch.tensor(lens, device=dev, dtype=torch.int64)[:, None]
    idxmat = (st + ar[None, :]).clamp(max=N * C - 1)
    dense = torch.where(ar[None, :] < ln, masked[idxmat], masked.new_full((), float("-inf")))
    top = dense.topk(min(int(topk), maxlen), dim=1)
    flat = idxmat.gather(1, top.indices)
    counts = (top.values > float("-inf")).sum(1).tolist()
    return [flat[b, :counts[b]] for b in range(B)]


def paste_masks(masks, boxes, H, W, threshold=0.5):
    """paste_masks_in_image [3P]: soft masks (R, S, S) or (R, 1, S, S) -> (R, H, W) bool, one launch for the batch."""
    R = masks.shape[0]
    out = torch.empty(R, int(H), int(W), dtype=torch.uint8, device=masks.device)
    if R:
        m = masks.reshape(R, masks.shape[-2], masks.shape[-1]).float().contiguous()
        call("ttdg_paste_masks", ptr(m), ptr(boxes.float().contiguous()), R, m.shape[-1], int(H), int(W), float(threshold), ptr(out), stream())
    return out.view(torch.bool)


def mask_pair_counts(pred_ptrs, gt_ptrs, cy, cx, H, W, device):
    """(npairs, 12) int32 quadrant counts {n(p&g), n(p), n(g)} x 4 for pairs of H x W byte masks given by device address
    (DiceEvaluator; dice_metric.py:25-92).  ``pred_ptrs`` / ``gt_ptrs`` / ``cy`` / ``cx``: python int lists, uploaded as one
    small table.  The caller keeps the mask tensors alive until the stream has consumed them."""
    n = len(pred_ptrs)
    counts = torch.empty(n, 12, device=device, dtype=torch.int32)
    if n == 0:
        return counts
    table = torch.tensor([pred_ptrs, gt_ptrs], dtype=torch.int64).to(device, non_blocking=True)
    cuts = torch.tensor([cy, cx], dtype=torch.int32).to(device, non_blocking=True)
    call("ttdg_mask_pair_counts", ptr(table), ptr(table) + 8 * n, ptr(cuts), ptr(cuts) + 4 * n, n, int(H), int(W), ptr(counts), stream())
    return counts


def mask_pair_measures(pred_ptrs, gt_ptrs, cy, cx, owner, H, W, best, alpha=0.5):
    """mask_pair_counts + the float64 closed forms + the maximum over a prediction's same-class ground truths, on the device:
    best[owner[i]] = max(best[owner[i]], 100 * (Dice, E-measure, S-measure) of pair i); ``best`` (npred, 3) float64, zero-initialised
    by the caller.  Two launches and two small table uploads for all pairs of one mask size.  Returns the (npairs, 12) counts."""
    n = len(pred_ptrs)
    counts = torch.empty(n, 12, device=best.device, dtype=torch.int32)
    if n == 0:
        return counts
    if best.dtype != torch.float64 or not best.is_contiguous() or best.shape[1] != 3:
        raise TypeError("mask_pair_measures: best must be a contiguous (npred, 3) float64 tensor")
    table = torch.tensor([pred_ptrs, gt_ptrs], dtype=torch.int64).to(best.device, non_blocking=True)
    cuts = torch.tensor([cy, cx, owner], dtype=torch.int32).to(best.device, non_blocking=True)
    call("ttdg_mask_pair_counts", ptr(table), ptr(table) + 8 * n, ptr(cuts), ptr(cuts) + 4 * n, n, int(H), int(W), ptr(counts), stream())
    call("ttdg_mask_measures", ptr(counts), ptr(cuts), ptr(cuts) + 4 * n, ptr(cuts) + 8 * n, n, int(H), int(W), float(alpha), ptr(best), stream())
    return counts


def nms(boxes, scores, thr, group=None):
    """Greedy NMS; returns kept indices (into the input order) sorted by descending score."""
    ng = None if group is None else int(group.max().item()) + 1 if group.numel() else 1
    return nms_collect([nms_launch(boxes, scores, thr, group, ng)])[0]
