"""GPU parity tests (run on the MI355X box with `-m gpu`): every HIP operator, called through the C ABI,
against the CPU oracle and the committed golden vectors on the same seeded inputs.

Bars (BASELINE.json north_star / SURVEY.md §8d): <=1e-4 fp32 on Wds, V, loss and gradients; identical
permutation matrices; bit-exact for integer/index work (sampler, LAP)."""
import copy
import numpy as np
import pytest
import torch

import cases
from ttdg_mgm_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4

# Which statement did a data-dependent test actually make?  Every `if` that chooses between a strong assertion (identical
# permutations, identical counts) and a weaker one records its branch here; test_statement_ledger (last test of the file)
# ASSERTS how often the strong branches ran and writes the ledger to gpurun_out/parity_ledger.json (a
# green run must say which statements were made).
import collections
LEDGER = collections.Counter()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from ttdg_mgm_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def maxerr(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


def derived_gate(what, dev_val, ref32, truth64, scale=1.0, factor=2.0, quiet=False):
    """Where the 1e-4 bar is below what fp32 arithmetic can deliver (log-domain magnitudes of 1/tau = 20..160 times the
    input, iterated maps), the gate is DERIVED, not chosen: `truth64` is the same oracle computation in float64 on the same
    fp32 inputs, e_ref = |fp32 oracle - truth| is what the reference's own fp32 path loses, and the device must stay within
    max(1e-4 * scale, factor * e_ref) of the truth.  Both errors are printed."""
    t = truth64.detach().cpu().double()
    e_ref = float((ref32.detach().cpu().double() - t).abs().max())
    e_dev = float((dev_val.detach().cpu().double() - t).abs().max())
    bound = max(TOL * scale, factor * e_ref)
    if not quiet:
        print("%s: device vs fp64 truth %.3e, fp32 oracle vs fp64 truth %.3e, gate %.3e" % (what, e_dev, e_ref, bound))
    assert e_dev <= bound, (what, e_dev, e_ref, bound)
    return e_dev, e_ref


def planted_gold(golden, name):
    """mgm3.npz holds the planted cases of up to 32 nodes per graph, mgm3_big.npz the ones where the kernels branch."""
    return golden("mgm3_big" if name.startswith("pb_") else "mgm3")


ALL_PLANTED = cases.PLANTED_CASES + cases.PLANTED_BIG_CASES


def check_pgrad(gold, key, g, tol):
    flat = g.detach().reshape(-1).cpu()
    assert maxerr(flat[::cases.PSTRIDE], gold[key + "__sample"]) <= tol, key
    n0 = float(gold[key + "__norm"])
    assert abs(float(flat.double().norm()) - n0) <= tol * max(1.0, n0) * 10, key


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (7, 5, 3), (64, 64, 16), (120, 512, 256), (131, 257, 70), (2048, 32, 256)])
def test_gemm_all_layouts(dev, M, N, K):
    from ttdg_mgm_amd import ops
    g = synth.gen(M * 1000 + N * 10 + K)
    A, B = synth.normal(g, (M, K)).to(dev), synth.normal(g, (N, K)).to(dev)
    bias = synth.normal(g, (N,)).to(dev)
    ref = A.double() @ B.double().t()
    tol = 2e-6 * K ** 0.5 * 4 + 1e-6
    C = torch.empty(M, N, device=dev)
    ops.gemm(A, K, 1, B, K, 1, C, N, 1, M, N, K)                                       # NT
    assert maxerr(C, ref.float()) <= tol * 4
    At, Bt = A.t().contiguous(), B.t().contiguous()
    C2 = torch.empty(M, N, device=dev)
    ops.gemm(At, 1, M, Bt, 1, N, C2, N, 1, M, N, K, bias=bias)                         # TN with bias
    assert maxerr(C2, (ref + bias.double()).float()) <= tol * 4
    C3 = torch.full((N, M), 2.0, device=dev)
    ops.gemm(A, K, 1, B, K, 1, C3, 1, M, M, N, K, alpha=0.5, beta=1.0)                 # transposed C, alpha/beta
    assert maxerr(C3, (0.5 * ref.t() + 2.0).float()) <= tol * 4


@pytest.mark.parametrize("M,N,K", [(2048, 512, 256), (256, 256, 2048), (70, 130, 33), (64, 64, 32), (65, 63, 31), (5, 300, 129),
                                   (64, 64, 36), (128, 64, 68), (64, 128, 96), (72, 68, 100), (64, 64, 132), (60, 64, 164), (64, 64, 4)])   # 1..6 slabs: every tail of the ring of three
def test_gemm_pipelined_staging_every_operand_layout(dev, M, N, K):
    """[r4] csrc/gemm.hip stages the next K slab global -> registers -> LDS behind the MFMAs, with 16-byte loads where the operand
    allows them.  Every staging path - k-contiguous, m-contiguous and general strides for either operand, 16-byte loads legal or not
    (odd leading dimension, base pointer off by one element), ragged edges in M, N and K, the split-K entry (256 x 256 x 2048) -
    against the float64 product; and the k order of an accumulator does not depend on the path: all layouts of one product return
    the SAME bits."""
    from ttdg_mgm_amd import ops
    g = synth.gen(M * 7 + N * 3 + K)
    A, B = synth.normal(g, (M, K)).to(dev), synth.normal(g, (N, K)).to(dev)
    ref = (A.double() @ B.double().t()).float()
    tol = (2e-6 * K ** 0.5 * 4 + 1e-6) * 4
    outs = []

    def run(a, sam, sak, b, sbn, sbk, a_off=0, b_off=0):
        C = torch.empty(M, N, device=dev)
        ops.gemm(a, sam, sak, b, sbn, sbk, C, N, 1, M, N, K, a_off=a_off, b_off=b_off)
        assert maxerr(C, ref) <= tol
        outs.append(C)
    At, Bt = A.t().contiguous(), B.t().contiguous()
    run(A, K, 1, B, K, 1)                                    # NT: both k-contiguous
    run(At, 1, M, Bt, 1, N)                                  # TN: both m-contiguous
    run(A, K, 1, Bt, 1, N)                                   # NN
    run(At, 1, M, B, K, 1)                                   # TT
    # 16-byte loads illegal: odd leading dimension / base pointer off by one element
    Aw = torch.zeros(M, K + 1, device=dev); Aw[:, :K] = A
    Bw = torch.zeros(N, K + 3, device=dev); Bw[:, 1:K + 1] = B
    run(Aw, K + 1, 1, Bw, K + 3, 1, b_off=1)
    Atw = torch.zeros(K, M + 1, device=dev); Atw[:, 1:] = At
    run(Atw, 1, M + 1, Bt, 1, N, a_off=1)
    # general strides: neither index has unit stride
    A2 = torch.zeros(M, 2 * K, device=dev); A2[:, ::2] = A
    B2 = torch.zeros(N, 3 * K, device=dev); B2[:, ::3] = B
    run(A2, 2 * K, 2, B2, 3 * K, 3)
    run(A2, 2 * K, 2, B, K, 1)
    for C in outs[1:]:
        assert torch.equal(C, outs[0]), "the accumulation order must not depend on the operand layout"


@pytest.mark.parametrize("M,N", [(2048, 512), (7, 3), (130, 70), (513, 16), (64, 17)])
def test_colsum_bias_gradient_kernel(dev, M, N):
    """csrc/gemm.hip colsum_kernel (bias gradients of the affinity MLP, reference utils/affinity.py:37-40 through autograd):
    against the float64 column sums, and deterministic (fixed-order LDS reduction: two runs return the same bits)."""
    from ttdg_mgm_amd import ops
    X = synth.normal(synth.gen(M * 31 + N), (M, N)).to(dev)
    a, b = ops.colsum(X), ops.colsum(X)
    assert torch.equal(a, b)
    assert maxerr(a, X.double().sum(0).float()) <= 1e-6 * M ** 0.5 * 4 + 1e-6


def test_gemm_grouped_all_layouts_and_two_segments(dev):
    """csrc/gemm_grouped.hip: up to eight products in one launch (32 x 32 tiles, K split over the wavefronts), every operand
    layout (NT / TN / NN, transposed C, column offsets), bias / alpha / beta, a second K segment, empty and ragged shapes - each
    against the float64 product, and bit-for-bit against itself when launched alone (tile / group independent)."""
    from ttdg_mgm_amd import ops
    g = synth.gen(4711)
    shapes = [(1, 1, 1), (7, 5, 3), (120, 256, 256), (131, 257, 70), (33, 32, 130), (512, 256, 119), (120, 32, 256), (64, 64, 64)]
    descs, checks, keep = [], [], []
    for i, (M, N, K) in enumerate(shapes):
        A, B, bias = synth.normal(g, (M, K)).to(dev), synth.normal(g, (N, K)).to(dev), synth.normal(g, (N,)).to(dev)
        ref = A.double() @ B.double().t()
        tol = (2e-6 * K ** 0.5 * 4 + 1e-6) * 4
        if i % 4 == 0:        # NT, bias
            C = torch.empty(M, N, device=dev)
            descs.append(ops.gdesc(A, K, 1, B, K, 1, C, N, 1, M, N, K, bias=bias))
            checks.append((C, (ref + bias.double()), tol))
        elif i % 4 == 1:      # TN (both operands m-contiguous), alpha / beta
            At, Bt = A.t().contiguous(), B.t().contiguous()
            C = torch.full((M, N), 2.0, device=dev)
            descs.append(ops.gdesc(At, 1, M, Bt, 1, N, C, N, 1, M, N, K, alpha=0.5, beta=1.0))
            checks.append((C, 0.5 * ref + 2.0, tol))
            keep += [At, Bt]
        elif i % 4 == 2:      # transposed C, B read at a column offset of a wider matrix
            Bw = torch.cat((synth.normal(g, (N, 9)).to(dev), B), 1).contiguous()
            C = torch.empty(N, M, device=dev)
            descs.append(ops.gdesc(A, K, 1, Bw, K + 9, 1, C, 1, M, M, N, K, b_off=9))
            checks.append((C, ref.t(), tol))
            keep.append(Bw)
        else:                 # two K segments: A B^T + A2 B2^T, second one NN
            K2 = 37
            A2, B2 = synth.normal(g, (M, K2)).to(dev), synth.normal(g, (K2, N)).to(dev)
            C = torch.empty(M, N, device=dev)
            descs.append(ops.gdesc(A, K, 1, B, K, 1, C, N, 1, M, N, K, second=(A2, K2, 1, B2, 1, N, K2)))
            checks.append((C, ref + A2.double() @ B2.double(), tol * 2))
            keep += [A2, B2]
        keep += [A, B, bias]
    ops.gemm_grouped(descs)
    for (C, ref, tol), shp in zip(checks, shapes):
        assert maxerr(C, ref.float()) <= tol, shp
    # launched alone: same bits (the tile's arithmetic does not depend on the group)
    M, N, K = shapes[2]
    A, B = synth.normal(g, (M, K)).to(dev), synth.normal(g, (N, K)).to(dev)
    Ca, Cb = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.gemm_grouped([ops.gdesc(A, K, 1, B, K, 1, Ca, N, 1, M, N, K)])
    ops.gemm_grouped([ops.gdesc(B, K, 1, A, K, 1, torch.empty(N, M, device=dev), M, 1, N, M, K), ops.gdesc(A, K, 1, B, K, 1, Cb, N, 1, M, N, K)])
    assert torch.equal(Ca, Cb)
    ops.gemm_grouped([ops.gdesc(A, K, 1, B, K, 1, Ca, N, 1, 0, N, K)])          # an empty product is skipped


def test_linear_autograd(dev):
    from ttdg_mgm_amd import ops
    g = synth.gen(77)
    x, W, b = (synth.normal(g, s).to(dev).requires_grad_() for s in ((45, 256), (512, 256), (512,)))
    y = ops.LinearFn.apply(x, W, b)
    y.square().sum().backward()
    xr, Wr, br = (t.detach().clone().requires_grad_() for t in (x, W, b))
    torch.nn.functional.linear(xr, Wr, br).square().sum().backward()
    for got, ref in ((x.grad, xr.grad), (W.grad, Wr.grad), (b.grad, br.grad)):       # sums of 45..512 products of O(10) terms
        assert maxerr(got, ref) <= 2e-5 * float(ref.abs().max())


# ------------------------------------------------------------------------------------------- A4
@pytest.mark.parametrize("ci", range(len(cases.AFF_CASES)))
def test_affinity_golden(dev, golden, ci):
    from ttdg_mgm_amd.GModule.utils.affinity import Affinity
    gold = golden("affinity")
    m = Affinity(256).to(dev)
    sd = {k[len("node_affinity."):]: v for k, v in synth.mgm3_params(cases.AFF_PARAM_SEED).items() if k.startswith("node_affinity.")}
    m.load_state_dict(sd, strict=True)
    X, Y, R = (t.to(dev) for t in cases.aff_inputs(ci))
    X.requires_grad_(), Y.requires_grad_()
    M = m(X, Y)
    (M * R).sum().backward()
    assert maxerr(M, gold[f"c{ci}_M"]) <= 1e-5
    assert maxerr(X.grad, gold[f"c{ci}_dX"]) <= TOL and maxerr(Y.grad, gold[f"c{ci}_dY"]) <= TOL
    for k, p in m.named_parameters():
        check_pgrad(gold, f"c{ci}_d_{k}", p.grad, 5e-4)


def test_affinity_large_matches_oracle(dev):
    """cfg-3 node count per graph (256) with K split and tile edges; oracle = materialised-MLP formulation."""
    from oracle import gmodule as og
    from ttdg_mgm_amd.GModule.utils.affinity import Affinity
    p = synth.mgm3_params(31)
    g = synth.gen(32)
    X, Y = synth.normal(g, (200, 256), 0.5), synth.normal(g, (131, 256), 0.5)
    ref = og.affinity(p, X, Y)
    m = Affinity(256).to(dev)
    m.load_state_dict({k[len("node_affinity."):]: v for k, v in p.items() if k.startswith("node_affinity.")})
    assert maxerr(m(X.to(dev), Y.to(dev)), ref) <= 2e-5


@pytest.mark.parametrize("sizes,seed", [((130, 70, 257, 33), 51), ((64, 64, 64), 52), ((200, 9), 53), ((96,) * 5, 54)])
def test_affinity_pairwise_bwd_ragged_vs_float64(dev, sizes, seed):
    """[r5] the stand-alone affinity backward (csrc/affinity.hip: both passes in one launch, slices of the block-triangular range that
    exist only where it has data, partial planes summed by the finish kernel) on ragged batches: node counts that are no multiple of
    the 64-row tile or of 4, row tiles that span two graphs (per-row limits inside a tile), a first graph without any dP range and a
    last one without any dQ range.  Reference: S[i,k] = sum_{j in earlier graphs} dM[i,j] [P[i,k] > -Q[j,k]] (the fp32 comparison the
    kernel makes: reference utils/affinity.py:44-57 through relu'), R likewise over later graphs, dP = w2 S, dQ = w2 R,
    dw2 = sum P S + sum Q R, db2 = sum dM - accumulated in float64."""
    from ttdg_mgm_amd import ops
    g = synth.gen(seed)
    M, H = sum(sizes), 512
    P, Q = synth.normal(g, (M, H), 0.5), synth.normal(g, (M, H), 0.5)
    w2 = synth.normal(g, (H,), 0.3)
    dM = synth.normal(g, (M, M), 1.0)
    gid = torch.cat([torch.full((n,), k) for k, n in enumerate(sizes)])
    wanted = gid[:, None] > gid[None, :]
    dMw = torch.where(wanted, dM, torch.zeros(()))
    S, R = torch.zeros(M, H, dtype=torch.float64), torch.zeros(M, H, dtype=torch.float64)
    off = [0]
    for n in sizes:
        off.append(off[-1] + n)
    for a in range(len(sizes)):
        for b in range(a):
            ia, ib = slice(off[a], off[a + 1]), slice(off[b], off[b + 1])
            step = (P[ia][:, None, :] > -Q[ib][None, :, :]).double()                    # (na, nb, H)
            D = dM[ia, ib].double()
            S[ia] += torch.einsum("ij,ijk->ik", D, step)
            R[ib] += torch.einsum("ij,ijk->jk", D, step)
    ref = dict(dP=S * w2.double(), dQ=R * w2.double(), dw2=(P.double() * S).sum(0) + (Q.double() * R).sum(0), db2=dMw.double().sum().reshape(1))
    # garbage outside the wanted blocks must not matter (the header: dM is read only where g(i) > g(j))
    dM_dev = torch.where(wanted, dM, torch.full((), float("nan"))).to(dev).contiguous()
    dP, dQ, dw2, db2 = ops.affinity_pairwise_bwd(P.to(dev), Q.to(dev), w2.to(dev), dM_dev, ops.graphs(list(sizes)))
    for name, got in (("dP", dP), ("dQ", dQ), ("dw2", dw2), ("db2", db2)):
        r = ref[name]
        err = float((got.cpu().double() - r).abs().max()) / max(1.0, float(r.abs().max()))
        assert err <= 1e-5, (name, err)


# ------------------------------------------------------------------------------------------- A3
@pytest.mark.parametrize("ci", range(len(cases.MHA_CASES)))
def test_mha_adjacency_golden(dev, golden, ci):
    from ttdg_mgm_amd.GModule.utils.attentions import MultiHeadAttention
    gold = golden("mha")
    m = MultiHeadAttention(256, 1, dropout=0.1, version='v2').to(dev).eval()
    sd = {k[len("intra_domain_graph."):]: v for k, v in synth.mgm3_params(cases.MHA_PARAM_SEED).items() if k.startswith("intra_domain_graph.")}
    m.load_state_dict(sd, strict=True)
    x = cases.mha_input(ci).to(dev)
    out, adj = m([x, x, x])
    assert maxerr(adj, gold[f"c{ci}_adj"]) <= 1e-5
    assert out.shape == x.shape


def test_mha_dropout_statistics(dev):
    from ttdg_mgm_amd import ops
    n, d = 64, 256
    g = synth.gen(5)
    q, k = synth.normal(g, (n, d), 0.1).to(dev), synth.normal(g, (n, d), 0.1).to(dev)
    a0 = ops.mha_adjacency(q, k, ops.graphs([n]), [n], d ** -0.5, 0.0, 1, zero_diag=False).view(n, n)
    a1 = ops.mha_adjacency(q, k, ops.graphs([n]), [n], d ** -0.5, 0.1, 1, zero_diag=False).view(n, n)
    a2 = ops.mha_adjacency(q, k, ops.graphs([n]), [n], d ** -0.5, 0.1, 2, zero_diag=False).view(n, n)
    kept = a1 != 0
    assert 0.85 < float(kept.float().mean()) < 0.95                      # keep probability 0.9
    assert maxerr(a1[kept], (a0 / 0.9)[kept]) <= 1e-6                     # survivors are rescaled
    assert not torch.equal(a1 != 0, a2 != 0)                              # seed changes the mask


def test_mha_adjacency_large_graphs_vs_float64(dev):
    """[r5] graphs of 128+ nodes take another route (csrc/mha.hip: scores by the grouped MFMA GEMM straight into the packed blocks,
    row softmax in place): ragged batch incl. a small graph beside large ones, against softmax(q k^T * 256^-0.5) in float64
    (reference utils/attentions.py:72-86), diagonal zeroed (multi_graph_matching.py:502); with dropout the survivors are the
    rescaled values and the mask is the one the per-row kernel draws for the same (seed, graph, row, column)."""
    from ttdg_mgm_amd import ops
    sizes, d = [256, 130, 300, 64], 256
    g = synth.gen(77)
    M = sum(sizes)
    q, k = synth.normal(g, (M, d), 0.3), synth.normal(g, (M, d), 0.3)
    gr = ops.graphs(sizes)
    a = ops.mha_adjacency(q.to(dev), k.to(dev), gr, sizes, d ** -0.5, 0.0, 1, zero_diag=True).cpu()
    ad = ops.mha_adjacency(q.to(dev), k.to(dev), gr, sizes, d ** -0.5, 0.1, 9, zero_diag=True).cpu()
    o = oa = 0
    for gi, n in enumerate(sizes):
        ref = torch.softmax((q[o:o + n].double() @ k[o:o + n].double().t()) * d ** -0.5, dim=1)
        ref.fill_diagonal_(0.0)
        blk, blkd = a[oa:oa + n * n].view(n, n), ad[oa:oa + n * n].view(n, n)
        assert float((blk.double() - ref).abs().max()) <= 1e-5, gi
        kept = blkd != 0
        off_diag = ~torch.eye(n, dtype=torch.bool)
        assert 0.85 < float(kept[off_diag].float().mean()) < 0.95
        assert float((blkd[kept].double() - (ref / 0.9)[kept]).abs().max()) <= 1e-5
        o += n
        oa += n * n
    # mask identity across the two routes: graph 0 of a batch [64] (per-row kernel) against graph 0 of a batch [64, 256] (GEMM route)
    q2, k2 = q[:64 + 256].to(dev), k[:64 + 256].to(dev)
    m_small = ops.mha_adjacency(q2[:64], k2[:64], ops.graphs([64]), [64], d ** -0.5, 0.1, 5, zero_diag=True).view(64, 64) != 0
    m_large = ops.mha_adjacency(q2, k2, ops.graphs([64, 256]), [64, 256], d ** -0.5, 0.1, 5, zero_diag=True)[:64 * 64].view(64, 64) != 0
    assert torch.equal(m_small, m_large)


# ------------------------------------------------------------------------------------------- A5
SK_CASES = [  # b, r, c, dummy, tau, iters, n1, n2
    (1, 9, 14, True, 0.05, 20, None, None), (1, 14, 9, True, 0.05, 20, None, None), (3, 16, 16, False, 0.5, 21, None, None),
    (4, 22, 32, True, 0.1, 20, None, None), (3, 32, 40, True, 0.00625, 20, None, None), (1, 1, 5, True, 0.1, 20, None, None),
    (3, 30, 32, True, 0.1, 20, (12, 30, 7), None), (3, 40, 32, True, 0.1, 20, (22, 40, 35), None),
    (2, 100, 130, True, 0.05, 20, None, None), (1, 256, 256, True, 0.05, 20, None, None),
]


@pytest.mark.parametrize("b,r,c,dummy,tau,iters,n1,n2", SK_CASES)
def test_sinkhorn_batched_vs_oracle(dev, b, r, c, dummy, tau, iters, n1, n2):
    from oracle.sinkhorn_spec import sinkhorn as osk
    from ttdg_mgm_amd import ops
    s = synth.normal(synth.gen(b * 7 + r * 3 + c), (b, r, c), 0.3)
    if n1 is not None:   # zero padding beyond the valid rows, as pad_tensor produces
        for i, n in enumerate(n1):
            s[i, n:] = 0
    t1 = None if n1 is None else torch.tensor(n1)
    ref = osk(s, n1=t1, dummy_row=dummy, max_iter=iters, tau=tau, batched_operation=True)
    got = ops.sinkhorn_batched(s.to(dev), t1, None, dummy, tau, iters)
    assert maxerr(got, ref) <= TOL


@pytest.mark.parametrize("name", [c[0] for c in cases.SKREF_CASES])
def test_sinkhorn_vs_reference_tree_log_sinkhorn(dev, golden, name):
    """ttdg_sinkhorn_batched_fwd against the reference tree's OWN log-Sinkhorn (graph_matching.py:828-839; fixture
    tests/golden/sinkhorn_ref.npz): probabilities <= 1e-4 and, reconstructed from the potentials the kernel logs
    (log P_ij = s_ij / tau - (f_i + g_j) ln 2), the LOG-domain output.  Log-domain gate: 1e-4 + 2 * sweeps * ulp(max|s/tau|) -
    every sweep rounds a potential of that magnitude once on either side."""
    from ttdg_mgm_amd import ops
    ref = torch.from_numpy(golden("sinkhorn_ref")[name + "_log"])
    s, tau = cases.skref_input(name)
    b, r, c = s.shape
    iters = 2 * cases.SKREF_SWEEPS
    out, pot = ops.sinkhorn_batched(s.to(dev), None, None, r < c, tau, iters, want_pot=True)
    ref = ref[:, :r]
    assert maxerr(out, ref.exp()) <= TOL
    f, g = pot[:, iters - 2, :r].cpu(), pot[:, iters - 1, :c].cpu()
    logp = s / tau - (f[:, :, None] + g[:, None, :]) * float(np.log(2.0))
    live = ref > -60.0                                   # below that the probability is < 1e-26: no information
    err = float((logp - ref)[live].abs().max())
    ulp = float((s / tau).abs().max()) * 2.0 ** -23
    print(name, "log-domain |d| = %.3e, ulp(max|s/tau|) = %.3e, P-domain |d| = %.3e" % (err, ulp, maxerr(out, ref.exp())))
    assert err <= TOL + 2 * iters * ulp


@pytest.mark.parametrize("b,r,c,dummy,tau,n1", [(2, 9, 14, True, 0.1, None), (3, 14, 9, True, 0.05, None), (2, 12, 12, False, 0.2, None),
                                                 (3, 40, 32, True, 0.05, (40, 25, 33)), (2, 20, 32, True, 0.1, (20, 11)),
                                                 (1, 150, 32, True, 0.1, None)])
def test_sinkhorn_differentiable_vs_autograd_through_the_oracle(dev, b, r, c, dummy, tau, n1):
    """Stand-alone Sinkhorn with gradient (ttdg_sinkhorn_batched_bwd, potentials logged by the forward) against autograd
    through the oracle's unrolled sweeps: plain, transposed, ragged (n1) and > 128-column problems."""
    from oracle.sinkhorn_spec import sinkhorn as osk
    from ttdg_mgm_amd.GModule.utils.sinkhorn import Sinkhorn
    g = synth.gen(b * 11 + r * 5 + c)
    s = synth.normal(g, (b, r, c), 0.3)
    w = synth.normal(g, (b, r, c), 1.0)
    if n1 is not None:
        for i, n in enumerate(n1):
            s[i, n:] = 0
    t1 = None if n1 is None else torch.tensor(n1)
    sr = s.clone().requires_grad_()
    ref = osk(sr, n1=t1, dummy_row=dummy, max_iter=20, tau=tau, batched_operation=True)
    (ref * w).sum().backward()
    sd = s.to(dev).requires_grad_()
    out = Sinkhorn(max_iter=20, tau=tau, batched_operation=True)(sd, t1, None, dummy_row=dummy)
    (out * w.to(dev)).sum().backward()
    assert maxerr(out, ref) <= TOL
    valid = torch.ones(b, r, c, dtype=torch.bool)
    if n1 is not None:
        for i, n in enumerate(n1):
            valid[i, n:] = False
    assert torch.isfinite(sd.grad).all()
    if n1 is None:
        s64 = s.double().requires_grad_()
        (osk(s64, dummy_row=dummy, max_iter=20, tau=tau, batched_operation=True) * w.double()).sum().backward()
        derived_gate("sinkhorn bwd (%d,%d,%d) tau %g" % (b, r, c, tau), sd.grad, sr.grad, s64.grad)
    else:
        # autograd through the unrolled batched spec turns the -inf padding of a ragged batch into NaN gradients
        # (0 * inf in the logsumexp backward), so every matrix is checked against its own stand-alone problem instead
        # (same orientation as inside the batch for n != 32)
        for i, n in enumerate(n1):
            si = s[i, :n].clone().requires_grad_()
            (osk(si, dummy_row=dummy, max_iter=20, tau=tau) * w[i, :n]).sum().backward()
            s64 = s[i, :n].double().requires_grad_()
            (osk(s64, dummy_row=dummy, max_iter=20, tau=tau) * w[i, :n].double()).sum().backward()
            derived_gate("ragged sinkhorn bwd %d" % i, sd.grad[i, :n], si.grad, s64.grad)
            assert float(sd.grad[i, n:].abs().max()) == 0 if n < r else True


def test_sinkhorn_module_2d_and_transposed_view(dev):
    from oracle.sinkhorn_spec import sinkhorn as osk
    from ttdg_mgm_amd.GModule.utils.sinkhorn import Sinkhorn
    s = synth.normal(synth.gen(91), (12, 5), 0.3)
    sk = Sinkhorn(max_iter=20, tau=0.05, epsilon=1e-10, batched_operation=False)
    assert maxerr(sk(s.to(dev), dummy_row=True), osk(s, dummy_row=True, max_iter=20, tau=0.05)) <= TOL
    assert maxerr(sk(s.to(dev).t(), dummy_row=True), osk(s.t(), dummy_row=True, max_iter=20, tau=0.05)) <= TOL


@pytest.mark.parametrize("sizes,ks,spike", [((9, 14), 2, None), ((22, 22, 22), 2, None), ((22, 35, 28, 40), 2, None), ((5, 3), 2, None), ((70, 120), 1, None),
                                            ((140, 200, 256), 1, None), ((256, 130), 1, None), ((200, 31, 200), 1, None), ((150, 129), 2, None),
                                            ((200, 150), 1, -12.0), ((256, 256), 1, -10.0), ((256, 3, 2), 1, None), ((130, 1, 3), 1, None)])
def test_pair_sinkhorn_forward_backward(dev, sizes, ks, spike):
    """Pair stage (Wds) and its backward against autograd through the oracle's Sinkhorn.  Sizes above 128 nodes with a
    single K plane run the register-resident kernels (matrix / dY in the register file of one workgroup); with two
    planes they fall back to the LDS / L2 kernels.  ``spike`` (one column and one row of every block shifted: the scaling form's
    potentials leave its range) sends the register kernels' BACKWARD through the log-domain fall-back, where the incoming gradient and
    the result share one buffer; graphs of 1 - 3 nodes next to a graph above 128 exercise the blocks without a whole 16-byte piece."""
    from oracle import gmodule as og
    from ttdg_mgm_amd import ops
    G, M = len(sizes), sum(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)])
    g = synth.gen(sum(sizes) * 13)
    Mraw = synth.normal(g, (M, M), 0.1)
    if spike is not None:
        for a in range(G):
            Mraw[:, off[a] + 3] += spike
            Mraw[off[a] + 5, :] += spike
    Rw = synth.normal(g, (M, M), 1.0)
    b2 = torch.tensor([0.03])
    # oracle: same loop as multi_graph_matching.py:504-525 on the raw affinities
    Mr = Mraw.clone().requires_grad_()
    Wds = torch.zeros(M, M)
    for a in range(G):
        for b in range(G):
            if a < b:
                continue
            blk = Mr[off[a]:off[a + 1], off[b]:off[b + 1]] + b2
            ds = og.sinkhorn_pair(blk) if sizes[b] >= sizes[a] else og.sinkhorn_pair(blk.t()).t()
            Wds[off[a]:off[a + 1], off[b]:off[b + 1]] += ds
            if a != b:
                Wds[off[b]:off[b + 1], off[a]:off[a + 1]] += ds.t()
    # loss only touches the a<b blocks (rows of graph a, cols of graph b)
    mask = torch.zeros(M, M)
    for a in range(G):
        for b in range(a + 1, G):
            mask[off[a]:off[a + 1], off[b]:off[b + 1]] = 1
    (Wds * Rw * mask).sum().backward()
    # the same in float64: what the exact arithmetic gives on these fp32 inputs
    M64 = Mraw.double().requires_grad_()
    W64 = torch.zeros(M, M, dtype=torch.float64)
    for a in range(G):
        for b in range(a + 1):
            blk = M64[off[a]:off[a + 1], off[b]:off[b + 1]] + b2.double()
            ds = og.sinkhorn_pair(blk) if sizes[b] >= sizes[a] else og.sinkhorn_pair(blk.t()).t()
            W64[off[a]:off[a + 1], off[b]:off[b + 1]] += ds
            if a != b:
                W64[off[b]:off[b + 1], off[a]:off[a + 1]] += ds.t()
    (W64 * (Rw * mask).double()).sum().backward()

    gr = ops.graphs(sizes)
    part = (torch.stack([Mraw * 0.25, Mraw * 0.75]) if ks == 2 else Mraw.unsqueeze(0)).to(dev).contiguous()   # K-slices that sum to Mraw
    Wd, pot = ops.sinkhorn_pairs_fwd(part, b2.to(dev), gr, list(sizes), 0.05, 20)
    assert maxerr(Wd, Wds) <= TOL
    dM = ops.sinkhorn_pairs_bwd(part, b2.to(dev), pot, (Rw * mask).to(dev), gr, 0.05, 20)
    low = torch.zeros(M, M)
    for a in range(G):
        for b in range(a):
            low[off[a]:off[a + 1], off[b]:off[b + 1]] = 1
    derived_gate("pair sinkhorn bwd %s" % (sizes,), torch.where(low > 0, dM.cpu(), torch.zeros(())), Mr.grad * low, M64.grad * low)     # (dM is unspecified outside the a > b blocks: select, do not multiply)


@pytest.mark.parametrize("sizes,spike", [((256, 256), None), ((140, 200, 256), None), ((200, 31, 200), None), ((256, 256), -10.0), ((200, 150), -12.0),
                                         ((256, 190), 9.0)])
def test_pair_sinkhorn_scaling_form_log_domain_and_range_fallback(dev, sizes, spike):
    """[r5] The pair stage for graphs of 129-256 nodes runs in the SCALING form (csrc/sinkhorn.hip: K = the row-normalised matrix after
    sweep 0, then s_p = sum_q K v, c_q = sum_p K u - one exponential per entry in all; utils/sinkhorn.py:58-87 / SURVEY Appendix B in
    exact arithmetic).  Checked in the LOG domain - y = L - f - g rebuilt from the potentials the kernel logs for its backward, against
    log of the float64 oracle - on every block, at the gate of test_sinkhorn_vs_reference_tree_log_sinkhorn (1e-4 + 40 ulp of max |s / tau|).
    ``spike``: one column (and one row) of every block is shifted by that much: at tau = 0.05 a shift of -10 is 2^-288 against the
    rest of its rows - the scaling form's column sum underflows, the workgroup must detect it and take the log-domain sweeps instead
    (the oracle has no such limit); +9 overflows the other way round only if the row maximum were not removed first."""
    from oracle import gmodule as og
    from ttdg_mgm_amd import ops
    G, M = len(sizes), sum(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)])
    g = synth.gen(sum(sizes) * 17 + (0 if spike is None else 1))
    Mraw = synth.normal(g, (M, M), 0.1)
    if spike is not None:
        for a in range(G):
            Mraw[:, off[a] + 3] += spike
            Mraw[off[a] + 5, :] += spike
    b2 = torch.tensor([0.03])
    tau, iters = 0.05, 20
    gr = ops.graphs(sizes)
    Wd, pot = ops.sinkhorn_pairs_fwd(Mraw.unsqueeze(0).to(dev).contiguous(), b2.to(dev), gr, list(sizes), tau, iters)
    Wd, pot = Wd.cpu(), pot.cpu()
    assert torch.isfinite(Wd).all()
    pair, worst_p, worst_l = 0, 0.0, 0.0
    for a in range(G):
        for b in range(a + 1):
            blk = (Mraw[off[a]:off[a + 1], off[b]:off[b + 1]] + b2)
            flip = sizes[b] < sizes[a]                        # rows <= cols (multi_graph_matching.py:518-522)
            x = blk.t() if flip else blk
            r, c = x.shape
            ref32 = og.sinkhorn_pair(x)
            ref64 = og.sinkhorn_pair(x.double())
            out = Wd[off[a]:off[a + 1], off[b]:off[b + 1]]
            out = out.t() if flip else out
            worst_p = max(worst_p, float((out - ref32).abs().max()))
            assert float((out - ref32).abs().max()) <= TOL, (a, b)
            if a != b:
                assert torch.equal(Wd[off[b]:off[b + 1], off[a]:off[a + 1]], Wd[off[a]:off[a + 1], off[b]:off[b + 1]].t())
            L2 = x.double() * (1.4426950408889634 / tau)
            f, gq = pot[pair, iters - 2, :r].double(), pot[pair, iters - 1, :c].double()
            y2 = L2 - f[:, None] - gq[None, :]
            live = ref64 > 1e-300
            d = float((y2[live] - ref64[live].log2()).abs().max())
            ulp = float(np.spacing(np.float32(float(L2.abs().max()))))
            worst_l = max(worst_l, d)
            assert d <= (1e-4 + 40 * ulp) * 1.4426950408889634, (a, b, d, ulp)
            pair += 1
    print("pair sinkhorn %s spike %s: max |Wds - oracle| %.2e, log2 domain %.2e" % (sizes, spike, worst_p, worst_l))


@pytest.mark.parametrize("sizes", [(9, 14), (22, 22, 22), (22, 35, 28, 40), (5, 3), (64, 64), (1, 1), (64, 1, 33), (40,),
                                   (30, 27, 33, 25, 38, 21)])
def test_fused_pair_stage_forward_backward(dev, sizes):
    """csrc/pair_stage.hip (round 3): affinity + pair Sinkhorn in ONE launch, block resident on the CU.  From P / Q rows to
    Wds against the oracle's formulation (M_ij = sum_k w2_k relu(P_ik + Q_jk) + b2, then multi_graph_matching.py:504-525),
    the affinity plane it leaves for the backward, the backward against autograd through the oracle (derived gate: float64
    statement of the same computation), and both halves against the round-2 two-launch kernels on the same inputs."""
    from oracle import gmodule as og
    from ttdg_mgm_amd import ops
    G, M, H = len(sizes), sum(sizes), 512
    off = np.concatenate([[0], np.cumsum(sizes)])
    g = synth.gen(sum(sizes) * 17 + G)
    P, Q = synth.normal(g, (M, H), 0.3), synth.normal(g, (M, H), 0.3)
    w2, b2 = synth.normal(g, (H,), 0.05), torch.tensor([0.03])
    Rw = synth.normal(g, (M, M), 1.0)
    mask, low = torch.zeros(M, M), torch.zeros(M, M)
    for a in range(G):
        for b in range(a + 1, G):
            mask[off[a]:off[a + 1], off[b]:off[b + 1]] = 1
            low[off[b]:off[b + 1], off[a]:off[a + 1]] = 1

    def host(dt):
        Pd, Qd, wd = P.to(dt), Q.to(dt), w2.to(dt)
        Mr = (torch.relu(Pd[:, None, :] + Qd[None, :, :]) * wd).sum(-1).detach().requires_grad_()        # without b2
        W = torch.zeros(M, M, dtype=dt)
        for a in range(G):
            for b in range(a + 1):
                blk = Mr[off[a]:off[a + 1], off[b]:off[b + 1]] + b2.to(dt)
                ds = og.sinkhorn_pair(blk) if sizes[b] >= sizes[a] else og.sinkhorn_pair(blk.t()).t()
                W[off[a]:off[a + 1], off[b]:off[b + 1]] += ds
                if a != b:
                    W[off[b]:off[b + 1], off[a]:off[a + 1]] += ds.t()
        (W * (Rw * mask).to(dt)).sum().backward()
        return Mr, W
    M32, W32 = host(torch.float32)
    M64, W64 = host(torch.float64)
    gr = ops.graphs(sizes)
    Pd, Qd, wd, bd = P.to(dev), Q.to(dev), w2.to(dev), b2.to(dev)
    aff, Wd, pot = ops.pair_stage_fwd(Pd, Qd, wd, bd, gr, list(sizes), 0.05, 20)
    tri = torch.zeros(M, M)
    for a in range(G):
        for b in range(a + 1):
            tri[off[a]:off[a + 1], off[b]:off[b + 1]] = 1
    scale = max(1.0, float(M64.detach().abs().max()))
    assert maxerr(aff[0].cpu() * tri, M64.detach().float() * tri) <= 1e-5 * scale
    assert maxerr(Wd, W64.float()) <= TOL
    print("fused pair stage %s: |Wds - fp64| %.2e (fp32 oracle %.2e)" % (sizes, maxerr(Wd, W64.float()), maxerr(W32, W64.float())))
    if G > 1:
        dM = ops.pair_stage_bwd(aff, bd, pot, (Rw * mask).to(dev), gr, 0.05, 20)
        derived_gate("fused pair stage bwd %s" % (sizes,), torch.where(low > 0, dM.cpu(), torch.zeros(())), M32.grad * low, M64.grad * low)
    # against the two-launch form on the same inputs (different summation order: ~1e-6)
    part = ops.affinity_pairwise_fwd(Pd, Qd, wd, gr, 2)
    W2, pot2 = ops.sinkhorn_pairs_fwd(part, bd, gr, list(sizes), 0.05, 20)
    assert maxerr(W2, Wd) <= 2e-5
    assert maxerr(part.sum(0).cpu() * tri, aff[0].cpu() * tri) <= 1e-5 * scale
    if G > 1:
        dM2 = ops.sinkhorn_pairs_bwd(aff, bd, pot, (Rw * mask).to(dev), gr, 0.05, 20)        # the old backward reads the new plane + log
        zero = torch.zeros(())                                  # (dM is only written where g(i) > g(j): mask by selection, not by 0 * garbage)
        assert maxerr(torch.where(low > 0, dM2.cpu(), zero), torch.where(low > 0, dM.cpu(), zero)) <= 1e-4 * max(1.0, float((M64.grad * low).abs().max()))
    with pytest.raises(RuntimeError):
        big = (70, 20)
        ops.pair_stage_fwd(torch.zeros(90, H, device=dev), torch.zeros(90, H, device=dev), wd, bd, ops.graphs(big), list(big), 0.05, 20)


# ------------------------------------------------------------------------------------------- A7
@pytest.mark.parametrize("ci", range(len(cases.HUNG_CASES)))
def test_hungarian_golden(dev, golden, ci):
    from ttdg_mgm_amd.GModule.utils.hungarian import hungarian
    gold = golden("hungarian")
    assert maxerr(hungarian(cases.hung_input(ci).to(dev)), gold[f"c{ci}_x"]) == 0


def test_hungarian_ties_and_errors(dev, golden):
    from ttdg_mgm_amd.GModule.utils.hungarian import hungarian
    gold = golden("hungarian")
    assert maxerr(hungarian(torch.from_numpy(gold["ties_s"]).to(dev)), gold["ties_x"]) == 0
    with pytest.raises(ValueError):
        hungarian(torch.zeros(3, device=dev))


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("shape", [(6, 6), (12, 32), (40, 32), (32, 95), (64, 64), (256, 32), (20, 33), (32, 129), (200, 32), (60, 250), (100, 128)])
def test_lap_batched_vs_scipy_incl_ties(dev, shape, variant):
    """All three settings of the register-resident solver (0: cost column in registers + compiler-lowered fp64 DPP minimum,
    1: hand-scheduled minimum, 2: costs read per step); other shapes take the same code under every setting."""
    import scipy.optimize
    from ttdg_mgm_amd import _lib, ops
    _lib.load().ttdg_debug_set_lap_variant(variant)
    mats = []
    for seed in range(24):
        g = synth.gen(9300 + seed)
        if seed % 2:
            m = g.integers(0, 3, size=shape).astype(np.float32)
            if seed % 3 == 0:
                m[:, g.integers(0, shape[1], size=max(1, shape[1] // 3))] = 0
        else:
            m = g.standard_normal(shape).astype(np.float32)
        mats.append(m)
    s = torch.from_numpy(np.stack(mats))
    try:
        x = ops.lap_batched(s.to(dev)).cpu().numpy()
    finally:
        _lib.load().ttdg_debug_set_lap_variant(0)
    for i, m in enumerate(mats):
        r, c = scipy.optimize.linear_sum_assignment(m.astype(np.float64) * -1)
        ref = np.zeros(shape, np.float32)
        ref[r, c] = 1
        assert np.array_equal(x[i], ref), i


# ------------------------------------------------------------------------------------------- A6
# Solver parity is stated three ways (DESIGN.md "Solver parity"):
#  (1) the MAP is the reference's: one iteration from any state of the oracle's own trajectory reproduces the
#      oracle's next state (Sinkhorn projector <= 1e-4, Hungarian projector identical) - no chaotic compounding;
#  (2) on planted (trained-like) inputs, where the reference converges and is rounding-stable, the full solve
#      returns IDENTICAL permutations and per-stage iteration counts;
#  (3) on random-weight inputs the reference itself is rounding-unstable (different permutations with 1 vs 8 CPU
#      threads): there we check every converged stage and the validity of the output.
def _pack(A, sizes):
    off, blocks = 0, []
    for n in sizes:
        blocks.append(A[off:off + n, off:off + n].reshape(-1))
        off += n
    return torch.cat(blocks)


def _oracle_trajectory(A, W, U0, sizes, max_keep=40):
    """Oracle states (projector, tau, U_t, U_{t+1}, V_t) along the reference's own schedule."""
    from oracle import gmodule as og
    G, ms = len(sizes), list(sizes)
    U, lastU, tau, proj, out = U0, torch.zeros_like(U0), 0.1, "sinkhorn", []
    while True:
        for i in range(200):
            lastU2, lastU = lastU, U
            V = (torch.linalg.multi_dot([A, U @ U.t(), A, U]) * 0.5 * 2 + W @ U) / G
            if proj == "hungarian":
                parts, s0 = [], 0
                for m in ms:
                    parts.append(og.hungarian(V[s0:s0 + m, :32]))
                    s0 += m
                Un = torch.cat(parts)
            else:
                Un = og._project_sinkhorn(V, ms, 32, tau, 20)
            if G == 2:
                Un[:ms[0]] = torch.eye(ms[0], 32)
            out.append((proj, tau, U, Un, V))
            U = Un
            if torch.norm(U - lastU) < 1e-3 or torch.norm(U - lastU2) == 0:
                break
        if proj == "hungarian":
            break
        elif tau > 1e-2:
            tau *= 0.5
        else:
            proj = "hungarian"
    step = max(1, len(out) // max_keep)
    return out[::step] + [o for o in out if o[0] == "hungarian"][:6]


# extra one-step cases (no golden file needed): a 32-node block inside an unequal batch whose largest graph exceeds the
# universe keeps rows = universe in the Sinkhorn projector (Appendix B steps 1-3); with max n_g <= 32 it does not
# ... and graphs above 128 nodes, which run on the multi-workgroup solver (csrc/gagm_large.hip)
ONE_STEP_EXTRA = (("uneq_with32", (32, 40, 27), 506), ("uneq_le32", (32, 20, 32), 507), ("eq32_and_big", (32, 32, 70), 508),
                  ("large_uneq", (130, 150, 20, 32), 509), ("large_g2", (150, 129), 510), ("large_eq", (160, 160, 160), 511))


@pytest.mark.parametrize("name,sizes,seed", cases.GAGM_CASES + ONE_STEP_EXTRA)
def test_gagm_one_step_map_along_oracle_trajectory(dev, name, sizes, seed):
    from oracle import gmodule as og
    from ttdg_mgm_amd import ops
    A, W, U0 = cases.gagm_inputs(sizes, seed)
    ap, Wd, gr = _pack(A, sizes).to(dev), W.to(dev), ops.graphs(sizes)
    nh, worst = 0, (0.0, 0.0)
    for proj, tau, Ut, Unext, V in _oracle_trajectory(A, W, U0, sizes):
        Ug, Vg = ops.gagm_one_step(ap, Wd, Ut.to(dev), gr, list(sizes), None if proj == "hungarian" else tau)
        scale = max(1.0, float(V.abs().max()))
        assert maxerr(Vg, V) <= TOL * scale
        if proj == "hungarian":
            # identical permutation unless the oracle's own LAP is decided below the fp32 resolution of V
            if not torch.equal(Ug.cpu(), Unext):
                off = 0
                for n in sizes:
                    v = V[off:off + n].double().numpy()
                    r1, c1 = np.nonzero(Unext[off:off + n].numpy())
                    r2, c2 = np.nonzero(Ug.cpu()[off:off + n].numpy())
                    assert abs(v[r1, c1].sum() - v[r2, c2].sum()) <= 1e-5 * scale, "LAP value gap"
                    off += n
            nh += 1
        else:
            # exact statement of the step on the same fp32 state: V and the projection in float64
            Ad, Wdd, Ud = A.double(), W.double(), Ut.double()
            V64 = (torch.linalg.multi_dot([Ad, Ud @ Ud.t(), Ad, Ud]) + Wdd @ Ud) / len(sizes)
            U64 = og._project_sinkhorn(V64, list(sizes), 32, tau, 20)
            if len(sizes) == 2:
                U64[:sizes[0]] = torch.eye(sizes[0], 32, dtype=torch.float64)
            worst = max(worst, derived_gate("%s one step tau %g" % (name, tau), Ug, Unext, U64, quiet=True))
    print("%s: Sinkhorn-projector steps, worst (device vs fp64 truth, fp32 oracle vs fp64 truth) = (%.3e, %.3e), gate 1e-4" % ((name,) + worst))
    assert nh >= 1


@pytest.mark.parametrize("name,sizes,seed", ALL_PLANTED)
def test_gagm_planted_identical_permutations(dev, golden, name, sizes, seed):
    """Full solve on the solver inputs of a planted case (A, Wds, U0 from the oracle's front end).  The pb_* cases cover
    every solver branch against the REFERENCE's permutations: n_g > 32 (transposed Sinkhorn orientation), 64 < n_g <= 128
    (two LAP columns per lane, LDS cost matrix), the multi-workgroup solver (a graph above 128 nodes; >= 320 nodes in
    total), G = 2 with n > 32."""
    from oracle import gmodule as og
    from ttdg_mgm_amd.GModule.multi_graph_matching import GA_GM
    gold = planted_gold(golden, name)
    params, nodes, labels, U, _ = cases.mgm_inputs(name)
    otr = {}
    og.mgm3_unsup_forward(params, nodes, labels, U, trace=otr)
    solver = GA_GM(mgm_iter=[200], cluster_iter=10, sk_iter=20, sk_tau0=[0.1], sk_gamma=0.5, cluster_beta=[1.0, 0.0],
                   converge_tol=1.0e-3, min_tau=[1.0e-2], projector0=['sinkhorn', 'sinkhorn'])
    Ug, cluster = solver(otr["A"].to(dev), otr["Wds"].to(dev), otr["U0"].to(dev), torch.tensor(sizes, dtype=torch.int), 32, 0.5, 1)
    info = solver.last_info.cpu().tolist()
    print(name, "iterations per stage: device", info[:6], "oracle", otr["iters"])
    assert tuple(cluster.tolist()) == (0,) * len(sizes)
    assert np.array_equal(Ug.cpu().numpy(), gold[f"{name}_U"]), "permutation matrices differ from the reference"
    LEDGER["planted_solver.identical_permutations_asserted"] += 1
    LEDGER["planted_solver.hungarian_count_exact"] += int(info[5] == otr["iters"][5])
    # Sinkhorn stages: identical iteration counts; Hungarian stage: the same fixed point, reached within one
    # iteration of the reference's count (a sub-resolution LAP tie can cost or save one round trip)
    assert info[:5] == otr["iters"][:5] and abs(info[5] - otr["iters"][5]) <= 1 and info[7] == 6


@pytest.mark.parametrize("name,sizes,seed", cases.GAGM_CASES)
def test_gagm_random_inputs_converged_stages_and_validity(dev, golden, name, sizes, seed):
    from oracle import gmodule as og
    from ttdg_mgm_amd import ops
    gold = golden("gagm")
    A, W, U0 = cases.gagm_inputs(sizes, seed)
    tr = {}
    og.gagm(A, W, U0.clone(), sizes, trace=tr)
    U, info, V0 = ops.gagm_solve(_pack(A, sizes).to(dev), W.to(dev), U0.to(dev), ops.graphs(sizes), list(sizes))
    it = info.cpu().tolist()
    print(name, "device", it[:7], "oracle", tr["iters"])
    assert maxerr(V0, gold[f"{name}_V0"]) <= TOL * max(1.0, float(np.abs(gold[f"{name}_V0"]).max()))
    k = 0
    while k < 6 and tr["iters"][k] < 200:   # identical counts up to the first stage the reference does not converge in
        assert it[k] == tr["iters"][k], (k, it[:6], tr["iters"])
        k += 1
    assert it[7] == 6
    Uc = U.cpu()
    assert set(np.unique(Uc.numpy())).issubset({0.0, 1.0})
    off = 0
    for n in sizes:       # every graph: a maximal partial permutation
        blk = Uc[off:off + n]
        assert float(blk.sum()) == min(n, 32) and float(blk.sum(0).max()) <= 1 and float(blk.sum(1).max()) <= 1
        off += n
    if len(sizes) == 2:
        assert torch.equal(Uc[:sizes[0]], torch.eye(sizes[0], 32))


@pytest.mark.parametrize("j", range(4))
def test_gagm_trained_regime_trajectory(dev, golden, j):
    """FREE-RUNNING solver parity on trained-regime inputs (A / Wds / U0 recorded from the synthetic checkpoint's own TTA
    steps, graphs of 21..38 nodes: both register-resident Sinkhorn projectors).  Through the first four stages of the
    schedule (tau 0.1 ... 0.0125; 15-50 iterations of the map) the trajectory is reproducible - float32 and float64 runs of
    the oracle agree to 1e-6 (tests/test_oracle_golden.py) - so the device must take the same number of iterations in every
    stage and land on the same state within the 1e-4 bar.  The last Sinkhorn stage (tau 0.00625) is chaotic under rounding for
    the reference itself; there the full solve is only required to return valid maximal partial permutations."""
    from oracle import gmodule as og
    from ttdg_mgm_amd import ops
    sizes, A, apack, W, U0 = cases.trained_solver_case(golden("trained_solver_inputs"), j)
    t64 = {}
    U64 = og.gagm(A.double(), W.double(), U0.double(), sizes, trace=t64, max_stages=4)
    U32 = og.gagm(A, W, U0, sizes, max_stages=4)
    cfg = ops.gagm_cfg(max_stages=4)
    U, info, _ = ops.gagm_solve(apack.to(dev), W.to(dev).contiguous(), U0.to(dev).contiguous(), ops.graphs(sizes), sizes, cfg)
    it = info.cpu().tolist()
    assert it[:4] == t64["iters"] and it[7] == 4, (it[:8], t64["iters"])
    derived_gate("trained-regime state after 4 stages %s" % (sizes,), U, U32, U64)
    Uf, info, _ = ops.gagm_solve(apack.to(dev), W.to(dev).contiguous(), U0.to(dev).contiguous(), ops.graphs(sizes), sizes)
    it = info.cpu().tolist()
    assert it[:4] == t64["iters"] and it[7] == 6
    Uc, off = Uf.cpu(), 0
    assert set(np.unique(Uc.numpy())).issubset({0.0, 1.0})
    for n in sizes:
        blk = Uc[off:off + n]
        assert float(blk.sum()) == min(n, 32) and float(blk.sum(0).max()) <= 1 and float(blk.sum(1).max()) <= 1
        off += n


def test_gagm_rejects_unknown_modes(dev):
    from ttdg_mgm_amd.GModule.multi_graph_matching import GA_GM
    with pytest.raises(NameError):
        GA_GM(projector0=('nope',))._cfg(1.0)
    with pytest.raises(NotImplementedError):
        GA_GM()(torch.zeros(2, 2), torch.zeros(2, 2), torch.zeros(2, 32), torch.tensor([1, 1]), 32, num_clusters=2)


# ------------------------------------------------------------------------------------------- A8/A9
@pytest.mark.parametrize("ci", range(3))
def test_perm_loss_kernel_vs_golden(dev, golden, ci):
    """loss.hip on a 2-graph problem: the a<b block of Wds is the golden score matrix (clamp edges included),
    the pseudo-label is U_a U_b^T for random one-hot assignments; checked against the oracle formula, which
    tests/test_oracle_golden.py pins to the reference's PermutationLoss."""
    from oracle import gmodule as og
    from ttdg_mgm_amd import ops
    gold = golden("loss")
    s = gold[f"c{ci}_s"]
    na, nb = s.shape
    g = synth.gen(40 + ci)
    Ua = torch.zeros(na, 32); Ub = torch.zeros(nb, 32)
    Ua[torch.arange(na), torch.from_numpy(g.integers(0, 32, na))] = 1
    Ub[torch.arange(nb), torch.from_numpy(g.integers(0, 32, nb))] = 1
    M = na + nb
    Wds = torch.zeros(M, M)
    Wds[:na, na:] = torch.from_numpy(s)
    sr = torch.from_numpy(s).clone().requires_grad_()
    ref = og.permutation_loss(sr.unsqueeze(0), (Ua @ Ub.t()).unsqueeze(0))
    ref.backward()
    loss, dW, flag = ops.perm_loss_fwd_bwd(Wds.to(dev), torch.cat([Ua, Ub]).to(dev), ops.graphs([na, nb]), 2)
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-6
    assert maxerr(dW[:na, na:], sr.grad) <= 1e-6 and int(flag.item()) == 0
    assert float(dW[na:, :].abs().max()) == 0 and float(dW[:na, :na].abs().max()) == 0
    bad = Wds.clone(); bad[0, na] = 1.5
    _, _, flag = ops.perm_loss_fwd_bwd(bad.to(dev), torch.cat([Ua, Ub]).to(dev), ops.graphs([na, nb]), 2)
    assert int(flag.item()) == 1


def _run_mgm3(dev, name, forced=None):
    from ttdg_mgm_amd.GModule import MGM3_unsup
    params, nodes, labels, U, sizes = cases.mgm_inputs(name)
    m = MGM3_unsup(2, 32).to(dev).eval()
    m.load_state_dict(params, strict=True)
    dn = [x.to(dev).requires_grad_() for x in nodes]
    tr = {}
    loss = m(dn, [l.to(dev) for l in labels], U.to(dev), trace=tr, forced_U=forced)
    loss.backward()
    return m, dn, loss, tr


def _check_against_gold(gold, name, m, dn, loss):
    assert abs(float(loss.detach()) - float(gold[f"{name}_loss"])) <= TOL
    for gi, x in enumerate(dn):
        assert maxerr(x.grad, gold[f"{name}_dnode{gi}"]) <= TOL, gi
    for k, p in m.named_parameters():
        if f"{name}_nograd_{k}" in gold:
            assert p.grad is None, k
        else:
            check_pgrad(gold, f"{name}_d_{k}", p.grad, TOL)


@pytest.mark.parametrize("name", [c[0] for c in ALL_PLANTED])
def test_mgm3_end_to_end_planted_golden(dev, golden, name):
    """Free-running: node features -> loss, gradients and the permutation matrices, all against the reference."""
    gold = planted_gold(golden, name)
    m, dn, loss, tr = _run_mgm3(dev, name)
    print(name, "gagm iterations", tr["info"].cpu().tolist()[:7], "loss", float(loss.detach()), "ref", float(gold[f"{name}_loss"]))
    assert np.array_equal(tr["Ub"].cpu().numpy(), gold[f"{name}_U"]), "permutation matrices differ from the reference"
    LEDGER["planted_e2e.identical_permutations_asserted"] += 1
    _check_against_gold(gold, name, m, dn, loss)


# ------------------------------------------------------------------------------------------- cfg-3 at full size, planted
@pytest.mark.parametrize("name", [c[0] for c in cases.PLANTED_CFG3_CASES])
def test_cfg3_planted_solver_identical_permutations(dev, golden, name):
    """BASELINE.json cfg-3 at FULL SIZE (8 graphs x 256 nodes) against the REFERENCE's own output (VERDICT r4 item 2;
    tests/golden/mgm3_cfg3.npz: multi_graph_matching.py:300-389 run on the planted case and admitted by
    tests/golden/admission.py with the reference's GA_GM).  Solver inputs (A, Wds, U0) from the oracle's front end, then the
    multi-workgroup solver free-running: first V, the permutation matrices, every Sinkhorn-stage iteration count and the
    Hungarian-stage count as the reference's print_helper reported them."""
    from oracle import gmodule as og
    from ttdg_mgm_amd import ops
    gold = golden("mgm3_cfg3")
    params, nodes, labels, U, sizes = cases.mgm_inputs(name)
    otr = {}
    og.mgm3_unsup_forward(params, nodes, labels, U, trace=otr)
    ref_iters = gold[f"{name}_iters"].tolist()
    assert otr["iters"] == ref_iters, ("the oracle does not walk the reference's trajectory", otr["iters"], ref_iters)
    assert maxerr(otr["U0"], gold[f"{name}_U0"]) <= 1e-6 * max(1.0, float(np.abs(gold[f"{name}_U0"]).max()))
    Ug, info, V0 = ops.gagm_solve(_pack(otr["A"], sizes).to(dev), otr["Wds"].to(dev), otr["U0"].to(dev), ops.graphs(sizes), list(sizes))
    it = info.cpu().tolist()
    print(name, "iterations per stage: device", it[:6], "reference", ref_iters, "| certified LAPs / fallbacks", it[12], it[13])
    assert maxerr(V0, gold[f"{name}_V0"]) <= TOL * max(1.0, float(np.abs(gold[f"{name}_V0"]).max()))
    assert np.array_equal(cases.perm_to_columns(Ug.cpu().numpy()), gold[f"{name}_U"]), "permutation matrices differ from the reference"
    assert it[:5] == ref_iters[:5] and abs(it[5] - ref_iters[5]) <= 1 and it[7] == 6
    LEDGER["cfg3_planted.identical_permutations_asserted"] += 1
    LEDGER["cfg3_planted.hungarian_count_exact"] += int(it[5] == ref_iters[5])


@pytest.mark.parametrize("name", [c[0] for c in cases.PLANTED_CFG3_CASES])
def test_cfg3_planted_end_to_end_golden(dev, golden, name):
    """The same case end to end and free-running through MGM3_unsup (multi_graph_matching.py:487-569): node features -> Wds
    (strided sample + norm), U0, the reference's permutation matrices, loss, node gradients and parameter gradients."""
    gold = golden("mgm3_cfg3")
    m, dn, loss, tr = _run_mgm3(dev, name)
    it = tr["info"].cpu().tolist()
    print(name, "gagm iterations", it[:7], "loss", float(loss.detach()), "ref", float(gold[f"{name}_loss"]))
    assert np.array_equal(cases.perm_to_columns(tr["Ub"].cpu().numpy()), gold[f"{name}_U"]), "permutation matrices differ from the reference"
    assert it[:5] == gold[f"{name}_iters"].tolist()[:5]
    W = tr["Wds"].detach().reshape(-1).cpu()
    assert maxerr(W[::cases.CFG3_WSTRIDE], gold[f"{name}_Wds__sample"]) <= TOL
    assert abs(float(W.double().norm()) - float(gold[f"{name}_Wds__norm"])) <= TOL * float(gold[f"{name}_Wds__norm"])
    assert maxerr(tr["U0"], gold[f"{name}_U0"]) <= TOL * max(1.0, float(np.abs(gold[f"{name}_U0"]).max()))
    assert abs(float(loss.detach()) - float(gold[f"{name}_loss"])) <= TOL
    for gi, x in enumerate(dn):
        check_pgrad(gold, f"{name}_dnode{gi}", x.grad, TOL)
    for k, p in m.named_parameters():
        if f"{name}_nograd_{k}" in gold:
            assert p.grad is None, k
        else:
            check_pgrad(gold, f"{name}_d_{k}", p.grad, TOL)
    LEDGER["cfg3_planted_e2e.identical_permutations_asserted"] += 1


def test_large_graph_linear_products_on_both_engines(dev, golden):
    """ops.LARGE_GEMM_ENGINE: the seventeen nn.Linear-shaped products of a matching step beyond 512 stacked nodes run on the streaming
    product of csrc/pointwise.hip ("mm": 64 x 64 tiles, fixed but not ascending k order, weight gradients over 8 row slices) or on
    gemm_f32 ("gemm": ascending k).  Same teacher-forced cfg-3 case through both: Wds, U0, loss, every node and parameter gradient
    agree to fp32 rounding of 256- / 512- / 2048-term sums, and both meet the reference golden at the suite's tolerance."""
    from ttdg_mgm_amd import ops
    name = cases.PLANTED_CFG3_CASES[0][0]
    gold = golden("mgm3_cfg3")
    forced = torch.from_numpy(cases.columns_to_perm(gold[f"{name}_U"])).to(dev) if hasattr(cases, "columns_to_perm") else None
    res = {}
    keep = ops.LARGE_GEMM_ENGINE
    for eng in ("mm_grouped", "mm", "gemm"):
        ops.LARGE_GEMM_ENGINE = eng
        try:
            m, dn, loss, tr = _run_mgm3(dev, name, forced=forced)
        finally:
            ops.LARGE_GEMM_ENGINE = keep
        res[eng] = (tr["Wds"].detach().clone(), tr["U0"].detach().clone(), float(loss.detach()), [x.grad.clone() for x in dn],
                    {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        assert abs(res[eng][2] - float(gold[f"{name}_loss"])) <= TOL
    # the grouped launches run the same tiles in the same order as the one-product launches: bit-identical
    ga = res["mm_grouped"]
    assert torch.equal(ga[0], res["mm"][0]) and torch.equal(ga[1], res["mm"][1]) and ga[2] == res["mm"][2]
    assert all(torch.equal(x, y) for x, y in zip(ga[3], res["mm"][3])) and all(torch.equal(ga[4][k], res["mm"][4][k]) for k in ga[4])
    a, b = res["mm"], res["gemm"]
    assert maxerr(a[0], b[0]) <= 1e-5 and maxerr(a[1], b[1]) <= 1e-5 * max(1.0, float(b[1].abs().max())) and abs(a[2] - b[2]) <= 1e-6
    # gradients: the two engines' 1e-6 differences in P / Q pass through the Sinkhorn backward at tau = 0.05 and the relu masks of the
    # affinity backward (measured between the engines: 2e-4 of the largest entry at 8 x 256); both meet the golden at TOL above / below
    for x, y in zip(a[3], b[3]):
        assert maxerr(x, y) <= 2e-3 * max(1e-30, float(y.abs().max()))
    assert set(a[4]) == set(b[4])
    for k in a[4]:
        # (+ 1e-7 absolute: the gradient of the affinity's output bias is a sum that cancels to ~1e-9 - a constant shift of every
        # block leaves the Sinkhorn output unchanged - so its two values are two roundings of zero)
        assert maxerr(a[4][k], b[4][k]) <= 2e-3 * float(b[4][k].abs().max()) + 1e-7, k
    for eng in ("mm_grouped", "mm", "gemm"):
        for gi, x in enumerate(res[eng][3]):
            check_pgrad(gold, f"{name}_dnode{gi}", x, TOL)


@pytest.mark.parametrize("name", [c[0] for c in cases.MGM_CASES])
def test_mgm3_end_to_end_random_teacher_forced(dev, golden, name):
    """Random-weight cases: the reference's own permutations are rounding noise there (it returns different ones
    with 1 vs 8 CPU threads), so loss/gradient parity is taken with the golden run's pseudo-labels supplied."""
    gold = golden("mgm3")
    m, dn, loss, tr = _run_mgm3(dev, name, forced=torch.from_numpy(gold[f"{name}_U"]).to(dev))
    _check_against_gold(gold, name, m, dn, loss)


@pytest.mark.parametrize("name", [c[0] for c in cases.MGM_CASES + ALL_PLANTED])
def test_mgm3_intermediates_vs_oracle(dev, name):
    from oracle import gmodule as og
    from ttdg_mgm_amd.GModule import MGM3_unsup
    params, nodes, labels, U, sizes = cases.mgm_inputs(name)
    otr = {}
    og.mgm3_unsup_forward(params, nodes, labels, U, trace=otr)
    m = MGM3_unsup(2, 32).to(dev).eval()
    m.load_state_dict(params, strict=True)
    tr = {}
    with torch.no_grad():
        m([x.to(dev) for x in nodes], [l.to(dev) for l in labels], U.to(dev), trace=tr)
    assert maxerr(tr["Wds"], otr["Wds"]) <= TOL
    assert maxerr(tr["U0"], otr["U0"]) <= TOL * max(1.0, float(otr["U0"].abs().max()))
    assert maxerr(tr["V0"], otr["V0"]) <= TOL * max(1.0, float(otr["V0"].abs().max()))
    assert maxerr(tr["apack"], _pack(otr["A"], sizes)) <= 1e-5
    if name in [c[0] for c in ALL_PLANTED]:
        assert torch.equal(tr["Ub"].cpu(), otr["Ub"])
        it = tr["info"].cpu().tolist()
        assert it[:5] == otr["iters"][:5] and abs(it[5] - otr["iters"][5]) <= 1


def test_mgm3_none_and_train_mode(dev):
    from ttdg_mgm_amd.GModule import MGM3_unsup
    m = MGM3_unsup(2, 32).to(dev)
    nodes, labels = synth.node_sets(1, (9,))
    U = synth.universe(2).to(dev)
    assert m([nodes[0].to(dev)], [labels[0].to(dev)], U) is None
    assert m(None, None, U) is None
    nodes, labels = synth.node_sets(3, (20, 24, 17), scale=0.5)
    m.load_state_dict(synth.mgm3_params(9), strict=True)
    m.train()                                     # dropout on the attention: stochastic but finite
    l1 = m([x.to(dev) for x in nodes], [l.to(dev) for l in labels], U)
    assert torch.isfinite(l1) and float(l1.detach()) > 0


# ------------------------------------------------------------------------------------------- A2
@pytest.mark.parametrize("ci", range(len(cases.PROTO_CASES)))
def test_prototype_computation_golden(dev, golden, ci):
    from oracle.ref_import import FakeInstances
    from ttdg_mgm_amd.GModule import PrototypeComputation
    gold = golden("proto")
    name, feats, boxes, classes = cases.proto_inputs(ci)
    pc = PrototypeComputation(2, 10)
    fd = [f.to(dev).requires_grad_() for f in feats]
    nodes, labels = pc(fd, [FakeInstances(b.to(dev), c.to(dev)) for b, c in zip(boxes, classes)])
    if f"{name}_none" in gold:
        assert nodes is None and labels is None
        return
    assert [len(n) for n in nodes] == gold[f"{name}_count"].tolist()
    for gi, (n, l) in enumerate(zip(nodes, labels)):
        assert maxerr(n, gold[f"{name}_nodes{gi}"]) == 0, gi
        assert np.array_equal(l.cpu().numpy(), gold[f"{name}_labels{gi}"]) and l.dtype == torch.int64
    # gather backward = scatter: gradient of sum(nodes * R) lands exactly on the selected points
    tot = sum((n * (gi + 1)).sum() for gi, n in enumerate(nodes))
    tot.backward()
    nsel = sum(len(n) for n in nodes)
    nz = sum(int((f.grad != 0).sum()) for f in fd)
    assert nz == nsel * 256


# ------------------------------------------------------------------------------------------- A11
def test_fused_sgd_vs_oracle(dev):
    from oracle import gmodule as og
    from ttdg_mgm_amd.optim import FusedSGD
    g = synth.gen(123)
    shapes = [(512, 512), (512,), (1, 512), (1,), (256, 256), (3, 3, 64, 64), (7,), (1000003,)]
    ps = [synth.normal(g, s) for s in shapes]
    ref_p = [p.clone() for p in ps]
    bufs = [None] * len(ps)
    dp = [torch.nn.Parameter(p.clone().to(dev)) for p in ps]
    opt = FusedSGD([{"params": [p], "weight_decay": (0.0 if i == 1 else 1e-4)} for i, p in enumerate(dp)], lr=0.005, momentum=0.9)
    for step in range(3):
        grads = [synth.normal(g, s) for s in shapes]
        if step == 1:
            grads[4] = None                     # tensors without a gradient are skipped
        for p, gr in zip(dp, grads):
            p.grad = None if gr is None else gr.to(dev)
        opt.step()
        # oracle step with per-tensor weight decay
        for i, (p, gr) in enumerate(zip(ref_p, grads)):
            b = [bufs[i]]
            og.sgd_step([p], [gr], b, 0.005, 0.9, 0.0 if i == 1 else 1e-4)
            bufs[i] = b[0]
    for p, r in zip(dp, ref_p):
        assert maxerr(p, r) <= 1e-6


def test_fused_sgd_follows_a_layout_change_of_the_parameter(dev):
    """the backbone moves its filters to channels-last in place on its first fp32 GPU forward (and ``load_state_dict``
    brings momentum buffers in the layout they were saved in).  A buffer made BEFORE such a change must be re-laid: the kernel is
    element-wise over storage order.  Two steps with a layout change of the parameter (and an NCHW gradient) in between, against
    the oracle's step on the logical values."""
    from oracle import gmodule as og
    from ttdg_mgm_amd.optim import FusedSGD
    g = synth.gen(321)
    shape = (24, 16, 3, 3)
    p0 = synth.normal(g, shape)
    ref_p, buf = [p0.clone()], [None]
    dp = torch.nn.Parameter(p0.clone().to(dev))
    opt = FusedSGD([dp], lr=0.005, momentum=0.9, weight_decay=1e-4)
    for step in range(3):
        gr = synth.normal(g, shape)
        if step == 1:                           # the parameter changes its storage order; the buffer of step 0 is NCHW
            dp.data = dp.data.contiguous(memory_format=torch.channels_last)
        if step == 2:                           # a buffer restored in the other layout (load_state_dict)
            st = opt.state[dp]
            st["momentum_buffer"] = st["momentum_buffer"].contiguous()
        dp.grad = gr.to(dev)                    # NCHW gradient for a channels-last parameter from step 1 on
        opt.step()
        og.sgd_step(ref_p, [gr], buf, 0.005, 0.9, 1e-4)
        assert dp.is_contiguous(memory_format=torch.channels_last) == (step >= 1)
        assert maxerr(dp, ref_p[0]) <= 1e-6, step
        assert maxerr(opt.state[dp]["momentum_buffer"], buf[0]) <= 1e-6, step


# ------------------------------------------------------------------------------------------- detection helpers (N1)
def test_nms_vs_oracle(dev):
    from oracle import detection as od
    from ttdg_mgm_amd import ops
    for seed, n in ((1, 1), (2, 77), (3, 2000), (4, 8400)):
        g = synth.gen(seed)
        xy = g.uniform(0, 700, size=(n, 2)); wh = g.uniform(5, 200, size=(n, 2))
        b = torch.from_numpy(np.concatenate([xy, xy + wh], 1).astype(np.float32))
        s = torch.from_numpy(g.permutation(n).astype(np.float32))          # distinct scores: unique order
        grp = torch.from_numpy(g.integers(0, 5, size=n).astype(np.int32))
        ref = od.nms(b, s, 0.7, grp)
        got = ops.nms(b.to(dev), s.to(dev), 0.7, grp.to(dev))
        assert torch.equal(got.cpu(), ref), (seed, n)
    assert ops.nms(torch.zeros(0, 4, device=dev), torch.zeros(0, device=dev), 0.5).numel() == 0


def test_roi_align_vs_oracle(dev):
    from oracle import detection as od
    from ttdg_mgm_amd import ops
    g = synth.gen(11)
    f = synth.normal(g, (2, 16, 50, 50))
    rois = torch.tensor([[0, 10., 12., 120., 90.], [1, 0., 0., 399., 399.], [1, 30., 30., 31., 31.], [0, -20., -30., 500., 460.],
                         [1, 200., 100., 260., 390.]])
    for P, scale in ((7, 0.125), (14, 0.125), (7, 0.03125)):
        assert maxerr(ops.roi_align(f.to(dev), rois.to(dev), scale, P), od.roi_align(f, rois, scale, P)) <= 1e-5


@pytest.mark.parametrize("name,sizes,seed", cases.GAGM_CASES)
def test_gagm_cycle_shortcut_is_exact(dev, name, sizes, seed):
    """The Hungarian-stage cycle shortcut must return exactly what running every iteration returns."""
    from ttdg_mgm_amd import ops
    A, W, U0 = cases.gagm_inputs(sizes, seed)
    args = (_pack(A, sizes).to(dev), W.to(dev), U0.to(dev), ops.graphs(sizes), list(sizes))
    U1, i1, _ = ops.gagm_solve(*args, ops.gagm_cfg())
    U2, i2, _ = ops.gagm_solve(*args, ops.gagm_cfg(no_cycle_skip=True))
    assert torch.equal(U1, U2)
    assert i1.cpu().tolist()[:8] == i2.cpu().tolist()[:8]


def test_gagm_cycle_shortcut_on_feature_derived_inputs(dev):
    """Same check on solver inputs that come from node features (random weights: the regime with long Hungarian cycles)."""
    from ttdg_mgm_amd import ops
    from ttdg_mgm_amd.GModule import MGM3_unsup
    hits = 0
    for seed, sizes in ((700, (22, 22, 22)), (704, (22, 35, 28, 40)), (711, (30, 27, 33, 25)), (705, (18, 25))):
        nodes, labels = synth.node_sets(seed, sizes, scale=0.5)
        m = MGM3_unsup(2, 32).to(dev).eval()
        m.load_state_dict(synth.mgm3_params(seed + 50))
        tr = {}
        with torch.no_grad():
            m([x.to(dev) for x in nodes], [l.to(dev) for l in labels], synth.universe(seed + 70).to(dev), trace=tr)
        U2, i2, _ = ops.gagm_solve(tr["apack"], tr["Wds"], tr["U0"], ops.graphs(sizes), list(sizes), ops.gagm_cfg(no_cycle_skip=True))
        assert torch.equal(tr["Ub"], U2)
        assert tr["info"].cpu().tolist()[:8] == i2.cpu().tolist()[:8]
        hits += int(i2.cpu().tolist()[5] == 200)
    assert hits >= 1      # at least one case actually exercised a capped Hungarian stage


# ------------------------------------------------------------------------------------------- cfg-3 scale (8 x 256 nodes)
@pytest.mark.parametrize("sizes,seed", [((256,) * 8, 600), ((200, 150, 31, 32, 140, 257), 601), ((130, 140), 602), ((129, 64, 300), 603),
                                        ((30,) * 12, 604), ((22, 35, 28, 33, 19, 32, 27, 31, 24, 34, 29, 26, 21, 30, 25, 23), 605)])
def test_large_solver_matches_host_driven_statement(dev, sizes, seed):
    """Graphs above 128 nodes, or many small graphs with >= 320 nodes in total (the gathered multi-graph of Mode S): the native multi-workgroup solver (two launches per iteration, stage machine on the device)
    against the host-driven statement of the same schedule on the stand-alone operators (one host decision per
    iteration).  (1) from every state of the host-driven trajectory one native iteration gives the same V and the same
    projection (Sinkhorn projector: both statements within 1e-4 of the float64 statement of the step, Hungarian identical or
    equal LAP value); (2) free-running: same
    iteration count for every stage up to the first one that hits the 200-iteration cap (a capped stage is the
    rounding-chaotic regime of DESIGN.md §4), and identical permutations when no stage is capped."""
    from ttdg_mgm_amd import ops
    A, W, U0 = cases.gagm_inputs(sizes, seed)
    ap, Wd, U0d, gr = _pack(A, sizes).to(dev), W.to(dev), U0.to(dev), ops.graphs(sizes)
    states = []
    cfg = ops.gagm_cfg()
    Uh, info_h, V0h = ops.gagm_solve_hostloop(ap, Wd, U0d, list(sizes), cfg, states=states)
    Un, info_n, V0n = ops.gagm_solve(ap, Wd, U0d, gr, list(sizes), cfg)
    assert maxerr(V0n, V0h) <= TOL * max(1.0, float(V0h.abs().max()))
    n_i, h_i = info_n.cpu().tolist(), info_h.cpu().tolist()
    capped = [k for k in range(6) if h_i[k] >= 200 or n_i[k] >= 200]
    upto = capped[0] if capped else 6
    # the Hungarian stage (index 5) may part ways at a near-tie of the LAP (V differs by summation order between the two
    # statements; the one-step check below accepts exactly that: identical assignment OR equal LAP value), so its count is
    # only compared when the two runs end on the same permutations
    same_end = torch.equal(Un, Uh)
    assert n_i[:min(upto, 5)] == h_i[:min(upto, 5)], (n_i, h_i)
    assert not same_end or upto < 6 or n_i[5] == h_i[5], (n_i, h_i)
    LEDGER["large_solver.cases"] += 1
    LEDGER["large_solver.sinkhorn_stage_counts_compared"] += min(upto, 5)
    LEDGER["large_solver.no_stage_capped"] += int(not capped)
    LEDGER["large_solver.identical_permutations"] += int(same_end)
    LEDGER["large_solver.hungarian_count_compared"] += int(same_end and upto == 6)
    Unc = Un.cpu()
    assert set(np.unique(Unc.numpy())).issubset({0.0, 1.0})
    off = 0
    for n in sizes:
        blk = Unc[off:off + n]
        assert float(blk.sum()) == min(n, 32) and float(blk.sum(0).max()) <= 1 and float(blk.sum(1).max()) <= 1
        off += n
    if not capped and n_i[5] == h_i[5]:
        assert same_end
        LEDGER["large_solver.identical_permutations_asserted"] += 1
    # one native iteration from sampled states of the host-driven trajectory
    step = max(1, len(states) // 12)
    picked = states[::step] + [st for st in states if st[0]][:4]
    from oracle import gmodule as og
    Ad, Wdd, worst = A.double(), W.double(), (0.0, 0.0)
    for hung, tau, Ub, Ua, V in picked:
        Ug, Vg = ops.gagm_one_step(ap, Wd, Ub.contiguous(), gr, list(sizes), None if hung else tau)
        scale = max(1.0, float(V.abs().max()))
        assert maxerr(Vg, V) <= TOL * scale
        if not hung:
            # exact (float64) statement of the step from the same fp32 state; native and host-driven results both within 1e-4
            Ud = Ub.double().cpu()
            B64 = Ad @ Ud
            V64 = (B64 @ (Ud.t() @ B64) + Wdd @ Ud) / len(sizes)
            U64 = og._project_sinkhorn(V64, list(sizes), 32, tau, 20)
            if len(sizes) == 2:
                U64[:sizes[0]] = torch.eye(sizes[0], 32, dtype=torch.float64)
            worst = max(worst, derived_gate("large one step tau %g" % tau, Ug, Ua, U64, quiet=True))
            assert maxerr(Ua, U64.float()) <= TOL
        elif not torch.equal(Ug, Ua):
            LEDGER["large_solver.hungarian_step_equal_value_only"] += 1
            o = 0
            for n in sizes:
                v = V[o:o + n].double().cpu().numpy()
                r1, c1 = np.nonzero(Ua[o:o + n].cpu().numpy())
                r2, c2 = np.nonzero(Ug[o:o + n].cpu().numpy())
                assert abs(v[r1, c1].sum() - v[r2, c2].sum()) <= 1e-5 * scale, "LAP value gap"
                o += n
        else:
            LEDGER["large_solver.hungarian_step_identical"] += 1
    print(sizes, "Sinkhorn-projector steps: worst (native, host-driven) deviation from the fp64 statement = (%.3e, %.3e)" % worst)


@pytest.mark.parametrize("sizes,seed", [((256,) * 8, 610), ((129, 64, 300, 33), 611), ((512, 40, 65), 612), ((520, 100), 613)])
def test_large_solver_sinkhorn_projectors_agree_with_the_float64_step(dev, sizes, seed):
    """[r4] The multi-workgroup solver's Sinkhorn projector is now the block-layout one over all wavefronts of the workgroup
    (gagm_large.hip: gl_project_blk - one exponential per entry per sweep pair, row sums met in LDS once per pair); round 3's
    column-per-thread projector stays behind cfg.variant = TTDG_GAGM_COLUMN_PROJECTOR (and serves graphs of more than 512 nodes).
    One solver step from U0 and from a sharpened state at every temperature of the schedule: BOTH within the derived gate of the
    float64 statement of the step, every block size class (CB = 2, 4, 8; ragged last block; a < 33-node graph beside large ones;
    the 1024-thread build)."""
    from oracle import gmodule as og
    from ttdg_mgm_amd import _lib, ops
    A, W, U0 = cases.gagm_inputs(sizes, seed)
    ap, Wd, gr = _pack(A, sizes).to(dev), W.to(dev), ops.graphs(sizes)
    Ad, Wdd = A.double(), W.double()
    state = U0
    for tau in (0.1, 0.05, 0.025, 0.0125, 0.00625):
        Ud = state.double()
        B64 = Ad @ Ud
        V64 = (B64 @ (Ud.t() @ B64) + Wdd @ Ud) / len(sizes)
        U64 = og._project_sinkhorn(V64, list(sizes), 32, tau, 20)
        if len(sizes) == 2:
            U64[:sizes[0]] = torch.eye(sizes[0], 32, dtype=torch.float64)
        # the float32 oracle on the same state: what the reference's own fp32 path loses against the float64 statement
        B32 = A @ state
        U32 = og._project_sinkhorn((B32 @ (state.t() @ B32) + W @ state) / len(sizes), list(sizes), 32, tau, 20)
        if len(sizes) == 2:
            U32[:sizes[0]] = torch.eye(sizes[0], 32)
        for name, var in (("block", 0), ("column", _lib.GAGM_COLUMN_PROJECTOR)):
            Ug, Vg = ops.gagm_one_step(ap, Wd, state.to(dev).contiguous(), gr, list(sizes), tau, variant=var | _lib.GAGM_FORCE_LARGE)
            assert maxerr(Vg, V64.float()) <= TOL * max(1.0, float(V64.abs().max()))
            derived_gate("large solver, %s projector, tau %g" % (name, tau), Ug, U32, U64, quiet=True)
        state = U64.float()                       # the next temperature starts from the sharpened state


def _scipy_projection(V, sizes):
    """utils/hungarian.py:8-66 on every graph block of V: scipy.optimize.linear_sum_assignment on the negated block."""
    from scipy.optimize import linear_sum_assignment
    out, o = torch.zeros_like(V), 0
    for n in sizes:
        r, c = linear_sum_assignment(V[o:o + n].numpy() * -1)
        out[o + torch.from_numpy(r), torch.from_numpy(c)] = 1.0
        o += n
    return out


@pytest.mark.parametrize("sizes,seed", [((256,) * 8, 620), ((129, 64, 300, 33, 40), 621), ((512, 40, 65), 622), ((520, 100, 257), 623)])
def test_large_solver_certified_lap_equals_scipy_and_the_scipy_order_solver(dev, sizes, seed):
    """[r4] Hungarian stage of the multi-workgroup solver: warm-started workgroup LAP with a uniqueness certificate
    (csrc/lap_certified.h), the one-wavefront scipy-order solver only where the certificate fails (info[13]).
    (a) one step from U0: the projection IS scipy's on the device's own V (reference utils/hungarian.py:63), both LAP paths;
    (b) a whole Hungarian stage entered directly, cycle shortcut off (every iteration a LAP, the duals carried over): identical
        permutations and iteration counts with cfg.variant = TTDG_GAGM_SCIPY_ORDER_LAP; the certified count is reported;
    (c) exact ties (every node of a graph identical: uniform adjacency, W = 0): no certificate can exist - every LAP must fall
        back, and the answer is still the scipy-order one."""
    from ttdg_mgm_amd import _lib, ops
    A, W, U0 = cases.gagm_inputs(sizes, seed)
    ap, Wd, gr = _pack(A, sizes).to(dev), W.to(dev), ops.graphs(sizes)
    U0d = U0.to(dev).contiguous()
    big = sum(1 for n in sizes if 32 < n <= 512)
    # (a)
    for var in (0, _lib.GAGM_SCIPY_ORDER_LAP):
        Ug, Vg = ops.gagm_one_step(ap, Wd, U0d, gr, list(sizes), None, variant=var | _lib.GAGM_FORCE_LARGE)
        assert torch.equal(Ug.cpu(), _scipy_projection(Vg.cpu(), sizes)), var
    # (b)
    res = {}
    for var in (0, _lib.GAGM_SCIPY_ORDER_LAP):
        cfg = ops.gagm_cfg(start_hungarian=True, max_stages=1, no_cycle_skip=True, max_iter=24, variant=var | _lib.GAGM_FORCE_LARGE)
        Ub, info, _ = ops.gagm_solve(ap, Wd, U0d, gr, list(sizes), cfg)
        res[var] = (Ub.cpu(), info.cpu().tolist())
    (Uf, inf_f), (Us, inf_s) = res[0], res[_lib.GAGM_SCIPY_ORDER_LAP]
    assert torch.equal(Uf, Us) and inf_f[:8] == inf_s[:8], (inf_f, inf_s)
    assert inf_s[12] == 0 and inf_s[13] == 0
    assert inf_f[12] + inf_f[13] + inf_f[21] == big * inf_f[6], (inf_f, big)
    assert inf_f[12] >= 0.7 * big * inf_f[6], inf_f                   # generic inputs: the optimum is unique (the rest: attempts given up after the pricing step)
    print(sizes, "Hungarian stage: %d iterations, %d certified LAPs, %d fallbacks" % (inf_f[6], inf_f[12], inf_f[13]))
    # (c)
    At = torch.zeros_like(A)
    o = 0
    for n in sizes:
        At[o:o + n, o:o + n] = 1.0 / n
        o += n
    apt, Wt = _pack(At, sizes).to(dev), torch.zeros_like(Wd)
    res = {}
    for var in (0, _lib.GAGM_SCIPY_ORDER_LAP):
        cfg = ops.gagm_cfg(start_hungarian=True, max_stages=1, no_cycle_skip=True, max_iter=3, variant=var | _lib.GAGM_FORCE_LARGE)
        Ub, info, V0 = ops.gagm_solve(apt, Wt, U0d, gr, list(sizes), cfg)
        res[var] = (Ub.cpu(), info.cpu().tolist(), V0.cpu())
    (Uf, inf_f, V0f), (Us, inf_s, _) = res[0], res[_lib.GAGM_SCIPY_ORDER_LAP]
    o = 0
    for n in sizes:
        assert float((V0f[o:o + n] - V0f[o:o + 1]).abs().max()) == 0.0, "the tie construction needs bit-identical rows of V"
        o += n
    assert torch.equal(Uf, Us) and inf_f[:8] == inf_s[:8], (inf_f, inf_s)
    assert inf_f[12] == 0 and inf_f[13] + inf_f[21] == big * inf_f[6] and inf_f[21] >= big, inf_f      # [r5] constant blocks: the integer solver
    U1 = ops.gagm_one_step(apt, Wt, U0d, gr, list(sizes), None, variant=_lib.GAGM_FORCE_LARGE)[0]
    assert torch.equal(U1.cpu(), _scipy_projection(V0f, sizes)), "all-ties block: scipy's own tie rules decide"


@pytest.mark.parametrize("sizes,seed", [((256,) * 8, 640), ((129, 64, 300, 33, 40), 641), ((512, 40, 65), 642), ((520, 100, 257), 643)])
def test_large_solver_integer_lap_on_narrow_range_blocks_equals_scipy(dev, sizes, seed):
    """[r5] The block behind a collapsed Sinkhorn stage: U ~ 1 / n, V agrees along the universe index to an ulp and across the nodes
    to a few thousand ulp; no uniqueness certificate exists and scipy's tie rules decide over 528 Dijkstra steps per graph (1.2 M
    cycles in the fp64 step-by-step solver: a quarter of the BASELINE cfg-3 solve).  Blocks of a narrow value range take the INTEGER
    statement of the scipy-order solver (csrc/lap_device.h: lap_wave_solve_int; exact because scipy's own float64 arithmetic is
    exact on such a block - argument in the header).  Two constructions, a Hungarian iteration entered from each:
      U0 exactly uniform                       -> V bit-constant along the universe index, distinct node values;
      U0 uniform x (1 + 1e-4 noise) per entry   -> the real thing: near-ties everywhere, rounding decides;
    on both the projection IS scipy.optimize.linear_sum_assignment on the device's own V (reference utils/hungarian.py:63), it is
    identical to the fp64 step-by-step solver's (cfg.variant = TTDG_GAGM_NO_INT_LAP), and info[21] counts one integer solve per
    graph of 33 .. 512 nodes (a 520-node graph keeps the one-wavefront fp64 solver)."""
    from ttdg_mgm_amd import _lib, ops
    g = torch.Generator().manual_seed(seed)
    M = sum(sizes)
    # near-uniform adjacency blocks and affinities (relative noise 1e-4): the node values of V agree to ~1e-5, i.e. to a few hundred ulp
    A = torch.zeros(M, M)
    o = 0
    for n in sizes:
        A[o:o + n, o:o + n] = (1.0 / n) * (1.0 + 1e-4 * torch.randn(n, n, generator=g))
        o += n
    A.fill_diagonal_(0.0)
    W = 0.3 * (1.0 + 1e-4 * torch.randn(M, M, generator=g))
    W = 0.5 * (W + W.t())
    ap, Wd, gr = _pack(A, sizes).to(dev), W.to(dev).contiguous(), ops.graphs(sizes)
    big = sum(1 for n in sizes if 32 < n <= 512)
    for noise in (0.0, 1e-4):
        U0 = torch.cat([torch.full((n, 32), 1.0 / n) * (1.0 + noise * torch.randn(n, 32, generator=g)) for n in sizes]).to(dev).contiguous()
        res = {}
        for var in (0, _lib.GAGM_NO_INT_LAP):
            cfg = ops.gagm_cfg(start_hungarian=True, max_stages=1, no_cycle_skip=True, max_iter=1, variant=var | _lib.GAGM_FORCE_LARGE)
            Ub, info, V0 = ops.gagm_solve(ap, Wd, U0, gr, list(sizes), cfg)
            res[var] = (Ub.cpu(), info.cpu().tolist(), V0.cpu())
        (Uf, inf_f, V0f), (Us, inf_s, V0s) = res[0], res[_lib.GAGM_NO_INT_LAP]
        assert torch.equal(V0f, V0s)
        if noise == 0.0:
            assert float((V0f - V0f[:, :1]).abs().max()) == 0.0, "V constant along the universe index, bit for bit"
        assert torch.equal(Uf, Us) and inf_f[:8] == inf_s[:8], (noise, inf_f, inf_s)
        assert torch.equal(Uf, _scipy_projection(V0f, sizes)), noise
        assert inf_f[21] == big and inf_f[12] == 0 and inf_f[13] == 0, (noise, inf_f)
        assert inf_s[21] == 0 and inf_s[13] == big, (noise, inf_s)
        LEDGER["large_solver.integer_lap_blocks_equal_to_scipy"] += inf_f[21]


@pytest.mark.parametrize("sizes,seed", [((256,) * 8, 630), ((129, 64, 300, 33, 40), 631), ((520, 100, 257), 632), ((200, 150), 633),
                                        ((20, 33, 40), 634), ((96,) * 12, 635)])
def test_large_solver_one_cooperative_launch_equals_the_two_launch_form(dev, sizes, seed):
    """[r4] cfg.variant = TTDG_GAGM_ONE_LAUNCH runs the multi-workgroup solver's whole schedule in ONE cooperative launch
    (gagm_large_persistent_kernel: device-side iteration loop, grid barrier between the mul and the projection phase, no host
    synchronisation); the default stays round 2's form (two launches per iteration, enqueued in chunks with a host read per chunk -
    measured faster, see the kernel's header).  Same arithmetic in the same order: the
    permutations, every iteration count, the first-iteration V and the first projected U must be IDENTICAL BITS - on the full
    schedule, on a schedule stopped after the Sinkhorn stages (fractional U), and on a Hungarian stage entered directly.
    Covers the 512- and the 1024-thread build, G = 2 (identity pin), graphs below 33 nodes beside large ones, more graphs than
    K slices, and a workgroup count below / above the number of (row tile, K slice) items."""
    from ttdg_mgm_amd import _lib, ops
    A, W, U0 = cases.gagm_inputs(sizes, seed)
    ap, Wd, gr = _pack(A, sizes).to(dev), W.to(dev), ops.graphs(sizes)
    U0d = U0.to(dev).contiguous()
    for kw in (dict(max_iter=30), dict(max_stages=3), dict(start_hungarian=True, max_stages=1, max_iter=12, no_cycle_skip=True),
               dict(start_hungarian=True, max_stages=1, max_iter=40)):
        out = []
        for var in (_lib.GAGM_ONE_LAUNCH, 0):
            Ub, info, V0 = ops.gagm_solve(ap, Wd, U0d, gr, list(sizes), ops.gagm_cfg(variant=var | _lib.GAGM_FORCE_LARGE, **kw))
            out.append((Ub.cpu(), info.cpu().tolist(), V0.cpu().clone(), ops.gagm_solve.last_U1.cpu().clone()))
        (Ua, ia, Va, U1a), (Ub_, ib, Vb, U1b) = out
        assert ia[8] == 0 and ib[8] == 0, (ia, ib)
        assert ia[:8] == ib[:8] and ia[12:16] == ib[12:16], (kw, ia, ib)
        assert torch.equal(Va, Vb) and torch.equal(U1a, U1b) and torch.equal(Ua, Ub_), kw
        # [r6] info[22]: iterations EXECUTED (launch pairs that did work) - the count itself unless a Hungarian-stage cycle jump skipped the rest
        assert ia[22] == ib[22] and 0 < ib[22] <= ib[6], (kw, ia, ib)
        assert ib[22] == ib[6] or ib[14] >= 3, (kw, ib)


def test_large_solver_launch_hint_does_not_touch_results(dev):
    """[r6] The two-launch form sizes its first chunk of enqueued iterations from the iterations the PREVIOUS solve of the host thread
    executed (csrc/gagm_large.hip: last_exec; follow-up chunks of 4, 8, 16, 32).  Whatever the hint - after a long solve, after a short
    one, first solve of a fresh size - U, the first-iteration V and every info word are the same bits."""
    from ttdg_mgm_amd import _lib, ops
    probs = []
    for sizes, seed, kw in (((256,) * 4, 640, dict()), ((130, 140), 641, dict(max_stages=1, max_iter=3)), ((200, 150, 129), 642, dict(max_iter=60)),
                            ((256,) * 4, 640, dict(start_hungarian=True, max_stages=1, max_iter=12, no_cycle_skip=True))):
        A, W, U0 = cases.gagm_inputs(sizes, seed)
        probs.append((_pack(A, sizes).to(dev), W.to(dev), U0.to(dev).contiguous(), ops.graphs(sizes), list(sizes), kw))
    ref = {}
    for order in ((0, 1, 2, 3), (3, 2, 1, 0), (1, 1, 0, 2, 3, 0, 3, 1)):
        for k in order:
            ap, Wd, U0d, gr, sizes, kw = probs[k]
            Ub, info, V0 = ops.gagm_solve(ap, Wd, U0d, gr, sizes, ops.gagm_cfg(variant=_lib.GAGM_FORCE_LARGE, **kw))
            got = (Ub.cpu(), info.cpu().tolist(), V0.cpu().clone())
            if k not in ref:
                ref[k] = got
                assert got[1][22] > 0
            else:
                assert torch.equal(got[0], ref[k][0]) and got[1] == ref[k][1] and torch.equal(got[2], ref[k][2]), (order, k, got[1], ref[k][1])


def test_cfg3_scale_front_end_and_large_solver(dev):
    """BASELINE cfg-3 operator shapes: 8 graphs x 256 nodes.  Wds / A / U0 / V0 against the oracle on a 3-graph
    slice the CPU finishes in seconds, then the full 8 x 256 forward+backward through the large-graph solver with
    size-independent properties (doubly-stochastic blocks, symmetric mirror, valid partial permutations, finite grads)."""
    from oracle import gmodule as og
    from ttdg_mgm_amd.GModule import MGM3_unsup
    params = synth.mgm3_params(3003)
    U = synth.universe(3004)
    m = MGM3_unsup(2, 32).to(dev).eval()
    m.load_state_dict(params, strict=True)
    nodes, labels = synth.node_sets(3000, (256, 256, 256), scale=0.1)
    otr = {}
    og.mgm3_unsup_forward(params, nodes, labels, U, trace=otr)
    tr = {}
    with torch.no_grad():
        m([x.to(dev) for x in nodes], [l.to(dev) for l in labels], U.to(dev), trace=tr)
    assert maxerr(tr["Wds"], otr["Wds"]) <= TOL
    assert maxerr(tr["V0"], otr["V0"]) <= TOL * max(1.0, float(otr["V0"].abs().max()))
    assert maxerr(tr["apack"], _pack(otr["A"], (256, 256, 256))) <= 1e-5
    # full cfg-3 size, forward + backward
    sizes = (256,) * 8
    nodes, labels = synth.node_sets(3001, sizes, scale=0.1)
    dn = [x.to(dev).requires_grad_() for x in nodes]
    tr = {}
    loss = m(dn, [l.to(dev) for l in labels], U.to(dev), trace=tr)
    loss.backward()
    W = tr["Wds"]
    assert float(W.min()) >= 0 and float(W.max()) <= 1 + 1e-6
    # (diagonal blocks are Sinkhorn outputs of a graph against itself and need not be symmetric; the off-diagonal mirrors are)
    for a in range(8):
        for b in range(a):
            blk = W[a * 256:(a + 1) * 256, b * 256:(b + 1) * 256]
            assert torch.equal(blk, W[b * 256:(b + 1) * 256, a * 256:(a + 1) * 256].t())
            assert maxerr(blk.sum(0), torch.ones(256)) <= 1e-3        # 20 sweeps end on a column normalisation
    Ub = tr["Ub"].cpu()
    assert set(np.unique(Ub.numpy())).issubset({0.0, 1.0})
    for g in range(8):
        blk = Ub[g * 256:(g + 1) * 256]
        assert float(blk.sum()) == 32 and float(blk.sum(0).max()) <= 1 and float(blk.sum(1).max()) <= 1
    assert torch.isfinite(loss) and all(torch.isfinite(x.grad).all() for x in dn)
    assert all(torch.isfinite(p.grad).all() for k, p in m.named_parameters() if k.startswith("node_affinity"))


def test_cfg3_full_size_forward_backward_against_the_oracle(dev):
    """BASELINE cfg-3 at FULL size - 8 graphs x 256 nodes - against the oracle itself, not only through
    size-independent properties: every block of Wds (36 pairs incl. the diagonal), A, U0, the first V, the loss and ALL
    gradients (d nodes, the six affinity tensors), with the device's own pseudo-labels supplied to both sides (the free-running
    solve of random weights is rounding noise in the reference too).  The oracle's formulation is the reference's: materialised
    (256, 256, 1024) affinity MLP per pair and a per-pair Sinkhorn; pairs are differentiated one at a time (the reference keeps
    two 256 x 256 x 512 tensors per pair alive: 9.7 GB for the 36 pairs)."""
    import itertools
    from oracle import gmodule as og
    from ttdg_mgm_amd.GModule import MGM3_unsup
    import os
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(64, os.cpu_count() or 1))          # beyond 64 intra-op threads torch's CPU kernels stop scaling here
    G, n = 8, 256
    sizes = (n,) * G
    off = [g * n for g in range(G + 1)]
    M = G * n
    params = synth.mgm3_params(3003)
    U = synth.universe(3004)
    nodes, labels = synth.node_sets(3001, sizes, scale=0.1)
    m = MGM3_unsup(2, 32).to(dev).eval()
    m.load_state_dict(params, strict=True)
    dn = [x.to(dev).requires_grad_() for x in nodes]
    tr = {}
    loss = m(dn, [l.to(dev) for l in labels], U.to(dev), trace=tr)
    loss.backward()
    Ub = tr["Ub"].cpu()
    # ---- oracle, pair by pair
    p = {k: v.clone().requires_grad_() for k, v in params.items()}
    rn = [x.clone().requires_grad_() for x in nodes]
    Wds = torch.zeros(M, M)
    npairs = G * (G - 1) // 2
    ref_loss = 0.0
    for a in range(G):
        for b in range(a + 1):
            if a == b:
                with torch.no_grad():
                    Wds[off[a]:off[a + 1], off[a]:off[a + 1]] = og.sinkhorn_pair(og.affinity(p, rn[a], rn[a]))
                continue
            ds = og.sinkhorn_pair(og.affinity(p, rn[a], rn[b]))             # equal sizes: no transposition (:518-522)
            Wds[off[a]:off[a + 1], off[b]:off[b + 1]] = ds.detach()
            Wds[off[b]:off[b + 1], off[a]:off[a + 1]] = ds.detach().t()
            # the loss reads Wds[i-rows, j-cols] for i < j: i = b, j = a -> ds^T, pseudo-label U_i U_j^T (:615-631)
            lp = og.permutation_loss(ds.t().unsqueeze(0), (Ub[off[b]:off[b + 1]] @ Ub[off[a]:off[a + 1]].t()).unsqueeze(0)) / npairs
            lp.backward()
            ref_loss += float(lp.detach())
    A = torch.zeros(M, M)
    with torch.no_grad():
        for g in range(G):
            A[off[g]:off[g + 1], off[g]:off[g + 1]] = og.mha_adjacency(p, rn[g])
        A.fill_diagonal_(0)
        U0 = torch.cat([x @ U.t() for x in rn])
        V0 = (torch.linalg.multi_dot([A, U0 @ U0.t(), A, U0]) + Wds @ U0) / G
    print("cfg-3 full size: |Wds| %.2e  |A| %.2e  |U0| rel %.2e  |V0| rel %.2e  loss %.6f vs %.6f" % (
        maxerr(tr["Wds"], Wds), maxerr(tr["apack"], _pack(A, sizes)), maxerr(tr["U0"], U0) / max(1.0, float(U0.abs().max())),
        maxerr(tr["V0"], V0) / max(1.0, float(V0.abs().max())), float(loss.detach()), ref_loss))
    assert maxerr(tr["Wds"], Wds) <= TOL
    assert maxerr(tr["apack"], _pack(A, sizes)) <= 1e-5
    assert maxerr(tr["U0"], U0) <= TOL * max(1.0, float(U0.abs().max()))
    assert maxerr(tr["V0"], V0) <= TOL * max(1.0, float(V0.abs().max()))
    assert abs(float(loss.detach()) - ref_loss) <= TOL
    for a, b in zip(dn, rn):
        assert maxerr(a.grad, b.grad) <= TOL * max(1.0, float(b.grad.abs().max()))
    for k, q in m.named_parameters():
        if k.startswith("node_affinity."):
            g_ref = p[k].grad
            assert maxerr(q.grad, g_ref) <= TOL * max(1.0, float(g_ref.abs().max())), k
    torch.set_num_threads(nthreads)


# ------------------------------------------------------------------------------------------- N1: fused box pipelines
def _cpu_backend():
    from oracle import tta_cpu
    return tta_cpu._CpuBackend


def test_rpn_decode_and_batched_selection_match_host_formulation(dev):
    """ttdg_rpn_decode + the batch-level (image, level)-grouped NMS / top-k against the detectron2 formulation in plain
    torch (oracle.tta_cpu._CpuBackend): same boxes, same validity, same kept proposals in the same order."""
    from ttdg_mgm_amd import ops
    cb = _cpu_backend()
    g = synth.gen(7100)
    B, A = 3, 3
    shapes, strides, pre, post = [(24, 24), (12, 12), (6, 6)], [4, 8, 16], 300, 200
    sizes = [(96, 96), (90, 96), (96, 80)]
    anchors, logits, deltas = [], [], []
    for (h, w), s in zip(shapes, strides):
        ys, xs = np.meshgrid(np.arange(h) * s, np.arange(w) * s, indexing="ij")
        base = np.array([[-8, -4, 8, 4], [-6, -6, 6, 6], [-4, -8, 4, 8]], np.float32) * (s / 4)
        an = (np.stack((xs, ys, xs, ys), -1).reshape(-1, 1, 4) + base[None]).reshape(-1, 4).astype(np.float32)
        anchors.append(torch.from_numpy(an))
        logits.append(synth.normal(g, (B, A, h, w), 1.0))
        d = synth.normal(g, (B, A * 4, h, w), 0.5)
        d[0, 0, 0, 0] = float("nan")                        # a non-finite candidate must be dropped, not propagated
        deltas.append(d)
    ks = [min(pre, A * h * w) for h, w in shapes]
    K = sum(ks)
    lvl = torch.cat([torch.full((k,), l, dtype=torch.int64) for l, k in enumerate(ks)])
    res = {}
    for name, be, dv in (("cpu", cb, torch.device("cpu")), ("gpu", ops, dev)):
        st = be.image_sizes_tensor(sizes, dv)
        boxes, scores = torch.empty(B, K, 4, device=dv), torch.empty(B, K, device=dv)
        col = 0
        for lg, dl, an, k in zip(logits, deltas, anchors, ks):
            sc, idx = lg.to(dv).permute(0, 2, 3, 1).reshape(B, -1).topk(k, dim=1)
            be.rpn_decode(dl.to(dv), an.to(dv), idx, sc, st, boxes, scores, col)
            col += k
        keep, counts = be.nms_batched(boxes, scores, lvl.to(dv), len(shapes), 0.7, pre, post)
        res[name] = (boxes.cpu(), scores.cpu(), keep.cpu(), counts)
        if name == "gpu":
            # the RPN's own call: per-level blocks already in descending score order -> the sort-free sweep, same selection
            keep2, counts2 = be.nms_batched(boxes, scores, lvl.to(dv), len(shapes), 0.7, pre, post, level_sizes=ks)
            assert counts2 == counts
            for b in range(B):
                assert torch.equal(keep2[b, :counts2[b]], keep[b, :counts[b]]), b
    (bc, sc_, kc, cc), (bg, sg, kg, cg) = res["cpu"], res["gpu"]
    live = sc_ > float("-inf")
    assert torch.equal(live, sg > float("-inf")) and int((~live).sum()) >= 1
    assert maxerr(bg[live], bc[live]) <= 1e-3 and torch.equal(sg[live], sc_[live])
    assert cc == cg
    for b in range(B):
        assert torch.equal(kc[b, :cc[b]], kg[b, :cg[b]]), b


def test_fused_rpn_selection_equals_per_level_topk_and_decode(dev):
    """ttdg_rpn_select (one launch: exact radix select + LDS sort + decode for every (image, level)) against the path it
    replaces - per level permute + torch.topk + ttdg_rpn_decode: scores bit-identical in every slot, boxes bit-identical, the
    selected SET identical; and on the CPU host backend's formulation.  Cases: the bench's level shapes scaled down and one
    level larger than a workgroup's pass (3 x 120 x 100 = 36000 logits, k = 2000), k == n (tiny level), k == 1, exact ties
    across the k-th rank (quantised logits: more ties than slots -> the deterministic lowest-index rule), constant logits,
    +-inf and NaN logits, NaN deltas."""
    from ttdg_mgm_amd import ops
    cb = _cpu_backend()
    g = synth.gen(7150)
    A = 3

    def run(shapes, pre, B, quant=None, special=False, sizes=None):
        anchors, logits, deltas = [], [], []
        for li, (h, w) in enumerate(shapes):
            s_ = 4 * 2 ** li
            ys, xs = np.meshgrid(np.arange(h) * s_, np.arange(w) * s_, indexing="ij")
            base = np.array([[-8, -4, 8, 4], [-6, -6, 6, 6], [-4, -8, 4, 8]], np.float32) * (s_ / 4)
            anchors.append(torch.from_numpy((np.stack((xs, ys, xs, ys), -1).reshape(-1, 1, 4) + base[None]).reshape(-1, 4).astype(np.float32)).to(dev))
            lg = synth.normal(g, (B, A, h, w), 1.0)
            if quant is not None:
                lg = (lg / quant).round() * quant if quant > 0 else torch.zeros_like(lg)
            d = synth.normal(g, (B, A * 4, h, w), 0.5)
            if special and h * w > 8:
                lg[0, 0, 0, 0], lg[0, 1, 0, 1], lg[0, 2, 1, 0] = float("inf"), float("-inf"), float("nan")
                d[0, 0, 0, 2] = float("nan")
                lg[0, 0, 0, 2] = 50.0                                    # (so the NaN-delta candidate is certainly selected)
            logits.append(lg.to(dev)); deltas.append(d.to(dev))
        ks = [min(pre, A * h * w) for h, w in shapes]
        K = sum(ks)
        sizes = sizes or [(shapes[0][0] * 4, shapes[0][1] * 4)] * B
        st = ops.image_sizes_tensor(sizes, dev)
        out = {}
        for fused in (True, False):
            ops.RPN_SELECT = fused
            try:
                boxes, scores = torch.full((B, K, 4), -7.0, device=dev), torch.full((B, K), -7.0, device=dev)
                ops.rpn_select(logits, deltas, anchors, ks, st, boxes, scores)
                out[fused] = (boxes.cpu(), scores.cpu())
            finally:
                ops.RPN_SELECT = True
        (bf, sf), (bp, sp) = out[True], out[False]
        # channels-last head outputs (what the channels-last backbone hands over) take the kNhwc instantiation: same slots, bit for bit
        CL = torch.channels_last
        bn, sn = torch.full((B, K, 4), -7.0, device=dev), torch.full((B, K), -7.0, device=dev)
        ops.rpn_select([t.contiguous(memory_format=CL) for t in logits], [t.contiguous(memory_format=CL) for t in deltas], anchors, ks, st, bn, sn)
        assert torch.equal(bn.cpu(), bf) and torch.equal(sn.cpu().nan_to_num(nan=7.0), sf.nan_to_num(nan=7.0))
        # (a) the kernel's own rule - descending score, ascending (h, w, a) index on ties, -0.0 == +0.0, NaN first - restated on
        #     the CPU: a stable descending sort of the raster, then the host backend's decode.  Every slot must coincide.
        bc, sc_ = torch.empty(B, K, 4), torch.empty(B, K)
        stc = cb.image_sizes_tensor(sizes, torch.device("cpu"))
        col = 0
        for lg, dl, an, k in zip(logits, deltas, anchors, ks):
            v, order = torch.sort(lg.cpu().permute(0, 2, 3, 1).reshape(B, -1), dim=1, descending=True, stable=True)
            cb.rpn_decode(dl.cpu(), an.cpu(), order[:, :k].contiguous(), v[:, :k].contiguous(), stc, bc, sc_, col)
            col += k
        live = sc_ > float("-inf")
        assert torch.equal(live, sf > float("-inf")), "live slots differ: %s" % (shapes,)
        assert torch.equal(sf[live], sc_[live]) and maxerr(bf[live], bc[live]) <= 1e-3 and bool((bf[~live] == 0).all())
        # (b) the path it replaces (per level permute + torch.topk + ttdg_rpn_decode), where that path is determined: no designed
        #     ties, and not in the slots where random fp32 logits happen to collide (7500 per row: now and then)
        if quant is None:
            assert torch.equal(torch.isnan(sf), torch.isnan(sp)) and torch.equal(sf.nan_to_num(nan=7.0), sp.nan_to_num(nan=7.0)), "scores differ: %s" % (shapes,)
            uq = torch.ones_like(live)
            uq[:, 1:] &= sf[:, 1:] != sf[:, :-1]
            uq[:, :-1] &= sf[:, :-1] != sf[:, 1:]
            assert torch.equal(bf[uq], bp[uq])
        return sf

    run([(50, 50), (25, 25), (13, 13), (7, 7), (4, 4)], 1000, 4)                 # the bench's pyramid at a quarter of the size (k = 1000; 48 = k == n on top)
    run([(120, 100), (2, 2)], 2000, 2)                                            # 36000 logits per row, k = 2000; a level with k == n == 12
    run([(24, 24), (12, 12)], 1, 3)                                               # k == 1
    run([(40, 40), (20, 20)], 700, 2, quant=0.25)                                 # ties across the k-th rank
    s0 = run([(16, 16)], 100, 2, quant=0)                                         # constant logits: 100 of 768 equal keys
    assert bool((s0[s0 > float("-inf")] == 0).all())
    run([(30, 30), (15, 15)], 500, 2, special=True)                               # +-inf / NaN logits, NaN deltas
    with pytest.raises(RuntimeError):
        lv = ops._lib.RpnLevel()
        ops.call("ttdg_rpn_select", (ops._lib.RpnLevel * 1)(lv), 1, 1, 3, 0, 0, 0, 0, ops.stream())


def test_box_inference_and_ragged_nms_match_host_formulation(dev):
    from ttdg_mgm_amd import ops
    cb = _cpu_backend()
    g = synth.gen(7200)
    C, per_image, sizes = 3, [40, 25, 0, 33], [(96, 96), (80, 96), (96, 96), (64, 64)]
    N = sum(per_image)
    xy = np.abs(synth.normal(g, (N, 2), 30.0).numpy())
    wh = np.abs(synth.normal(g, (N, 2), 12.0).numpy()) + 2
    img = np.repeat(np.arange(len(per_image)), per_image).astype(np.float32)
    rois = torch.from_numpy(np.concatenate((img[:, None], xy, xy + wh), 1).astype(np.float32))
    logits, deltas = synth.normal(g, (N, C + 1), 2.0), synth.normal(g, (N, 4 * C), 1.0)
    deltas[3, 1] = float("nan")
    out = {}
    for name, be, dv in (("cpu", cb, torch.device("cpu")), ("gpu", ops, dev)):
        st = be.image_sizes_tensor(sizes, dv)
        boxes, scores = be.box_inference(logits.to(dv), deltas.to(dv), rois.to(dv), st, C, (10.0, 10.0, 5.0, 5.0), 0.05)
        flat = be.nms_ragged(boxes, scores, per_image, C, 0.5, 20)
        out[name] = (boxes.cpu(), scores.cpu(), [f.cpu() for f in flat])
    (bc, sc_, fc), (bg, sg, fg) = out["cpu"], out["gpu"]
    live = sc_ > float("-inf")
    assert torch.equal(live, sg > float("-inf")) and not bool(live[3].any())
    assert maxerr(bg[live], bc[live]) <= 1e-3 and maxerr(sg[live], sc_[live]) <= 1e-6
    for a, b in zip(fc, fg):
        assert torch.equal(a, b)
    assert len(fg[2]) == 0


def test_paste_masks_matches_grid_sample(dev):
    from ttdg_mgm_amd import ops
    from ttdg_mgm_amd.modeling.detector import paste_masks_in_image_torch
    g = synth.gen(7300)
    R, S, H, W = 9, 28, 96, 80
    masks = torch.sigmoid(synth.normal(g, (R, 1, S, S), 2.0))
    xy = np.abs(synth.normal(g, (R, 2), 20.0).numpy())
    wh = np.abs(synth.normal(g, (R, 2), 25.0).numpy()) + 3
    boxes = torch.from_numpy(np.concatenate((xy - 5, xy + wh), 1).astype(np.float32))     # some boxes stick out of the image
    ref = paste_masks_in_image_torch(masks, boxes, (H, W), 0.5)
    got = ops.paste_masks(masks.to(dev), boxes.to(dev), H, W, 0.5).cpu()
    assert got.dtype == torch.bool and got.shape == ref.shape
    # a pixel may flip only where the interpolated value sits within rounding of the threshold
    assert int((got != ref).sum()) <= 1e-4 * ref.numel()
    assert int(ref.sum()) > 0


def test_fused_bias_residual_relu_epilogue(dev):
    """ttdg_bias_act against the three torch passes it replaces, vectorised (H*W % 4 == 0) and scalar layouts; and the
    fused no-grad backbone forward against the unfused module code on the same weights."""
    from ttdg_mgm_amd import ops
    from ttdg_mgm_amd.modeling import backbone as bb
    g = synth.gen(7400)
    for shape in ((2, 8, 12, 12), (3, 5, 5, 5)):
        y, r = synth.normal(g, shape, 1.0).to(dev), synth.normal(g, shape, 1.0).to(dev)
        b, b2 = synth.normal(g, (shape[1],), 1.0).to(dev), synth.normal(g, (shape[1],), 1.0).to(dev)
        ref = torch.relu(y + b.view(1, -1, 1, 1) + r + b2.view(1, -1, 1, 1))
        assert maxerr(ops.bias_act_(y.clone(), b, r, b2), ref) <= 1e-6
        assert maxerr(ops.bias_act_(y.clone(), b, None, None, relu=False), y + b.view(1, -1, 1, 1)) <= 1e-6
    # one bottleneck with a projection shortcut and one with the identity shortcut (few convolution configurations:
    # MIOpen builds kernels for shapes it has not seen, which is slow on a fresh box)
    torch.manual_seed(0)
    blocks = torch.nn.Sequential(bb.Bottleneck(64, 256, 64, 1), bb.Bottleneck(256, 256, 64, 1)).to(dev).eval()
    x = synth.normal(g, (2, 64, 56, 56), 1.0).to(dev)
    with torch.no_grad():
        fused = blocks(x)
        bb.FUSED_EPILOGUE = False
        try:
            plain = blocks(x)
        finally:
            bb.FUSED_EPILOGUE = True
    assert maxerr(fused, plain) <= 1e-4 * max(1.0, float(plain.abs().max()))


def test_multi_tensor_filter_fold_is_the_per_filter_multiply(dev):
    """csrc/fold.hip: the FrozenBN scale folded into all filters of a stage in one launch.  The kernel against torch's
    broadcast multiply (bit-exact: one fp32 product per value) on vector, scalar (row length % 4 != 0), unaligned and empty
    tensors, more than 64 tensors per call, and through autograd; then the ResNet trunk with the switch on and off - outputs
    bit-identical, every filter gradient equal to the vendor kernels' reproducibility in a TTA-style pass (gradients flow through res3 - res5) and in the
    no-grad pass, including after the filters have moved in place."""
    from ttdg_mgm_amd import ops
    from ttdg_mgm_amd.modeling import backbone as bb
    g = synth.gen(7420)
    shapes = [(8, 4, 3, 3), (5, 3, 7, 7), (16, 64, 1, 1), (3, 5), (0, 4, 1, 1), (130, 64, 3, 3), (7, 1, 1, 1)] + [(4, 8, 1, 1)] * 70
    ts = [synth.normal(g, sh, 1.0).to(dev) for sh in shapes]
    ts[2] = torch.cat([torch.zeros(1, device=dev), ts[2].flatten()])[1:].view(shapes[2])        # 4-byte aligned only
    sc = [synth.normal(g, (sh[0],), 1.0).to(dev) for sh in shapes]
    outs = ops.row_scale_multi(ts, sc)
    assert len(outs) == len(ts)
    for t, s_, o in zip(ts, sc, outs):
        assert o.shape == t.shape and torch.equal(o, t * s_.view([-1] + [1] * (t.dim() - 1)))
    ws = [t.clone().requires_grad_() for t in ts[:6]]
    got = ops.FoldFiltersFn.apply(tuple(sc[:6]), *ws)
    up = [synth.normal(g, t.shape, 1.0).to(dev) for t in ws]
    sum((o * u).sum() for o, u in zip(got[:5], up[:5])).backward()            # the sixth result is unused: its filter gets no gradient
    for i, (w, s_, u) in enumerate(zip(ws, sc, up)):
        if i == 5:
            assert w.grad is None
        else:
            assert torch.equal(w.grad, u * s_.view([-1] + [1] * (w.dim() - 1)))
    with pytest.raises(ValueError):
        ops.row_scale_multi(ts[:2], sc[:1])
    with pytest.raises(ValueError):
        ops.row_scale_multi([ts[0]], [sc[1]])

    torch.manual_seed(2)
    net = bb.ResNet50(2).to(dev).train()
    for m in net.modules():
        if isinstance(m, bb.FrozenBatchNorm2d):
            m.weight.copy_(synth.normal(g, m.weight.shape, 0.1).to(dev) + 1.0)
            m.running_mean.copy_(synth.normal(g, m.bias.shape, 0.1).to(dev))
    x0 = synth.normal(g, (1, 3, 64, 64), 1.0).to(dev)
    res = {}
    for multi in (True, False, None):                            # None: the per-filter path a second time
        bb.MULTI_FOLD = bool(multi)
        try:
            net.zero_grad(set_to_none=True)
            outs = net(x0)
            sum(o.square().mean() for o in outs).backward()
            grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
            with torch.no_grad():
                quiet = [o.clone() for o in net(x0)]
                keep = [p.clone() for p in net.res4.parameters()]
                for p in net.res4.parameters():
                    p.mul_(1.01)                                 # the filters move in place: the cached folds are stale
                moved = [o.clone() for o in net(x0)]
                for p, k in zip(net.res4.parameters(), keep):
                    p.copy_(k)
            res[multi] = ([o.detach().clone() for o in outs], grads, quiet, moved)
        finally:
            bb.MULTI_FOLD = True
    assert all(c._staged_w is None for c in net.modules() if isinstance(c, bb.ConvNorm))
    assert len(res[True][1]) == len(res[False][1]) > 40
    flat = {m: r[0] + r[2] + r[3] for m, r in res.items()}
    # are the vendor convolutions bit-reproducible on these shapes at all?  (small GEMM-shaped 1x1 convolutions may split K with atomics)
    reproducible = all(torch.equal(a, b) for a, b in zip(flat[False], flat[None]))
    print("trunk with per-filter folds, run twice: %s" % ("bit-identical" if reproducible else "NOT bit-identical (max %.2e)" % max(
        maxerr(a, b) for a, b in zip(flat[False], flat[None]))))
    for k, (a, b) in enumerate(zip(flat[True], flat[False])):
        if reproducible:
            assert torch.equal(a, b), "output %d differs by %.3e" % (k, maxerr(a, b))
        else:
            assert maxerr(a, b) <= 2e-5 * max(1.0, float(b.abs().max())), "output %d differs by %.3e" % (k, maxerr(a, b))
    for n, gr in res[False][1].items():                          # (MIOpen's weight-gradient kernels may split K with atomics: not bit-reproducible)
        assert maxerr(res[True][1][n], gr) <= 1e-5 * max(1e-30, float(gr.abs().max())), n
    assert not torch.equal(res[True][3][2], res[True][2][2])


def test_channels_last_layout_is_the_same_arithmetic(dev):
    """The backbone runs in channels-last memory (modeling.backbone.CHANNELS_LAST).  Every kernel behind it on an (N, H, W, C)
    activation against the NCHW form of the same operator, exactly (layout changes addresses, not sums): bias_act_ with bias /
    residual / second bias / ReLU (vector body and tail sizes), BiasActFn + relu_bwd, the multi-tensor filter fold on
    (O, kh, kw, I) filters, the node gather / scatter, the pooler fed channels-last maps.  Then ResNet-50 + FPN as a whole in both
    layouts: outputs and filter gradients to the vendor convolutions' accuracy (the two layouts run different MIOpen solvers),
    gradient strides matching the filters', and the fused SGD step applied to channels-last parameters."""
    from ttdg_mgm_amd import ops
    from ttdg_mgm_amd.modeling import backbone as bb
    from ttdg_mgm_amd.optim import FusedSGD
    g = synth.gen(7440)
    CL = torch.channels_last
    for shape in ((2, 8, 12, 12), (3, 64, 5, 7), (1, 256, 50, 50), (5, 4, 1, 3)):
        y, r = synth.normal(g, shape, 1.0).to(dev), synth.normal(g, shape, 1.0).to(dev)
        b, b2 = synth.normal(g, (shape[1],), 1.0).to(dev), synth.normal(g, (shape[1],), 1.0).to(dev)
        for args in ((b, r, b2, True), (b, None, None, True), (b, r, None, False), (None, r, b2, True), (b, None, None, False)):
            bb_, rr, b2_, relu = args
            want = ops.bias_act_(y.clone(), bb_, rr, b2_, relu=relu)
            ycl = y.clone().contiguous(memory_format=CL)
            got = ops.bias_act_(ycl, bb_, None if rr is None else rr.contiguous(memory_format=CL), b2_, relu=relu)
            assert got.data_ptr() == ycl.data_ptr() and (ops.is_channels_last(got) or min(shape[1], shape[2] * shape[3]) == 1)
            assert torch.equal(got, want), (shape, relu)
        # with gradients
        res = {}
        for cl in (False, True):
            fmt = CL if cl else torch.contiguous_format
            y0, r0 = y.clone().contiguous(memory_format=fmt).requires_grad_(), r.clone().contiguous(memory_format=fmt).requires_grad_()
            out = ops.BiasActFn.apply((y0 * 1.0).contiguous(memory_format=fmt), b, (r0 * 1.0).contiguous(memory_format=fmt), b2)
            (out * r).sum().backward()
            res[cl] = (out.detach().clone(), y0.grad.clone(), r0.grad.clone())
        for a, c in zip(res[True], res[False]):
            assert torch.equal(a, c)
    with pytest.raises(ValueError):
        ops.bias_act_(y.clone().contiguous(memory_format=CL), b, r)                # residual in the other layout
    # filter fold on channels-last filters
    ws = [synth.normal(g, sh, 1.0).to(dev) for sh in ((8, 4, 3, 3), (16, 64, 1, 1), (12, 8, 7, 7))]
    sc = [synth.normal(g, (w.shape[0],), 1.0).to(dev) for w in ws]
    outs = ops.row_scale_multi([w.contiguous(memory_format=CL) for w in ws], sc)
    for w, s_, o in zip(ws, sc, outs):
        assert torch.equal(o, w * s_.view(-1, 1, 1, 1)) and o.is_contiguous(memory_format=CL)
    # node gather / scatter
    feats = [synth.normal(g, (2, 16, s_, s_ + 1), 1.0).to(dev) for s_ in (8, 4)]
    img = torch.tensor([0, 1, 1, 0, 1], dtype=torch.int32, device=dev)
    pid = torch.tensor([5, 70, (1 << 28) | 3, (1 << 28) | 19, 0], dtype=torch.int32, device=dev)
    up = synth.normal(g, (5, 16), 1.0).to(dev)
    res = {}
    for cl in (False, True):
        fs = [f.clone().contiguous(memory_format=CL if cl else torch.contiguous_format).requires_grad_() for f in feats]
        rows = ops.NodeGatherFn.apply(img, pid, *fs)
        (rows * up).sum().backward()
        res[cl] = (rows.detach().clone(), [f.grad.clone() for f in fs], [f.grad.is_contiguous(memory_format=CL) for f in fs])
    assert torch.equal(res[True][0], res[False][0]) and all(torch.equal(a, c) for a, c in zip(res[True][1], res[False][1])) and all(res[True][2])
    # [r6] no node selected in the whole batch (detections too small to contain an FPN point: a random-initialised detector produced that on one
    # fresh box, profiles/r06_gpu_suite_box13_failed.txt): empty index tensors carry null pointers - empty rows, zero gradients, no error
    for cl in (False, True):
        fs = [f.clone().contiguous(memory_format=CL if cl else torch.contiguous_format).requires_grad_() for f in feats]
        e32 = torch.empty(0, dtype=torch.int32, device=dev)
        rows = ops.NodeGatherFn.apply(e32, e32, *fs)
        assert rows.shape == (0, 16)
        (rows.sum() + sum(f.sum() * 0 for f in fs)).backward()
        assert all(f.grad is not None and not bool(f.grad.any()) for f in fs)
    # the same for an empty ROI set / empty candidate set
    maps0 = [synth.normal(g, (2, 32, 16 // k, 16 // k), 1.0).to(dev) for k in (1, 2)]
    r0 = torch.empty(0, 5, device=dev)
    assert ops.roi_align_multilevel(maps0, r0, [4, 8], 7).shape == (0, 32, 7, 7)
    assert ops.roi_align_multilevel(maps0, r0, [4, 8], 7, nhwc=ops.to_nhwc(maps0)).shape == (0, 32, 7, 7)
    b0, s0 = ops.box_inference(torch.empty(0, 3, device=dev), torch.empty(0, 8, device=dev), torch.empty(0, 5, device=dev),
                               ops.image_sizes_tensor([(64, 64)], dev), 2, (10.0, 10.0, 5.0, 5.0), 0.05)
    assert b0.shape == (0, 2, 4) and s0.shape == (0, 2)
    # pooler: channels-last maps are used as they are
    maps = [synth.normal(g, (2, 32, 16 // k, 16 // k), 1.0).to(dev) for k in (1, 2)]
    rois = torch.tensor([[0, 1.0, 2.0, 30.0, 40.0], [1, 5.0, 5.0, 20.0, 12.0], [1, 0.0, 0.0, 63.0, 63.0]], device=dev)
    a = ops.roi_align_multilevel(maps, rois, [4, 8], 7, nhwc=ops.to_nhwc(maps))
    mcl = [m.contiguous(memory_format=CL) for m in maps]
    views = ops.to_nhwc(mcl)
    assert all(v.data_ptr() == m.data_ptr() for v, m in zip(views, mcl))
    assert torch.equal(ops.roi_align_multilevel(mcl, rois, [4, 8], 7, nhwc=views), a)

    # the whole trunk + FPN in both layouts
    torch.manual_seed(5)
    x0 = synth.normal(g, (2, 3, 64, 96), 1.0).to(dev)
    res = {}
    keep = bb.CHANNELS_LAST
    for cl in (False, True):
        bb.CHANNELS_LAST = cl
        try:
            torch.manual_seed(6)
            net = bb.FPN(2).to(dev).train()
            for m in net.modules():
                if isinstance(m, bb.FrozenBatchNorm2d):
                    m.weight.copy_(1.0 + 0.1 * torch.sin(torch.arange(m.weight.numel(), device=dev).float()))
                    m.running_mean.copy_(0.1 * torch.cos(torch.arange(m.weight.numel(), device=dev).float()))
            outs = net(x0)
            sum(v.square().mean() for v in outs.values()).backward()
            w = net.bottom_up.res4[0].conv2.weight
            assert w.is_contiguous(memory_format=CL) == cl or not cl
            if cl:
                assert ops.is_channels_last(outs["p3"]) and ops.is_channels_last(w) and w.grad.stride() == w.stride()
            grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
            with torch.no_grad():
                quiet = net(x0)
            opt = FusedSGD([p for p in net.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=1e-4)
            before = {n: p.detach().clone() for n, p in net.named_parameters()}
            opt.step()
            moved = {n: (p.detach() - before[n]) for n, p in net.named_parameters() if p.grad is not None}
            res[cl] = ({k: v.detach().clone() for k, v in outs.items()}, grads, moved, {k: v.clone() for k, v in quiet.items()})
        finally:
            bb.CHANNELS_LAST = keep
    for k in res[False][0]:
        sc_ = max(1.0, float(res[False][0][k].abs().max()))
        assert maxerr(res[True][0][k], res[False][0][k]) <= 2e-4 * sc_ and maxerr(res[True][3][k], res[False][3][k]) <= 2e-4 * sc_, k
    assert len(res[True][1]) == len(res[False][1]) > 50
    # gradients of a randomly initialised 50-layer trunk amplify the rounding of the first layers: the two layouts (different
    # MIOpen solvers in every convolution) are compared in the Frobenius norm, and entry-wise against the largest entry
    worst = (0.0, None)
    for n, gr in res[False][1].items():
        rel = float((res[True][1][n] - gr).norm()) / max(1e-30, float(gr.norm()))
        worst = max(worst, (rel, n))
        assert rel <= 2e-2 and maxerr(res[True][1][n], gr) <= 5e-2 * max(1e-30, float(gr.abs().max())), (n, rel)
        mv = res[False][2][n]                                                                   # first SGD step: -lr * (g + wd p)
        assert float((res[True][2][n] - mv).norm()) <= 2e-2 * max(1e-30, float(mv.norm())), n
    print("channels-last vs NCHW trunk: worst relative gradient difference %.2e (%s)" % worst)


@pytest.mark.parametrize("quant", [None, 0.25])
def test_rpn_heads_as_one_product_per_level(dev, quant):
    """[r6] RPNHead's packed path (modeling/detector.py: FUSED_RPN_HEADS): the two 1 x 1 heads of a level as ONE streaming product with split
    output, the 3 x 3 filter's bias + ReLU applied at the operand fetch, objectness padded 3 -> 4 anchors (pad logit -inf), deltas 12 -> 16.
      (a) head outputs against the float64 statement conv3x3 -> +bias -> ReLU -> the two 1 x 1 filters, next to the vendor path's distance from it;
      (b) the pad columns: logit -inf, deltas exactly 0;
      (c) the selection (ops.rpn_select with A = 4, padded anchors, k from the three real anchors) on the packed outputs returns, bit for bit,
          what it returns on the three real channels copied out (A = 3) - also with quantised logits (ties across the k-th rank)."""
    from ttdg_mgm_amd import ops
    from ttdg_mgm_amd.modeling import detector as det
    torch.manual_seed(5)
    rpn = det.PseudoLabRPN().to(dev).eval()
    head = rpn.rpn_head
    with torch.no_grad():
        for m in (head.conv, head.objectness_logits, head.anchor_deltas):
            m.weight.normal_(0, 0.05)
            m.bias.normal_(0, 0.3)
    B = 2
    shapes = [(40, 56), (20, 28), (10, 14), (5, 7), (3, 4)]
    feats = [torch.randn(B, 256, h, w, device=dev).contiguous(memory_format=torch.channels_last) for h, w in shapes]
    with torch.no_grad():
        assert det.FUSED_RPN_HEADS
        lg_p, dl_p = head(feats)
        det.FUSED_RPN_HEADS = False
        try:
            lg_v, dl_v = head(feats)
        finally:
            det.FUSED_RPN_HEADS = True
    h64 = copy.deepcopy(head).double()
    for x, lp, dp, lv, dv in zip(feats, lg_p, dl_p, lg_v, dl_v):
        assert lp.shape[1] == 4 and dp.shape[1] == 16 and lv.shape[1] == 3 and dv.shape[1] == 12
        assert lp.is_contiguous(memory_format=torch.channels_last) and dp.is_contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            t = torch.relu(h64.conv(x.double()))
            l64, d64 = h64.objectness_logits(t), h64.anchor_deltas(t)
        derived_gate("rpn heads, objectness %s" % (tuple(x.shape[-2:]),), lp[:, :3], lv, l64, quiet=True)
        derived_gate("rpn heads, deltas %s" % (tuple(x.shape[-2:]),), dp[:, :12], dv, d64, quiet=True)
        assert bool(torch.isneginf(lp[:, 3]).all()) and bool((dp[:, 12:] == 0).all())
    if quant is not None:            # ties across the k-th rank: quantise the packed logits in place (pad stays -inf)
        for lp in lg_p:
            lp[:, :3] = (lp[:, :3] / quant).round() * quant
    pre = 600
    ks = [min(pre, 3 * h * w) for h, w in shapes]
    K = sum(ks)
    st = ops.image_sizes_tensor([(160, 224)] * B, dev)
    a3 = rpn._anchors(shapes, dev)
    a4 = rpn._anchors(shapes, dev, pad_to=4)
    b4, s4 = torch.full((B, K, 4), -7.0, device=dev), torch.full((B, K), -7.0, device=dev)
    ops.rpn_select(lg_p, dl_p, a4, ks, st, b4, s4)
    CL = torch.channels_last
    b3, s3 = torch.full((B, K, 4), -7.0, device=dev), torch.full((B, K), -7.0, device=dev)
    ops.rpn_select([t[:, :3].contiguous(memory_format=CL) for t in lg_p], [t[:, :12].contiguous(memory_format=CL) for t in dl_p], a3, ks, st, b3, s3)
    assert torch.equal(s4, s3) and torch.equal(b4, b3)
    assert bool((s4 > float("-inf")).any())


def test_fused_bias_epilogues_of_fpn_rpn_head_and_mask_head(dev):
    """modeling.backbone.FUSED_HEADS: the FPN's lateral / output convolutions apply bias (+ the top-down sum) through the
    in-place epilogue kernel (ops.BiasAddFn when gradients flow), the RPN head runs without a tape inside the TTA step and the
    mask head fuses bias + ReLU.  Same sums in the same order as conv-with-bias, add, relu: outputs and every gradient against
    the plain torch formulation (vendor convolutions see identical operands; their own run-to-run rounding is the tolerance)."""
    from ttdg_mgm_amd import ops
    from ttdg_mgm_amd.modeling import backbone as bb
    from ttdg_mgm_amd.modeling import detector as det
    g = synth.gen(7430)
    # BiasAddFn against torch, exactly
    y0, r0, b0 = synth.normal(g, (2, 6, 8, 8), 1.0).to(dev), synth.normal(g, (2, 6, 8, 8), 1.0).to(dev), synth.normal(g, (6,), 1.0).to(dev)
    up = synth.normal(g, (2, 6, 8, 8), 1.0).to(dev)
    for with_res in (True, False):
        y, r, b = y0.clone().requires_grad_(), r0.clone().requires_grad_(), b0.clone().requires_grad_()
        out = ops.BiasAddFn.apply(y * 1.0, b, r * 1.0 if with_res else None)
        (out * up).sum().backward()
        y2, r2, b2 = y0.clone().requires_grad_(), r0.clone().requires_grad_(), b0.clone().requires_grad_()
        ref = (y2 + b2.view(1, -1, 1, 1)) + r2 if with_res else y2 + b2.view(1, -1, 1, 1)
        (ref * up).sum().backward()
        assert torch.equal(out, ref) and torch.equal(y.grad, y2.grad) and maxerr(b.grad, b2.grad) <= 1e-5
        assert (torch.equal(r.grad, r2.grad) if with_res else r.grad is None)
    # the FPN on top of a fixed pyramid (the trunk is covered elsewhere): train-style pass and no-grad pass
    torch.manual_seed(3)
    fpn = bb.FPN(2).to(dev).train()
    for m in fpn.modules():
        if isinstance(m, torch.nn.Conv2d) and m.bias is not None:
            m.bias.data.copy_(synth.normal(g, m.bias.shape, 0.5).to(dev))
    x0 = synth.normal(g, (1, 3, 64, 64), 1.0).to(dev)
    res = {}
    for fused in (True, False):
        bb.FUSED_HEADS = fused
        try:
            fpn.zero_grad(set_to_none=True)
            outs = fpn(x0)
            sum(v.square().mean() for v in outs.values()).backward()
            grads = {n: p.grad.clone() for n, p in fpn.named_parameters() if p.grad is not None and n.startswith("fpn_")}
            with torch.no_grad():
                quiet = fpn(x0)
            res[fused] = ({k: v.detach().clone() for k, v in outs.items()}, grads, {k: v.clone() for k, v in quiet.items()})
        finally:
            bb.FUSED_HEADS = True
    assert len(res[True][1]) == len(res[False][1]) == 16
    for k in res[False][0]:
        sc_ = max(1.0, float(res[False][0][k].abs().max()))
        assert maxerr(res[True][0][k], res[False][0][k]) <= 2e-5 * sc_ and maxerr(res[True][2][k], res[False][2][k]) <= 2e-5 * sc_, k
    for n, gr in res[False][1].items():
        assert maxerr(res[True][1][n], gr) <= 1e-4 * max(1e-30, float(gr.abs().max())), n
    # mask head (inference only) and RPN head without a tape
    torch.manual_seed(4)
    mh, rh = det.MaskRCNNConvUpsampleHead(3).to(dev).eval(), det.RPNHead().to(dev)
    for m in list(mh.modules()) + list(rh.modules()):
        if getattr(m, "bias", None) is not None and isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            m.bias.data.copy_(synth.normal(g, m.bias.shape, 0.5).to(dev))
            m.weight.data.mul_(20.0 if m.weight.std() < 0.02 else 1.0)
    xm = synth.normal(g, (5, 256, 14, 14), 1.0).to(dev)
    feats = [synth.normal(g, (2, 256, s_, s_), 1.0).to(dev) for s_ in (16, 8)]
    out = {}
    for fused in (True, False):
        bb.FUSED_HEADS = fused
        try:
            with torch.no_grad():
                out[fused] = (mh(xm).clone(), [t.clone() for ts in rh(feats) for t in ts])
        finally:
            bb.FUSED_HEADS = True
    assert out[True][0].shape == (5, 3, 28, 28) and maxerr(out[True][0], out[False][0]) <= 2e-5 * max(1.0, float(out[False][0].abs().max()))
    for a, b in zip(out[True][1], out[False][1]):
        assert maxerr(a, b) <= 2e-5 * max(1.0, float(b.abs().max()))


def test_fused_epilogue_with_gradients_is_the_plain_torch_block(dev):
    """The adapted bottlenecks (gradients flow) take ops.BiasActFn - in-place shift / residual / ReLU with a one-pass backward.
    Same arithmetic in the same order as conv-with-bias, add, F.relu_: outputs, input gradient and every filter gradient must
    coincide with the plain torch formulation (the only difference allowed is the rounding inside the vendor convolutions,
    which see bit-identical operands)."""
    from ttdg_mgm_amd.modeling import backbone as bb
    g = synth.gen(7410)
    torch.manual_seed(1)
    blocks = torch.nn.Sequential(bb.Bottleneck(64, 256, 64, 1), bb.Bottleneck(256, 256, 64, 1)).to(dev).train()
    for m in blocks.modules():
        if isinstance(m, bb.FrozenBatchNorm2d):
            m.weight.copy_(synth.normal(g, m.weight.shape, 0.2).to(dev) + 1.0)
            m.bias.copy_(synth.normal(g, m.bias.shape, 0.2).to(dev))
            m.running_mean.copy_(synth.normal(g, m.bias.shape, 0.2).to(dev))
    x0 = synth.normal(g, (2, 64, 56, 56), 1.0).to(dev)
    wts = synth.normal(g, (2, 256, 56, 56), 1.0).to(dev)
    res = {}
    for fused in (True, False):
        bb.FUSED_EPILOGUE = fused
        try:
            x = x0.clone().requires_grad_()
            blocks.zero_grad(set_to_none=True)
            out = blocks(x)
            (out * wts).sum().backward()
            res[fused] = (out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in blocks.parameters()])
        finally:
            bb.FUSED_EPILOGUE = True
    assert torch.equal(res[True][0], res[False][0])
    assert maxerr(res[True][1], res[False][1]) <= 1e-5 * max(1.0, float(res[False][1].abs().max()))
    assert len(res[True][2]) == 7 and all(t is not None for t in res[True][2])
    for a, b in zip(res[True][2], res[False][2]):
        assert maxerr(a, b) <= 1e-5 * max(1.0, float(b.abs().max()))


def _mm_bound(A64, B64):
    """fp32 MFMA chain against float64: one rounding per product and per addition; |error| <= c * eps * sum_k |a_k b_k| with c growing
    like the chain length's square root in practice (MI355X guide: 0.75-1.5e-7 at K <= 1024, 3.5e-7 at K = 4096) - gate 1e-6."""
    return 1e-6 * (A64.abs() @ B64.abs().t()) + 1e-30


@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (1, 4, 4), (70, 132, 100), (129, 64, 36), (1000, 256, 64), (257, 520, 2048), (5000, 128, 256)])
def test_pointwise_product_every_tile_vs_float64(dev, M, N, K):
    """ttdg_mm_f32 (csrc/pointwise.hip), k-contiguous operands: every tile code, ragged M / N / K edges (K % 32 != 0 exercises the
    register-zeroed tail slab), bias + residual + second bias + ReLU epilogue, the fused input activation, against float64."""
    from ttdg_mgm_amd import ops
    g = synth.gen(7600 + M + N + K)
    A, Bm = synth.normal(g, (M, K), 1.0).to(dev), synth.normal(g, (N, K), 1.0).to(dev)
    b, b2, pb = synth.normal(g, (N,), 1.0).to(dev), synth.normal(g, (N,), 1.0).to(dev), synth.normal(g, (K,), 1.0).to(dev)
    R = synth.normal(g, (M, N), 1.0).to(dev)
    A64, B64 = A.double(), Bm.double()
    plain = A64 @ B64.t()
    for tile in (0, 1, 2, 3, 4):
        out = torch.full((M, N), float("nan"), device=dev)
        ops.mm(A, Bm, out, M, N, K, K, K, N, tile=tile)
        assert bool(((out.double() - plain).abs() <= _mm_bound(A64, B64)).all()), (tile, float((out.double() - plain).abs().max()))
        ops.mm(A, Bm, out, M, N, K, K, K, N, bias=b, res=R, ldres=N, bias2=b2, relu=True, tile=tile)
        want = (plain + b.double() + (R.double() + b2.double())).relu()
        assert bool(((out.double() - want).abs() <= _mm_bound(A64, B64) + 1e-6 * (1 + want.abs())).all()), tile
        Ap = (A64 + pb.double()).relu()
        ops.mm(A, Bm, out, M, N, K, K, K, N, bias=b, pbias=pb, prelu=True, tile=tile)
        assert bool(((out.double() - (Ap @ B64.t() + b.double())).abs() <= _mm_bound(Ap, B64) + 1e-6).all()), tile
    # two launches of one shape give the same bits (fixed order of additions)
    o1, o2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.mm(A, Bm, o1, M, N, K, K, K, N, bias=b)
    ops.mm(A, Bm, o2, M, N, K, K, K, N, bias=b)
    assert torch.equal(o1, o2)


def test_pointwise_product_backward_layouts_row_maps_and_split(dev):
    """The other operand layouts of ttdg_mm_f32 (dX = dY W with the filter read n-contiguous, dW = dY^T X with both operands read
    through the strided reduction index and the pixel axis split over workgroup planes), the strided A row map of a stride-2
    convolution and the up-sampled residual of the FPN's top-down sum, against float64 / the torch formulation."""
    from ttdg_mgm_amd import ops
    g = synth.gen(7650)
    for M, Cin, Cout in ((300, 64, 128), (1030, 132, 68), (4100, 256, 512)):
        dy, x, w = synth.normal(g, (M, Cout), 1.0).to(dev), synth.normal(g, (M, Cin), 1.0).to(dev), synth.normal(g, (Cout, Cin), 1.0).to(dev)
        for tile in (0, 1, 2, 3, 4):
            dx = torch.full((M, Cin), float("nan"), device=dev)
            ops.mm(dy, w, dx, M, Cin, Cout, Cout, Cin, Cin, b_layout=1, tile=tile)
            want = dy.double() @ w.double()
            assert bool(((dx.double() - want).abs() <= _mm_bound(dy.double(), w.double().t())).all()), ("dx", M, tile)
            for ks in (0, 2, 5, M // 256):
                if ks == 1:
                    continue
                dw = torch.full((Cout, Cin), float("nan"), device=dev)
                ops.mm(dy, x, dw, Cout, Cin, M, Cout, Cin, Cin, a_layout=1, b_layout=1, kslices=ks, tile=tile)
                want = dy.double().t() @ x.double()
                assert bool(((dw.double() - want).abs() <= _mm_bound(dy.double().t(), x.double().t())).all()), ("dw", M, tile, ks)
    # strided row map + up-sampled residual through the convolution wrappers
    CL = torch.channels_last
    for (B, Cin, Cout, H, W, s) in ((2, 64, 128, 14, 18, 2), (1, 32, 64, 7, 9, 2), (3, 16, 32, 8, 8, 1)):
        x = synth.normal(g, (B, Cin, H, W), 1.0).to(dev).contiguous(memory_format=CL)
        w = synth.normal(g, (Cout, Cin, 1, 1), 0.2).to(dev)
        b = synth.normal(g, (Cout,), 1.0).to(dev)
        got = ops.pointwise_conv(x, w, b, relu=True, stride=s)
        want = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), s).relu()
        assert got.shape == want.shape and ops.is_channels_last(got) and maxerr(got, want) <= 1e-5 * float(want.abs().max())
        Ho, Wo = want.shape[2], want.shape[3]
        if Ho % 2 == 0 and Wo % 2 == 0:
            xs = x[:, :, ::s, ::s].contiguous(memory_format=CL)
            coarse = synth.normal(g, (B, Cout, Ho // 2, Wo // 2), 1.0).to(dev).contiguous(memory_format=CL)
            got = ops.pointwise_conv(xs, w, b, residual=coarse, res_up=True)
            want = torch.nn.functional.conv2d(xs.double(), w.double(), b.double()) + torch.nn.functional.interpolate(coarse.double(), scale_factor=2.0, mode="nearest")
            assert maxerr(got, want) <= 1e-5 * float(want.abs().max())
    # second reduction segment: conv3(relu(y + b2)) + shortcut(x[::s, ::s]) in one product (a bottleneck's first block)
    for (B, Cmid, Cin2, Cout, H, W, s) in ((2, 64, 64, 256, 12, 20, 1), (1, 128, 256, 512, 14, 18, 2), (2, 512, 1024, 2048, 6, 10, 2)):
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        y = synth.normal(g, (B, Cmid, Ho, Wo), 1.0).to(dev).contiguous(memory_format=CL)
        x2 = synth.normal(g, (B, Cin2, H, W), 1.0).to(dev).contiguous(memory_format=CL)
        w3, ws = synth.normal(g, (Cout, Cmid, 1, 1), 0.1).to(dev), synth.normal(g, (Cout, Cin2, 1, 1), 0.1).to(dev)
        b3, bs, b2 = synth.normal(g, (Cout,), 1.0).to(dev), synth.normal(g, (Cout,), 1.0).to(dev), synth.normal(g, (Cmid,), 1.0).to(dev)
        got = ops.pointwise_conv(y, w3, b3, bias2=bs, relu=True, pbias=b2, prelu=True, second=(x2, ws, s))
        f64 = torch.nn.functional.conv2d
        want = (f64((y.double() + b2.double().view(1, -1, 1, 1)).relu(), w3.double(), b3.double()) + f64(x2.double(), ws.double(), bs.double(), s)).relu()
        assert got.shape == want.shape and maxerr(got, want) <= 2e-6 * float(want.abs().max()) * max(1.0, (Cmid + Cin2) ** 0.5 / 8), (B, Cmid, Cin2, Cout, s)
        two = ops.pointwise_conv(ops.bias_act_(y.clone(), b2), w3, b3, residual=ops.pointwise_conv(x2, ws, bs, stride=s), relu=True)
        assert maxerr(got, two) <= 1e-5 * float(want.abs().max())
    with pytest.raises(ValueError):
        ops.pointwise_conv(y, w3, b3, residual=y, second=(x2, ws, s))               # a second segment takes no residual
    with pytest.raises(TypeError):
        ops.pointwise_conv(x.contiguous(), w)                                     # NCHW activation
    with pytest.raises(ValueError):
        ops.pointwise_conv(x, w, residual=x)                                      # residual of the wrong shape


def test_pointwise_convolution_with_gradients_vs_float64_autograd(dev):
    """ops.PointwiseConvFn (forward product with fused shift / residual / ReLU, backward mask + dX + split dW + bias column sums +
    residual gradient incl. the 2 x 2 sum behind the up-sampled residual, strided input compaction) against torch autograd on the
    float64 formulation of the same block."""
    from ttdg_mgm_amd import ops
    g = synth.gen(7660)
    CL = torch.channels_last
    cases_ = [(2, 64, 128, 12, 16, 1, True, False, True), (2, 64, 128, 12, 16, 2, False, False, True), (1, 128, 64, 20, 20, 1, True, True, False),
              (2, 32, 32, 30, 34, 1, False, False, False), (3, 256, 64, 16, 16, 2, True, False, True)]
    for (B, Cin, Cout, H, W, s, with_res, up, relu), engine in [(c, e) for c in cases_ for e in ("own", "vendor")]:
        x0 = synth.normal(g, (B, Cin, H, W), 1.0).to(dev)
        w0 = synth.normal(g, (Cout, Cin, 1, 1), 0.1).to(dev)
        b0 = synth.normal(g, (Cout,), 0.5).to(dev)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        rshape = (B, Cout, Ho // 2, Wo // 2) if up else (B, Cout, Ho, Wo)
        r0 = synth.normal(g, rshape, 1.0).to(dev) if with_res else None
        up_w = synth.normal(g, (B, Cout, Ho, Wo), 1.0).to(dev)
        x = x0.clone().contiguous(memory_format=CL).requires_grad_()
        w, b = w0.clone().requires_grad_(), b0.clone().requires_grad_()
        r = r0.clone().contiguous(memory_format=CL).requires_grad_() if with_res else None
        keep = ops.POINTWISE_BACKWARD
        ops.POINTWISE_BACKWARD = engine            # who computes dX / dW: the streaming product's backward layouts or MIOpen
        try:
            out = ops.PointwiseConvFn.apply(x, w, b, r, None, relu, s, up)
            (out * up_w).sum().backward()
        finally:
            ops.POINTWISE_BACKWARD = keep
        x2, w2, b2 = x0.double().requires_grad_(), w0.double().requires_grad_(), b0.double().requires_grad_()
        r2 = r0.double().requires_grad_() if with_res else None
        ref = torch.nn.functional.conv2d(x2, w2, b2, s)
        if with_res:
            ref = ref + (torch.nn.functional.interpolate(r2, scale_factor=2.0, mode="nearest") if up else r2)
        if relu:
            ref = ref.relu()
        (ref * up_w.double()).sum().backward()
        key = (B, Cin, Cout, H, W, s, with_res, up, relu, engine)
        assert ops.is_channels_last(out) and maxerr(out, ref) <= 2e-6 * float(ref.abs().max()) * max(1.0, Cin ** 0.5 / 8), key
        for name, a, c in (("dx", x.grad, x2.grad), ("dw", w.grad, w2.grad), ("db", b.grad, b2.grad)) + ((("dres", r.grad, r2.grad),) if with_res else ()):
            assert a.shape == c.shape, (name, key)
            assert maxerr(a, c) <= 3e-6 * float(c.abs().max()) * max(1.0, (B * Ho * Wo) ** 0.5 / 16), (name, key, maxerr(a, c), float(c.abs().max()))
        assert w.grad.shape == w.shape and (engine == "vendor" or w.grad.stride() == w.stride())


def test_own_pointwise_backbone_vs_vendor_backbone_and_float64(dev):
    """modeling.backbone.OWN_POINTWISE: ResNet-50 + FPN in channels-last memory with the pointwise convolutions on the streaming
    product (fused epilogues, own backward products) against the same network on vendor convolutions + epilogue kernels: feature
    maps of the no-grad pass and of the TTA-style pass, every filter gradient.  Neither arm is the truth: both are held to the
    float64 CPU network (same weights, same input), and the own arm may not be further from it than twice the vendor arm
    (+ a rounding floor)."""
    from ttdg_mgm_amd.modeling import backbone as bb
    g = synth.gen(7670)
    torch.manual_seed(11)
    net = bb.FPN(2).train()
    for m in net.modules():
        if isinstance(m, bb.FrozenBatchNorm2d):
            m.weight.copy_(1.0 + 0.1 * torch.sin(torch.arange(m.weight.numel()).float()))
            m.running_mean.copy_(0.1 * torch.cos(torch.arange(m.weight.numel()).float()))
        if isinstance(m, torch.nn.Conv2d) and m.bias is not None:
            m.bias.data.copy_(synth.normal(g, m.bias.shape, 0.1))
    x0 = synth.normal(g, (2, 3, 96, 128), 1.0)
    import copy
    net64 = copy.deepcopy(net).double()
    outs64 = net64(x0.double())
    sum(v.square().mean() for v in outs64.values()).backward()
    g64 = {n: p.grad for n, p in net64.named_parameters() if p.grad is not None}
    from ttdg_mgm_amd import ops
    res = {}
    keep, keep_b, keep_p, keep_a, keep_s = bb.OWN_POINTWISE, ops.POINTWISE_BACKWARD, bb.POINTWISE_MIN_PIXELS, bb.FUSED_INPUT_ACTIVATION, bb.FUSED_SHORTCUT
    # MIOpen restricted to its deterministic solvers for all three arms: with the default choice its backward kernels for these small
    # maps differ from run to run by 1e-3 ... 1e-2 relative in single filter gradients (recorded in round 5: profiles/r05_graph_probe.txt,
    # "default" 2.7e-3 ... 2.2e-2 against 2.4e-6 "deterministic"; seen again here with every hand-written kernel switched off:
    # res3.0.conv1.weight 1.7e-3 from float64 in 4 of 6 identical runs) - a vendor property that would otherwise decide this comparison
    keep_det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    for own in (True, "own-backward", False):
        bb.OWN_POINTWISE = bool(own)
        bb.POINTWISE_MIN_PIXELS = 0                  # every stage of this small input on the streaming product (the bench routes res5's 2500 pixels to the vendor)
        bb.FUSED_INPUT_ACTIVATION = bb.FUSED_SHORTCUT = own is True      # the second arm applies conv2's epilogue with the in-place kernel and runs the shortcut as its own product
        ops.POINTWISE_BACKWARD = "own" if own == "own-backward" else "vendor"
        try:
            nd = copy.deepcopy(net).to(dev)
            outs = nd(x0.to(dev))
            sum(v.square().mean() for v in outs.values()).backward()
            grads = {n: p.grad.detach().cpu() for n, p in nd.named_parameters() if p.grad is not None}
            with torch.no_grad():
                quiet = nd(x0.to(dev))
            res[own] = ({k: v.detach().cpu() for k, v in outs.items()}, grads, {k: v.cpu() for k, v in quiet.items()})
        finally:
            bb.OWN_POINTWISE, ops.POINTWISE_BACKWARD, bb.POINTWISE_MIN_PIXELS, bb.FUSED_INPUT_ACTIVATION, bb.FUSED_SHORTCUT = keep, keep_b, keep_p, keep_a, keep_s
            if own is False:
                torch.backends.cudnn.deterministic = keep_det
    assert set(res[True][1]) == set(res["own-backward"][1]) == set(res[False][1]) == set(g64) and len(g64) > 50
    for arm in (True, "own-backward"):
        worst = {"out": (0.0, 0.0), "grad": (0.0, 0.0)}
        for k, t in outs64.items():
            sc = float(t.abs().max())
            for idx in (0, 2):
                eo, ev = maxerr(res[arm][idx][k], t.detach()) / sc, maxerr(res[False][idx][k], t.detach()) / sc
                worst["out"] = max(worst["out"], (eo, ev))
                assert eo <= 2.0 * ev + 2e-6, (arm, k, idx, eo, ev)
        for n, t in g64.items():
            nrm = float(t.norm())
            eo, ev = float((res[arm][1][n].double() - t).norm()) / max(nrm, 1e-30), float((res[False][1][n].double() - t).norm()) / max(nrm, 1e-30)
            worst["grad"] = max(worst["grad"], (eo, ev))
            assert eo <= 2.0 * ev + 2e-6, (arm, n, eo, ev)
        print("own pointwise (%s) vs float64: worst relative output error %.2e (vendor %.2e), worst relative gradient error %.2e (vendor %.2e)"
              % (("forward, vendor backward" if arm is True else "forward and backward",) + worst["out"] + worst["grad"]))


def test_roi_align_multilevel_matches_per_level_pooler(dev):
    """ttdg_roi_align_multilevel (level chosen inside the kernel) against detectron2's ROIPooler formulation
    (level by level: assign, compact, ROIAlign, scatter) on the host."""
    from ttdg_mgm_amd import ops
    cb = _cpu_backend()
    g = synth.gen(7500)
    B, C = 2, 16
    feats = [synth.normal(g, (B, C, 64 // s, 64 // s), 1.0) for s in (1, 2, 4, 8)]      # strides 4, 8, 16, 32 of a 256 px image
    R = 60
    xy = np.abs(synth.normal(g, (R, 2), 60.0).numpy())
    wh = np.abs(synth.normal(g, (R, 2), 80.0).numpy()) + 1.0          # from a few pixels to beyond the canonical 224
    img = g.integers(0, B, size=R).astype(np.float32)
    rois = torch.from_numpy(np.concatenate((img[:, None], xy, xy + wh), 1).astype(np.float32))
    rois[0, 1:] = torch.tensor([10.0, 10.0, 10.0, 10.0])             # degenerate box
    from ttdg_mgm_amd import _lib
    for P in (7, 14):
        ref = cb.roi_align_multilevel(feats, rois, [4, 8, 16, 32], P)
        for mode in (2, 1, 0):           # separable table kernel (default), direct kernel XCD-sliced, direct kernel flat
            _lib.load().ttdg_debug_set_roi_align_sliced(mode)
            try:
                got = ops.roi_align_multilevel([f.to(dev) for f in feats], rois.to(dev), [4, 8, 16, 32], P)
            finally:
                _lib.load().ttdg_debug_set_roi_align_sliced(2)
            assert maxerr(got, ref) <= 1e-4, (P, mode)
        fd = [f.to(dev) for f in feats]                # channels-last kernel (lane = channel) on transposed copies
        assert maxerr(ops.roi_align_multilevel(fd, rois.to(dev), [4, 8, 16, 32], P, nhwc=ops.to_nhwc(fd)), ref) <= 1e-4, P
    # a channel count that is not a multiple of 8 (one slice), ROIs beyond the table limits (> 8 samples per bin: direct
    # formula inside the kernel), ROIs hanging over every border, a one-pixel map
    f5 = [f[:, :5].contiguous() for f in feats]
    assert maxerr(ops.roi_align_multilevel([f.to(dev) for f in f5], rois.to(dev), [4, 8, 16, 32], 7),
                  cb.roi_align_multilevel(f5, rois, [4, 8, 16, 32], 7)) <= 1e-4
    big = torch.tensor([[0, -300.0, -200.0, 900.0, 800.0], [1, 0.0, 0.0, 256.0, 256.0], [1, 250.0, 250.0, 700.0, 262.0], [0, -50.0, 100.0, 20.0, 400.0],
                        [1, 3.0, 3.0, 3.5, 3.5], [0, 255.0, 255.0, 256.0, 256.0],
                        # unclipped boxes whose x samples ALL fall outside the map while some y samples do not, and vice versa
                        # (every bin is 0; the separable kernel once contracted an unwritten LDS tile here)
                        [0, 300.0, 40.0, 340.0, 120.0], [1, -90.0, 20.0, -30.0, 200.0], [0, 30.0, 290.0, 150.0, 330.0],
                        [1, 20.0, -80.0, 90.0, -20.0]])
    fd = [f.to(dev) for f in feats]
    tf = ops.to_nhwc(fd)
    assert all(torch.equal(t, f.permute(0, 2, 3, 1)) for t, f in zip(tf, fd))                     # the transposition itself
    for P in (7, 14):
        want = cb.roi_align_multilevel(feats, big, [4, 8, 16, 32], P)
        assert maxerr(ops.roi_align_multilevel(fd, big.to(dev), [4, 8, 16, 32], P), want) <= 1e-4, P
        assert maxerr(ops.roi_align_multilevel(fd, big.to(dev), [4, 8, 16, 32], P, nhwc=tf), want) <= 1e-4, P
    f5d = [f.to(dev) for f in f5]
    assert maxerr(ops.roi_align_multilevel(f5d, rois.to(dev), [4, 8, 16, 32], 7, nhwc=ops.to_nhwc(f5d)),
                  cb.roi_align_multilevel(f5, rois, [4, 8, 16, 32], 7)) <= 1e-4


def test_roi_align_four_channels_per_lane_is_the_one_channel_kernel(dev):
    """[r6] roi_align_nhwc4_kernel (a lane owns four consecutive channels: one 16-byte load per tap, the ROI's (C, P, P) block leaves
    as one contiguous run) against roi_align_nhwc_kernel (one channel per lane): same taps, same weights, same order of additions -
    bit for bit, for P = 7 and 14, channel counts 256 / 64 / 260 (a partial last group of 256), ROIs over every border, beyond the
    table limits and degenerate; and both against the host statement."""
    from ttdg_mgm_amd import _lib, ops
    cb = _cpu_backend()
    g = synth.gen(7510)
    rois = torch.tensor([[0, 1.0, 2.0, 30.0, 40.0], [1, 5.0, 5.0, 20.0, 12.0], [1, 0.0, 0.0, 63.0, 63.0], [0, -300.0, -200.0, 900.0, 800.0],
                         [1, 250.0, 250.0, 700.0, 262.0], [0, 10.0, 10.0, 10.0, 10.0], [1, 3.0, 3.0, 3.5, 3.5], [0, 300.0, 40.0, 340.0, 120.0],
                         [1, -90.0, 20.0, -30.0, 200.0]] + [[int(i % 2), float(3 * i), float(2 * i), float(3 * i + 20 + 5 * i), float(2 * i + 30 + 3 * i)] for i in range(40)])
    for C in (256, 64, 260):
        feats = [synth.normal(g, (2, C, 64 // s, 64 // s), 1.0) for s in (1, 2, 4, 8)]
        fd = [f.to(dev).contiguous(memory_format=torch.channels_last) for f in feats]
        views = ops.to_nhwc(fd)
        for P in (7, 14):
            out = {}
            for wide in (True, False):
                _lib.load().ttdg_debug_set_roi_align_sliced(2 | (0 if wide else 128))
                try:
                    out[wide] = ops.roi_align_multilevel(fd, rois.to(dev), [4, 8, 16, 32], P, nhwc=views)
                finally:
                    _lib.load().ttdg_debug_set_roi_align_sliced(2)
            assert torch.equal(out[True], out[False]), (C, P, maxerr(out[True], out[False]))
            assert maxerr(out[True], cb.roi_align_multilevel(feats, rois, [4, 8, 16, 32], P)) <= 1e-4, (C, P)


# ------------------------------------------------------------------------------------------- N3: HiPPI / U_sup
@pytest.mark.parametrize("name,sizes,seed,proj", cases.HIPPI_CASES)
def test_hippi_golden(dev, golden, name, sizes, seed, proj):
    """HiPPI.forward (multi_graph_matching.py:414-449) against the reference's own result on planted similarities."""
    from ttdg_mgm_amd.GModule.multi_graph_matching import HiPPI
    gold = golden("usup")
    W, U0 = cases.hippi_inputs(sizes, seed)
    h = HiPPI()
    V0 = h.power_step(W.to(dev), U0.to(dev))
    ref = gold[f"hippi_{name}_V0"]
    assert maxerr(V0, ref) <= 1e-5 * float(np.abs(ref).max())
    U = h(W.to(dev), U0.to(dev), torch.tensor(sizes), 32, projector=proj)
    if proj == "hungarian":
        assert maxerr(U, gold[f"hippi_{name}_U"]) == 0.0, h.last_iters
    else:
        # up to 50 projected power iterations at tau = 1/200: gate derived from the float64 run of the same iteration
        from oracle import gmodule as og
        U64 = og.hippi(W.double(), U0.double(), sizes, 32, projector=proj)
        derived_gate("HiPPI %s (%d iterations)" % (name, h.last_iters), U, torch.from_numpy(gold[f"hippi_{name}_U"]), U64)


@pytest.mark.parametrize("sizes,seed", [((22, 30, 26, 19), 1), ((40, 12, 33), 2), ((32, 32), 3), ((5,), 4)])
def test_hippi_one_step(dev, sizes, seed):
    """One projected power step from a random state against the oracle (ragged graphs, some with more nodes than the
    universe): the well-posed unit of the iteration."""
    from oracle import gmodule as og
    from ttdg_mgm_amd.GModule.multi_graph_matching import HiPPI
    g = synth.gen(9000 + seed)
    M = sum(sizes)
    W = torch.from_numpy(g.uniform(0, 1, size=(M, M)).astype(np.float32)) / M
    U0 = torch.from_numpy(g.uniform(0, 1, size=(M, 32)).astype(np.float32)) / 8      # V = O(1): V / tau = O(200), fp32 ulp 1e-5
    ref = og.hippi(W, U0, sizes, 32, max_iter=1)
    h = HiPPI(max_iter=1)
    U = h(W.to(dev), U0.to(dev), torch.tensor(sizes), 32)
    assert maxerr(U, ref) <= 2e-4


def test_hippi_bad_projector(dev):
    from ttdg_mgm_amd.GModule.multi_graph_matching import HiPPI
    W, U0 = cases.hippi_inputs((5, 6), 1)
    with pytest.raises(NameError):
        HiPPI()(W.to(dev), U0.to(dev), torch.tensor([5, 6]), 32, projector="nope")


@pytest.mark.parametrize("name,sizes,seed", cases.USUP_CASES)
def test_u_sup_forward_backward(dev, golden, name, sizes, seed):
    """U_sup.forward (:136-169) with the detached HiPPI target taken from the reference run: N, Sinkhorn(N), loss and every
    gradient against the reference; the free-running loss (rounding-driven edges) must be finite."""
    from ttdg_mgm_amd.GModule.multi_graph_matching import U_sup
    gold = golden("usup")
    m = U_sup(2, 32).to(dev).eval()
    m.load_state_dict(synth.usup_params(cases.USUP_PARAM_SEED), strict=True)
    nodes, labels = cases.usup_inputs(sizes, seed)
    nodes = [x.to(dev).requires_grad_() for x in nodes]
    labels = [l.to(dev) for l in labels]
    tr = {}
    loss = m(nodes, labels, forced_target=torch.from_numpy(gold[f"usup_{name}_target"]).to(dev), trace=tr)
    loss.backward()
    assert maxerr(tr["N"], gold[f"usup_{name}_N"]) <= TOL
    assert maxerr(tr["Us"], gold[f"usup_{name}_Us"]) <= TOL
    assert abs(float(loss.detach()) - float(gold[f"usup_{name}_loss"])) <= 1e-6
    for g, x in enumerate(nodes):
        assert maxerr(x.grad, gold[f"usup_{name}_dnode{g}"]) <= 1e-6
    assert maxerr(m.U.grad, gold[f"usup_{name}_d_U"]) <= 1e-6
    for k in ("linear_k.weight", "linear_v.weight", "linear_q.weight", "linear_final.weight", "linear_final.bias", "layer_norm.weight"):
        check_pgrad(gold, f"usup_{name}_d_Net_U.g_gene.{k}", getattr(m.Net_U.g_gene, k.split(".")[0]).__getattr__(k.split(".")[1]).grad, 1e-6)
    with torch.no_grad():
        free = m([x.detach() for x in nodes], labels)
    assert torch.isfinite(free) and m.matching.last_iters >= 1


def test_u_sup_label_matrix(dev):
    from ttdg_mgm_amd.GModule.multi_graph_matching import U_sup
    m = U_sup(3, 32).to(dev)
    la, lb = torch.tensor([1, 3, 0], device=dev), torch.tensor([3, 2], device=dev)
    W = m.label_matrix([la, lb]).cpu()
    oh = torch.cat([m.one_hot(la), m.one_hot(lb)]).cpu()
    assert torch.equal(W, oh @ oh.t())                  # label 0 wraps to the last class, as eye[x - 1] does
    with pytest.raises(IndexError):
        m.label_matrix([torch.tensor([4], device=dev)])


# ------------------------------------------------------------------------------------------- A12 on the device
@pytest.mark.parametrize("i", range(6))
def test_dice_e_s_measures_on_device_vs_reference_golden(dev, golden, i):
    """DiceEvaluator's device reductions (CUDA tensors) against the reference's own numpy functions (dice_metric.py:54-66,
    110-240; tests/golden/dice.npz): the same gates as the host test."""
    from ttdg_mgm_amd.evaluation import dice_tensor, enhanced_align_tensor, structure_measure_tensor
    gold = golden("dice")
    p, g = (torch.from_numpy(m).to(dev) for m in cases.dice_mask_pairs()[i])
    assert p.is_cuda
    assert abs(float(dice_tensor(p, g)) - float(gold[f"c{i}_dice"])) <= 1e-9
    assert abs(float(enhanced_align_tensor(p, g)) - float(gold[f"c{i}_ea"])) <= 1e-9
    assert abs(float(structure_measure_tensor(p, g)) - float(gold[f"c{i}_sm"])) <= 1e-6


def test_dice_evaluator_on_device_vs_reference_golden(dev, golden):
    """The evaluator's whole process/evaluate path on CUDA tensors: six golden mask pairs as six predictions of one image
    (score threshold, same-class best-of-GT, x100, means) against the reference numbers."""
    from ttdg_mgm_amd.evaluation import DiceEvaluator
    from ttdg_mgm_amd.modeling.structures import Boxes, Instances
    gold = golden("dice")
    pairs = cases.dice_mask_pairs()
    dd = [dict(image_id=i, annotations=[dict(category_id=0, mask=torch.from_numpy(g), bbox=torch.zeros(4))]) for i, (p, g) in enumerate(pairs)]
    ev = DiceEvaluator("golden_pairs", 0.9, dataset_dicts=dd)
    outs = []
    for i, (p, g) in enumerate(pairs):
        pm = torch.from_numpy(np.stack([p, ~p])).to(dev)
        outs.append({"instances": Instances(p.shape, pred_boxes=Boxes(torch.zeros(2, 4, device=dev)), scores=torch.tensor([0.95, 0.5], device=dev),
                                            pred_classes=torch.tensor([0, 0], device=dev), pred_masks=pm)})
    ev.process([{"image_id": i} for i in range(len(pairs))], outs)
    res = ev.evaluate()
    assert len(ev.dice_scores) == len(pairs)                       # the 0.5-score prediction of every image is dropped
    for k, key in (("dice", "Dice Coefficient"), ("ea", "Enhanced Alignment Metric"), ("sm", "Structural Similarity Metric")):
        want = 100.0 * float(np.mean([float(gold[f"c{i}_{k}"]) for i in range(len(pairs))]))
        assert abs(res[key] - want) <= 1e-4, (key, res[key], want)


@pytest.mark.parametrize("H,W", [(96, 80), (33, 17), (64, 61), (512, 512), (1, 7)])
def test_mask_pair_counts_kernel_exact(dev, H, W):
    """ttdg_mask_pair_counts against the same twelve counts in plain torch: vectorised (W % 4 == 0) and byte paths, cuts at
    the borders, on a word boundary and inside a word, pairs that share a prediction."""
    from ttdg_mgm_amd import ops
    from ttdg_mgm_amd.evaluation import quadrant_counts
    g = synth.gen(H * 1000 + W)
    P = torch.from_numpy(g.uniform(size=(5, H, W)) < 0.4)
    G = torch.from_numpy(g.uniform(size=(3, H, W)) < 0.6)
    Pd, Gd = P.to(dev).contiguous(), G.to(dev).contiguous()
    pairs = [(0, 0, 0, 0), (1, 1, H, W), (2, 2, H // 2, W // 2), (3, 0, min(H, 5), min(W, 3)), (4, 1, 1, 4 if W >= 4 else 1), (0, 2, H - 1, W - 1)]
    cnt = ops.mask_pair_counts([Pd[a].data_ptr() for a, _, _, _ in pairs], [Gd[b].data_ptr() for _, b, _, _ in pairs],
                               [c for _, _, c, _ in pairs], [c for _, _, _, c in pairs], H, W, dev).cpu().tolist()
    for row, (a, b, cy, cx) in zip(cnt, pairs):
        assert row == quadrant_counts(P[a], G[b], cy, cx), (a, b, cy, cx)


@pytest.mark.parametrize("H,W", [(96, 80), (33, 17), (512, 512)])
def test_mask_measures_kernel_equals_the_float64_closed_forms(dev, H, W):
    """ttdg_mask_measures (float64 closed forms of Dice / E-measure / S-measure from the twelve counts + the maximum over a
    prediction's same-class ground truths, one thread per pair) against evaluation.measures_from_counts on the SAME counts
    (torch float64 on the CPU - itself held to dice_metric.py's numpy functions by tests/test_host.py): to 1e-12 relative.
    Edge cases: empty prediction, empty / full ground truth, cuts at 0, at the border and beyond it, one-pixel quadrants,
    a prediction with three ground truths (maximum), a prediction with none (stays 0)."""
    from ttdg_mgm_amd import ops
    from ttdg_mgm_amd.evaluation import measures_from_counts
    g = synth.gen(H * 77 + W)
    P = torch.from_numpy(g.uniform(size=(6, H, W)) < 0.4)
    G = torch.from_numpy(g.uniform(size=(5, H, W)) < 0.6)
    P[4] = False                                   # empty prediction
    G[3], G[4] = False, True                       # empty and full ground truth
    yy, xx = np.mgrid[0:H, 0:W]
    P[5] = torch.from_numpy((yy - H / 2) ** 2 + (xx - W / 2) ** 2 < (min(H, W) / 3) ** 2)      # a disc against ...
    G[2] = torch.from_numpy((yy - H / 2 - 2) ** 2 + (xx - W / 2 + 1) ** 2 < (min(H, W) / 3) ** 2)      # ... a shifted disc (high scores)
    Pd, Gd = P.to(dev).contiguous(), G.to(dev).contiguous()
    # (prediction, ground truth, cy, cx, owner)
    pairs = [(0, 0, 0, 0, 0), (1, 1, H, W, 1), (2, 2, H // 2, W // 2, 2), (3, 0, min(H, 5), min(W, 3), 3), (4, 1, 1, 1, 4), (0, 3, H - 1, W - 1, 5),
             (1, 4, H // 3, W // 3, 6), (5, 2, H // 2, W // 2, 7), (5, 0, H // 2 + 1, W // 2, 7), (5, 1, 1, W, 7), (4, 3, H + 3, W + 9, 8), (2, 4, H, 0, 9)]
    npred = 11                                     # owner 10 has no pair
    best = torch.zeros(npred, 3, dtype=torch.float64, device=dev)
    counts = ops.mask_pair_measures([Pd[a].data_ptr() for a, *_ in pairs], [Gd[b].data_ptr() for _, b, *_ in pairs], [p[2] for p in pairs],
                                    [p[3] for p in pairs], [p[4] for p in pairs], H, W, best)
    want = measures_from_counts(counts.cpu(), H, W, [p[2] for p in pairs], [p[3] for p in pairs]) * 100
    ref = torch.zeros(npred, 3, dtype=torch.float64)
    for row, p in zip(want, pairs):
        ref[p[4]] = torch.maximum(ref[p[4]], row)
    got = best.cpu()
    assert float(got[10].abs().max()) == 0.0
    assert float(got[7, 0]) > 60.0                                                             # the two discs
    # a one-pixel quadrant (cut at row 1 / column 1) has no sample covariance: NaN in the reference's S-measure, in the host
    # statement and - propagated through the maximum, as torch.maximum does - on the device
    assert torch.equal(torch.isnan(got), torch.isnan(ref)) and bool(torch.isnan(ref[4, 2])) and int(torch.isnan(ref).sum()) <= 2
    ok = ~torch.isnan(ref)
    err = (got[ok] - ref[ok]).abs() / ref[ok].abs().clamp(min=1.0)
    assert float(err.max()) <= 1e-12, (err.max(), got, ref)
    with pytest.raises(TypeError):
        ops.mask_pair_measures([Pd[0].data_ptr()], [Gd[0].data_ptr()], [1], [1], [0], H, W, torch.zeros(1, 3, device=dev))


# ------------------------------------------------------------------------------------------- N4: loader on the device
@pytest.mark.parametrize("shape,min_size,max_size", [((3, 512, 512), 800, 1333), ((3, 384, 384), 384, 1333), ((3, 640, 480), 320, 1333),
                                                     ((3, 300, 500), 800, 1333), ((1, 97, 131), 211, 260), ((3, 800, 800), 511, 1333)])
def test_device_resize_matches_host_mapper(dev, shape, min_size, max_size):
    """csrc/resize.hip against the host mapper (data.map_for_test = F.interpolate on float32, antialiased when shrinking, round,
    uint8): never more than 1 LSB apart, and equal on all but a sliver of pixels (the taps are summed in another order, so a
    value that lands within rounding of x.5 may go the other way)."""
    from ttdg_mgm_amd import data, ops
    g = synth.gen(shape[1] * 7 + shape[2])
    img = torch.from_numpy(g.integers(0, 256, size=(2,) + shape, dtype=np.uint8))
    # smooth content too: half of the batch is a blurred ramp (x.5 cases are common on flat gradients)
    yy, xx = np.mgrid[0:shape[1], 0:shape[2]]
    img[1] = torch.from_numpy(((yy * 3 + xx * 2) % 256).astype(np.uint8))[None].expand(shape[0], -1, -1)
    h, w = shape[1], shape[2]
    nh, nw = data.mapped_size(h, w, min_size, max_size)
    host = torch.stack([data.map_for_test(dict(image=img[k], height=h, width=w, image_id=k, annotations=[]), min_size, max_size)["image"] for k in range(2)])
    got = ops.resize_u8(img.to(dev), nh, nw).cpu()
    assert got.shape == host.shape and got.dtype == torch.uint8
    d = (got.int() - host.int()).abs()
    frac = float((d > 0).float().mean())
    print("resize %s -> %s: max |d| %d LSB, %.2e of the pixels differ" % (shape, (nh, nw), int(d.max()), frac))
    assert int(d.max()) <= 1 and frac <= 2e-3


# ------------------------------------------------------------------------------------------- what was actually asserted
def test_statement_ledger():
    """Must run last.  Asserts how often the strong branch of every data-dependent test fired (full-file runs only: a
    `-k` selection skips), and leaves the ledger in gpurun_out/parity_ledger.json."""
    import json
    import os
    print("parity ledger:", dict(LEDGER))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_ledger.json"), "w") as f:
        json.dump(dict(LEDGER), f, indent=1, sort_keys=True)
    nplanted = len(ALL_PLANTED)
    if LEDGER["planted_solver.identical_permutations_asserted"] == 0 and LEDGER["large_solver.cases"] == 0:
        pytest.skip("solver tests were not part of this selection")
    # every planted golden: the reference's permutation matrices, twice (solver alone, and free-running end to end)
    assert LEDGER["planted_solver.identical_permutations_asserted"] == nplanted
    assert LEDGER["planted_e2e.identical_permutations_asserted"] == nplanted
    assert LEDGER["planted_solver.hungarian_count_exact"] >= nplanted - 2          # +-1 tolerated on at most two
    # BASELINE.json cfg-3 at full size (8 x 256): the reference's permutation matrices, twice
    assert LEDGER["cfg3_planted.identical_permutations_asserted"] == len(cases.PLANTED_CFG3_CASES)
    assert LEDGER["cfg3_planted_e2e.identical_permutations_asserted"] == len(cases.PLANTED_CFG3_CASES)
    # multi-workgroup solver vs its host-driven statement on RANDOM inputs (the rounding-chaotic regime; its reference
    # permutations are pinned by the planted cases pb_n132 / pb_12x30 above): the Sinkhorn-stage counts are compared on
    # every case, and the strong end-state statement must have been made at least `LARGE_MIN_IDENTICAL` times
    assert LEDGER["large_solver.cases"] == 6
    assert LEDGER["large_solver.sinkhorn_stage_counts_compared"] >= LARGE_MIN_STAGE_COUNTS
    assert LEDGER["large_solver.identical_permutations"] >= LARGE_MIN_IDENTICAL
    assert LEDGER["large_solver.hungarian_step_identical"] >= 1


LARGE_MIN_STAGE_COUNTS, LARGE_MIN_IDENTICAL = 18, 1       # recorded: 21 stage counts compared, 1 case free of capped stages and identical (profiles/r03_parity_ledger.json)
