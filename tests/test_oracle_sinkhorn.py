"""oracle/sinkhorn_spec.py (the stand-in for the absent pygmtools==0.3.8) against the reference tree's own
log-Sinkhorn (fixture tests/golden/sinkhorn_ref.npz <- graph_matching.py:828-839) and against properties.
What stays unpinned against the real package is listed in the spec's header."""
import numpy as np
import pytest
import torch

from oracle import gmodule as og
from oracle.sinkhorn_spec import sinkhorn
from ttdg_mgm_amd import synth


import cases
from oracle.sinkhorn_spec import log_sinkhorn


@pytest.mark.parametrize("name", [c[0] for c in cases.SKREF_CASES])
@pytest.mark.parametrize("batched", [False, True])
def test_spec_matches_reference_tree_log_sinkhorn(golden, name, batched):
    """Log domain, fp32, both code paths of the spec: <= 4 ulp of the largest log-domain magnitude (the reference function
    and the spec run the same torch.logsumexp sequence; with dummy rows the spec's bookkeeping differs, the arithmetic not)."""
    ref = torch.from_numpy(golden("sinkhorn_ref")[name + "_log"])
    s, tau = cases.skref_input(name)
    b, r, c = s.shape
    got = log_sinkhorn(s, dummy_row=r < c, max_iter=2 * cases.SKREF_SWEEPS, tau=tau, batched_operation=batched)
    assert got.shape == (b, r, c)
    ulp = float((s / tau).abs().max()) * 2.0 ** -23
    err = float((got - ref[:, :r]).abs().max())
    print(name, "log-domain |d| = %.3e (ulp of max|s/tau| = %.3e)" % (err, ulp))
    assert err <= max(4 * ulp, 1e-6)
    assert float((got.exp() - ref[:, :r].exp()).abs().max()) <= 1e-6


def test_ragged_batch_equals_reference_on_the_valid_blocks(golden):
    """n1 masking (gagm's unequal-size branch): every matrix of a zero-padded batch equals the pinned stand-alone problem."""
    ref = torch.from_numpy(golden("sinkhorn_ref")["dm_t10_log"])
    s, tau = cases.skref_input("dm_t10")                      # (4, 22, 32)
    pad = torch.zeros(4, 30, 32)
    pad[:, :22] = s
    sizes = torch.tensor([22, 22, 22, 22])
    got = log_sinkhorn(pad, n1=sizes, dummy_row=True, max_iter=20, tau=tau, batched_operation=True)
    assert float((got[:, :22] - ref[:, :22]).abs().max()) <= 1e-5 and bool(torch.isinf(got[:, 22:]).all())


def test_transposed_problem_equals_reference(golden):
    """rows > cols: the spec solves the transposed problem (Appendix B step 1) - pinned through the fixture's transpose."""
    ref = torch.from_numpy(golden("sinkhorn_ref")["dm_t05_log"])          # (1, 25, 25), real rows 18
    s, tau = cases.skref_input("dm_t05")                                   # (1, 18, 25)
    got = log_sinkhorn(s.transpose(1, 2).contiguous(), dummy_row=True, max_iter=20, tau=tau)
    assert float((got.transpose(1, 2) - ref[:, :18]).abs().max()) <= 1e-5


def rnd(seed, shape, scale=1.0):
    return synth.normal(synth.gen(seed), shape, scale)


def test_square_rows_cols_sum_to_one():
    s = rnd(1, (3, 16, 16))
    for bo in (False, True):
        p = sinkhorn(s, max_iter=200, tau=0.5, batched_operation=bo)
        assert torch.allclose(p.sum(1), torch.ones(3, 16), atol=1e-4)
        assert torch.allclose(p.sum(2), torch.ones(3, 16), atol=1e-4)


def test_even_iters_end_on_column_normalisation():
    p = sinkhorn(rnd(2, (9, 14)), dummy_row=True, max_iter=20, tau=0.05)
    assert p.shape == (9, 14)
    assert float(p.min()) >= 0 and float(p.max()) <= 1
    assert bool((p.sum(0) <= 1 + 1e-5).all())          # dummy rows carry the rest of each column
    q = sinkhorn(rnd(2, (9, 14)), dummy_row=True, max_iter=19, tau=0.05)   # odd count ends on a row sweep
    assert torch.allclose(q.sum(1), torch.ones(9), atol=1e-5)


def test_shift_invariance():
    s = rnd(3, (7, 7))
    a = sinkhorn(s, max_iter=20, tau=0.3)
    b = sinkhorn(s + rnd(4, (7, 1)), max_iter=20, tau=0.3)          # first sweep is a row sweep: exact
    assert torch.allclose(a, b, atol=1e-5)
    a = sinkhorn(s, max_iter=400, tau=0.3)
    b = sinkhorn(s + rnd(4, (7, 1)) + rnd(5, (1, 7)), max_iter=400, tau=0.3)   # both shifts: converged limit
    assert torch.allclose(a, b, atol=1e-4)


def test_transposed_input_is_transposed_back():
    s = rnd(6, (12, 5))
    a = sinkhorn(s, dummy_row=True, max_iter=20, tau=0.1)
    b = sinkhorn(s.t().contiguous(), dummy_row=True, max_iter=20, tau=0.1).t()
    assert a.shape == (12, 5) and torch.allclose(a, b, atol=1e-6)


def test_batched_equals_per_matrix_and_padding_is_zero():
    blocks = [rnd(10 + i, (n, 32)) for i, n in enumerate((12, 30, 7))]
    st = torch.stack(og.pad_tensor(blocks))
    a = sinkhorn(st, n1=torch.tensor([12, 30, 7]), dummy_row=True, max_iter=20, tau=0.1, batched_operation=True)
    for i, blk in enumerate(blocks):
        n = blk.shape[0]
        b = sinkhorn(blk, dummy_row=True, max_iter=20, tau=0.1)
        assert torch.allclose(a[i, :n], b, atol=1e-5)
        assert float(a[i, n:].abs().sum()) == 0


def test_per_matrix_transpose_when_rows_exceed_cols():
    # unequal graphs with max n > 32: whole batch is transposed, small graphs re-transposed (Appendix B step 3)
    sizes = (22, 40, 35)
    blocks = [rnd(20 + i, (n, 32)) for i, n in enumerate(sizes)]
    st = torch.stack(og.pad_tensor(blocks))
    a = sinkhorn(st, n1=torch.tensor(sizes), dummy_row=True, max_iter=20, tau=0.1, batched_operation=True)
    for i, blk in enumerate(blocks):
        n = blk.shape[0]
        b = sinkhorn(blk, dummy_row=True, max_iter=20, tau=0.1)   # 2-D: auto-transposed when n > 32
        assert torch.allclose(a[i, :n], b, atol=1e-5), i


def test_small_tau_approaches_hungarian():
    s = rnd(30, (10, 10))
    p = sinkhorn(s, max_iter=400, tau=0.005)
    h = og.hungarian(s)
    assert torch.equal((p > 0.5).float(), h)


def test_gradient_matches_finite_difference():
    s = rnd(31, (4, 6)).double().requires_grad_()
    w = rnd(32, (4, 6)).double()
    f = lambda x: (sinkhorn(x, dummy_row=True, max_iter=20, tau=0.5) * w).sum()
    assert torch.autograd.gradcheck(f, (s,), eps=1e-6, atol=1e-5)
