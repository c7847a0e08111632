"""Property tests pinning oracle/sinkhorn_spec.py (the stand-in for the absent
pygmtools==0.3.8; PARITY UNPINNED against the real package — SURVEY.md §8c)."""
import numpy as np
import pytest
import torch

from oracle import gmodule as og
from oracle.sinkhorn_spec import sinkhorn
from ttdg_mgm_amd import synth


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
