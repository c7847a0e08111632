"""Pin oracle/lap.c (restatement of scipy's rectangular LSAP, incl. tie rules) against the installed scipy."""
import numpy as np
import pytest
import scipy.optimize

from oracle import lap as olap
from ttdg_mgm_amd import synth


def scipy_cols(cost, maximize):
    r, c = scipy.optimize.linear_sum_assignment(cost, maximize=maximize)
    out = np.full(cost.shape[0], -1, np.int64)
    out[r] = c
    return out


@pytest.mark.parametrize("shape", [(5, 32), (32, 32), (40, 32), (22, 35), (1, 7), (7, 1), (64, 64), (32, 95), (256, 32)])
@pytest.mark.parametrize("maximize", [False, True])
def test_random_float32_inputs(shape, maximize):
    for seed in range(5):
        g = synth.gen(9000 + seed)
        cost = g.standard_normal(shape).astype(np.float32).astype(np.float64)
        assert np.array_equal(olap.lap(cost, maximize), scipy_cols(cost, maximize))


@pytest.mark.parametrize("shape", [(6, 6), (12, 32), (32, 12), (40, 32), (32, 32), (20, 33)])
def test_tie_heavy_inputs(shape):
    # small-integer costs and exact-zero columns/rows: the outcome is decided by the tie rules only
    for seed in range(40):
        g = synth.gen(9100 + seed)
        cost = g.integers(0, 3, size=shape).astype(np.float64)
        if seed % 3 == 0:
            cost[:, g.integers(0, shape[1], size=shape[1] // 3)] = 0
        if seed % 4 == 0:
            cost[g.integers(0, shape[0], size=shape[0] // 3), :] = 0
        for maximize in (False, True):
            assert np.array_equal(olap.lap(cost, maximize), scipy_cols(cost, maximize)), (seed, maximize)


def test_constant_matrix_is_identity():
    assert np.array_equal(olap.lap(np.ones((9, 9))), np.arange(9))
    assert np.array_equal(scipy_cols(np.ones((9, 9)), False), np.arange(9))
