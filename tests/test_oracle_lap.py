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


# ---- the uniqueness certificate of csrc/lap_certified.h (restated in oracle/lap_certificate.py), checked by brute force ----------
def _all_assignments(nr, nc):
    import itertools
    return itertools.permutations(range(nc), nr)


def _brute(c):
    nr, nc = c.shape
    costs = sorted((float(sum(c[i, p[i]] for i in range(nr))), p) for p in _all_assignments(nr, nc))
    return costs


@pytest.mark.parametrize("seed", range(6))
def test_lap_certificate_is_sound_by_brute_force(seed):
    """The claim the multi-workgroup solver's Hungarian stage rests on (lap_certified.h): an ACYCLIC tight-entry graph proves that
    the assignment is the UNIQUE optimum - then it is what scipy returns, whatever its tie rules.  Small instances (3 x 3 ... 4 x 6),
    real-valued and tie-laden integer costs, duals from a shortest-path solver AND deliberately damaged duals / assignments:
    whenever the certificate says yes, enumeration of all assignments must agree (strict gap to the runner-up), and it must agree
    with scipy; on generic costs the certificate must say yes (otherwise it would be useless), on all-ties costs no."""
    from oracle import lap_certificate as lc
    rng = np.random.default_rng(100 + seed)
    yes = no = yes_generic = 0
    for trial in range(120):
        nr = int(rng.integers(3, 5))
        nc = int(rng.integers(nr, 7))
        kind = trial % 4
        if kind == 0:
            c = rng.normal(size=(nr, nc))                                   # generic
        elif kind == 1:
            c = rng.integers(0, 3, size=(nr, nc)).astype(np.float64)        # ties everywhere
        elif kind == 2:
            c = np.repeat(rng.normal(size=(1, nc)), nr, axis=0) + 1e-9 * rng.integers(0, 2, size=(nr, nc))   # near-constant rows
        else:
            c = rng.normal(size=(nr, nc)).astype(np.float32).astype(np.float64)
            c[:, -1] = c[:, 0]                                              # a duplicated column (a duplicated node)
        m, u, v = lc.ssp_duals(c)
        variants = [(m, v)]
        v2 = v.copy(); v2[rng.integers(nc)] -= abs(rng.normal())          # a damaged dual
        variants.append((m, v2))
        m3 = m.copy(); m3[0], m3[1] = m3[1], m3[0]                          # a (usually) non-optimal assignment with the true duals
        variants.append((m3, v))
        ranked = _brute(c)
        for mm, vv in variants:
            if lc.certificate(c, mm, vv):
                yes += 1
                assert tuple(int(x) for x in mm) == ranked[0][1], "certified an assignment that is not optimal"
                assert ranked[1][0] - ranked[0][0] > 0.0, "certified an optimum that is not unique"
                r, cc = scipy.optimize.linear_sum_assignment(c)
                assert np.array_equal(cc, mm)
            else:
                no += 1
        if kind == 0:
            assert lc.certificate(c, m, v), "generic costs with the solver's own duals must certify"
            yes_generic += 1
        if kind == 1 and ranked[1][0] == ranked[0][0]:
            assert not lc.certificate(c, m, v)
    assert yes >= 30 and no >= 60 and yes_generic == 30, (yes, no, yes_generic)


def test_lap_certificate_on_solver_sized_blocks():
    """32 x 64 ... 32 x 256 blocks of generic fp32 values (the shape of one graph's Hungarian-stage projection): the shortest-path
    duals certify, and the certified assignment is scipy's; a block of identical columns (every node the same) never certifies."""
    from oracle import lap_certificate as lc
    rng = np.random.default_rng(7)
    for nc in (64, 130, 256):
        c = -rng.normal(size=(32, nc)).astype(np.float32).astype(np.float64)
        m, u, v = lc.ssp_duals(c)
        assert lc.certificate(c, m, v)
        assert np.array_equal(scipy.optimize.linear_sum_assignment(c)[1], m)
        same = np.repeat(rng.normal(size=(32, 1)), nc, axis=1)
        m2, _, v2 = lc.ssp_duals(same)
        assert not lc.certificate(same, m2, v2)


# ---- [r5] the integer statement of the scipy-order LAP for narrow-range blocks (csrc/lap_device.h: lap_wave_solve_int) -------------
def _scipy_slots(V):
    """What the reference does with an (n x 32) block (utils/hungarian.py:34-63): node of every universe slot."""
    r, c = scipy.optimize.linear_sum_assignment(np.asarray(V, np.float32) * -1)
    out = np.full(32, -1, np.int64)
    out[c] = r
    return out


def _narrow_blocks():
    g = synth.gen(9500)
    for n in (33, 40, 64, 100, 256):
        for kind in range(6):
            base = np.float32(2.0 ** g.integers(-10, 4)) * (1.0 + g.random())
            ulp = np.spacing(np.float32(base))
            if kind == 0:      # what follows a collapsed Sinkhorn stage: node values a few thousand ulp apart, +- 1 ulp along the universe
                V = base + ulp * (g.integers(0, 4000, size=(n, 1)) + g.integers(0, 2, size=(n, 32)))
            elif kind == 1:    # every entry equal: scipy's tie rules alone
                V = np.full((n, 32), base)
            elif kind == 2:    # a handful of levels
                V = base + ulp * g.integers(0, 3, size=(n, 32))
            elif kind == 3:    # generic inside the admitted range
                V = base + ulp * g.integers(0, 100000, size=(n, 32))
            elif kind == 4:    # identical rows of the transposed problem (cost independent of the slot), distinct nodes
                V = np.repeat(base + ulp * g.permutation(n)[:, None], 32, axis=1)
            else:              # negative values
                V = -(base + ulp * g.integers(0, 500, size=(n, 32)))
            yield n, kind, V.astype(np.float32)


def test_integer_lap_statement_is_scipy_on_admitted_blocks():
    from oracle import lap_int
    done = 0
    for n, kind, V in _narrow_blocks():
        C = lap_int.admit(V)
        assert C is not None, (n, kind)
        assert C.min() == 0 and C.max() < (1 << lap_int.RANGE_BITS)
        assert np.array_equal(lap_int.solve(C), _scipy_slots(V)), (n, kind)
        done += 1
    assert done == 30


def test_integer_lap_admission_declines_what_the_argument_does_not_cover():
    from oracle import lap_int
    g = synth.gen(9501)
    V = (1.0 + 1e-4 * g.random((64, 32))).astype(np.float32)
    assert lap_int.admit(V) is not None
    W = V.copy(); W[3, 5] = 0.0
    assert lap_int.admit(W) is None                       # zero
    W = V.copy(); W[3, 5] = np.float32("inf")
    assert lap_int.admit(W) is None
    W = V.copy(); W[3, 5] = np.float32(1e-30)
    assert lap_int.admit(W) is None                       # exponent span
    W = (1.0 + 0.5 * g.random((64, 32))).astype(np.float32)
    assert lap_int.admit(W) is None                       # range: 2^22 quanta
    W = V.copy(); W[3, 5] = np.float32(1e-39)
    assert lap_int.admit(W) is None                       # denormal
