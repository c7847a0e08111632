"""TEST INFRASTRUCTURE (see oracle/__init__.py).  numpy restatement of the INTEGER scipy-order LAP of the multi-workgroup solver's
Hungarian stage (csrc/lap_device.h: lap_wave_solve_int, csrc/lap_certified.h: lap_int_admit) - the path a block takes when no uniqueness
certificate exists and its values span a narrow range (what follows a collapsed Sinkhorn stage; reference multi_graph_matching.py:326-328
-> utils/hungarian.py:34-63 -> scipy.optimize.linear_sum_assignment on the transposed 32 x n problem).

The argument the device relies on, restated so that the CPU tests can check it against scipy itself:
  * admission: every entry a normal float32, binary exponents spanning <= 6, (max - min) / 2^(emin - 150) < 2^17.  Then every entry is
    an integer multiple of q = 2^(emin - 150) below 2^30 q and scipy's float64 arithmetic on the block is exact;
  * shifting all costs by a constant changes no decision, so the solver runs on c' = (max V - V) / q >= 0 (minimisation of -V);
  * the scan's arg-min - smallest shortest-path value, then scipy's tie rule (some minimal column unassigned -> the LAST such position
    of `remaining`, else the FIRST minimal position) - is the minimum of ONE integer key
        (min(spc, 2^19 - 1) << 11) | (1023 - pos  if unassigned else  1024 + pos).
`solve` below is scipy's successive-shortest-path algorithm (rectangular_lsap.cpp, as restated in oracle/lap.c) with exactly these
two substitutions."""
import numpy as np

RANGE_BITS = 17
CAP = (1 << 19) - 1
BIG = 1 << 30


def admit(V):
    """V: (n, 32) float32 block, n > 32.  -> (32, n) int64 shifted costs, or None where the device declines."""
    V = np.ascontiguousarray(V, np.float32)
    bits = V.view(np.int32).astype(np.int64)
    e = (bits >> 23) & 0xff
    if np.any(e == 0) or np.any(e == 255):
        return None
    emin = int(e.min())
    if int(e.max()) - emin > 6:
        return None
    m = ((bits & 0x7fffff) | 0x800000) << (e - emin)
    iv = np.where(bits < 0, -m, m)
    if int(iv.max()) - int(iv.min()) >= (1 << RANGE_BITS):
        return None
    return (int(iv.max()) - iv).T.copy()            # rows = universe slots, columns = nodes


def solve(C):
    """C: (32, n) non-negative int64 costs.  -> col4row (32,) as scipy.optimize.linear_sum_assignment returns it."""
    nr, nc = C.shape
    u = np.zeros(nr, np.int64)
    v = np.zeros(nc, np.int64)
    col4row = np.full(nr, -1, np.int64)
    row4col = np.full(nc, -1, np.int64)
    for cur in range(nr):
        pos = nc - 1 - np.arange(nc)                 # position of every column in scipy's `remaining`; -1 once scanned
        spc = np.full(nc, BIG, np.int64)
        path = np.full(nc, -1, np.int64)
        SR = np.zeros(nr, bool)
        minval, i, sink, nrem = 0, cur, -1, nc
        while sink == -1:
            SR[i] = True
            act = pos >= 0
            r = np.where(act, minval - u[i] + (C[i] - v), BIG)
            better = r < spc
            path[better] = i
            spc = np.minimum(spc, r)
            code = np.where(row4col < 0, 1023 - pos, 1024 + pos)
            key = np.where(act, (np.minimum(spc, CAP) << 11) | code, np.iinfo(np.int64).max)
            gkey = int(key.min())
            assert gkey < (1 << 31), "the key must fit a signed 32-bit word"
            minval = gkey >> 11
            c = gkey & 2047
            selpos = 1023 - c if c < 1024 else c - 1024
            nrem -= 1
            j = int(np.nonzero(pos == selpos)[0][0])
            pos[pos == nrem] = selpos                # the last position moves into the hole (a no-op when the hole is the last one)
            pos[j] = -1
            if row4col[j] < 0:
                sink = j
            else:
                i = int(row4col[j])
        assert minval < (1 << (RANGE_BITS + 1))
        scanned = pos < 0
        u[cur] += minval
        for k in np.nonzero(SR)[0]:
            if k != cur:
                u[k] += minval - spc[col4row[k]]
        v[scanned] -= minval - spc[scanned]
        j = sink
        while True:
            k = int(path[j])
            row4col[j] = k
            col4row[k], j = j, col4row[k]
            if k == cur:
                break
    return col4row
