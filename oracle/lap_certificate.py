"""TEST INFRASTRUCTURE (see oracle/__init__.py): numpy restatement of the uniqueness certificate of
ttdg-mgm_amd/csrc/lap_certified.h (step 3), which decides whether the warm-started workgroup LAP of the multi-workgroup GA-MGM
solver may stand in for scipy.optimize.linear_sum_assignment (reference utils/hungarian.py:34-63) - plus a plain successive-
shortest-path solver that returns the duals the certificate is fed with.  Minimisation form, nr <= nc, every row assigned.

    certificate(c, m, v):  u_i := c[i, m_i] - v[m_i],  S = max|c| + max|v|
      (i)   m injective, v <= 0, v = 0 on unmatched columns
      (ii)  rc_ij = c_ij - u_i - v_j >= -1e-12 S for every non-matched entry
      (iii) entries with rc_ij < 1e-9 S are TIGHT; directed graph on rows + one node F:  i -> i' if (i, m_i') is tight,
            i -> F if i has a tight entry in an unmatched column, F -> i if v[m_i] > -1e-9 S;  the graph must be acyclic.
    True  =>  m is the unique optimal assignment (gap ~1e-9 S): tests/test_oracle_lap.py checks that claim by brute force."""
import numpy as np


def ssp_duals(c):
    """Successive shortest augmenting paths (the algorithm family of scipy's solver, without its tie rules).
    -> (col_of_row, u, v) with c_ij - u_i - v_j >= 0, equality on the matching, v <= 0, v = 0 on unmatched columns."""
    c = np.asarray(c, dtype=np.float64)
    nr, nc = c.shape
    u, v = np.zeros(nr), np.zeros(nc)
    col4row, row4col = -np.ones(nr, dtype=np.int64), -np.ones(nc, dtype=np.int64)
    for cur in range(nr):
        spc = np.full(nc, np.inf)
        path = -np.ones(nc, dtype=np.int64)
        scanned = np.zeros(nc, dtype=bool)
        in_tree = np.zeros(nr, dtype=bool)
        i, min_val, sink = cur, 0.0, -1
        while sink < 0:
            in_tree[i] = True
            r = min_val + c[i] - u[i] - v
            better = (~scanned) & (r < spc)
            spc[better], path[better] = r[better], i
            cand = np.where(~scanned, spc, np.inf)
            j = int(np.argmin(cand))
            min_val = cand[j]
            scanned[j] = True
            if row4col[j] < 0:
                sink = j
            else:
                i = int(row4col[j])
        u[cur] += min_val
        for r_ in range(nr):
            if in_tree[r_] and r_ != cur:
                u[r_] += min_val - spc[col4row[r_]]
        v[scanned] -= min_val - spc[scanned]
        j = sink
        while True:
            r_ = int(path[j])
            row4col[j] = r_
            col4row[r_], j = j, col4row[r_]
            if r_ == cur:
                break
    return col4row, u, v


def certificate(c, m, v):
    c, v = np.asarray(c, dtype=np.float64), np.asarray(v, dtype=np.float64)
    m = np.asarray(m, dtype=np.int64)
    nr, nc = c.shape
    if not (np.all(np.isfinite(c)) and np.all(np.isfinite(v))):
        return False
    if np.any(m < 0) or np.any(m >= nc) or len(set(m.tolist())) != nr:
        return False
    matched = np.zeros(nc, dtype=bool)
    matched[m] = True
    if np.any(v > 0.0) or np.any(v[~matched] != 0.0):
        return False
    S = float(np.abs(c).max() + np.abs(v).max())
    if not S > 0.0:
        return False
    t_lo, t_hi = 1e-12 * S, 1e-9 * S
    u = c[np.arange(nr), m] - v[m]
    rc = c - u[:, None] - v[None, :]
    own = np.zeros_like(rc, dtype=bool)
    own[np.arange(nr), m] = True
    if np.any(rc[~own] < -t_lo):
        return False
    tight = (rc < t_hi) & ~own
    row_of_col = -np.ones(nc, dtype=np.int64)
    row_of_col[m] = np.arange(nr)
    F = nr
    adj = [set() for _ in range(nr + 1)]
    for i, j in zip(*np.nonzero(tight)):
        adj[i].add(int(row_of_col[j]) if row_of_col[j] >= 0 else F)
    for i in range(nr):
        if v[m[i]] > -t_hi:
            adj[F].add(i)
    alive = set(range(nr + 1))                      # peel sinks: acyclic <=> everything peels off
    while True:
        sinks = [n for n in alive if not (adj[n] & alive)]
        if not sinks:
            break
        alive -= set(sinks)
    return not alive
