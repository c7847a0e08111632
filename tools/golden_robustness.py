"""How robust is the reference permutation of each planted golden against rounding-sized changes of Wds?  (CPU, oracle only.)
For every planted case the oracle's front end gives A, Wds, U0; the solver is then re-run on Wds + noise, noise uniform in
[-eps, eps] on the non-zero entries (eps = 2e-6: the size of the difference between two equally accurate fp32 pair-stage
kernels), and the result compared with the unperturbed permutation.   usage: golden_robustness.py [trials] [eps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import cases  # noqa: E402
from oracle import gmodule as og  # noqa: E402


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    eps = float(sys.argv[2]) if len(sys.argv) > 2 else 2e-6
    torch.set_num_threads(8)
    names = [c[0] for c in cases.PLANTED_CASES + cases.PLANTED_BIG_CASES]
    for name in names:
        params, nodes, labels, U, sizes = cases.mgm_inputs(name)
        tr = {}
        og.mgm3_unsup_forward(params, nodes, labels, U, trace=tr)
        A, W, U0, Ub, it0 = tr["A"], tr["Wds"], tr["U0"], tr["Ub"], tr["iters"]
        g = torch.Generator().manual_seed(7)
        flips, its = 0, []
        for _ in range(trials):
            noise = (torch.rand(W.shape, generator=g) * 2 - 1) * eps
            t = {}
            Up = og.gagm(A, W + noise * (W != 0), U0, list(sizes), trace=t)
            flips += int(not torch.equal(Up, Ub))
            its.append(t["iters"][4])
        print("%-9s sizes %-28s stage iterations %s  permutation changed in %d / %d trials (tau=0.00625 stage took %d..%d iterations)"
              % (name, sizes, it0, flips, trials, min(its), max(its)), flush=True)


if __name__ == "__main__":
    main()
