"""Which planted goldens sit on a rounding edge of the reference algorithm?  (CPU, oracle only.)  Runs the admission test of
tests/golden/admission.py - structured rounding-sized perturbations of the front end (float64, reversed summation order,
per-block +-2 ulp), the solve in float64, and relative noise multiplied into every Sinkhorn-stage projection - on every
planted case and prints what each survives.  make_golden.py applies the same test with the REFERENCE's solver and refuses to
write a case that fails.   usage: golden_robustness.py [case,case,...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import admission  # noqa: E402
import cases  # noqa: E402


def main():
    torch.set_num_threads(8)
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else [c[0] for c in cases.PLANTED_CASES + cases.PLANTED_BIG_CASES]
    for name in names:
        params, nodes, labels, U, sizes = cases.mgm_inputs(name)
        t0 = time.perf_counter()
        tr = admission.front(params, nodes, labels, U)
        ok, res = admission.check(params, nodes, labels, U, sizes)
        print("%-9s sizes %-28s stage iterations %-24s %s  (%.1f s)" % (name, sizes, tr["iters"], "admitted" if ok else
              "REFUSED: " + ", ".join(k for k, v in res.items() if not v), time.perf_counter() - t0), flush=True)


if __name__ == "__main__":
    main()
