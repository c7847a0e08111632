"""GA-MGM solver at Mode S sizes: G graphs of ~30 nodes (G = 4 ... 32), single-workgroup kernel vs multi-workgroup solver.
    python tools/bench_gagm_scale.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from ttdg_mgm_amd import _lib, ops, synth  # noqa: E402
import cases  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
for G in (4, 8, 16, 32):
    g = synth.gen(40 + G)
    sizes = tuple(int(x) for x in g.integers(22, 36, size=G))
    A, W, U0 = cases.gagm_inputs(sizes, 900 + G)
    off, blocks = 0, []
    for n in sizes:
        blocks.append(A[off:off + n, off:off + n].reshape(-1))
        off += n
    apack = torch.cat(blocks).to(dev)
    W, U0 = W.to(dev), U0.to(dev)
    gr = ops.graphs(sizes)
    row = []
    for name, var in (("single-workgroup", _lib.GAGM_FORCE_SINGLE), ("multi-workgroup", _lib.GAGM_FORCE_LARGE)):
        cfg = ops.gagm_cfg(variant=var)
        U, info, _ = ops.gagm_solve(apack, W, U0, gr, sizes, cfg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            U, info, _ = ops.gagm_solve(apack, W, U0, gr, sizes, cfg)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        it = info.cpu().tolist()
        iters = sum(it[:6])
        row.append((name, dt * 1e3, iters, dt / max(iters, 1) * 1e6, float(U.sum())))
    print("G=%2d M=%4d " % (G, sum(sizes)) + "  ".join("%s %.2f ms (%d it, %.1f us/it, |U|=%d)" % r for r in row), flush=True)
