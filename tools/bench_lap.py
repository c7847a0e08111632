"""A/B micro-benchmark of the on-device LAP (one wavefront per matrix): well-separated vs degenerate (near-constant) costs."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from ttdg_mgm_amd import ops, synth, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
g = synth.gen(1)
def mats(kind, b, r, c):
    if kind == "random":
        return torch.from_numpy(g.standard_normal((b, r, c)).astype(np.float32))
    base = g.uniform(0.05, 0.051, size=(b, r, 1)).astype(np.float32)              # rows nearly constant (collapsed U)
    return torch.from_numpy((base + g.standard_normal((b, r, c)).astype(np.float32) * 1e-8).astype(np.float32))
for kind in ("random", "degenerate"):
    for (r, c) in ((30, 32), (32, 32), (32, 60)):
        s = mats(kind, 4, r, c).to(dev)
        for variant in (0, 2, 0, 2):
            lib.ttdg_debug_set_lap_variant(variant)
            ops.lap_batched(s); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(50): ops.lap_batched(s)
            torch.cuda.synchronize()
            print("%-10s %dx%d variant %d: %.1f us per batch of 4" % (kind, r, c, variant, (time.perf_counter() - t) / 50 * 1e6))
lib.ttdg_debug_set_lap_variant(0)
