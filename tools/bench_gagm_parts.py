"""Cycle counts of the in-kernel projections (Sinkhorn / LAP) in isolation: one wavefront per graph, LDS-resident V."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from ttdg_mgm_amd import _lib, synth
from ttdg_mgm_amd._lib import call, ptr, stream
dev = torch.device("cuda:0")
g = synth.gen(3)
for G, n in ((1, 30), (4, 30), (4, 22), (2, 25), (4, 40)):
    V = torch.from_numpy(g.uniform(0, 1, size=(G * n, 32)).astype(np.float32)).to(dev)
    U = torch.empty_like(V); ticks = torch.zeros(1, dtype=torch.int64, device=dev)
    for mode, name in ((0, "sinkhorn x20 sweeps"), (1, "lap")):
        reps = 50
        call("ttdg_debug_project", ptr(V), n, G, 0.05, 20, reps, mode, ptr(U), ptr(ticks), stream())
        torch.cuda.synchronize()
        print("G=%d n=%d %-20s %8.0f cycles per projection" % (G, n, name, ticks.item() / reps))
