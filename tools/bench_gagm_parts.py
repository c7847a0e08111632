"""Cycle counts of the in-kernel projections (Sinkhorn / LAP) in isolation: one wavefront per graph, LDS-resident V; also
prints how far the block-layout Sinkhorn projector is from the round-1 one on the same input (sanity, not a parity test)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from ttdg_mgm_amd import _lib, synth
from ttdg_mgm_amd._lib import call, ptr, stream
dev = torch.device("cuda:0")
g = synth.gen(3)
MODES = ((0, "sinkhorn x20 sweeps (block layout)"), (2, "sinkhorn x20, round 1"), (1, "lap"))
for G, n in ((1, 30), (4, 30), (4, 22), (4, 32), (4, 33), (4, 38), (4, 48), (4, 64)):
    V = torch.from_numpy(g.uniform(0, 1, size=(G * n, 32)).astype(np.float32)).to(dev)
    ticks = torch.zeros(1, dtype=torch.int64, device=dev)
    for tau in (0.05, 0.00625):
        outs = {}
        for mode, name in MODES:
            if mode == 1 and tau != 0.05:
                continue
            reps = 50
            U = torch.zeros_like(V)
            call("ttdg_debug_project", ptr(V), n, G, tau, 20, reps, mode, ptr(U), ptr(ticks), stream())
            torch.cuda.synchronize()
            outs[mode] = U
            print("G=%d n=%d tau=%.5f %-36s %8.0f cycles per projection" % (G, n, tau, name, ticks.item() / reps))
        print("      max |block - round1| %.2e" % float((outs[0] - outs[2]).abs().max()))
