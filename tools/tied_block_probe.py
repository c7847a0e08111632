"""Diagnostics (not product): the Hungarian-stage block behind the collapsed Sinkhorn stages of BASELINE cfg-3 with random weights.
Runs the Sinkhorn stages of the solve (cfg.max_stages = 5), then ONE Hungarian iteration from that state, and prints how constant the
state U and the resulting V are along the universe index - the structure the step-by-step scipy-order LAP spends 1.2 M cycles on.
usage: tied_block_probe.py [sizes e.g. 256x256x256x256x256x256x256x256]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden")
import numpy as np
import torch
from ttdg_mgm_amd import _lib, ops, synth
from ttdg_mgm_amd.GModule import MGM3_unsup

dev = torch.device("cuda:0")
sizes = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "256x256x256x256x256x256x256x256").split("x"))
nodes, labels = synth.node_sets(5, sizes, scale=0.5)
params, U = synth.mgm3_params(6), synth.universe(7)
m = MGM3_unsup(2, 32).to(dev).eval()
m.load_state_dict(params)
tr = {}
with torch.no_grad():
    m([x.to(dev) for x in nodes], [l.to(dev) for l in labels], U.to(dev), trace=tr)
gr = ops.graphs(list(sizes))
Us, info, _ = ops.gagm_solve(tr["apack"], tr["Wds"], tr["U0"], gr, list(sizes), ops.gagm_cfg(max_stages=5))
print("Sinkhorn stages:", info.cpu().tolist()[:8])
Uc = Us.cpu().numpy()
o = 0
for g, n in enumerate(sizes):
    blk = Uc[o:o + n]
    print("graph %d: U min %.9g max %.9g  distinct values %d  (1/n = %.9g)" % (g, blk.min(), blk.max(), len(np.unique(blk)), 1.0 / n))
    o += n
Un, V = ops.gagm_one_step(tr["apack"], tr["Wds"], Us, gr, list(sizes), None, variant=_lib.GAGM_FORCE_LARGE)
Vc = V.cpu().numpy()
o = 0
for g, n in enumerate(sizes):
    blk = Vc[o:o + n]
    spread = (blk.max(1) - blk.min(1))
    ulp = np.spacing(np.abs(blk).max(1).astype(np.float32))
    nd = np.array([len(np.unique(r)) for r in blk])
    srt = np.sort(blk.mean(1))[::-1]
    gap = np.diff(srt[:34]).__abs__()
    print("graph %d: V rows: max spread along the universe %.3g (%.1f ulp), rows bit-constant %d of %d, distinct values per row max %d; "
          "node-value gaps among the 34 largest: min %.3g (%.1f ulp)" % (g, spread.max(), (spread / ulp).max(), int((nd == 1).sum()), n, nd.max(),
                                                                          gap.min(), gap.min() / ulp.max()))
    o += n
np.save("gpurun_out/tied_block_V.npy", Vc)
np.save("gpurun_out/tied_block_U.npy", Uc)
