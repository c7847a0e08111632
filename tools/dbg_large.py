"""Debug: native large-graph solver vs the host-driven statement, one step at a time along the host-driven trajectory."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden")
import torch
from ttdg_mgm_amd import synth, ops
from ttdg_mgm_amd.GModule import MGM3_unsup
dev = torch.device("cuda:0")
sizes = (256,) * 8
nodes, labels = synth.node_sets(5, sizes, scale=0.5)
params, U = synth.mgm3_params(6), synth.universe(7)
m = MGM3_unsup(2, 32).to(dev).eval(); m.load_state_dict(params)
tr = {}
with torch.no_grad():
    m([x.to(dev) for x in nodes], [l.to(dev) for l in labels], U.to(dev), trace=tr)
print("native info", tr["info"].cpu().tolist()[:8])
ap, W, U0 = tr["apack"], tr["Wds"], tr["U0"]
gr = ops.graphs(sizes)
st = []
Uh, info_h, V0h = ops.gagm_solve_hostloop(ap, W, U0, list(sizes), ops.gagm_cfg(), states=st)
print("host info", info_h.cpu().tolist()[:8])
for k, (hung, tau, Ub, Ua, V) in enumerate(st):
    Un, Vn = ops.gagm_one_step(ap, W, Ub.contiguous(), gr, list(sizes), None if hung else tau)
    dv = float((Vn - V).abs().max()); du = float((Un - Ua).abs().max())
    print(k, "hung" if hung else "tau=%g" % tau, "dV %.2e (|V| %.2e)  dU %.2e  rows differing %d" % (dv, float(V.abs().max()), du, int(((Un - Ua).abs().sum(1) > 1e-3).sum())))
