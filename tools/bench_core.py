"""Micro-benchmark of the matching core (MGM3_unsup fwd+bwd) on the GPU; used with rocprofv3 --stats."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden")
import torch
import cases
from ttdg_mgm_amd import synth
from ttdg_mgm_amd.GModule import MGM3_unsup

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "p4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if which.startswith("rand"):
    sizes = tuple(int(x) for x in which[4:].split("x")) if len(which) > 4 else (30, 27, 33, 25)
    nodes, labels = synth.node_sets(5, sizes, scale=0.5)
    params, U = synth.mgm3_params(6), synth.universe(7)
else:
    params, nodes, labels, U, sizes = cases.mgm_inputs(which)
m = MGM3_unsup(2, 32).to(dev).eval()
m.load_state_dict(params)
dn = [x.to(dev).requires_grad_() for x in nodes]
dl = [l.to(dev) for l in labels]
Ud = U.to(dev)
for _ in range(3):
    tr = {}
    loss = m(dn, dl, Ud, trace=tr); loss.backward()
torch.cuda.synchronize()
info = tr["info"].cpu().tolist()
print("sizes", sizes, "gagm iters", info[:7], "loss", float(loss))
if max(sizes) > 128 and any(info[12:14]):
    print("multi-workgroup solver, Hungarian stage: %d LAPs with a uniqueness certificate, %d by the scipy-order solver" % (info[12], info[13]))
t0 = time.perf_counter()
for _ in range(reps):
    loss = m(dn, dl, Ud); loss.backward()
torch.cuda.synchronize()
print("fwd+bwd ms/step: %.3f" % ((time.perf_counter() - t0) / reps * 1e3))
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(reps):
        loss = m(dn, dl, Ud)
torch.cuda.synchronize()
print("fwd only ms/step: %.3f" % ((time.perf_counter() - t0) / reps * 1e3))
