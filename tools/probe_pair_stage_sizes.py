"""Does the fused pair stage (csrc/pair_stage.hip) lose more than the fp32 oracle on batches with a graph above ~40 nodes?  (Both fresh boxes that
failed the trajectory gate - r05 box 2, r06 box 6 - had 43 - 45-node graphs in the step where the device left the host walkers.)  Per size set and seed:
|Wds - fp64|, |dM - fp64| for the device and for the fp32 oracle, on random inputs AND on peaked inputs (scaled up: a trained affinity)."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import torch
from oracle import gmodule as og
from ttdg_mgm_amd import ops, synth
dev = torch.device("cuda:0")
H = 512


def run(sizes, seed, gain):
    G, M = len(sizes), sum(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)])
    g = synth.gen(seed)
    P, Q = synth.normal(g, (M, H), 0.3) * gain, synth.normal(g, (M, H), 0.3) * gain
    w2, b2 = synth.normal(g, (H,), 0.05), torch.tensor([0.03])
    Rw = synth.normal(g, (M, M), 1.0)
    mask, low = torch.zeros(M, M), torch.zeros(M, M)
    for a in range(G):
        for b in range(a + 1, G):
            mask[off[a]:off[a + 1], off[b]:off[b + 1]] = 1
            low[off[b]:off[b + 1], off[a]:off[a + 1]] = 1

    def host(dt):
        Pd, Qd, wd = P.to(dt), Q.to(dt), w2.to(dt)
        Mr = (torch.relu(Pd[:, None, :] + Qd[None, :, :]) * wd).sum(-1).detach().requires_grad_()
        W = torch.zeros(M, M, dtype=dt)
        for a in range(G):
            for b in range(a + 1):
                blk = Mr[off[a]:off[a + 1], off[b]:off[b + 1]] + b2.to(dt)
                ds = og.sinkhorn_pair(blk) if sizes[b] >= sizes[a] else og.sinkhorn_pair(blk.t()).t()
                W[off[a]:off[a + 1], off[b]:off[b + 1]] += ds
                if a != b:
                    W[off[b]:off[b + 1], off[a]:off[a + 1]] += ds.t()
        (W * (Rw * mask).to(dt)).sum().backward()
        return Mr, W
    M32, W32 = host(torch.float32)
    M64, W64 = host(torch.float64)
    gr = ops.graphs(sizes)
    aff, Wd, pot = ops.pair_stage_fwd(P.to(dev), Q.to(dev), w2.to(dev), b2.to(dev), gr, list(sizes), 0.05, 20)
    dM = ops.pair_stage_bwd(aff, b2.to(dev), pot, (Rw * mask).to(dev), gr, 0.05, 20)
    t = (M64.grad * low)
    e_dev = float((torch.where(low > 0, dM.cpu(), torch.zeros(())).double() - t).abs().max())
    e_ref = float(((M32.grad * low).double() - t).abs().max())
    w_dev, w_ref = float((Wd.cpu().double() - W64).abs().max()), float((W32.double() - W64).abs().max())
    return w_dev, w_ref, e_dev, e_ref, float(t.abs().max())


for sizes in ((22, 22, 22, 22), (35, 37, 38, 37), (22, 38, 21, 45), (44, 43, 23, 22), (22, 38, 21, 40), (22, 38, 21, 48), (22, 38, 21, 49), (64, 38, 21, 33)):
    for gain in (1.0, 3.0):
        rows = [run(sizes, 1000 + s, gain) for s in range(3)]
        print("%-18s gain %.0f: Wds dev/ref %.2e / %.2e | dM dev/ref %s (|dM| max %.1e)" % (
            sizes, gain, max(r[0] for r in rows), max(r[1] for r in rows),
            ", ".join("%.1e / %.1e" % (r[2], r[3]) for r in rows), max(r[4] for r in rows)), flush=True)

sizes = (35, 37, 38, 37)
g = synth.gen(5)
M = sum(sizes)
P, Q = synth.normal(g, (M, H), 0.3).to(dev), synth.normal(g, (M, H), 0.3).to(dev)
w2, b2 = synth.normal(g, (H,), 0.05).to(dev), torch.tensor([0.03]).to(dev)
gr = ops.graphs(sizes)
for _ in range(3):
    ops.pair_stage_fwd(P, Q, w2, b2, gr, list(sizes), 0.05, 20)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    ops.pair_stage_fwd(P, Q, w2, b2, gr, list(sizes), 0.05, 20)
b.record(); torch.cuda.synchronize()
print("pair_stage_fwd %s: %.1f us per call" % (sizes, a.elapsed_time(b) * 1e3 / 50))
