"""Operator-level benchmark at BASELINE cfg-3 shapes (8 graphs x 256 nodes, d=256): per-kernel time and roofline
fractions of the hand-written kernels of the matching front end (affinity, pair Sinkhorn, GEMMs, loss, adjacency)."""
import sys, json
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd import ops, synth

dev = torch.device("cuda:0")
G, n, d, H = 8, 256, 256, 512
sizes = [n] * G
M = G * n
gr = ops.graphs(sizes)
g = synth.gen(33)
P, Q = synth.normal(g, (M, H), 0.3).to(dev), synth.normal(g, (M, H), 0.3).to(dev)
w2, b2 = synth.normal(g, (H,), 0.05).to(dev), torch.zeros(1, device=dev)
X = synth.normal(g, (M, d), 0.1).to(dev)
Wl = synth.normal(g, (H, d), 0.05).to(dev)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3, r


out = {}
pairs_ge = G * (G + 1) // 2
pairs_gt = G * (G - 1) // 2
# affinity forward: 4 FLOP per (i,j,k) (4 h n_i n_j), 3 VALU lane-ops
ks = ops.pick_ksplit(M, H, n)
t, part = timed(lambda: ops.affinity_pairwise_fwd(P, Q, w2, gr, ks))
fl = 4.0 * H * pairs_ge * n * n
out["affinity_fwd"] = dict(ms=t * 1e3, tflops=fl / t / 1e12, frac_fp32_peak=fl / t / 157.3e12,
                           lane_ops_frac=3.0 * H * pairs_ge * n * n / t / 78.6e12)
Wds, pot = ops.sinkhorn_pairs_fwd(part, b2, gr, sizes, 0.05, 20)
t, _ = timed(lambda: ops.sinkhorn_pairs_fwd(part, b2, gr, sizes, 0.05, 20))
out["sinkhorn_pairs_fwd"] = dict(ms=t * 1e3, algorithmic_GBps=8.0 * pairs_ge * n * n / t / 1e9, flops_T=5.0 * 20 * pairs_ge * n * n / t / 1e12)
dW = torch.rand(M, M, device=dev) * 1e-3
t, dM = timed(lambda: ops.sinkhorn_pairs_bwd(part, b2, pot, dW, gr, 0.05, 20))
out["sinkhorn_pairs_bwd"] = dict(ms=t * 1e3)
t, _ = timed(lambda: ops.affinity_pairwise_bwd(P, Q, w2, dM, gr))
flb = 2 * 4.0 * H * pairs_gt * n * n
out["affinity_bwd"] = dict(ms=t * 1e3, tflops=flb / t / 1e12, frac_fp32_peak=flb / t / 157.3e12,
                           lane_ops_frac=6.0 * H * pairs_gt * n * n / t / 78.6e12)
t, _ = timed(lambda: ops.linear_raw(X, Wl))
out["gemm_f32 (2048x256 @ 256x512)"] = dict(ms=t * 1e3, tflops=2.0 * M * d * H / t / 1e12, frac_fp32_peak=2.0 * M * d * H / t / 157.3e12)
U = torch.zeros(M, 32, device=dev); U[torch.arange(M), torch.arange(M) % 32] = 1
t, _ = timed(lambda: ops.perm_loss_fwd_bwd(Wds, U, gr, G))
out["perm_loss_fwd_bwd"] = dict(ms=t * 1e3, GBps=(8.0 * pairs_gt * n * n + 4.0 * M * M) / t / 1e9)
q, k = synth.normal(g, (M, d), 0.1).to(dev), synth.normal(g, (M, d), 0.1).to(dev)
t, _ = timed(lambda: ops.mha_adjacency(q, k, gr, sizes, d ** -0.5))
out["mha_adjacency"] = dict(ms=t * 1e3, tflops=2.0 * G * n * n * d / t / 1e12)
print(json.dumps(out, indent=1))
