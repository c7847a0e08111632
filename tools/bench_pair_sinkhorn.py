"""Diagnostics (not product): the pair-stage Sinkhorn kernels alone at BASELINE cfg-3 size (8 graphs x 256 nodes -> 36 forward / 28
backward blocks of 256 x 256), HIP-event timing per launch for two sweep counts - the difference is the cost of the sweeps, the rest
is load / initialisation / output.   usage: bench_pair_sinkhorn.py [G=8] [n=256] [reps=20]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ttdg_mgm_amd import ops, synth  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = torch.device("cuda:0")
    sizes = [n] * G
    M = sum(sizes)
    g = synth.gen(3)
    part = synth.normal(g, (1, M, M), 0.1).to(dev).contiguous()
    b2 = torch.tensor([0.03], device=dev)
    dW = synth.normal(g, (M, M), 1.0).to(dev)
    gr = ops.graphs(sizes)
    out = {}
    for iters in (2, 20):
        W, pot = ops.sinkhorn_pairs_fwd(part, b2, gr, sizes, 0.05, iters)
        out["fwd_%d" % iters] = timed(lambda: ops.sinkhorn_pairs_fwd(part, b2, gr, sizes, 0.05, iters), reps)
        out["fwd_nopot_%d" % iters] = timed(lambda: ops.sinkhorn_pairs_fwd(part, b2, gr, sizes, 0.05, iters, want_pot=False), reps)
        out["bwd_%d" % iters] = timed(lambda: ops.sinkhorn_pairs_bwd(part, b2, pot, dW, gr, 0.05, iters), reps)
    print("G %d n %d: " % (G, n) + ", ".join("%s %.1f us" % kv for kv in out.items()))
    print("per sweep: fwd %.2f us, bwd %.2f us; fixed part: fwd %.1f us, bwd %.1f us" %
          ((out["fwd_20"] - out["fwd_2"]) / 18, (out["bwd_20"] - out["bwd_2"]) / 18, out["fwd_2"] - 2 * (out["fwd_20"] - out["fwd_2"]) / 18,
           out["bwd_2"] - 2 * (out["bwd_20"] - out["bwd_2"]) / 18))


if __name__ == "__main__":
    main()
