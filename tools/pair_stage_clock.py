"""Diagnostics: where does ttdg_pair_stage_fwd spend its time?  In-kernel shader clock (s_memtime) per workgroup at the phase
boundaries (begin / affinity block done / 20 sweeps done / end), plus HIP-event time of the launch.  usage: pair_stage_clock.py [sizes]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ttdg_mgm_amd import _lib, ops, synth  # noqa: E402


def main():
    sizes = tuple(int(x) for x in sys.argv[1].split("x")) if len(sys.argv) > 1 else (21, 22, 38, 38)
    dev = torch.device("cuda:0")
    G, M, H = len(sizes), sum(sizes), 512
    g = synth.gen(11)
    P, Q = synth.normal(g, (M, H), 0.3).to(dev), synth.normal(g, (M, H), 0.3).to(dev)
    w2, b2 = synth.normal(g, (H,), 0.05).to(dev), torch.tensor([0.03], device=dev)
    gr = ops.graphs(sizes)
    npairs = G * (G + 1) // 2
    prof = torch.zeros(npairs * 4, dtype=torch.int64, device=dev)
    lib = _lib.load()
    for _ in range(3):
        ops.pair_stage_fwd(P, Q, w2, b2, gr, list(sizes), 0.05, 20)
    lib.ttdg_debug_set_pair_stage_profile(prof.data_ptr())
    ops.pair_stage_fwd(P, Q, w2, b2, gr, list(sizes), 0.05, 20)
    torch.cuda.synchronize()
    lib.ttdg_debug_set_pair_stage_profile(None)
    t = prof.cpu().view(npairs, 4)
    print("sizes", sizes, "shader clocks per workgroup: affinity / sweeps / epilogue / total")
    for k in range(npairs):
        b, a, s, e = t[k].tolist()
        print("  pair %2d: %7d %7d %7d %8d" % (k, a - b, s - a, e - s, e - b))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.pair_stage_fwd(P, Q, w2, b2, gr, list(sizes), 0.05, 20)
    e1.record()
    torch.cuda.synchronize()
    print("launch, back to back: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 50))
    aff, Wds, pot = ops.pair_stage_fwd(P, Q, w2, b2, gr, list(sizes), 0.05, 20)
    dW = torch.randn(M, M, device=dev)
    e0.record()
    for _ in range(50):
        ops.pair_stage_bwd(aff, b2, pot, dW, gr, 0.05, 20)
    e1.record()
    torch.cuda.synchronize()
    print("backward launch, back to back: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 50))
    part = ops.affinity_pairwise_fwd(P, Q, w2, gr, 16)
    e0.record()
    for _ in range(50):
        part = ops.affinity_pairwise_fwd(P, Q, w2, gr, 16)
        ops.sinkhorn_pairs_fwd(part, b2, gr, list(sizes), 0.05, 20)
    e1.record()
    torch.cuda.synchronize()
    print("two-launch form (affinity ksplit 16 + sinkhorn_pairs_fwd): %.1f us" % (e0.elapsed_time(e1) * 1e3 / 50))


if __name__ == "__main__":
    main()
