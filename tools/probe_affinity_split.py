"""Pairwise forward kernel at cfg-3-like shapes: does the placement (576 equal tiles on 256 CUs are dealt 2 or 3 per CU) cost a quarter of its time?
The same kernel with the hidden dimension split over blockIdx.z (ksplit planes): shorter workgroups, dealt dynamically."""
import sys
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd import ops, synth
dev = torch.device("cuda:0")
for G, n in ((8, 256), (8, 200), (6, 256), (16, 128), (4, 256), (12, 64)):
    H = 512
    sizes = [n] * G; M = G * n; gr = ops.graphs(sizes); g = synth.gen(33)
    P, Q = synth.normal(g, (M, H), 0.3).to(dev), synth.normal(g, (M, H), 0.3).to(dev)
    w2 = synth.normal(g, (H,), 0.05).to(dev)
    for ks in (1, 2, 4):
        fn = lambda: ops.affinity_pairwise_fwd(P, Q, w2, gr, ks)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        print("G %2d n %3d  ksplit %d: %7.1f us" % (G, n, ks, a.elapsed_time(b) * 1e3 / 20), flush=True)
