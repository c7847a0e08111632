"""Ablation of ttdg_mm_f32 (csrc/pointwise.hip): the same launch with (1) only the first K slab loaded, (2) no MFMAs, (3) neither - which of HBM/L2 traffic, the K loop and the fixed part (launch, first load, epilogue) a layer pays for.  Results of the ablated launches are WRONG by construction."""
import sys
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, K, N, res) in ((40000, 512, 128, 0), (40000, 128, 512, 1), (10000, 1024, 256, 0), (160000, 64, 256, 1), (160000, 256, 256, 0)):
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5, torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev)
    for tile in (4, 2, 1):
        row = []
        for dbg in (0, 1, 2, 3):
            t = timed(lambda: ops.mm(x, w, out, M, N, K, K, K, N, bias=b, res=r, ldres=N, relu=True, tile=tile + 256 * dbg))
            row.append(t)
        print("M=%6d K=%4d N=%4d res=%d tile %d: full %6.1f us | no loads after slab 0 %6.1f | no MFMA %6.1f | neither %6.1f   (%.1f TF)" % (M, K, N, res, tile, *row, 2.0 * M * K * N / row[0] / 1e6), flush=True)
