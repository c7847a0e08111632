"""Steady-state rate of the pointwise product: long-K and zero-filled operands (DVFS probe)."""
import sys
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, K, N) in ((40000, 512, 128), (40000, 4096, 128), (4096, 4096, 4096), (160000, 256, 256), (8192, 2048, 2048)):
    for fill in ("randn", "zeros", "uniform"):
        x = {"randn": torch.randn, "zeros": torch.zeros, "uniform": lambda *a, **k: torch.rand(*a, **k) * 2 - 1}[fill](M, K, device=dev)
        w = {"randn": torch.randn, "zeros": torch.zeros, "uniform": lambda *a, **k: torch.rand(*a, **k) * 2 - 1}[fill](N, K, device=dev)
        out = torch.empty(M, N, device=dev)
        for tile in (4, 2, 1):
            t = timed(lambda: ops.mm(x, w, out, M, N, K, K, K, N, tile=tile))
            print("M=%6d K=%5d N=%5d %-8s tile %d: %8.1f us  %6.1f TF" % (M, K, N, fill, tile, t, 2.0 * M * K * N / t / 1e6), flush=True)
        if fill == "randn":
            xt = x.view(1, M, 1, K).permute(0, 3, 1, 2)
            wt = w.view(N, 1, 1, K).permute(0, 3, 1, 2)
            t = timed(lambda: torch.nn.functional.conv2d(xt, wt))
            print("   vendor conv2d (NHWC 1x1): %8.1f us  %6.1f TF" % (t, 2.0 * M * K * N / t / 1e6), flush=True)
            t = timed(lambda: torch.mm(x, w.t()))
            print("   torch.mm (hipBLASLt):     %8.1f us  %6.1f TF" % (t, 2.0 * M * K * N / t / 1e6), flush=True)
