"""Fused multi-tensor SGD on the parameter set of the TTA step (27 M fp32 values): HBM roofline + PMC target."""
import sys
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd.optim import FusedSGD
dev = torch.device("cuda:0")
shapes = [(512, 256, 1, 1), (512, 512, 3, 3), (2048, 512, 1, 1), (1024, 1024, 3, 3)] * 3 + [(256, 256, 3, 3)] * 8 + [(512, 512), (512,), (256, 256)]
ps = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.01) for s in shapes]
n = sum(p.numel() for p in ps)
opt = FusedSGD([{"params": [p], "weight_decay": 1e-4} for p in ps], lr=0.005, momentum=0.9)
for p in ps: p.grad = torch.randn_like(p)
for _ in range(3): opt.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): opt.step()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e-3
print("params %.1f M, %.1f us/step, %.2f TB/s algorithmic (20 B/param), %.1f%% of 8 TB/s" % (n / 1e6, t * 1e6, n * 20 / t / 1e12, n * 20 / t / 8e12 * 100))
