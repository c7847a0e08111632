"""One pointwise-product shape, a few launches (for rocprofv3 --pmc): probe_pointwise.py M K N tile [res] [reps]"""
import sys
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd import ops
M, K, N, tile = (int(a) for a in sys.argv[1:5])
res = len(sys.argv) > 5 and sys.argv[5] == "1"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) / K ** 0.5, torch.randn(N, device=dev)
r = torch.randn(M, N, device=dev) if res else None
out = torch.empty(M, N, device=dev)
for _ in range(reps):
    ops.mm(x, w, out, M, N, K, K, K, N, bias=b, res=r, ldres=N, relu=True, tile=tile)
torch.cuda.synchronize()
