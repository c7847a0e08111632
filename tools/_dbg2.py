import sys
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
M = 2048
for (N, K) in ((256, 256), (512, 256), (32, 256)):
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
    o1, o2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    t1 = timed(lambda: ops.gemm(x, K, 1, w, K, 1, o1, N, 1, M, N, K, bias=b))
    t2 = timed(lambda: ops.mm(x, w, o2, M, N, K, K, K, N, bias=b))
    print("fwd  x W^T  %dx%dx%d: gemm_f32 %.1f us, mm %.1f us, maxdiff %.2e" % (M, N, K, t1, t2, float((o1 - o2).abs().max())))
# dX = dP W  (M x 256 from K = 512)
dP, W1 = torch.randn(M, 512, device=dev), torch.randn(512, 512, device=dev)
o1, o2 = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
t1 = timed(lambda: ops.gemm(dP, 512, 1, W1, 1, 512, o1, 256, 1, M, 256, 512))
t2 = timed(lambda: ops.mm(dP, W1, o2, M, 256, 512, 512, 512, 256, b_layout=1))
print("dX = dP W1[:, :256]: gemm_f32 %.1f us, mm %.1f us, maxdiff %.2e" % (t1, t2, float((o1 - o2).abs().max())))
# dW = dP^T Xs (512 x 256, reduction 2048)
Xs = torch.randn(M, 256, device=dev)
o1, o2 = torch.empty(512, 512, device=dev), torch.empty(512, 512, device=dev)
t1 = timed(lambda: ops.gemm(dP, 1, 512, Xs, 1, 256, o1, 512, 1, 512, 256, M))
for ks in (0, 2, 4, 8):
    t2 = timed(lambda: ops.mm(dP, Xs, o2, 512, 256, M, 512, 256, 512, a_layout=1, b_layout=1, kslices=ks))
    print("dW1 = dP^T Xs: gemm_f32(splitk) %.1f us, mm ks=%d %.1f us, maxdiff %.2e" % (t1, ks, t2, float((o1[:, :256] - o2[:, :256]).abs().max())))
