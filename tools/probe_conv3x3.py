"""Steady-state rate of the vendor's 3 x 3 convolution (fp32, NHWC, stride 1, pad 1; MIOpen with the shipped find-db) on the
shapes of the bench's batch (4 x 800 x 800): forward, data gradient, weight gradient, each alone, 20 launches between two events.
The figure an own implicit-product kernel has to beat per shape.  python tools/probe_conv3x3.py [own]"""
import sys
sys.path.insert(0, ".")
import torch, torch.nn.functional as F
import ttdg_mgm_amd  # noqa: F401  (stages the find-db)

dev = "cuda:0"
SHAPES = [("res2.conv2", 4, 200, 200, 64, 64), ("res3.conv2", 4, 100, 100, 128, 128), ("res4.conv2", 4, 50, 50, 256, 256), ("res5.conv2", 4, 25, 25, 512, 512),
          ("fpn/rpn p2", 4, 200, 200, 256, 256), ("fpn/rpn p3", 4, 100, 100, 256, 256), ("fpn/rpn p4", 4, 50, 50, 256, 256), ("fpn/rpn p5", 4, 25, 25, 256, 256),
          ("rpn p6", 4, 13, 13, 256, 256), ("mask head 100 rois", 100, 14, 14, 256, 256), ("mask head 400 rois", 400, 14, 14, 256, 256)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


own = len(sys.argv) > 1 and sys.argv[1] == "own"
if own:
    from ttdg_mgm_amd import ops
print("%-22s %28s %10s %8s | %10s %8s | %10s %8s" % ("layer", "N x H x W x Cin -> Cout", "fwd us", "TF", "dX us", "TF", "dW us", "TF"))
for name, n, h, w, ci, co in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(n, ci, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    g = torch.randn(n, co, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    gf = 2.0 * n * h * w * 9 * ci * co / 1e6     # MFLOP -> us * TF
    tf = timeit(lambda: F.conv2d(x, wt, None, 1, 1))
    tdx = timeit(lambda: torch.ops.aten.convolution_backward(g, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    tdw = timeit(lambda: torch.ops.aten.convolution_backward(g, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    line = "%-22s %28s %10.1f %8.1f | %10.1f %8.1f | %10.1f %8.1f" % (name, "%d x %d x %d x %d -> %d" % (n, h, w, ci, co), tf, gf / tf, tdx, gf / tdx, tdw, gf / tdw)
    if own:
        to = timeit(lambda: ops.conv3x3(x, wt))
        line += " | own fwd %8.1f us %6.1f TF" % (to, gf / to)
    print(line, flush=True)
