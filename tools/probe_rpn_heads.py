"""RPN head (3 x 3 filter + the two 1 x 1 heads) on the five levels of the bench batch (4 x 800 x 800), whole forward of modeling.detector.RPNHead:
packed path (one streaming product per level, bias + ReLU at the operand fetch) against round 5's (bias + ReLU pass, two vendor convolutions + bias)."""
import sys
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd.modeling import detector as det
dev = "cuda:0"


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


torch.manual_seed(0)
head = det.RPNHead().to(dev).eval()
feats = [torch.randn(4, 256, h, h, device=dev).contiguous(memory_format=torch.channels_last) for h in (200, 100, 50, 25, 13)]
with torch.no_grad():
    for rep in range(2):
        det.FUSED_RPN_HEADS = True
        tp = timeit(lambda: head(feats))
        det.FUSED_RPN_HEADS = False
        tv = timeit(lambda: head(feats))
        det.FUSED_RPN_HEADS = True
        print("RPNHead.forward over five levels: packed heads %.1f us, bias + ReLU pass + two vendor convolutions %.1f us" % (tp, tv), flush=True)
    for x in feats:
        det.FUSED_RPN_HEADS = True
        tp = timeit(lambda: head([x]))
        det.FUSED_RPN_HEADS = False
        tv = timeit(lambda: head([x]))
        det.FUSED_RPN_HEADS = True
        print("  level %3d x %3d: packed %.1f us, round 5 %.1f us" % (x.shape[-2], x.shape[-1], tp, tv), flush=True)
