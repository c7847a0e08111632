"""Mask head (four 3 x 3 filters + 2 x 2 stride-2 deconvolution + 1 x 1 predictor) at the ROI counts of the bench's Dice pass: time per layer, vendor."""
import sys
sys.path.insert(0, ".")
import torch, torch.nn.functional as F
import ttdg_mgm_amd  # noqa
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
head = det.MaskRCNNConvUpsampleHead(2).to(dev).eval()
with torch.no_grad():
    for R in (4, 8, 16, 32, 64, 128):
        x = torch.randn(R, 256, 14, 14, device=dev).contiguous(memory_format=torch.channels_last)
        w = head.mask_fcn1.weight
        t3 = timeit(lambda: F.conv2d(x, w, None, 1, 1))
        y = F.conv2d(x, w, None, 1, 1)
        td = timeit(lambda: F.conv_transpose2d(y, head.deconv.weight, None, 2))
        z = F.conv_transpose2d(y, head.deconv.weight, None, 2)
        tp = timeit(lambda: head.predictor(z))
        th = timeit(lambda: head(x))
        gf = 2.0 * R * 196 * 2304 * 256 / 1e6
        print("R %3d: 3 x 3 filter %6.1f us (%5.1f TF) | deconvolution %6.1f us | predictor %5.1f us | whole head %7.1f us" % (R, t3, gf / t3, td, tp, th), flush=True)
