"""ROIPooler at the box head's shape (4 x 256-channel FPN maps of an 800 x 800 batch, ~4000 proposals, 7 x 7): one channel per lane
(roi_align_nhwc_kernel, round 2-5) against four channels per lane (roi_align_nhwc4_kernel, round 6)."""
import sys
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd import _lib, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, C = 4, 256
feats = [torch.randn(B, C, 800 // s, 800 // s, device=dev).contiguous(memory_format=torch.channels_last) for s in (4, 8, 16, 32)]
views = ops.to_nhwc(feats)
def rois(n, spread):
    ctr = torch.rand(n, 2, device=dev) * 200 + 300 if not spread else torch.rand(n, 2, device=dev) * 700 + 50
    wh = torch.rand(n, 2, device=dev) * 120 + 30
    img = torch.arange(n, device=dev).float().div(n / B).floor()
    return torch.cat([img[:, None], ctr - wh / 2, ctr + wh / 2], 1).contiguous()
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n, spread, P in ((4000, False, 7), (4000, True, 7), (2000, False, 7), (400, False, 14)):
    r = rois(n, spread)
    row = []
    for mode in (2 | 128, 2):
        _lib.load().ttdg_debug_set_roi_align_sliced(mode)
        row.append(timed(lambda: ops.roi_align_multilevel(feats, r, [4, 8, 16, 32], P, nhwc=views)))
    _lib.load().ttdg_debug_set_roi_align_sliced(2)
    print("R=%d %s P=%d: one channel per lane %.1f us, four channels per lane %.1f us (x%.2f)" % (n, "spread" if spread else "clustered", P, row[0], row[1], row[0] / row[1]), flush=True)
