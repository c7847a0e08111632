"""Which convolutions of the stand-in backbone are slow?  Times every Conv2d forward (fp32, eval batch of 4 x 800 x 800)."""
import sys, time
sys.path.insert(0, ".")
import torch, torch.nn as nn, torch.nn.functional as F
from ttdg_mgm_amd.config import get_cfg
from ttdg_mgm_amd.engine import BaselineTrainer
cfg = get_cfg(); cfg.merge_from_file("configs/test_segment.yaml"); cfg.MODEL.DEVICE = "cuda:0"
torch.manual_seed(0)
model = BaselineTrainer.build_model(cfg).eval()
rec = []
def hook(name):
    def pre(m, a):
        torch.cuda.synchronize(); m._t0 = time.perf_counter()
    def post(m, a, o):
        torch.cuda.synchronize(); rec.append((time.perf_counter() - m._t0, name, tuple(a[0].shape), tuple(m.weight.shape), m.stride))
    return pre, post
for n, m in model.named_modules():
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        p, q = hook(n); m.register_forward_pre_hook(p); m.register_forward_hook(q)
x = torch.randn(4, 3, 800, 800, device="cuda:0")
with torch.no_grad():
    for it in range(3):
        rec.clear()
        f = model.backbone(x)
        model.proposal_generator.rpn_head([f[k] for k in ("p2", "p3", "p4", "p5", "p6")])
rec.sort(reverse=True)
tot = sum(r[0] for r in rec)
print("total conv wall (with sync overhead) %.2f ms over %d convs" % (tot * 1e3, len(rec)))
for t, n, xs, ws, st in rec[:14]:
    print("%7.3f ms  %-40s in %s  w %s stride %s" % (t * 1e3, n, xs, ws, st))
