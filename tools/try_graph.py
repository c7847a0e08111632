import sys, time
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd import data
from ttdg_mgm_amd.config import get_cfg
from ttdg_mgm_amd.engine import BaselineTrainer
from ttdg_mgm_amd.modeling import calibrate_frozen_bn
cfg = get_cfg(); cfg.merge_from_file("configs/test_segment.yaml"); cfg.DATASETS.TEST = ["pe"]
data.register_synthetic("pe", 8)
torch.manual_seed(0)
model = BaselineTrainer.build_model(cfg); model.teacher_forced = True
BaselineTrainer.device = torch.device("cuda:0")
batches = list(BaselineTrainer.build_test_loader(cfg, "pe"))
calibrate_frozen_bn(model, batches[0])
model.train()
images = model.preprocess_image(batches[0])
x = images.tensor
class Wrap(torch.nn.Module):
    def __init__(s, bb): super().__init__(); s.bb = bb
    def forward(s, x):
        f = s.bb(x); return tuple(f[k] for k in ("p2", "p3", "p4", "p5", "p6"))
w = Wrap(model.backbone)
def run(fn, n=5):
    for _ in range(2):
        outs = fn(x); sum(o.sum() for o in outs).backward()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        outs = fn(x); sum(o.sum() for o in outs).backward()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("eager  fwd+bwd %.2f ms" % run(w))
g = torch.cuda.make_graphed_callables(w, (x.clone().requires_grad_(False),))
print("graphed fwd+bwd %.2f ms" % run(g))
