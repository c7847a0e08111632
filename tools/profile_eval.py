"""Time the phases of one TTA step and one eval pass on the GPU (torch.cuda events)."""
import sys, time
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd import data
from ttdg_mgm_amd.config import get_cfg
from ttdg_mgm_amd.engine import BaselineTrainer
from ttdg_mgm_amd.modeling import calibrate_frozen_bn
from ttdg_mgm_amd.modeling.detector import detector_postprocess
from ttdg_mgm_amd.evaluation import DiceEvaluator

cl = "--channels-last" in sys.argv
cfg = get_cfg(); cfg.merge_from_file("configs/test_segment.yaml"); cfg.DATASETS.TEST = ["pe"]
data.register_synthetic("pe", 8)
torch.manual_seed(0)
model = BaselineTrainer.build_model(cfg); model.teacher_forced = True
BaselineTrainer.device = torch.device("cuda:0")
batches = list(BaselineTrainer.build_test_loader(cfg, "pe"))
calibrate_frozen_bn(model, batches[0])
if cl:
    model.backbone.to(memory_format=torch.channels_last)
opt = BaselineTrainer.build_optimizer(cfg, model)

def T(fn, n=3):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r

b = batches[0]
model.train()
for _ in range(2): BaselineTrainer.tta_step(model, opt, b)
print("tta step            %.1f ms" % T(lambda: BaselineTrainer.tta_step(model, opt, b))[0])
def fwd():
    images = model.preprocess_image(b); return images, model._backbone(images.tensor)
ms, (images, feats) = T(fwd); print("  preprocess+backbone fwd %.1f ms" % ms)
def fb():
    images = model.preprocess_image(b); f = model._backbone(images.tensor); sum(v.sum() for v in f.values()).backward()
print("  backbone fwd+bwd    %.1f ms" % T(fb)[0])
ms, (props, _) = T(lambda: model.proposal_generator(images, feats, None, compute_loss=False)); print("  rpn (train topk)    %.1f ms" % ms)
ms, (dets, _) = T(lambda: model.roi_heads(images, feats, props, None, compute_loss=False, branch="TTT")); print("  box head (TTT)      %.1f ms" % ms)
model.eval()
with torch.no_grad():
    print("eval inference      %.1f ms" % T(lambda: model(b))[0])
    ms, (images, feats) = T(fwd); print("  backbone fwd        %.1f ms" % ms)
    ms, (props, _) = T(lambda: model.proposal_generator(images, feats, None, compute_loss=False)); print("  rpn (test topk)     %.1f ms" % ms)
    ms, (res, _) = T(lambda: model.roi_heads(images, feats, props, None, compute_loss=False, branch="")); print("  box+mask heads      %.1f ms" % ms)
    ms, _ = T(lambda: [detector_postprocess(r, 512, 512) for r in res]); print("  postprocess (paste) %.1f ms" % ms)
    ev = DiceEvaluator("pe", 0.9)
    outs = model(b)
    ms, _ = T(lambda: (ev.process(b, outs), ev.evaluate())); print("  dice process        %.1f ms" % ms)
