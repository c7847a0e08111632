"""End-to-end TTA on synthetic fundus data (GPU): a few adaptation steps + the Dice pass, with timings."""
import sys, time
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd import data
from ttdg_mgm_amd.config import get_cfg
from ttdg_mgm_amd.engine import BaselineTrainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = get_cfg(); cfg.merge_from_file("configs/test_segment.yaml")
cfg.DATASETS.TEST = ["synthfundus_a"]
data.register_synthetic("synthfundus_a", n)
torch.manual_seed(0)
model = BaselineTrainer.build_model(cfg)
model.teacher_forced = "--free" not in sys.argv
opt = BaselineTrainer.build_optimizer(cfg, model)
BaselineTrainer.device = torch.device("cuda:0")
loader = BaselineTrainer.build_test_loader(cfg, "synthfundus_a")
from ttdg_mgm_amd.modeling import calibrate_frozen_bn
calibrate_frozen_bn(model, next(iter(loader)))
for i, inputs in enumerate(loader):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = BaselineTrainer.tta_step(model, opt, inputs)
    torch.cuda.synchronize()
    info = model.multi_matching_unsup.ga_mgmc.last_info
    print("step", i, "loss", None if loss is None else float(loss), "ms %.1f" % ((time.perf_counter() - t0) * 1e3),
          "gagm", None if info is None else info.cpu().tolist()[:7])
t = {}
res = BaselineTrainer.test(cfg, model, opt, timers=t)
print(res, t)
