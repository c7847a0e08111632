"""ATen-op view of one TTA step and one eval batch (torch.profiler): GPU time per op, biggest first, and the count of
small glue launches.  python tools/profile_ops_summary.py"""
import sys
sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
from ttdg_mgm_amd import data
from ttdg_mgm_amd.config import get_cfg
from ttdg_mgm_amd.engine import BaselineTrainer
from ttdg_mgm_amd.evaluation import DiceEvaluator
from ttdg_mgm_amd.modeling import calibrate_frozen_bn
cfg = get_cfg(); cfg.merge_from_file("configs/test_segment.yaml"); cfg.DATASETS.TEST = ["pe"]
data.register_synthetic("pe", 12)
torch.manual_seed(0)
model = BaselineTrainer.build_model(cfg); model.teacher_forced = True
BaselineTrainer.device = torch.device("cuda:0")
batches = list(BaselineTrainer.build_test_loader(cfg, "pe"))
calibrate_frozen_bn(model, batches[0])
opt = BaselineTrainer.build_optimizer(cfg, model)
ev = DiceEvaluator("pe", 0.9)


def tta():
    model.train()
    BaselineTrainer.tta_step(model, opt, batches[1])


def evalb():
    model.eval()
    with torch.no_grad():
        ev.process(batches[2], model(batches[2]))
    ev.evaluate()


for name, fn in (("TTA step", tta), ("eval batch", evalb)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    print("=====", name)
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=28, max_name_column_width=60))
