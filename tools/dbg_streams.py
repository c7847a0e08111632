import sys, copy
sys.path.insert(0, ".")
import numpy as np, torch
from ttdg_mgm_amd import data
from ttdg_mgm_amd.config import get_cfg
from ttdg_mgm_amd.engine import BaselineTrainer
from ttdg_mgm_amd.engine.trainer import run_eval_batches
from ttdg_mgm_amd.evaluation import DiceEvaluator
from ttdg_mgm_amd.modeling import calibrate_frozen_bn
cfg = get_cfg(); cfg.TEST.BATCH = 2; cfg.INPUT.MIN_SIZE_TEST = 384; cfg.MODEL.DEVICE = "cuda:0"
data.register_synthetic("e2e_ds", 10, size=256)
torch.manual_seed(0)
model = BaselineTrainer.build_model(cfg); model.teacher_forced = True
BaselineTrainer.rank, BaselineTrainer.world, BaselineTrainer.device = 0, 1, torch.device("cuda:0")
batches = list(BaselineTrainer.build_test_loader(cfg, "e2e_ds"))
calibrate_frozen_bn(model, batches[0])
model.eval()
def run(streams):
    ev = DiceEvaluator("e2e_ds", 0.0)
    run_eval_batches(model, batches, ev, streams=streams, coalesce=1)
    ev.evaluate()
    return sorted(zip(ev.dice_scores, ev.ea_scores, ev.sm_scores))
runs = [("seq", run(1)), ("seq", run(1)), ("two", run(2)), ("two", run(2)), ("three", run(3))]
base = np.array(runs[0][1])
for name, r in runs:
    a = np.array(r)
    if a.shape != base.shape:
        print(name, "shape differs", a.shape, base.shape); continue
    d = np.abs(a - base); print(name, "n", len(a), "max diff", np.nanmax(d), "rows differing", int((d.max(1) > 1e-4).sum()))
