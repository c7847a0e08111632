import sys, os
sys.path.insert(0, ".")
import torch
from ttdg_mgm_amd.GModule import multi_graph_matching as _mgm
_mgm.GAGM_PROFILE = 1          # in-kernel phase clocks of every solve (info[8..13])
from ttdg_mgm_amd import data
from ttdg_mgm_amd.config import get_cfg
from ttdg_mgm_amd.engine import BaselineTrainer
from ttdg_mgm_amd.modeling import calibrate_frozen_bn
cfg = get_cfg(); cfg.merge_from_file("configs/test_segment.yaml"); cfg.DATASETS.TEST = ["pe"]; cfg.MODEL.DEVICE = "cuda:0"
data.register_synthetic("pe", 24, cfg_id=2)
torch.manual_seed(0)
model = BaselineTrainer.build_model(cfg); model.teacher_forced = True
BaselineTrainer.rank, BaselineTrainer.world, BaselineTrainer.device = 0, 1, torch.device("cuda:0")
batches = list(BaselineTrainer.build_test_loader(cfg, "pe"))
calibrate_frozen_bn(model, batches[0])
opt = BaselineTrainer.build_optimizer(cfg, model)
model.train(); model.multi_matching_unsup.keep_trace = True
tot = [0] * 5; its = 0
for b in batches:
    BaselineTrainer.tta_step(model, opt, b)
    info = model.multi_matching_unsup.last["info"].cpu().tolist()
    for k in range(5): tot[k] += info[9 + k]
    its += info[6]
s = sum(tot)
print("iterations", its, "phase share B %.1f%% S %.1f%% V %.1f%% proj %.1f%% conv %.1f%%" % tuple(100.0 * x / s for x in tot))
