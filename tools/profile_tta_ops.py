"""Map the GPU time of one TTA step to ATen ops with input shapes (torch.profiler): which convolution is the slow one?"""
import sys
sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
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
opt = BaselineTrainer.build_optimizer(cfg, model)
model.train()
for _ in range(2):
    BaselineTrainer.tta_step(model, opt, batches[0])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    BaselineTrainer.tta_step(model, opt, batches[1])
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=22, max_name_column_width=45, max_shapes_column_width=70))
