"""ATen-op inventory of one TTA step (torch.profiler, GPU self time, with and without input shapes): where the
element-wise glue around the vendor convolutions comes from."""
import sys
sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
from ttdg_mgm_amd import data
from ttdg_mgm_amd.config import get_cfg
from ttdg_mgm_amd.engine import BaselineTrainer
from ttdg_mgm_amd.modeling import calibrate_frozen_bn
cfg = get_cfg(); cfg.merge_from_file("configs/test_segment.yaml"); cfg.DATASETS.TEST = ["pe"]
data.register_synthetic("pe", 12)
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
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
ka = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::add", "aten::add_", "aten::mul", "aten::copy_", "aten::clamp_min_", "aten::threshold_backward", "aten::sum", "aten::fill_", "aten::zero_")]
ka.sort(key=lambda e: -e.self_device_time_total)
for e in ka[:40]:
    print("%-26s n=%3d  %8.1f us  %s" % (e.key, e.count, e.self_device_time_total, str(e.input_shapes)[:150]))
