"""ATen-op / kernel inventory (torch.profiler, GPU self time) of ONE free-running TTA step and ONE Dice-pass batch on the
trained-regime checkpoint: what runs between the vendor convolutions.   (Diagnostics.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import synth_checkpoint as sc  # noqa: E402
from ttdg_mgm_amd import data  # noqa: E402
from ttdg_mgm_amd.config import get_cfg  # noqa: E402
from ttdg_mgm_amd.engine import BaselineTrainer  # noqa: E402
from ttdg_mgm_amd.engine.checkpoint import load_weights  # noqa: E402
from ttdg_mgm_amd.evaluation import DiceEvaluator  # noqa: E402

cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "test_segment.yaml"))
dev = torch.device("cuda:0")
cfg.MODEL.DEVICE = str(dev)
path, rep = sc.get_or_make(cfg, dev, log=lambda m: None)
model = BaselineTrainer.build_model(cfg)
load_weights(model, path)
data.register_synthetic("pe", 24, cfg_id=2)
BaselineTrainer.rank, BaselineTrainer.world, BaselineTrainer.device = 0, 1, dev
loader = BaselineTrainer.build_test_loader(cfg, "pe")
batches = list(loader)
opt = BaselineTrainer.build_optimizer(cfg, model)
ev = DiceEvaluator("pe", 0.9, dataset_dicts=loader.dataset_dicts)
ev.prestage(dev)
SMALL = ("aten::add", "aten::add_", "aten::mul", "aten::copy_", "aten::clamp_min_", "aten::threshold_backward", "aten::sum", "aten::fill_",
         "aten::zero_", "aten::index", "aten::cat", "aten::_to_copy", "aten::sort", "aten::topk", "aten::nonzero", "aten::where", "aten::div",
         "aten::sub", "aten::sigmoid", "aten::index_select", "aten::gather", "aten::arange", "aten::item", "aten::_local_scalar_dense")


def show(prof, title):
    print("=" * 30, title)
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=58))
    ka = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in SMALL]
    ka.sort(key=lambda e: -e.self_device_time_total)
    for e in ka[:32]:
        print("%-26s n=%3d  %8.1f us  %s" % (e.key, e.count, e.self_device_time_total, str(e.input_shapes)[:140]))


model.train()
for b in batches[:2]:
    BaselineTrainer.tta_step(model, opt, b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    BaselineTrainer.tta_step(model, opt, batches[2])
    torch.cuda.synchronize()
show(prof, "TTA step (free-running, trained regime)")
model.eval()
with torch.no_grad():
    for b in batches[:2]:
        ev.process(b, model(b))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        ev.process(batches[3], model(batches[3]))
        torch.cuda.synchronize()
show(prof, "Dice-pass batch")
