"""Host-side cost of one eval batch: cProfile of the Python side (which wrappers burn the CPU between launches)."""
import cProfile
import pstats
import sys
import time
sys.path.insert(0, ".")
import torch
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
ev = DiceEvaluator("pe", 0.9)
model.eval()


def evalb(b):
    with torch.no_grad():
        ev.process(b, model(b))


for _ in range(3):
    evalb(batches[1])
ev.evaluate()
torch.cuda.synchronize()
# host-only time: launch everything without waiting for the GPU
t0 = time.perf_counter()
for _ in range(5):
    evalb(batches[2])
t_launch = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 5
print("eval batch: host-side %.1f ms per batch to issue, %.1f ms per batch wall incl. GPU drain" % (t_launch * 1e3, t_all * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    evalb(batches[2])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
