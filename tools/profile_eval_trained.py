"""Where one eval batch goes on the trained-regime checkpoint: inference vs evaluator, new images vs cached ground truth,
plus a cProfile of the host side.  (Diagnostics; the checkpoint comes from tools/synth_checkpoint.py's cache.)"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
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
data.register_synthetic("pe", 64, cfg_id=2)
BaselineTrainer.device = dev
loader = BaselineTrainer.build_test_loader(cfg, "pe")
batches = list(loader)
ev = DiceEvaluator("pe", 0.9, dataset_dicts=loader.dataset_dicts)
model.eval()


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


with torch.no_grad():
    for b in batches[:3]:
        ev.process(b, model(b))
    ev.evaluate()
    t_inf = t_ev = 0.0
    for b in batches[3:11]:           # new images: ground truth staged inside the loop
        t0 = sync()
        out = model(b)
        t1 = sync()
        ev.process(b, out)
        t2 = sync()
        t_inf += t1 - t0
        t_ev += t2 - t1
    print("new images   : inference %.1f ms, evaluator %.1f ms per batch" % (t_inf / 8 * 1e3, t_ev / 8 * 1e3))
    t_inf = t_ev = 0.0
    for b in batches[3:11]:           # same images again: ground truth cached on the device
        t0 = sync()
        out = model(b)
        t1 = sync()
        ev.process(b, out)
        t2 = sync()
        t_inf += t1 - t0
        t_ev += t2 - t1
    print("cached GT    : inference %.1f ms, evaluator %.1f ms per batch" % (t_inf / 8 * 1e3, t_ev / 8 * 1e3))
    t0 = sync()
    for b in batches[3:11]:
        ev.process(b, model(b))
    t1 = sync()
    print("pipelined    : %.1f ms per batch" % ((t1 - t0) / 8 * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for b in batches[11:16]:
        ev.process(b, model(b))
    pr.disable()
    sync()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
