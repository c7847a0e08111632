"""Diagnostics (not product): the GA-MGM solver in the trained regime.  Fits / finds the synthetic checkpoint, runs a few
free-running TTA steps with the in-kernel phase counters on, prints iterations per stage and the share of each phase, and
saves the solver's inputs (A blocks, Wds, U0, sizes) of every step so that the solver can be re-run in isolation
(tools/bench_gagm_inputs.py) without the detector.   usage: gagm_trained_probe.py [n_steps] [out.pt]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from ttdg_mgm_amd.GModule import multi_graph_matching as _mgm
_mgm.GAGM_PROFILE = 1          # in-kernel phase clocks of every solve (info[8..13])

import synth_checkpoint as sc  # noqa: E402


def main():
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.engine import BaselineTrainer
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "trained_solver_inputs.pt")
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "test_segment.yaml"))
    dev = torch.device("cuda:0")
    cfg.MODEL.DEVICE = "cuda:0"
    path, rep = sc.get_or_make(cfg, dev)
    model = BaselineTrainer.build_model(cfg)
    from ttdg_mgm_amd.engine.checkpoint import load_weights
    load_weights(model, path)
    data.register_synthetic("probe_ds", nsteps * cfg.TEST.BATCH, size=512, cfg_id=2)
    BaselineTrainer.rank, BaselineTrainer.world, BaselineTrainer.device = 0, 1, dev
    cfg.DATASETS.TEST = ["probe_ds"]
    batches = list(BaselineTrainer.build_test_loader(cfg, "probe_ds"))
    opt = BaselineTrainer.build_optimizer(cfg, model)
    model.train()
    m = model.multi_matching_unsup
    m.keep_trace = True
    tot = [0] * 5
    its = 0
    dump = []
    for b in batches:
        BaselineTrainer.tta_step(model, opt, b)
        tr = m.last
        info = tr["info"].cpu().tolist()
        for k in range(5):
            tot[k] += info[9 + k]
        its += info[6]
        print("sizes", tr["sizes"], "stage iterations", info[:6], "total", info[6],
              "kcycles B/S/V/proj/conv", [round(info[9 + k] * 64 / 1e3, 1) for k in range(5)], flush=True)
        dump.append(dict(apack=tr["apack"].cpu(), Wds=tr["Wds"].cpu(), U0=tr["U0"].cpu(), sizes=list(tr["sizes"]), info=info,
                         Ub=tr["Ub"].cpu()))
    s = float(sum(tot))
    print("iterations", its, "cycles/iteration %.0f" % (s * 64 / its),
          "phase share B %.1f%% S %.1f%% V %.1f%% proj %.1f%% conv %.1f%%" % tuple(100.0 * x / s for x in tot))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    torch.save(dump, out)
    print("saved", out)


if __name__ == "__main__":
    main()
