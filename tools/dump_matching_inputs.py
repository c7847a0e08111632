"""Diagnostics (not product): fit the synthetic checkpoint, then dump what the matching step sees on a few test batches -
node features, labels, U and the multi_matching_unsup weights - so that the solver regime can be studied on the CPU oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth_checkpoint as sc  # noqa: E402


def main():
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.engine import BaselineTrainer
    steps, tta, nb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "test_segment.yaml"))
    dev = torch.device("cuda:0")
    model, rep = sc.make(cfg, dev, steps=steps, tta_steps=tta)
    data.register_synthetic("dump_ds", nb * cfg.TEST.BATCH, size=512, cfg_id=2)
    BaselineTrainer.rank, BaselineTrainer.world, BaselineTrainer.device = 0, 1, dev
    loader = BaselineTrainer.build_test_loader(cfg, "dump_ds")
    model.train()
    model.teacher_forced = False
    m = model.multi_matching_unsup
    m.keep_trace = True
    out = {"U": model.multi_matching_sup.U.detach().cpu(), "params": {k: v.detach().cpu() for k, v in m.state_dict().items()}, "batches": []}
    out["forced"] = []
    with torch.no_grad():
        for b in loader:
            for forced, key in ((False, "batches"), (True, "forced")):
                model.teacher_forced = forced
                loss, _, _, _ = model(b, branch="TTT")
                tr = m.last
                out[key].append(dict(X=tr["X"].cpu(), sizes=tr["sizes"], info=tr["info"].cpu(), loss=float(loss), Wds=tr["Wds"].cpu(), U0=tr["U0"].cpu(),
                                     Ub=tr["Ub"].cpu()))
                print("forced" if forced else "free  ", "sizes", tr["sizes"], "iters", tr["info"].cpu().tolist()[:6], "loss", float(loss))
    torch.save(out, os.path.join(ROOT, "gpurun_out", "match_dump.pt"))


if __name__ == "__main__":
    main()
