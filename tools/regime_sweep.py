"""Diagnostics (not product): which synthetic-checkpoint recipe puts the GA-MGM solve in a regime where the REFERENCE's own
algorithm is well defined under rounding?  For every variant: fit (tools/synth_checkpoint.make), run N continual free-running
TTA steps on the cfg-2 test stream, and for every step hand the solver's inputs (A, Wds, U0, sizes) to the CPU oracle in
float32, in float64 and under two 1e-7-relative perturbations; a batch is "stable" when all four return the same U U^T.
Prints one JSON line per variant and saves the recorded solver inputs.   usage: regime_sweep.py name=key:val,key:val ..."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import synth_checkpoint as sc  # noqa: E402


def unpack(apack, sizes):
    M = sum(sizes)
    A = torch.zeros(M, M)
    o = p = 0
    for n in sizes:
        A[o:o + n, o:o + n] = apack[p:p + n * n].reshape(n, n)
        o += n
        p += n * n
    return A


def stability(A, W, U0, sizes):
    """-> (stable, iters32, iters64, rows of U U^T that differ between float32 and float64)."""
    from oracle import gmodule as og
    from ttdg_mgm_amd import synth
    t32, t64 = {}, {}
    U32 = og.gagm(A, W, U0, sizes, trace=t32)
    U64 = og.gagm(A.double(), W.double(), U0.double(), sizes, trace=t64).float()
    X32 = U32 @ U32.t()
    rows = int(((X32 - U64 @ U64.t()).abs().sum(1) > 0).sum())
    ok = rows == 0
    for k in range(2):
        g = synth.gen(9100 + k)
        Wp = W * (1 + 1e-7 * synth.normal(g, tuple(W.shape)))
        U0p = U0 * (1 + 1e-7 * synth.normal(g, tuple(U0.shape)))
        Up = og.gagm(A, Wp, U0p, sizes)
        ok = ok and bool(torch.equal(Up @ Up.t(), X32))
    return ok, t32["iters"], t64["iters"], rows, U32


def main():
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.engine import BaselineTrainer, inference_on_dataset
    from ttdg_mgm_amd.evaluation import DiceEvaluator
    nsteps = int(os.environ.get("SWEEP_STEPS", "16"))
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "test_segment.yaml"))
    dev = torch.device("cuda:0")
    cfg.MODEL.DEVICE = "cuda:0"
    cfg.DATASETS.TEST = ["sweep_ds"]
    data.register_synthetic("sweep_ds", nsteps * cfg.TEST.BATCH, size=512, cfg_id=2)
    BaselineTrainer.rank, BaselineTrainer.world, BaselineTrainer.device = 0, 1, dev
    loader = BaselineTrainer.build_test_loader(cfg, "sweep_ds")
    batches = list(loader)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for spec in sys.argv[1:]:
        name, _, kv = spec.partition("=")
        kw = {}
        for item in filter(None, kv.split(",")):
            k, v = item.split(":")
            kw[k] = int(v) if v.lstrip("-").isdigit() else float(v)
        model, rep = sc.make(cfg, dev, log=lambda m: None, **kw)
        opt = BaselineTrainer.build_optimizer(cfg, model)
        model.train()
        model.teacher_forced = False
        m = model.multi_matching_unsup
        m.keep_trace = True
        dump, nstable, same_dev, its = [], 0, 0, []
        for b in batches:
            loss = BaselineTrainer.tta_step(model, opt, b)
            tr = m.last
            if loss is None or tr is None:
                continue
            sizes = list(tr["sizes"])
            A, W, U0 = unpack(tr["apack"].cpu(), sizes), tr["Wds"].cpu(), tr["U0"].cpu()
            ok, i32, i64, rows, U32 = stability(A, W, U0, sizes)
            Ud = tr["Ub"].cpu()
            dev_same = bool(torch.equal(Ud @ Ud.t(), U32 @ U32.t()))
            nstable += ok
            same_dev += dev_same
            its.append(tr["info"].cpu().tolist()[:6])
            dump.append(dict(X=tr["X"].cpu(), apack=tr["apack"].cpu(), Wds=W, U0=U0, sizes=sizes, info=tr["info"].cpu().tolist(), Ub=Ud, stable=ok, dev_same=dev_same,
                             it32=i32, it64=i64, rows=rows, loss=float(loss)))
        m.keep_trace = False
        m.last = None
        ev = DiceEvaluator("sweep_ds", cfg.TEST.DICE_THRES, dataset_dicts=loader.dataset_dicts)
        res, _ = inference_on_dataset(model, loader, ev, cfg)
        tot = [sum(x) for x in its]
        print(json.dumps(dict(variant=name, kw=kw, steps=len(dump), stable=nstable, device_equals_oracle=same_dev, iters_mean=sum(tot) / max(1, len(tot)),
                              iters_first4=its[:4], dice=res, kept=len(ev.dice_scores), fit_s=round(rep["seconds"], 1),
                              stage1_last=rep["stage1_last"])), flush=True)
        torch.save({"U": model.multi_matching_sup.U.detach().cpu(), "params": {k: v.detach().cpu() for k, v in m.state_dict().items()}}, os.path.join(ROOT, "gpurun_out", "sweep_%s_model.pt" % name))
        torch.save(dump, os.path.join(ROOT, "gpurun_out", "sweep_%s.pt" % name))


if __name__ == "__main__":
    main()
