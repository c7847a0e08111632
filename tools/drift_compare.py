"""Continual-TTA drift, device against the CPU port (second half): from the same checkpoint, K continual
free-running adaptation steps on the first K batches of the bench stream, then the Dice pass over the same K batches with the
adapted weights - once on the GPU (product), once with the CPU port (oracle/tta_cpu, test infrastructure).  The CPU side runs in a
child process with 64 threads.   usage: drift_compare.py [K]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    import synth_checkpoint as sc
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.engine import BaselineTrainer
    from ttdg_mgm_amd.engine.checkpoint import load_weights
    from ttdg_mgm_amd.engine.trainer import run_eval_batches
    from ttdg_mgm_amd.evaluation import DiceEvaluator
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "test_segment.yaml"))
    dev = torch.device("cuda:0")
    cfg.MODEL.DEVICE = "cuda:0"
    path, rep = sc.get_or_make(cfg, dev, log=lambda m: None)
    # the CPU port on the same checkpoint and images, in parallel with the GPU run
    code = ("import json, sys, torch; sys.path.insert(0, %r); torch.set_num_threads(64); from oracle import tta_cpu; "
            "r = tta_cpu.run(%d, 4, 512, teacher_forced=False, weights=%r, reps=1, warmup=0); print('CPUJSON' + json.dumps(r))" % (ROOT, K, path))
    env = dict(os.environ, OMP_NUM_THREADS="64", MKL_NUM_THREADS="64", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    t0 = time.perf_counter()
    child = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT)
    model = BaselineTrainer.build_model(cfg)
    load_weights(model, path)
    data.register_synthetic("synthfundus_cpu_baseline", K * 4, size=512, cfg_id=2)          # the stream oracle/tta_cpu uses
    BaselineTrainer.rank, BaselineTrainer.world, BaselineTrainer.device = 0, 1, dev
    loader = BaselineTrainer.build_test_loader(cfg, "synthfundus_cpu_baseline")
    batches = list(loader)
    opt = BaselineTrainer.build_optimizer(cfg, model)
    model.train()
    out = {"steps": K, "checkpoint": rep}
    ev = DiceEvaluator("synthfundus_cpu_baseline", cfg.TEST.DICE_THRES, dataset_dicts=loader.dataset_dicts)
    model.eval()
    run_eval_batches(model, batches, ev, 1, 1)
    out["gpu_before_adaptation"] = dict(ev.evaluate(), kept=len(ev.dice_scores))
    model.train()
    model.multi_matching_unsup.eval()          # attention dropout off, as in the CPU port
    for b in batches:
        BaselineTrainer.tta_step(model, opt, b)
    ev.reset()
    model.eval()
    run_eval_batches(model, batches, ev, 1, 1)
    out["gpu_after"] = dict(ev.evaluate(), kept=len(ev.dice_scores))
    stdout, _ = child.communicate(timeout=3000)
    r = json.loads([l for l in stdout.splitlines() if l.startswith("CPUJSON")][-1][7:])
    out["cpu_port_after"] = r["dice"]
    out["cpu_seconds"] = r["times"]
    out["wall_s"] = time.perf_counter() - t0
    out["abs_diff"] = {k: abs(out["gpu_after"][k] - out["cpu_port_after"][k]) for k in ("Dice Coefficient", "Enhanced Alignment Metric", "Structural Similarity Metric")}
    print(json.dumps(out, indent=1, default=str))


if __name__ == "__main__":
    main()
