"""cfg-5 (384 x 384 3-class polyp stream, bf16 backbone + fp32 matching): which part of the backbone has to stay fp32 for the Dice of
the bf16 run to agree with the fp32 run?  VERDICT r3 item 8.  For each of `fits` independently fitted checkpoints
(tools/synth_checkpoint.py kind = "polyp"; the fit is not bit-reproducible, every fit is a different checkpoint) the eval-mode
Dice / E / S over 48 held-out images with
    f32            the fp32 backbone (reference of the comparison)
    f32_eps1e-7_*  the fp32 backbone with every trainable tensor perturbed by 1e-7 relative (two draws): how well the metric is
                   defined by the fp32 path itself on this stream
    all            bf16 autocast over the whole backbone (round 2 / 3's cfg-5)
    res5 .. res2   bf16 up to and including that ResNet stage, fp32 behind it (later stages + FPN: a precision island)
and the eval-only images/s of every variant (median of 3 passes over the 48 images).
    python tools/cfg5_island.py [fits=3] [out.json] [variants, comma separated; "f32" is always run] [notime]"""
import json
import os
import statistics
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

VARIANTS = ("f32", "f32_eps1e-7_a", "f32_eps1e-7_b", "all", "res5", "res4", "res3", "res2")
KEYS = ("Dice Coefficient", "Enhanced Alignment Metric", "Structural Similarity Metric")


def main():
    fits = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    outp = sys.argv[2] if len(sys.argv) > 2 else None
    global VARIANTS
    if len(sys.argv) > 3 and sys.argv[3]:
        VARIANTS = ("f32",) + tuple(v for v in sys.argv[3].split(",") if v != "f32")
    notime = len(sys.argv) > 4 and sys.argv[4] == "notime"
    import synth_checkpoint as sc
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.engine import BaselineTrainer, inference_on_dataset
    from ttdg_mgm_amd.engine.checkpoint import load_weights
    from ttdg_mgm_amd.evaluation import DiceEvaluator
    dev = torch.device("cuda:0")
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "test_segment.yaml"))
    cfg.MODEL.DEVICE, cfg.MODEL.ROI_HEADS.NUM_CLASSES, cfg.INPUT.MIN_SIZE_TEST = "cuda:0", 3, 384
    data.register_synthetic("cfg5_island", 48, size=384, cfg_id=5, kind="polyp", num_cls=3)
    BaselineTrainer.rank, BaselineTrainer.world, BaselineTrainer.device = 0, 1, dev
    loader = BaselineTrainer.build_test_loader(cfg, "cfg5_island")
    rows = []
    for f in range(fits):
        with tempfile.TemporaryDirectory() as td:          # a fresh cache directory: a fresh fit
            path, rep = sc.get_or_make(cfg, dev, cache_dir=td, log=lambda m: None, kind="polyp", size=384, seed=f)
            res = {}
            for v in VARIANTS:
                m = BaselineTrainer.build_model(cfg)
                load_weights(m, path)
                m.autocast_backbone = False if v.startswith("f32") else v
                if v.startswith("f32_eps"):          # the fp32 path's OWN sensitivity: every trainable tensor times (1 + 1e-7 randn)
                    g = torch.Generator().manual_seed(11 if v.endswith("a") else 12)
                    with torch.no_grad():
                        for q in m.parameters():
                            if q.requires_grad:
                                q.mul_((1 + 1e-7 * torch.randn(q.shape, generator=g)).to(q.device))
                ev = DiceEvaluator("cfg5_island", cfg.TEST.DICE_THRES, dataset_dicts=loader.dataset_dicts)
                r, _ = inference_on_dataset(m, loader, ev, cfg)
                ts = []
                for _ in range(0 if notime else 3):
                    ev2 = DiceEvaluator("cfg5_island", cfg.TEST.DICE_THRES, dataset_dicts=loader.dataset_dicts)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    inference_on_dataset(m, loader, ev2, cfg)
                    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                res[v] = dict({k: r[k] for k in KEYS}, kept=len(ev.dice_scores), eval_images_per_s=(48.0 / statistics.median(ts)) if ts else 0.0)
                del m
            ref = res["f32"]
            for v in VARIANTS[1:]:
                res[v]["relative_difference"] = {k: abs(res[v][k] - ref[k]) / abs(ref[k]) for k in KEYS}
            rows.append(dict(fit=f, fit_seconds=rep.get("seconds"), variants=res))
            print("fit %d: " % f + "  ".join("%s dDice %.1e kept %d %.0f img/s" % (v, res[v].get("relative_difference", {}).get(KEYS[0], 0.0), res[v]["kept"],
                                                                                   res[v]["eval_images_per_s"]) for v in VARIANTS), file=sys.stderr, flush=True)
            doc = json.dumps(dict(fits=rows, worst_relative_dice_difference={v: max(r["variants"][v]["relative_difference"][KEYS[0]] for r in rows) for v in VARIANTS[1:]}), indent=1)
            if outp:
                with open(outp, "w") as fh:
                    fh.write(doc)
    print(doc)


if __name__ == "__main__":
    main()
