"""bench.py — adapted images/sec of the multi-graph-matching TTA hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): synthetic 512x512 2-class fundus stream, ResNet-50-FPN Mask R-CNN stand-in
with random-init weights (no checkpoints offline), TEST.BATCH = 4, 20-sweep Sinkhorn, one TTA step per batch, then
the eval-mode Dice pass over the same batches (reference order, engine/trainer.py:469-485).  Detections are
"teacher-forced" (GT boxes jittered +-2 px) because a random-init detector finds nothing; RPN and box
head still run inside the timed region.  One "step" = one adapted batch (TTA step + its share of the eval pass).
Each rank adapts its own shard (InferenceSampler semantics, no data-path collective): weak scaling; at N > 1 the one
collective of the path is the all-gather of the per-rank Dice score lists at the end of the eval pass (RCCL).  The eval
pass can feed its independent batches from several host threads on their own HIP streams (--eval-streams N; default 1:
the pass is GPU-bound since the detection pipelines were fused).

Prints ONE JSON line on rank 0 with the contract fields plus `roofline` (dominant hand-written kernel, timed live
with HIP events on the launch stream) and, at N = 1, `cpu_baseline` (the oracle "port" on the host cores, bounded
sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=1)
    ap.add_argument("--cpu-threads", type=int, default=64)
    ap.add_argument("--eval-streams", type=int, default=1, help="concurrent eval batches (HIP streams) in the Dice pass; 1 = sequential")
    ap.add_argument("--no-overlap-detector", action="store_true", help="A/B: keep the teacher-forced RPN + box head on the main stream")
    ap.add_argument("--eval-coalesce", type=int, default=1, help="loader batches merged into one inference call in the Dice pass; 1 = none")
    ap.add_argument("--free-running", action="store_true", help="use the detector's own boxes instead of teacher forcing")
    ap.add_argument("--bf16-backbone", action="store_true", help="cfg-5 style: bf16 autocast for the backbone only")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (validation on a 1-GPU box)")
    ap.add_argument("--share-device", action="store_true", help="validation only: every rank uses cuda:0")
    ap.add_argument("--sync-universe", action="store_true",
                    help="Mode S (SURVEY.md 8e): all ranks adapt on ONE multi-graph (RCCL all-gather of the node embeddings, gradient "
                         "all-reduce) = the single-GPU algorithm at batch N*B; default is Mode R (independent shards, as the reference)")
    return ap.parse_args()


def build(cfg_id, n_images, args, device, rank, world):
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.engine import BaselineTrainer
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "test_segment.yaml"))
    cfg.TEST.BATCH = args.batch
    name = "synthfundus_bench"
    data.register_synthetic(name, n_images, size=args.size, cfg_id=cfg_id)
    cfg.DATASETS.TEST = [name]
    cfg.MODEL.DEVICE = str(device)
    torch.manual_seed(0)
    model = BaselineTrainer.build_model(cfg)
    model.teacher_forced = not args.free_running
    if args.no_overlap_detector:
        from ttdg_mgm_amd.modeling import rcnn as _rcnn
        _rcnn.OVERLAP_DETECTOR = False
    model.autocast_backbone = args.bf16_backbone
    opt = BaselineTrainer.build_optimizer(cfg, model)
    BaselineTrainer.rank, BaselineTrainer.world, BaselineTrainer.device = rank, world, device
    loader = BaselineTrainer.build_test_loader(cfg, name)
    batches = list(loader)
    from ttdg_mgm_amd.modeling import calibrate_frozen_bn
    calibrate_frozen_bn(model, batches[0])
    if args.sync_universe and world > 1:
        model.sync_universe = True
        with torch.no_grad():          # replicas must start identical: the calibration batch differs per rank
            for t in list(model.parameters()) + list(model.buffers()):
                if dist.get_backend() == "nccl":
                    dist.broadcast(t, 0)
                else:
                    h = t.cpu()
                    dist.broadcast(h, 0)
                    t.copy_(h)
    return cfg, model, opt, batches, name, loader.dataset_dicts


def gpu_run(args, rank, world, device):
    from ttdg_mgm_amd.engine import BaselineTrainer
    from ttdg_mgm_amd.evaluation import DiceEvaluator
    from ttdg_mgm_amd import ops
    K, W, B = args.steps, args.warmup, args.batch
    # every rank owns (K + W) batches: the sampler shards a (world * (K+W) * B)-image stream contiguously
    cfg, model, opt, batches, name, local_dicts = build(2, world * (K + W) * B, args, device, rank, world)
    assert len(batches) >= K + W, (len(batches), K, W)
    dice = DiceEvaluator(name, cfg.TEST.DICE_THRES, dataset_dicts=local_dicts)

    def adapt(bs):
        for b in bs:
            BaselineTrainer.tta_step(model, opt, b)

    def evaluate(bs):
        from ttdg_mgm_amd.engine.trainer import run_eval_batches
        model.eval()
        dice.reset()
        if args.eval_streams == 0:        # A/B: the plain inline loop
            with torch.no_grad():
                for b in bs:
                    dice.process(b, model(b))
        else:
            run_eval_batches(model, bs, dice, args.eval_streams, args.eval_coalesce)      # independent batches: merged calls on concurrent HIP streams
        model.train()
        if world > 1:
            dice.gather_scores()     # Mode R's one collective: all-gather of the per-rank score lists over RCCL
        return dice.evaluate()

    model.train()
    adapt(batches[:W])
    evaluate(batches[:W])
    # ---- timed region: K adaptation steps, then the Dice pass over the same K batches ----
    stamps = []
    ops.KERNEL_TIMERS = stamps          # (name, start_event, end_event) pairs recorded around our dominant kernel
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    adapt(batches[W:W + K])
    torch.cuda.synchronize()
    t_mid = time.perf_counter()
    res = evaluate(batches[W:W + K])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    ops.KERNEL_TIMERS = None
    el, tta = t1 - t0, t_mid - t0
    if world > 1:
        t = torch.tensor([el, tta], device=device if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, tta = float(t[0]), float(t[1])
    return dict(elapsed=el, tta=tta, dice=res, stamps=stamps)


def roofline_from_stamps(run, K):
    """Dominant hand-written kernel = the GA-MGM solver (one launch per step).  Algorithmic FLOPs per launch
    (SURVEY.md §8d A6): per iteration 2u*sum(n_g^2) + 4Mu^2 + 2M^2u + projector (5*K_sk*sum(max(n_g,u)^2) for the
    Sinkhorn stages, ~n^2*u for the LAP stage), times the measured iteration count.  fp32 VALU work: the fp32
    vector peak equals the fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md), reported under bound 'mfma'."""
    ev = []
    for nm, a, b, sizes, info in run["stamps"]:
        if nm == "gagm":
            it = info.cpu().tolist()
            ev.append((a.elapsed_time(b) * 1e-3, sizes, sum(it[:5]), it[5], it[:6], it[14], it[15]))
    if not ev:
        return None
    u, ksk = 32, 20
    tot_t, tot_f, tot_it = 0.0, 0.0, 0
    for dt, sizes, iters_sk, iters_h, _, _, _ in ev:
        M = sum(sizes)
        tot_it += iters_sk + iters_h
        base = 2 * u * sum(n * n for n in sizes) + 4 * M * u * u + 2 * M * M * u
        f = (iters_sk + iters_h) * base + iters_sk * 5 * ksk * sum(max(n, u) ** 2 for n in sizes) \
            + iters_h * sum(min(n, u) ** 2 * max(n, u) for n in sizes)
        tot_t += dt
        tot_f += f
    peak = 157.3
    ach = tot_f / tot_t / 1e12
    return {"kernel": "gagm_kernel", "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "traffic": None, "launches": len(ev), "avg_launch_ms": tot_t / len(ev) * 1e3,
            "avg_iterations_per_launch": tot_it / len(ev), "us_per_iteration": tot_t / max(tot_it, 1) * 1e6,
            "avg_nodes_per_launch": sum(sum(e[1]) for e in ev) / len(ev), "graphs_per_launch": len(ev[0][1]),
            "avg_iterations_per_stage": [sum(e[4][k] for e in ev) / len(ev) for k in range(6)],
            "hungarian_cycle_periods": [e[5] for e in ev], "hungarian_cycle_detected_at": [e[6] for e in ev],
            "note": "single-workgroup latency-bound solver; fp32 VALU peak == fp32 MFMA peak"}


def pmc_traffic(kernel, streaming):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs, tools/pmc_summary.py -> profiles/r01_bench_pmc.json).  Counters cannot be read
    from inside the timed process, so this is the recorded figure, not a live one; None when the file is absent.
    `streaming`: apply the gfx950 x2 correction of FETCH_SIZE for wide coalesced reads."""
    path = os.path.join(ROOT, "profiles", "r01_bench_pmc.json")
    try:
        with open(path) as f:
            rec = json.load(f).get(kernel)
    except (OSError, ValueError):
        return None
    if not rec:
        return None
    return rec["hbm_bytes_per_launch_streaming_corrected" if streaming else "hbm_bytes_per_launch_raw"]


def cpu_baseline(args):
    from oracle import tta_cpu
    # intra-op threads: all host cores up to 64 (beyond that torch's CPU conv/GEMM kernels stop scaling on the
    # small tensors of this path and thread wake-ups dominate); `cores` reports what was actually used
    cores = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    t = tta_cpu.time_steps(args.cpu_steps, args.batch, args.size, teacher_forced=not args.free_running)
    return {"value": args.cpu_steps * args.batch / t, "unit": "adapted images/s", "cores": cores,
            "host_cores": os.cpu_count(), "kind": "port",
            "sample": "%d TTA step(s) + eval pass on %d synthetic %dx%d images, torch-CPU model + oracle GModule, %d threads"
                      % (args.cpu_steps, args.cpu_steps * args.batch, args.size, args.size, cores), "seconds": t}


def _finite(x):
    """NaN / inf -> null: the line must be strict JSON (the Dice means are NaN when no prediction of the random-init
    detector clears the 0.9 score threshold)."""
    if isinstance(x, float):
        return x if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    return x


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(args.dist_backend)   # RCCL; used for the barrier and the max-over-ranks timing only
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    if args.share_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    run = gpu_run(args, rank, world, device)
    if rank == 0:
        K, B = args.steps, args.batch
        images = world * K * B
        out = {
            "metric": "adapted images/sec (512x512, 2-class)", "value": images / run["elapsed"], "unit": "images/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": run["elapsed"] / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if not args.bf16_backbone else "bf16 backbone / f32 matching", "data": "synthetic",
            "config": {"workload": "cfg-2: %d-image synthetic %dx%d 2-class fundus stream per GPU, TEST.BATCH=%d, "
                                   "ResNet-50-FPN stand-in (random init) + 20-sweep Sinkhorn, 1 TTA step per batch + Dice pass, %s detections"
                                   % (K * B, args.size, args.size, B, "free-running" if args.free_running else "teacher-forced"),
                       "global_batch": world * B, "parallelism": ("dp%d (independent shards, no data-path collective)" % world) if not (args.sync_universe and world > 1)
                       else "dp%d synchronous universe graph (all-gather of node embeddings + gradient all-reduce)" % world},
            "tta_only_images_per_s": images / run["tta"], "dice": run["dice"],
        }
        out["roofline"] = roofline_from_stamps(run, K)
        if out["roofline"] is not None:
            out["roofline"]["traffic"] = pmc_traffic("gagm_kernel", False)
            out["roofline"]["traffic_note"] = ("HBM bytes per launch from profiles/r01_bench_pmc.json (separate rocprofv3 --pmc FETCH_SIZE / "
                                               "WRITE_SIZE passes of this command): the solver state is LDS-resident, ~0.4 MB per ~18 ms launch")
        sgd = [(a.elapsed_time(b) * 1e-3, nb) for nm, a, b, nb, _ in run["stamps"] if nm == "sgd"]
        if sgd:   # second hand-written kernel with a meaningful hardware bound: the fused SGD step streams 20 B/parameter
            t, nb = sum(x[0] for x in sgd), sum(x[1] for x in sgd)
            out["roofline_other_kernels"] = [{"kernel": "sgd_multi_tensor_kernel", "bound": "hbm", "achieved": nb / t / 1e9,
                                              "peak": 8000.0, "unit": "GB/s", "frac": nb / t / 8e12, "traffic": pmc_traffic("sgd_multi_tensor", True),
                                              "launches": len(sgd), "avg_launch_ms": t / len(sgd) * 1e3,
                                              "bytes_per_launch": nb / len(sgd)}]
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(_finite(out)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
