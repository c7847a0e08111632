#!/usr/bin/env python3
"""``train_net.py --eval-only`` for the MI355X path: the reference's command line (train_net.py:35-84 with detectron2's
``default_argument_parser``) on a box without detectron2.

    python train_net.py --eval-only --config-file configs/test_segment.yaml [--num-gpus N] \\
        [--register-coco NAME JSON IMAGE_ROOT]... [--register-synthetic NAME NUM_IMAGES]... \\
        [MODEL.WEIGHTS ckpt.pth DATASETS.TEST "('NAME',)" TEST.BATCH 4 OUTPUT_DIR out ...]

Only ``--eval-only`` exists: test-time adaptation + Dice evaluation (``BaselineTrainer.test``); source training is out of
scope.  One process per GPU: under ``torch.distributed.run`` the environment decides, otherwise
``--num-gpus N`` spawns N workers that rendezvous on 127.0.0.1 over RCCL.  Each rank adapts and evaluates its contiguous
shard (detectron2 ``InferenceSampler``), the score lists are all-gathered, rank 0 prints the result dict and appends it to
``OUTPUT_DIR/result_ap.txt`` in the reference's two-line format (train_net.py:77-80)."""
import argparse
import json
import logging
import os
import socket
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def argument_parser():
    ap = argparse.ArgumentParser(description="TTDG-MGM test-time adaptation + evaluation on MI355X")
    ap.add_argument("--config-file", default="", metavar="FILE")
    ap.add_argument("--eval-only", action="store_true")
    ap.add_argument("--resume", action="store_true", help="accepted for command-line compatibility (eval-only never resumes)")
    ap.add_argument("--num-gpus", type=int, default=1)
    ap.add_argument("--num-machines", type=int, default=1)
    ap.add_argument("--machine-rank", type=int, default=0)
    ap.add_argument("--dist-url", default="auto", help="accepted for compatibility; single-node rendezvous is 127.0.0.1")
    ap.add_argument("--register-coco", nargs=3, action="append", default=[], metavar=("NAME", "JSON", "IMAGE_ROOT"))
    ap.add_argument("--register-synthetic", nargs=2, action="append", default=[], metavar=("NAME", "NUM_IMAGES"))
    ap.add_argument("--sync-universe", action="store_true", help="Mode S: one joint adaptation step over all ranks (DESIGN.md §6)")
    ap.add_argument("opts", nargs=argparse.REMAINDER, default=[], help="KEY VALUE pairs merged into the config")
    return ap


def setup(args):
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import add_ateacher_config, get_cfg
    cfg = get_cfg()
    add_ateacher_config(cfg)
    if args.config_file:
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list([o for o in args.opts if o != "--"])
    for name, jf, root in args.register_coco:
        data.register_coco_instances(name, {}, jf, root, cfg.INPUT.FORMAT)
    for name, n in args.register_synthetic:
        data.register_synthetic(name, int(n))
    for name in cfg.DATASETS.TEST:       # the bundled yaml names synthetic streams: give them a default length
        if name.startswith("synthfundus") and name not in data._REGISTRY:
            data.register_synthetic(name, 16)
    os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)
    return cfg


def worker(rank, world, port, args):
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)       # "nccl" is RCCL on ROCm
    if not torch.cuda.is_available():
        raise RuntimeError("train_net.py runs the HIP path: no GPU is visible (there is no CPU fallback)")
    local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="[%(asctime)s %(name)s] %(message)s")
    cfg = setup(args)
    cfg.MODEL.DEVICE = "cuda:%d" % local
    from ttdg_mgm_amd.engine import BaselineTrainer
    from ttdg_mgm_amd.engine.checkpoint import load_weights
    if cfg.SEMISUPNET.Trainer == "ateacher":
        raise NotImplementedError("the mean-teacher trainer is source training (out of scope); its checkpoints are evaluated with "
                                  "SEMISUPNET.Trainer baseline (the student / teacher half is picked by TEST.EVAL_STU)")
    if cfg.SEMISUPNET.Trainer != "baseline":
        raise ValueError("Trainer Name is not found.")
    BaselineTrainer.rank, BaselineTrainer.world, BaselineTrainer.device = rank, world, torch.device("cuda", local)
    BaselineTrainer.resident_inputs = False       # real datasets: stream every pass with a bounded prefetch (data/__init__.py)
    from ttdg_mgm_amd import ops as _ops
    from ttdg_mgm_amd.modeling import detector as _det
    assert _det._backend is _ops, "train_net.py runs on the HIP operators only"
    torch.manual_seed(0)
    model = BaselineTrainer.build_model(cfg)
    load_weights(model, cfg.MODEL.WEIGHTS, prefer_student=bool(cfg.TEST.get("EVAL_STU", False)))
    if not cfg.MODEL.WEIGHTS and cfg.DATASETS.TEST:
        # random initialisation only: give the frozen BatchNorm layers statistics, as bench.py does (a checkpoint carries its own)
        from ttdg_mgm_amd.modeling import calibrate_frozen_bn
        first = next(iter(BaselineTrainer.build_test_loader(cfg, cfg.DATASETS.TEST[0])), None)
        if first is not None:
            calibrate_frozen_bn(model, first)
    model.sync_universe = bool(args.sync_universe and world > 1)
    if model.sync_universe:
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t, 0)
    model.train()
    res = BaselineTrainer.test(cfg, model, BaselineTrainer.build_optimizer(cfg, model))
    if rank == 0:
        print(res)
        with open(os.path.join(cfg.OUTPUT_DIR, "result_ap.txt"), "a") as f:
            f.write("loading data from: " + cfg.MODEL.WEIGHTS + "\n")
            f.write(json.dumps(res) + "\n")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return res


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main(argv=None):
    args = argument_parser().parse_args(argv)
    if not args.eval_only:
        raise NotImplementedError("only --eval-only (test-time adaptation + evaluation) is built; source training is out of scope")
    if args.num_machines != 1:
        raise NotImplementedError("single node only: one process per GPU over xGMI")
    if "WORLD_SIZE" in os.environ:                       # launched by torch.distributed.run
        return worker(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("MASTER_PORT", "29500")), args)
    if args.num_gpus <= 1:
        return worker(0, 1, 0, args)
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(args.num_gpus, free_port(), args), nprocs=args.num_gpus, join=True)
    return None


if __name__ == "__main__":
    main()
