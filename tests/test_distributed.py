"""World-size-2 gloo tests (CPU) of the multi-GPU path: contiguous InferenceSampler shards per rank, no data-path
collective, Dice score all-gather (SURVEY.md §8e Mode R), max-over-ranks timing reduction."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ttdg_mgm_amd import data
        from ttdg_mgm_amd.config import get_cfg
        from ttdg_mgm_amd.evaluation import DiceEvaluator
        cfg = get_cfg()
        cfg.TEST.BATCH = 2
        data.register_synthetic("dist_ds", 7, size=64)
        loader = data.build_detection_test_loader(cfg, "dist_ds", rank, world)
        ids = [d["image_id"] for b in loader for d in b]
        ev = DiceEvaluator("dist_ds", 0.9)
        ev.dice_scores = [float(i) for i in ids]          # stand-in scores: one per local image
        ev.ea_scores, ev.sm_scores = list(ev.dice_scores), list(ev.dice_scores)
        ev.gather_scores()
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # bench.py's max-over-ranks timing
        q.put((rank, ids, sorted(ev.dice_scores), float(t), len(loader)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_score_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, ids0, g0, t0, n0), (r1, ids1, g1, t1, n1) = res
    assert ids0 == [0, 1, 2, 3] and ids1 == [4, 5, 6]            # detectron2 InferenceSampler: ceil(7/2) per rank
    assert n0 == 2 and n1 == 2                                   # batches of 2, drop_last False
    assert g0 == g1 == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0]       # every rank ends up with all scores
    assert t0 == t1 == 2.0


def test_single_process_loader_matches_reference_sampler():
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    cfg = get_cfg()
    cfg.TEST.BATCH = 4
    data.register_synthetic("dist_ds2", 9, size=64)
    seen = []
    for r in range(4):
        seen.append([d["image_id"] for b in data.build_detection_test_loader(cfg, "dist_ds2", r, 4) for d in b])
    assert seen == [[0, 1, 2], [3, 4, 5], [6, 7, 8], []]
