"""World-size-2 gloo tests (CPU) of the multi-GPU path: contiguous InferenceSampler shards per rank, no data-path
collective, Dice score all-gather (SURVEY.md §8e Mode R), max-over-ranks timing reduction."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _ship(vals):
    """Tensors cross the process boundary BY VALUE (numpy arrays): a torch tensor on a multiprocessing queue travels as a
    shared-memory handle that the receiver fetches from the sender's resource sharer - which is gone when the worker has
    already exited (ConnectionResetError, seen under load)."""
    return tuple(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v for v in vals)


def _unship(vals):
    import numpy as np
    return tuple(torch.from_numpy(v) if isinstance(v, np.ndarray) else v for v in vals)


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


def _worker_cfg4(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ttdg_mgm_amd import data
        from ttdg_mgm_amd.config import get_cfg
        from ttdg_mgm_amd.evaluation import DiceEvaluator
        cfg = get_cfg()
        cfg.TEST.BATCH = 4
        cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 32, 64          # (the plumbing, not 2048 resizes to 800 x 800 on the CPU)
        data.register_synthetic("cfg4_ds", 2048, size=32, cfg_id=4)
        loader = data.build_detection_test_loader(cfg, "cfg4_ds", rank, world)
        nb, ids = 0, []
        for b in loader:
            nb += 1
            ids += [d["image_id"] for d in b]
        ev = DiceEvaluator("cfg4_ds", 0.9)
        ev.dice_scores = [float(i) for i in ids if i % (rank + 2) == 0]       # a DIFFERENT number of kept masks on every rank
        ev.ea_scores, ev.sm_scores = list(ev.dice_scores), list(ev.dice_scores)
        ev.gather_scores()
        t = torch.tensor([10.0 - rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, ids[0], ids[-1], len(ids), nb, len(ev.dice_scores), float(sum(ev.dice_scores)), float(t)))
    finally:
        dist.destroy_process_group()


def test_eight_rank_sharding_of_the_2048_image_stream():
    """BASELINE cfg-4 plumbing (no 8-GPU node is available to the builder): 2048 images over 8 ranks, TEST.BATCH = 4 - every rank
    owns the contiguous shard [256 r, 256 (r + 1)) (detectron2 InferenceSampler [3P], data/build.py:139), takes 64 steps, and the
    variable-length score lists are all-gathered so that every rank reports the same 2048-image means (SURVEY.md 8e Mode R); the
    bench's max-over-ranks timing reduction.  gloo on CPU; the same code path runs over RCCL on the GPUs."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_cfg4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect_kept = sum(1 for r in range(world) for i in range(256 * r, 256 * (r + 1)) if i % (r + 2) == 0)
    expect_sum = float(sum(i for r in range(world) for i in range(256 * r, 256 * (r + 1)) if i % (r + 2) == 0))
    for r, first, last, n, nb, kept, total, t in res:
        assert (first, last, n, nb) == (256 * r, 256 * r + 255, 256, 64), (r, first, last, n, nb)
        assert kept == expect_kept and total == expect_sum          # every rank holds every rank's scores
        assert t == 10.0


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


# ----------------------------------------------------------------------------- Mode S (engine/sync_universe.py)
def _toy_loss(nodes, w):
    """Stand-in for the replicated matching loss: couples every pair of graphs, has its own (replicated) parameter."""
    loss = 0
    for i in range(len(nodes)):
        for j in range(i + 1, len(nodes)):
            loss = loss + (nodes[i].mean(0) * nodes[j].mean(0) * w).sum()
    return loss


def _graphs(seed, sizes):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(n, 8, generator=g) for n in sizes], [torch.randint(1, 3, (n,), generator=g) for n in sizes]


def _sync_worker(rank, world, port, q, layout):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ttdg_mgm_amd.engine import sync_universe as su
        torch.manual_seed(0)
        backbone = torch.nn.Linear(8, 256)                     # "summed" parameters: see only the local graphs
        unused = torch.nn.Parameter(torch.zeros(3))            # never receives a gradient on any rank
        only0 = torch.nn.Parameter(torch.ones(256))            # receives a gradient on rank 0 only
        w = torch.nn.Parameter(torch.full((256,), 0.5))        # "replicated" parameter
        xs, labs = _graphs(100 + rank, layout[rank])
        nodes = [backbone(x) * (only0 if rank == 0 else 1.0) for x in xs] if xs else None
        all_nodes, all_labs = su.gather_graphs(nodes, labs if xs else None, torch.device("cpu"))
        loss = _toy_loss(all_nodes, w)
        loss.backward()
        su.allreduce_grads([backbone.weight, backbone.bias, unused, only0], [w])
        q.put(_ship((rank, float(loss), [tuple(t.shape) for t in all_nodes], [l.tolist() for l in all_labs], backbone.weight.grad.clone(),
                     backbone.bias.grad.clone(), w.grad.clone(), unused.grad is None, only0.grad.clone())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout", [((3, 5), (4,)), ((6, 2), ())])
def test_sync_universe_gather_and_gradient_allreduce(layout):
    """World-size-2 gloo: the gathered multi-graph is the rank-major concatenation, the loss is replicated, and after the
    all-reduce every rank holds exactly the gradients of the single-process computation on all graphs (a rank with no
    graph at all included); a parameter without gradient anywhere keeps grad None."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + len(layout[1])
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q, layout)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((_unship(q.get(timeout=120)) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on all graphs
    torch.manual_seed(0)
    backbone = torch.nn.Linear(8, 256)
    only0 = torch.nn.Parameter(torch.ones(256))
    w = torch.nn.Parameter(torch.full((256,), 0.5))
    nodes, labs = [], []
    for r in range(2):
        xs, ls = _graphs(100 + r, layout[r])
        nodes += [backbone(x) * (only0 if r == 0 else 1.0) for x in xs]
        labs += [l.tolist() for l in ls]
    loss = _toy_loss(nodes, w)
    loss.backward()
    for rank, l, shapes, glabs, gw, gb, gwr, unused_none, g0 in res:
        assert shapes == [tuple(n.shape) for n in nodes] and glabs == labs
        assert abs(l - float(loss.detach())) <= 1e-6 * max(1.0, abs(float(loss.detach())))
        assert torch.allclose(gw, backbone.weight.grad, rtol=1e-5, atol=1e-7)
        assert torch.allclose(gb, backbone.bias.grad, rtol=1e-5, atol=1e-7)
        assert torch.allclose(gwr, w.grad, rtol=1e-5, atol=1e-7)          # replicated: averaged, not summed
        assert torch.allclose(g0, only0.grad, rtol=1e-5, atol=1e-7)       # gradient on rank 0 only -> zeros from rank 1
        assert unused_none
    assert torch.equal(res[0][4], res[1][4]) and torch.equal(res[0][6], res[1][6])


def _overlap_worker(rank, world, port, q, layout):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ttdg_mgm_amd.engine import sync_universe as su

        def build():
            torch.manual_seed(0)
            net = torch.nn.Sequential(torch.nn.Linear(8, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 256))
            unused = torch.nn.Parameter(torch.zeros(3))
            w = torch.nn.Parameter(torch.full((256,), 0.5))
            sometimes = torch.nn.Parameter(torch.ones(256))        # data dependent: a gradient on steps 1 and 3 only (a late joiner,
            return net, unused, w, sometimes                       # and a live parameter without gradient anywhere on step 2)

        def step_loss(net, w, sometimes, step):
            xs, labs = _graphs(100 + rank + 10 * step, layout[step % len(layout)][rank])
            nodes = [net(x) * (sometimes if step % 2 == 1 else 1.0) for x in xs] if xs else None
            all_nodes, _ = su.gather_graphs(nodes, labs if xs else None, torch.device("cpu"))
            return _toy_loss(all_nodes, w)

        out = {}
        for mode in ("posthoc", "overlap"):
            net, unused, w, sometimes = build()
            summed, repl = list(net.parameters()) + [unused, sometimes], [w]
            red = su.OverlappedGradReducer(summed, repl, bucket_bytes=4096) if mode == "overlap" else None     # several buckets
            grads = []
            for step in range(4):
                for p in summed + repl:
                    p.grad = None
                loss = step_loss(net, w, sometimes, step)
                if red is None:
                    loss.backward()
                    su.allreduce_grads(summed, repl, bucket_bytes=4096)
                else:
                    red.prepare()
                    loss.backward()
                    red.finalize()
                grads.append([None if p.grad is None else p.grad.clone() for p in summed + repl])
                with torch.no_grad():
                    for p in summed + repl:
                        if p.grad is not None:
                            p -= 0.01 * p.grad
            out[mode] = grads
            if red is not None:
                out["launched_in_backward"] = red.overlapped_launches
                out["nbuckets"] = len(red.buckets)
                out["late"] = red.late_joins
                red.remove()
        same = all((a is None and b is None) or torch.equal(a, b) for ga, gb in zip(out["posthoc"], out["overlap"]) for a, b in zip(ga, gb))
        none_pattern = [[g is None for g in step] for step in out["overlap"]]
        q.put(_ship((rank, same, out["launched_in_backward"], out["nbuckets"], out["overlap"][-1][0].clone(), out["overlap"][-1][-3] is None, out["late"],
                     none_pattern == [[g is None for g in step] for step in out["posthoc"]], [step[-2] is None for step in out["overlap"]])))
    finally:
        dist.destroy_process_group()


def test_overlapped_gradient_allreduce_equals_posthoc_bit_for_bit():
    """Mode S, VERDICT r2 item 9: gradient buckets launched from autograd hooks in reverse layer order (the all-reduce
    overlaps the rest of the backward) give exactly the gradients of the post-hoc bucketed reduction, over several steps,
    including steps where one rank holds no graph at all (no hook fires there: everything is launched by finalize, in the
    same order), a parameter that never receives a gradient, and a parameter whose gradient is data dependent (it joins the live
    set late, and returns to ``grad is None`` on a step in which no rank differentiates it - as the single-GPU step would)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    layout = [((3, 5), (4,)), ((6, 2), ()), ((2,), (3, 3))]
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q, layout)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((_unship(q.get(timeout=180)) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, launched, nb, g_last, unused_none, late, same_none, sometimes_none in res:
        assert same, "overlapped reduction differs from the post-hoc one on rank %d" % rank
        assert nb >= 3 and unused_none and same_none
        assert late == 1                                   # `sometimes` joined the live set on step 1 ...
        assert sometimes_none == [True, False, True, False]    # ... and went back to grad None on step 2, where no rank had a gradient for it
    assert res[0][2] >= 3                        # rank 0 always has graphs: buckets did go out from inside backward
    assert torch.equal(res[0][4], res[1][4])     # replicas hold identical gradients


def _lockstep_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ttdg_mgm_amd import data
        from ttdg_mgm_amd.config import get_cfg
        from ttdg_mgm_amd.engine import BaselineTrainer

        class Model:
            sync_universe = True
            device = torch.device("cpu")

        cfg = get_cfg()
        cfg.TEST.BATCH = 2
        data.register_synthetic("lock_ds", 5, size=64)                # shards: rank 0 -> 3 images (2 batches), rank 1 -> 2 images (1 batch)
        loader = data.build_detection_test_loader(cfg, "lock_ds", rank, world)
        steps = BaselineTrainer.tta_batches(Model(), loader)
        capped = BaselineTrainer.tta_batches(Model(), loader, 1)
        plain = Model()
        plain.sync_universe = False
        q.put((rank, [None if b is None else [d["image_id"] for d in b] for b in steps], len(capped), len(list(BaselineTrainer.tta_batches(plain, loader)))))
    finally:
        dist.destroy_process_group()


def test_sync_universe_keeps_ranks_in_lockstep_on_uneven_shards():
    """Mode S: every rank takes max-over-ranks adaptation steps and feeds None once its shard is exhausted (the collectives
    inside a step must be entered by all ranks); Mode R keeps each rank's own count."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_lockstep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((_unship(q.get(timeout=120)) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [[0, 1], [2]] and res[1][1] == [[3, 4], None]
    assert res[0][2] == 1 and res[1][2] == 1
    assert res[0][3] == 2 and res[1][3] == 1


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the script spawns one rank per GPU on 127.0.0.1 as
    the reference's launch(main, num_gpus) does (train_net.py:94-101).  Here with gloo and --plumbing-only: rendezvous, shard
    arithmetic, barrier, max-over-ranks timing and the Dice all-gather - everything of the N > 1 path that is not GPU work -
    in the weak (--steps) and the strong (--images, cfg-4) scaling mode."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    for extra, want in ((["--steps", "3", "--warmup", "1"], dict(steps=3, scaling="weak", images_total=32)),
                        (["--images", "64", "--warmup", "2"], dict(steps=8, scaling="strong"))):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--plumbing-only"] + extra,
                           capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["backend"] == "gloo" and line["gathered_scores"] == 3
        assert abs(line["max_time"] - 0.002) < 1e-9
        st = line["strong"]                            # stand-in clocks: sharded 2 ms (max over the ranks), rank 0 alone 1.8 ms
        assert st["images"] == 512 and st["steps_per_rank"] == 64 and st["ranks"] == 2
        assert abs(st["value"] - 512 / 0.002) < 1e-6 and abs(st["efficiency_vs_n1"] - 0.0018 / (2 * 0.002)) < 1e-12
        for k, v in want.items():
            assert line[k] == v, (k, line)
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--plumbing-only", "--images", "30"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert bad.returncode != 0                         # 30 images do not split into 2 ranks x batches of 4
