"""BaselineTrainer — mirror of the reference trainer surface ``train_net.py --eval-only`` uses
(engine/trainer.py:430-529 ``test``; :1230-1360 ``inference_on_dataset``; build_* classmethods).

``test`` keeps the reference's order: for every dataset, one adaptation step per batch
(``loss = model(inputs, branch='TTT')``; skip on None; zero_grad / backward / step), THEN the Dice pass over the
same loader with the adapted weights; model and optimizer state carry over between datasets (continual TTA)."""
import logging
import time
from collections import OrderedDict, defaultdict

import torch

from ..data import build_detection_test_loader
from ..evaluation import DiceEvaluator
from ..modeling import build_model
from ..optim import FusedSGD


EVAL_STREAMS = 1      # concurrent eval batches per GPU (HIP streams, one host thread each); 1 = the plain loop.  With the fused
                      # detection pipelines an eval batch is ~5 ms of Python for ~19 ms of GPU work, so one stream is GPU-bound;
                      # two threads only add GIL traffic (1 core: 62.2 vs 59.4 images/s, >= 2 cores: equal).  2 pays when the
                      # host needs longer to issue a batch than the GPU to run it.
EVAL_COALESCE = 1     # loader batches merged into one inference call in the Dice pass (eval-mode inference is per image:
                      # FrozenBN, no cross-image op, so the merge is invisible in the results).  Off by default: measured on
                      # MI355X the fp32 convolutions gain nothing at batch 16 (44.8 vs 49.3 images/s) and a batch size the
                      # warm-up has not seen costs a one-time MIOpen solver search of ~25 s


def _eval_group(model, group, evaluator):
    """One inference call over the images of `group` (a list of loader batches); the evaluator still sees loader batches."""
    if len(group) == 1:
        evaluator.process(group[0], model(group[0]))
        return
    outs = model([x for b in group for x in b])
    start = 0
    for b in group:
        evaluator.process(b, outs[start:start + len(b)])
        start += len(b)


def run_eval_batches(model, batches, evaluator, streams=None, coalesce=None):
    """Eval-mode forward + evaluator.process over independent batches.  The weights are frozen during the Dice pass, so
    the batches do not depend on each other: they are spread over `streams` HIP streams, each fed by its own host thread
    (the ~2000 small kernels of one inference are launch/latency bound on a single stream, and a thread blocked in one of
    the few host reads of the detector - NMS keep lists, mask counts - no longer idles the GPU); `coalesce` consecutive
    loader batches share one inference call (fewer, fatter launches: sized for the GPU, not for the loader).  Results
    are identical to the sequential loop up to the order the evaluator receives the batches in."""
    streams = EVAL_STREAMS if streams is None else streams
    coalesce = EVAL_COALESCE if coalesce is None else coalesce
    dev_is_gpu = next(model.parameters()).is_cuda
    if not dev_is_gpu:
        coalesce = 1
    if (streams <= 1 or not dev_is_gpu) and coalesce <= 1:
        # the plain loop: the loader is consumed lazily, one batch alive at a time (a streaming loader stays bounded by its
        # own prefetch depth; only the multi-stream / coalescing modes below need the shard as a list)
        with torch.no_grad():
            for batch in batches:
                _eval_group(model, [batch], evaluator)
                del batch
        return
    batches = list(batches)
    batches = [batches[i:i + max(1, coalesce)] for i in range(0, len(batches), max(1, coalesce))]     # groups of loader batches
    if streams <= 1 or not dev_is_gpu or len(batches) < 3:
        with torch.no_grad():
            for group in batches:
                _eval_group(model, group, evaluator)
        return
    import threading
    with torch.no_grad():       # the first group runs on the caller's stream: every lazily built constant (folded filters,
        _eval_group(model, batches[0], evaluator)     # anchors, size tables) exists before the side streams start
    batches = batches[1:]
    main = torch.cuda.current_stream()
    side = [torch.cuda.Stream() for _ in range(streams)]
    errors = []

    def work(k):
        try:
            torch.cuda.set_device(main.device)
            with torch.cuda.stream(side[k]), torch.no_grad():
                for group in batches[k::streams]:
                    _eval_group(model, group, evaluator)
        except BaseException as e:       # surfaced on the caller's thread
            errors.append(e)

    for st in side:
        st.wait_stream(main)
    threads = [threading.Thread(target=work, args=(k,)) for k in range(streams)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for st in side:
        main.wait_stream(st)
    if errors:
        raise errors[0]


def inference_on_dataset(model, data_loader, evaluator, cfg=None):
    """Eval-mode, no-grad pass (reference :1230-1360); returns (results, evaluator)."""
    was_training = model.training
    model.eval()
    evaluator.reset()
    run_eval_batches(model, data_loader, evaluator)
    model.train(was_training)
    if hasattr(evaluator, "gather_scores"):
        evaluator.gather_scores()      # no-op on one rank; the all-gather the reference omits (it reports rank-local means)
    results = evaluator.evaluate()
    return (results if results is not None else {}), evaluator


class BaselineTrainer:
    rank, world = 0, 1            # set by the launcher (one process per GPU)
    device = None                 # inputs are uploaded to this device when set
    resident_inputs = True        # True: a shard is decoded / resized / uploaded once and stays in HBM (bench, tests);
                                  # False: streamed batch by batch with a bounded prefetch on every pass (train_net.py)

    @classmethod
    def build_model(cls, cfg):
        return build_model(cfg)

    @classmethod
    def build_optimizer(cls, cfg, model):
        """detectron2.solver.build_optimizer [3P]: SGD(momentum), one group per tensor, no weight decay on norms."""
        groups = []
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            wd = cfg.SOLVER.WEIGHT_DECAY_NORM if (".norm." in name or "layer_norm" in name) else cfg.SOLVER.WEIGHT_DECAY
            groups.append({"params": [p], "weight_decay": wd})
        return FusedSGD(groups, lr=cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)

    @classmethod
    def build_test_loader(cls, cfg, dataset_name):
        return build_detection_test_loader(cfg, dataset_name, cls.rank, cls.world, cls.device, resident=cls.resident_inputs)

    @classmethod
    def tta_step(cls, model, optimizer, inputs):
        """One adaptation step (reference :476-482).  Returns the loss tensor or None when skipped.
        With ``model.sync_universe`` (Mode S, engine/sync_universe.py) ``inputs`` may be None on a rank whose shard has
        run out: the skip decision is then global (the gathered multi-graph is the same on every rank) and the gradients
        are all-reduced before the step, so every replica takes the same step."""
        loss, _, _, _ = model(inputs, branch='TTT')
        if loss is None:
            return None
        optimizer.zero_grad(set_to_none=True)
        if getattr(model, "sync_universe", False):
            from . import sync_universe
            red = model.__dict__.get("_grad_reducer")
            if red is None:          # buckets launched from autograd hooks: the all-reduce overlaps the backbone backward
                red = model.__dict__["_grad_reducer"] = sync_universe.OverlappedGradReducer(*sync_universe.split_params(model))
            red.prepare()
            loss.backward()
            red.finalize()
        else:
            loss.backward()
        optimizer.step()
        return loss

    @classmethod
    def tta_batches(cls, model, data_loader, limit=None):
        """The batches one rank adapts on.  Mode S keeps the ranks in lockstep: every rank takes max-over-ranks steps and
        feeds None once its own shard is exhausted."""
        if not getattr(model, "sync_universe", False):
            def stream():           # nothing is held: the loader decides whether the shard is resident or streamed
                for bidx, inputs in enumerate(data_loader):
                    if limit is not None and bidx >= limit:
                        break
                    yield inputs
            return stream()
        batches = []
        for bidx, inputs in enumerate(data_loader):
            if limit is not None and bidx >= limit:
                break
            batches.append(inputs)
        if getattr(model, "sync_universe", False):
            import torch.distributed as dist
            n = torch.tensor([len(batches)], dtype=torch.int64)
            if dist.get_backend() == "nccl":
                n = n.to(model.device)
            dist.all_reduce(n, op=dist.ReduceOp.MAX)
            batches += [None] * (int(n) - len(batches))
        return batches

    @classmethod
    def test(cls, cfg, model, optimizer=None, evaluators=None, timers=None):
        logger = logging.getLogger(__name__)
        results = OrderedDict()
        for idx, dataset_name in enumerate(cfg.DATASETS.TEST):
            data_loader = cls.build_test_loader(cfg, dataset_name)
            t0 = time.perf_counter()
            if cfg.TEST.TTT:
                for inputs in cls.tta_batches(model, data_loader, cfg.TEST.MIN_BATCH_NUM):
                    cls.tta_step(model, optimizer, inputs)
            if timers is not None:
                torch.cuda.synchronize()
                timers.setdefault("tta_s", 0.0)
                timers["tta_s"] += time.perf_counter() - t0
            t1 = time.perf_counter()
            if evaluators is not None:
                dice = evaluators[idx]
            elif cls.resident_inputs:
                dice = DiceEvaluator(dataset_name, cfg.TEST.DICE_THRES, dataset_dicts=data_loader.dataset_dicts)   # ground truth of the local shard
                if cls.device is not None and torch.device(cls.device).type == "cuda":
                    dice.prestage(cls.device)
            else:
                dice = DiceEvaluator(dataset_name, cfg.TEST.DICE_THRES, lazy=True)      # ground truth travels with every streamed item
            results_i, _ = inference_on_dataset(model, data_loader, dice, cfg)
            if timers is not None:
                torch.cuda.synchronize()
                timers.setdefault("eval_s", 0.0)
                timers["eval_s"] += time.perf_counter() - t1
            results[dataset_name] = results_i
            assert isinstance(results_i, dict), "Evaluator must return a dict on the main process. Got {} instead.".format(results_i)
            logger.info("Evaluation results for {}: {}".format(dataset_name, results_i))
        fam = defaultdict(lambda: defaultdict(list))
        for key, value in results.items():
            for m in ("Dice Coefficient", "Enhanced Alignment Metric", "Structural Similarity Metric"):
                fam[key.split('_')[0]][m].append(value[m])
        for dname, metrics in fam.items():
            results[f"{dname}_mean"] = {m: sum(v) / len(v) for m, v in metrics.items()}
        return results
