"""CPU "port" baseline of one adapted batch (TEST INFRASTRUCTURE, see oracle/__init__.py): the same Mask R-CNN
stand-in modules run on the host with torch-CPU kernels, detection helpers from oracle/detection.py, node sampling /
matching loss / SGD from oracle/gmodule.py (the op-for-op restatement of the reference's GModule, including its
materialised affinity MLP, per-pair loops and per-iteration scipy LAPs).  Timed by bench.py's cpu_baseline leg on the
GPU box's host cores; never on the product path."""
import time

import torch

from . import detection as odet
from . import gmodule as og


class _CpuBackend:
    nms = staticmethod(odet.nms)
    roi_align = staticmethod(odet.roi_align)

    @staticmethod
    def nms_launch(boxes, scores, thr, group=None, ngroups=None, max_group=None, topk=None):
        keep = odet.nms(boxes, scores, thr, group)
        return keep if topk is None else keep[:topk]

    @staticmethod
    def nms_collect(launched):
        return list(launched)

    # ---- batch-level box pipelines (same interface as ttdg_mgm_amd.ops, detectron2 formulation in plain torch) ----
    @staticmethod
    def image_sizes_tensor(sizes, device):
        return torch.tensor([(float(h), float(w)) for h, w in sizes], dtype=torch.float32, device=device).reshape(-1, 2)

    @staticmethod
    def rpn_decode(deltas, anchors, idx, score, sizes_t, boxes, scores, col0):
        from ttdg_mgm_amd.modeling.detector import apply_deltas
        B, A4, H, W = deltas.shape
        k = idx.shape[1]
        dl = deltas.view(B, -1, 4, H, W).permute(0, 3, 4, 1, 2).reshape(B, -1, 4)
        bx = apply_deltas(dl.gather(1, idx[..., None].expand(-1, -1, 4)).reshape(-1, 4), anchors[idx.reshape(-1)],
                          (1.0, 1.0, 1.0, 1.0)).reshape(B, k, 4)
        ok = torch.isfinite(bx).all(-1) & torch.isfinite(score)
        h, w = sizes_t[:, 0, None], sizes_t[:, 1, None]
        bx = torch.stack((torch.minimum(bx[..., 0].clamp(min=0), w), torch.minimum(bx[..., 1].clamp(min=0), h),
                          torch.minimum(bx[..., 2].clamp(min=0), w), torch.minimum(bx[..., 3].clamp(min=0), h)), -1)
        ok &= (bx[..., 2] - bx[..., 0] > 0) & (bx[..., 3] - bx[..., 1] > 0)
        boxes[:, col0:col0 + k] = torch.nan_to_num(bx, nan=0.0)
        scores[:, col0:col0 + k] = torch.where(ok, score, score.new_full((), float("-inf")))

    @classmethod
    def rpn_select(cls, logits, deltas, anchors, ks, sizes_t, boxes, scores):
        col = 0
        for lg, dl, an, k in zip(logits, deltas, anchors, ks):
            sc, idx = lg.permute(0, 2, 3, 1).reshape(lg.shape[0], -1).topk(k, dim=1)
            cls.rpn_decode(dl, an, idx, sc.float(), sizes_t, boxes, scores, col)
            col += k

    @staticmethod
    def nms_batched(boxes, scores, lvl, nlvl, thr, max_group, topk, device_counts=False, level_sizes=None):
        B, K = scores.shape
        idx = torch.zeros(B, min(int(topk), K), dtype=torch.int64)
        counts = []
        for b in range(B):
            live = torch.nonzero(scores[b] > float("-inf")).squeeze(1)
            keep = live[odet.nms(boxes[b][live], scores[b][live], thr, lvl[live])][:idx.shape[1]]
            idx[b, :len(keep)] = keep
            counts.append(int(len(keep)))
        return idx, (torch.tensor(counts) if device_counts else counts)

    @staticmethod
    def box_inference(logits, deltas, rois, sizes_t, num_classes, weights, score_thresh):
        from ttdg_mgm_amd.modeling.detector import apply_deltas
        n = logits.shape[0]
        scores = torch.softmax(logits, dim=-1)[:, :-1]
        boxes = apply_deltas(deltas, rois[:, 1:], weights).view(n, num_classes, 4)
        img = rois[:, 0].long()
        h, w = sizes_t[img, 0][:, None], sizes_t[img, 1][:, None]
        boxes = torch.stack((torch.minimum(boxes[..., 0].clamp(min=0), w), torch.minimum(boxes[..., 1].clamp(min=0), h),
                             torch.minimum(boxes[..., 2].clamp(min=0), w), torch.minimum(boxes[..., 3].clamp(min=0), h)), -1)
        ok = torch.isfinite(boxes).all(-1).all(-1) & torch.isfinite(scores).all(-1)
        good = ok[:, None] & (scores > score_thresh)
        return torch.nan_to_num(boxes, nan=0.0), torch.where(good, scores, scores.new_full((), float("-inf")))

    @staticmethod
    def nms_ragged(boxes, scores, rois_per_image, num_classes, thr, topk):
        out, start = [], 0
        C = num_classes
        for n in rois_per_image:
            sc = scores[start:start + n].reshape(-1)
            bx = boxes[start:start + n].reshape(-1, 4)
            live = torch.nonzero(sc > float("-inf")).squeeze(1)
            keep = live[odet.nms(bx[live], sc[live], thr, live % C)][:topk]
            out.append(keep + start * C)
            start += n
        return out

    @staticmethod
    def roi_align_multilevel(feats, rois, strides, P, canonical_size=224, canonical_level=4, min_level=2):
        """detectron2 ROIPooler [3P], level by level (the formulation the fused kernel replaces)."""
        R = rois.shape[0]
        out = feats[0].new_zeros((R, feats[0].shape[1], P, P), dtype=torch.float32)
        if R == 0:
            return out
        area = (rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])
        lvl = torch.floor(canonical_level + torch.log2(torch.sqrt(area.clamp(min=0)) / canonical_size + 1e-8))
        lvl = lvl.clamp(min_level, min_level + len(feats) - 1).long() - min_level
        for l, (f, s) in enumerate(zip(feats, strides)):
            idx = torch.nonzero(lvl == l).squeeze(1)
            if idx.numel():
                out[idx] = odet.roi_align(f, rois[idx], 1.0 / s, P)
        return out

    @staticmethod
    def paste_masks(masks, boxes, H, W, threshold=0.5):
        from ttdg_mgm_amd.modeling.detector import paste_masks_in_image_torch
        return paste_masks_in_image_torch(masks.reshape(masks.shape[0], 1, masks.shape[-2], masks.shape[-1]), boxes, (H, W), threshold)


def _model_and_batches(n_steps, batch, size, teacher_forced, weights=None, first_batch=0, perturb=None):
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.modeling import build_model, detector
    cfg = get_cfg()
    cfg.MODEL.DEVICE = "cpu"
    cfg.TEST.BATCH = batch
    name = "synthfundus_cpu_baseline"
    data.register_synthetic(name, (first_batch + n_steps) * batch, size=size, cfg_id=2)        # the first images of bench.py's stream
    torch.manual_seed(0)
    model = build_model(cfg)
    model.teacher_forced = teacher_forced
    batches = list(data.build_detection_test_loader(cfg, name))[first_batch:]
    if weights:
        from ttdg_mgm_amd.engine.checkpoint import load_weights
        load_weights(model, weights)
    else:
        from ttdg_mgm_amd.modeling import calibrate_frozen_bn
        calibrate_frozen_bn(model, batches[0])
    if perturb:
        # a rounding-sized relative perturbation of every trainable tensor (drift studies: how far does the port's OWN
        # free-running trajectory move when its arithmetic is disturbed at the level of one fp32 ulp?)
        eps, seed = perturb
        g = torch.Generator().manual_seed(int(seed))
        with torch.no_grad():
            for q in model.parameters():
                if q.requires_grad:
                    q.mul_(1 + eps * torch.randn(q.shape, generator=g))
    return cfg, model, batches, detector, name


def tta_step(model, inputs, bufs, cfg):
    """One adaptation step, reference order (engine/trainer.py:476-482, meta_arch/rcnn.py:331-357)."""
    images = model.preprocess_image(inputs)
    features = model.backbone(images.tensor)
    props, _ = model.proposal_generator(images, features, None, compute_loss=False)
    dets, _ = model.roi_heads(images, features, props, None, compute_loss=False, branch="TTT")
    if model.teacher_forced:
        dets = [model._forced(x, sz) for x, sz in zip(inputs, images.image_sizes)]
    feats = [features[k] for k in ("p2", "p3", "p4", "p5", "p6")]
    nodes, labels = og.prototype_computation(feats, [d.pred_boxes.tensor for d in dets], [d.pred_classes for d in dets])
    p = dict(model.multi_matching_unsup.named_parameters())
    loss = og.mgm3_unsup_forward(p, nodes, labels, model.multi_matching_sup.U)
    if loss is None:
        return None
    params = [q for q in model.parameters() if q.requires_grad]
    for q in params:
        q.grad = None
    loss.backward()
    with torch.no_grad():
        og.sgd_step(params, [q.grad for q in params], bufs, cfg.SOLVER.BASE_LR, cfg.SOLVER.MOMENTUM, cfg.SOLVER.WEIGHT_DECAY)
    return loss


def _one_rep(model, cfg, batches, name):
    """n TTA steps, then the eval pass + Dice over the same batches (reference order); returns (seconds, Dice dict)."""
    from ttdg_mgm_amd.evaluation import DiceEvaluator
    model.train()
    bufs = [None] * len([q for q in model.parameters() if q.requires_grad])
    t0 = time.perf_counter()
    for b in batches:
        tta_step(model, b, bufs, cfg)
    model.eval()
    dice = DiceEvaluator(name, cfg.TEST.DICE_THRES, dataset_dicts=[it["dataset_dict"] for b in batches for it in b])
    with torch.no_grad():
        for b in batches:
            dice.process(b, model(b))
    res = dice.evaluate()
    dt = time.perf_counter() - t0
    res["kept_masks"] = len(dice.dice_scores)
    return dt, res


def run(n_steps, batch, size, teacher_forced=True, weights=None, reps=3, warmup=1, first_batch=0, perturb=None):
    """``warmup`` untimed-for-the-median repetitions, then ``reps`` timed ones, every repetition from the same initial
    weights (the state dict is restored outside the timed part).  Returns times, warm-up times and the Dice of the FIRST
    repetition (checkpoint -> n TTA steps -> eval), which is what bench.py compares with the GPU."""
    cfg, model, batches, detector, name = _model_and_batches(n_steps, batch, size, teacher_forced, weights, first_batch, perturb)
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    saved = detector._backend
    detector._backend = _CpuBackend
    try:
        times, wtimes, first = [], [], None
        for r in range(warmup + reps):
            model.load_state_dict(init)
            dt, res = _one_rep(model, cfg, batches, name)
            if first is None:
                first = res
            (wtimes if r < warmup else times).append(dt)
        return dict(times=times, warmup_times=wtimes, dice=first)
    finally:
        detector._backend = saved


def time_steps(n_steps, batch, size, teacher_forced=True):
    """Round-1 entry: one cold repetition."""
    return run(n_steps, batch, size, teacher_forced, reps=1, warmup=0)["times"][0]
