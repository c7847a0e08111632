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


def _model_and_batches(n_steps, batch, size, teacher_forced):
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.modeling import build_model, detector
    cfg = get_cfg()
    cfg.MODEL.DEVICE = "cpu"
    cfg.TEST.BATCH = batch
    name = "synthfundus_cpu_baseline"
    data.register_synthetic(name, n_steps * batch, size=size, cfg_id=2)
    torch.manual_seed(0)
    model = build_model(cfg)
    model.teacher_forced = teacher_forced
    batches = list(data.build_detection_test_loader(cfg, name))
    from ttdg_mgm_amd.modeling import calibrate_frozen_bn
    calibrate_frozen_bn(model, batches[0])
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


def time_steps(n_steps, batch, size, teacher_forced=True):
    """Seconds for n_steps adaptation steps followed by the eval pass over the same batches (no warm-up: the CPU
    path has no autotuning; the first step is representative)."""
    from ttdg_mgm_amd.evaluation import DiceEvaluator
    cfg, model, batches, detector, name = _model_and_batches(n_steps, batch, size, teacher_forced)
    saved = detector._backend
    detector._backend = _CpuBackend
    try:
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
        dice.evaluate()
        return time.perf_counter() - t0
    finally:
        detector._backend = saved
