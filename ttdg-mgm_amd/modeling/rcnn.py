"""DAobjTwoStagePseudoLabGeneralizedRCNN — mirror of the reference meta-architecture (meta_arch/rcnn.py:67-359)
for the two branches ``train_net.py --eval-only`` uses: ``branch='TTT'`` (:331-357) and eval-mode inference
(:179-182 -> detectron2 GeneralizedRCNN.inference [3P]).  Attribute names (backbone, proposal_generator, roi_heads,
D_img, graph_generator, multi_matching_sup, multi_matching_unsup) are the checkpoint contract."""
import torch
import torch.nn as nn

from ..GModule import MGM3_unsup, PrototypeComputation, U_sup
from .backbone import FPN
from .detector import PseudoLabRPN, StandardROIHeadsPseudoLab, detector_postprocess, detector_postprocess_batch  # noqa: F401
from .structures import Boxes, ImageList, Instances


class FCDiscriminator_img(nn.Module):
    """Image-level domain discriminator (rcnn.py:30-49); parameters only — the adversarial branch is training-time."""

    def __init__(self, num_classes, ndf1=256, ndf2=128):
        super().__init__()
        self.conv1 = nn.Conv2d(num_classes, ndf1, 3, padding=1)
        self.conv2 = nn.Conv2d(ndf1, ndf2, 3, padding=1)
        self.conv3 = nn.Conv2d(ndf2, ndf2, 3, padding=1)
        self.classifier = nn.Conv2d(ndf2, 1, 3, padding=1)


_SIDE_STREAMS = {}          # device -> the second HIP stream of the TTT branch
OVERLAP_DETECTOR = True     # teacher-forced TTT steps: run the (unused) RPN + box head on a second HIP stream next to the solver
DENSE_INFERENCE = True      # eval-mode inference on padded tensors (one host read per batch); False = the list-of-Instances path


class DAobjTwoStagePseudoLabGeneralizedRCNN(nn.Module):
    def __init__(self, *, backbone, proposal_generator, roi_heads, pixel_mean, pixel_std, input_format=None,
                 vis_period=0, dis_type="p2"):
        super().__init__()
        self.backbone = backbone
        self.proposal_generator = proposal_generator
        self.roi_heads = roi_heads
        self.input_format, self.vis_period, self.dis_type = input_format, vis_period, dis_type
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)
        self.D_img = FCDiscriminator_img(self.backbone._out_feature_channels[self.dis_type])
        sample_dist, univ_size = 10, 32                                            # rcnn.py:115-116
        self.graph_generator = PrototypeComputation(self.roi_heads.num_classes, sample_dist)
        self.multi_matching_sup = U_sup(self.roi_heads.num_classes, univ_size)
        self.multi_matching_unsup = MGM3_unsup(self.roi_heads.num_classes, univ_size)
        self.sync_universe = False      # Mode S (engine/sync_universe.py): all ranks adapt on one gathered multi-graph
        self.teacher_forced = False     # synthetic runs: replace detections by the jittered GT boxes the inputs carry
        self.autocast_backbone = False  # cfg-5: bf16 autocast for the backbone only (True / "all"), or up to a stage ("res2" .. "res5": the rest fp32)

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        raw = [x["image"] for x in batched_inputs]
        d = self.backbone.size_divisibility
        if len({tuple(t.shape) for t in raw}) == 1 and raw[0].shape[-2] % max(d, 1) == 0 and raw[0].shape[-1] % max(d, 1) == 0:
            # one normalisation for the batch (same-size, already divisible images: the test streams of this path)
            x = torch.stack([t.to(self.device, non_blocking=True) for t in raw]).float()
            return ImageList((x - self.pixel_mean) / self.pixel_std, [tuple(t.shape[-2:]) for t in raw])
        images = [(t.to(self.device, non_blocking=True).float() - self.pixel_mean) / self.pixel_std for t in raw]
        return ImageList.from_tensors(images, self.backbone.size_divisibility)

    def _backbone(self, x):
        mode = self.autocast_backbone
        if isinstance(mode, str) and mode != "all":
            # bf16 up to the named ResNet stage, fp32 behind it (a precision island: the FPN and the late stages decide the scores)
            self.backbone.bottom_up.autocast_upto = mode
            try:
                return self.backbone(x)
            finally:
                self.backbone.bottom_up.autocast_upto = None
        if mode:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                f = self.backbone(x)
            return {k: v.float() for k, v in f.items()}
        from . import graphed
        if graphed.ENABLED and x.is_cuda and not self.sync_universe:
            # A/B switch (off by default): fixed-shape fp32 batches replay the backbone's no-grad forward from a hipGraph
            g = self.__dict__.get("_graphed")
            if g is None:
                g = self.__dict__["_graphed"] = graphed.GraphedBackbone(self.backbone)
            f = g(x)
            if f is not None:
                return f
        return self.backbone(x)

    def forward(self, batched_inputs, branch=None, given_proposals=None, val_mode=False):
        if not self.training and not val_mode:
            return self.inference(batched_inputs)
        if batched_inputs is None:
            # Mode S only: this rank's shard has no batch left, but it still joins the all-gather and the replicated matching
            if not (branch == "TTT" and self.sync_universe):
                raise ValueError("batched_inputs=None is only meaningful for a synchronous-universe TTT step")
            from ..engine import sync_universe
            nodes, labels = sync_universe.gather_graphs(None, None, self.device)
            return self.multi_matching_unsup(nodes, labels, self.multi_matching_sup.U), [], [], []
        images = self.preprocess_image(batched_inputs)
        features = self._backbone(images.tensor)
        if branch == "TTT":
            if DENSE_INFERENCE:
                # RPN + box head on padded tensors: one host read (the detection counts) instead of two - none at all
                # when the detections are replaced by the teacher-forced boxes (the kernels still run)
                side = None
                if self.teacher_forced and OVERLAP_DETECTOR and images.tensor.is_cuda:
                    # teacher-forced detections: the RPN and the box head still run, but nothing downstream waits for
                    # them, so they go to a second HIP stream and overlap the (single-CU, latency-bound) matching solver;
                    # joined before the step returns.  With the detector's own boxes the chain is sequential and this
                    # branch is not taken.
                    cur = torch.cuda.current_stream()
                    side = _SIDE_STREAMS.get(images.tensor.device)
                    if side is None:
                        side = _SIDE_STREAMS[images.tensor.device] = torch.cuda.Stream(device=images.tensor.device)
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        boxes, scores, keep, pcounts = self.proposal_generator.forward_dense(features, images.image_sizes)
                        side_out = self.roi_heads.box_dense(features, boxes, scores, keep, images.image_sizes, pcounts)      # alive until the join
                else:
                    boxes, scores, keep, pcounts = self.proposal_generator.forward_dense(features, images.image_sizes)
                    dboxes, dscores, dcls, dcounts = self.roi_heads.box_dense(features, boxes, scores, keep, images.image_sizes, pcounts)
                if self.teacher_forced:
                    proposals_roih = [self._forced(x, sz) for x, sz in zip(batched_inputs, images.image_sizes)]
                else:
                    proposals_roih = [Instances(sz, pred_boxes=Boxes(dboxes[b, :n]), scores=dscores[b, :n], pred_classes=dcls[b, :n])
                                      for b, (n, sz) in enumerate(zip(dcounts.tolist(), images.image_sizes))]
            else:
                proposals_rpn, _ = self.proposal_generator(images, features, None, compute_loss=False)
                proposals_roih, _ = self.roi_heads(images, features, proposals_rpn, targets=None, compute_loss=False, branch=branch)
                if self.teacher_forced:
                    proposals_roih = [self._forced(x, sz) for x, sz in zip(batched_inputs, images.image_sizes)]
            feats = [features[k] for k in ("p2", "p3", "p4", "p5", "p6")]
            nodes, labels = self.graph_generator(feats, proposals_roih)
            if self.sync_universe:
                # Mode S (engine/sync_universe.py): one all-gather turns the per-rank graphs into the global multi-graph
                from ..engine import sync_universe
                nodes, labels = sync_universe.gather_graphs(nodes, labels, self.device)
            loss = self.multi_matching_unsup(nodes, labels, self.multi_matching_sup.U)
            if DENSE_INFERENCE and side is not None:
                torch.cuda.current_stream().wait_stream(side)      # join: backward / SGD start after the detector work
                del side_out
            return loss, [], [], feats
        if branch == "supervised":
            # source-training matching term only (rcnn.py:262-266): graph nodes sampled inside the
            # GROUND-TRUTH boxes, U_sup.forward -> loss_matching.  The detector's own RPN / ROI / mask losses of that
            # branch (rcnn.py:236-249) belong to source training proper and are out of scope (SURVEY.md §2 OUT).
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
            feats = [features[k] for k in ("p2", "p3", "p4", "p5", "p6")]
            nodes, labels = self.graph_generator(feats, gt_instances)
            if nodes is None:
                return {}, [], [], feats
            return {"loss_matching": self.multi_matching_sup(nodes, labels)}, [], [], feats
        raise NotImplementedError("branch {!r} is a source-training branch; only 'TTT', the matching term of 'supervised' and "
                                  "eval inference are built".format(branch))

    def _forced(self, x, size):
        b = x["tf_boxes"].to(self.device).float()
        return Instances(size, pred_boxes=Boxes(b), scores=torch.ones(len(b), device=self.device),
                         pred_classes=x["tf_classes"].to(self.device))

    @torch.no_grad()
    def inference(self, batched_inputs, do_postprocess=True):
        images = self.preprocess_image(batched_inputs)
        features = self._backbone(images.tensor)
        out_sizes = [(x.get("height", sz[0]), x.get("width", sz[1])) for x, sz in zip(batched_inputs, images.image_sizes)]
        if do_postprocess and DENSE_INFERENCE and len(set(out_sizes)) == 1:
            # padded tensors from the RPN to the pasted masks: one host read per batch instead of four
            boxes, scores, keep, pcounts = self.proposal_generator.forward_dense(features, images.image_sizes)
            res = self.roi_heads.inference_dense(features, boxes, scores, keep, images.image_sizes, out_sizes[0], counts=pcounts)
            return [{"instances": r} for r in res]
        proposals, _ = self.proposal_generator(images, features, None, compute_loss=False)
        results, _ = self.roi_heads(images, features, proposals, None, compute_loss=False, branch="")
        if not do_postprocess:
            return results
        sizes = [(x.get("height", r.image_size[0]), x.get("width", r.image_size[1])) for r, x in zip(results, batched_inputs)]
        return [{"instances": r} for r in detector_postprocess_batch(results, sizes)]


@torch.no_grad()
def calibrate_frozen_bn(model, batched_inputs):
    """Synthetic-weights helper (no pretrained checkpoints exist offline): give every FrozenBN the statistics a
    trained network would carry, by one forward pass that sets running_mean/var to the batch statistics layer by
    layer.  Without it random-init activations grow to ~1e3 through the 50 frozen-BN layers and the detector
    emits non-finite boxes.  Deterministic given the weights and the calibration batch."""
    import torch.nn.functional as F
    from .backbone import ConvNorm
    hooks = []

    def pre(mod, args):          # statistics of the raw convolution output, set before the folded conv runs
        y = F.conv2d(args[0].float(), mod.weight, None, mod.stride, mod.padding)
        mod.norm.running_mean.copy_(y.mean(dim=(0, 2, 3)))
        mod.norm.running_var.copy_(y.var(dim=(0, 2, 3), unbiased=False).clamp_min(1e-6))

    for m in model.modules():
        if isinstance(m, ConvNorm) and m.norm is not None:
            hooks.append(m.register_forward_pre_hook(pre))
    from . import backbone as _bb
    images = model.preprocess_image(batched_inputs)
    saved, _bb.FUSED_EPILOGUE = _bb.FUSED_EPILOGUE, False        # the statistics hooks sit on ConvNorm.forward
    try:
        model.backbone(images.tensor)
    finally:
        _bb.FUSED_EPILOGUE = saved
        for h in hooks:
            h.remove()
    return model


def build_model(cfg):
    """Trainer.build_model(cfg) equivalent for the keys test_segment.yaml sets (config.py:5-64, Base-RCNN-FPN.yaml)."""
    m = DAobjTwoStagePseudoLabGeneralizedRCNN(
        backbone=FPN(freeze_at=2),
        proposal_generator=PseudoLabRPN(),
        roi_heads=StandardROIHeadsPseudoLab(cfg.MODEL.ROI_HEADS.NUM_CLASSES, cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
                                            cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST, cfg.TEST.DETECTIONS_PER_IMAGE),
        pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, input_format=cfg.INPUT.FORMAT,
        dis_type=cfg.SEMISUPNET.DIS_TYPE)
    return m.to(torch.device(cfg.MODEL.DEVICE))
