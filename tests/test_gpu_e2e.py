"""GPU end-to-end checks of the TTA step on the Mask R-CNN stand-in (run with `-m gpu`): the device pipeline
(backbone on vendor kernels, our RPN/ROI helpers, sampler, matching loss, fused SGD) against the same modules run on
the host with the oracle's operators, on identical weights and inputs."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    assert torch.cuda.is_available()
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.engine import BaselineTrainer
    from ttdg_mgm_amd.modeling import calibrate_frozen_bn
    cfg = get_cfg()
    cfg.TEST.BATCH = 2
    cfg.INPUT.MIN_SIZE_TEST = 384          # smaller than the 800 of the bench: keeps the CPU side of the test quick
    data.register_synthetic("e2e_ds", 2, size=256)
    torch.manual_seed(0)
    cfg.MODEL.DEVICE = "cpu"
    cpu = BaselineTrainer.build_model(cfg)
    cpu.teacher_forced = True
    batch = next(iter(data.build_detection_test_loader(cfg, "e2e_ds")))
    calibrate_frozen_bn(cpu, batch)
    gpu = copy.deepcopy(cpu).to("cuda:0")
    return cfg, cpu, gpu, batch


def test_tta_forward_matches_host_pipeline(setup):
    from oracle import gmodule as og
    from oracle import tta_cpu
    from ttdg_mgm_amd.modeling import detector
    cfg, cpu, gpu, batch = setup
    gpu.train(), cpu.train()
    gpu.multi_matching_unsup.eval(), cpu.multi_matching_unsup.eval()      # attention dropout off on both sides
    gpu.multi_matching_unsup.keep_trace = True
    loss, _, _, feats = gpu(batch, branch="TTT")
    tr = gpu.multi_matching_unsup.last
    # host side: same modules, oracle operators
    saved = detector._backend
    detector._backend = tta_cpu._CpuBackend
    try:
        images = cpu.preprocess_image(batch)
        features = cpu.backbone(images.tensor)
        dets = [cpu._forced(x, sz) for x, sz in zip(batch, images.image_sizes)]
        hf = [features[k] for k in ("p2", "p3", "p4", "p5", "p6")]
        nodes, labels = og.prototype_computation(hf, [d.pred_boxes.tensor for d in dets], [d.pred_classes for d in dets])
    finally:
        detector._backend = saved
    for a, b in zip(feats, hf):                                          # vendor conv kernels: fp32, different summation order
        assert float((a.detach().cpu() - b).abs().max()) <= 2e-3 * max(1.0, float(b.abs().max()))
    assert tr["sizes"] == [len(n) for n in nodes]                        # identical node selection
    X = torch.cat(nodes)
    assert float((tr["X"].cpu() - X).abs().max()) <= 2e-3 * max(1.0, float(X.abs().max()))
    otr = {}
    p = dict(cpu.multi_matching_unsup.named_parameters())
    og.mgm3_unsup_forward(p, nodes, labels, cpu.multi_matching_sup.U, trace=otr)
    assert float((tr["Wds"].cpu() - otr["Wds"]).abs().max()) <= 2e-2     # Sinkhorn at tau=.05 amplifies the feature noise 20x
    assert torch.isfinite(loss) and float(loss) > 0


def test_tta_step_updates_exactly_the_reference_parameter_set(setup):
    from ttdg_mgm_amd.engine import BaselineTrainer
    cfg, cpu, gpu, batch = setup
    gpu = copy.deepcopy(gpu)
    gpu.train()
    opt = BaselineTrainer.build_optimizer(cfg, gpu)
    before = {k: v.detach().clone() for k, v in gpu.named_parameters()}
    loss = BaselineTrainer.tta_step(gpu, opt, batch)
    assert loss is not None and torch.isfinite(loss)
    changed = {k for k, v in gpu.named_parameters() if not torch.equal(v.detach(), before[k])}
    # res3-5, all FPN convs and node_affinity.* move; stem/res2, RPN/ROI heads, attention, U, D_img do not
    assert all(k.startswith(("backbone.bottom_up.res3", "backbone.bottom_up.res4", "backbone.bottom_up.res5", "backbone.fpn_",
                             "multi_matching_unsup.node_affinity.")) for k in changed), sorted(changed)[:5]
    assert any(k.startswith("backbone.bottom_up.res3") for k in changed) and any(k.startswith("backbone.fpn_lateral") for k in changed)
    # fc_M.2.bias (b2) is optional: Sinkhorn is invariant to a constant shift of the affinities, so d loss / d b2 is
    # rounding noise around 0 (starting from b2 = 0 the update may be exactly nothing)
    moved = {k for k in changed if k.startswith("multi_matching_unsup")} - {"multi_matching_unsup.node_affinity.fc_M.2.bias"}
    assert moved == {"multi_matching_unsup.node_affinity." + s for s in ("fc_M.0.weight", "fc_M.0.bias", "fc_M.2.weight",
                                                                        "project_sr.weight", "project_tg.weight")}
    assert torch.isfinite(torch.stack([v.detach().abs().max() for v in gpu.parameters()])).all()


def test_eval_pass_and_dice_run_on_device(setup):
    from ttdg_mgm_amd.engine import inference_on_dataset
    from ttdg_mgm_amd.evaluation import DiceEvaluator
    cfg, cpu, gpu, batch = setup
    ev = DiceEvaluator("e2e_ds", 0.0)               # threshold 0: exercise the Dice / E / S code on real predictions
    gpu.eval()
    with torch.no_grad():
        outs = gpu(batch)
    counts = [len(o["instances"]) for o in outs]
    assert sum(counts) > 0, counts
    assert all(o["instances"].pred_masks.shape[1:] == (256, 256) and o["instances"].pred_masks.dtype == torch.bool for o in outs)
    res, _ = inference_on_dataset(gpu, [batch], ev, cfg)
    assert len(ev.dice_scores) == sum(counts) and all(0 <= v <= 100 for v in ev.dice_scores)
    assert set(res) == {"Dice Coefficient", "Enhanced Alignment Metric", "Structural Similarity Metric"}
    assert gpu.training is True or gpu.training is False
