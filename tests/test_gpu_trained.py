"""Full-size end-to-end parity on the TRAINED-REGIME synthetic checkpoint (run with `-m gpu`).

BASELINE.json cfg-1 / cfg-2 shapes: 4 synthetic 512x512 2-class images -> 800x800 after the test mapper, TEST.BATCH = 4,
weights = the deterministic checkpoint tools/synth_checkpoint.py fits on a disjoint source stream (the same one bench.py
measures).  The device pipeline against the same modules on the host with the oracle's operators (the CPU side of cfg-1),
and the cfg-5 precision split (bf16 backbone / fp32 matching) against the fp32 backbone on the same checkpoint."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


@pytest.fixture(scope="module")
def trained():
    assert torch.cuda.is_available()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth_checkpoint as sc
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    from ttdg_mgm_amd.engine import BaselineTrainer
    from ttdg_mgm_amd.engine.checkpoint import load_weights
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "test_segment.yaml"))        # TEST.BATCH 4, 2 classes, 800 / 1333
    dev = torch.device("cuda:0")
    cfg.MODEL.DEVICE = str(dev)
    path, rep = sc.get_or_make(cfg, dev, log=lambda m: None)
    print("checkpoint:", rep)
    gpu = BaselineTrainer.build_model(cfg)
    load_weights(gpu, path)
    ccfg = cfg.clone()
    ccfg.MODEL.DEVICE = "cpu"
    cpu = BaselineTrainer.build_model(ccfg)
    load_weights(cpu, path)
    data.register_synthetic("trained_ds", 8, size=512, cfg_id=2)
    batches = list(data.build_detection_test_loader(ccfg, "trained_ds"))            # host-resident uint8 images
    return cfg, cpu, gpu, batches


def _host_backend():
    from oracle import tta_cpu
    from ttdg_mgm_amd.modeling import detector

    class swap:
        def __enter__(self):
            self.saved = detector._backend
            detector._backend = tta_cpu._CpuBackend

        def __exit__(self, *a):
            detector._backend = self.saved
    return swap()


def test_cfg2_full_size_tta_step_matches_host_pipeline(trained):
    """One TTA forward at the bench's size (4 x 800 x 800).  (1) the detector's own boxes agree between device and host
    (vendor convolutions differ in summation order: matched by IoU, not bit for bit); (2) with the SAME boxes on both sides
    the node selection is identical and the features agree to the vendor-kernel level; (3) the matching operators, fed the
    host's node features, reproduce Wds (probabilities and log domain), U0, the solver's first V, and - with the host run's
    permutations supplied - loss and gradients within 1e-4; (4) the free-running solve returns the host's permutations
    whenever the host's own solve converged in every stage."""
    from oracle import gmodule as og
    from ttdg_mgm_amd.modeling.structures import Boxes, Instances
    cfg, cpu, gpu, batches = trained
    batch = batches[0]
    gpu.train(), cpu.train()
    gpu.multi_matching_unsup.eval(), cpu.multi_matching_unsup.eval()        # attention dropout off on both sides
    # ---- host side: backbone, detector, sampler
    with _host_backend():
        images = cpu.preprocess_image(batch)
        features = cpu.backbone(images.tensor)
        props, _ = cpu.proposal_generator(images, features, None, compute_loss=False)
        dets, _ = cpu.roi_heads(images, features, props, None, compute_loss=False, branch="TTT")
    hf = [features[k] for k in ("p2", "p3", "p4", "p5", "p6")]
    nodes, labels = og.prototype_computation(hf, [d.pred_boxes.tensor for d in dets], [d.pred_classes for d in dets])
    # ---- device side, free-running
    gpu.teacher_forced = False
    gpu.multi_matching_unsup.keep_trace = True
    loss, _, _, feats = gpu(batch, branch="TTT")
    tr = gpu.multi_matching_unsup.last
    assert loss is not None and torch.isfinite(loss)
    for a, b in zip(feats, hf):
        rel = float((a.detach().cpu() - b).abs().max()) / max(1.0, float(b.abs().max()))
        assert rel <= 2e-3, rel
    # (1) confident detections agree
    gimg = gpu.preprocess_image(batch)
    with torch.no_grad():
        gfeat = gpu.backbone(gimg.tensor)
        gb, gs, gk, gc = gpu.proposal_generator.forward_dense(gfeat, gimg.image_sizes)
        dboxes, dscores, dcls, dcounts = gpu.roi_heads.box_dense(gfeat, gb, gs, gk, gimg.image_sizes, gc)
    import synth_checkpoint as sc
    for n, d in enumerate(dets):
        k = int(dcounts[n])
        hb, hs, hc = d.pred_boxes.tensor, d.scores, d.pred_classes
        conf = hs >= 0.5
        assert int(conf.sum()) >= 2, "the trained detector must find disc and cup"
        iou = sc.box_iou(hb[conf], dboxes[n, :k].cpu())
        best, arg = iou.max(1)
        assert float(best.min()) >= 0.9 and torch.equal(dcls[n, :k].cpu()[arg], hc[conf])
    # (2) same boxes on both sides -> identical node selection
    same = [Instances(d.image_size, pred_boxes=Boxes(d.pred_boxes.tensor.to("cuda:0")), scores=d.scores.to("cuda:0"),
                      pred_classes=d.pred_classes.to("cuda:0")) for d in dets]
    gn, gl = gpu.graph_generator([f.detach() for f in feats], same)
    assert [len(x) for x in gn] == [len(x) for x in nodes]
    for a, b in zip(gl, labels):
        assert torch.equal(a.cpu(), b)
    X = torch.cat(nodes)
    assert float((torch.cat(gn).cpu() - X).abs().max()) <= 2e-3 * max(1.0, float(X.abs().max()))
    # (3) matching operators on identical inputs
    p = dict(cpu.multi_matching_unsup.named_parameters())
    otr = {}
    ref_nodes = [x.detach().clone().requires_grad_() for x in nodes]
    ref_loss = og.mgm3_unsup_forward(p, ref_nodes, labels, cpu.multi_matching_sup.U, trace=otr)
    ref_loss.backward()
    m = gpu.multi_matching_unsup
    dn = [x.detach().to("cuda:0").requires_grad_() for x in nodes]
    dl = [l.to("cuda:0") for l in labels]
    tr2 = {}
    l2 = m(dn, dl, gpu.multi_matching_sup.U, trace=tr2, forced_U=otr["Ub"].to("cuda:0"))
    l2.backward()
    m.zero_grad()
    W, Wr = tr2["Wds"].cpu(), otr["Wds"]
    assert float((W - Wr).abs().max()) <= TOL
    live = Wr > 1e-20
    dlog = float((W[live].log() - Wr[live].log()).abs().max())
    print("Wds: |d| %.2e, log-domain |d| %.2e" % (float((W - Wr).abs().max()), dlog))
    assert dlog <= TOL
    assert float((tr2["U0"].cpu() - otr["U0"]).abs().max()) <= TOL * max(1.0, float(otr["U0"].abs().max()))
    assert abs(float(l2.detach()) - float(ref_loss.detach())) <= TOL
    for a, b in zip(dn, ref_nodes):
        assert float((a.grad.cpu() - b.grad).abs().max()) <= TOL * max(1.0, float(b.grad.abs().max()))
    # (4) free-running solve on identical inputs
    tr3 = {}
    with torch.no_grad():
        m([x.detach().to("cuda:0") for x in nodes], dl, gpu.multi_matching_sup.U, trace=tr3)
    it_dev, it_ref = tr3["info"].cpu().tolist()[:6], otr["iters"]
    print("solver iterations per stage: device", it_dev, "host", it_ref)
    assert float((tr3["V0"].cpu() - otr["V0"]).abs().max()) <= TOL * max(1.0, float(otr["V0"].abs().max()))
    # identical permutations wherever the host's own result is well defined: it converged in every stage AND does not change
    # under 1e-7-relative perturbations of its inputs (the rounding-stability criterion of tests/golden/make_golden.py)
    stable = max(it_ref) < 200
    if stable:
        from ttdg_mgm_amd import synth
        for k in range(2):
            g = synth.gen(4100 + k)
            pert = [x.detach() * (1 + 1e-7 * synth.normal(g, tuple(x.shape))) for x in nodes]
            t = {}
            og.mgm3_unsup_forward(p, pert, labels, cpu.multi_matching_sup.U, trace=t)
            stable = stable and torch.equal(t["Ub"], otr["Ub"])
    print("host solve rounding-stable:", stable)
    if stable:
        assert torch.equal(tr3["Ub"].cpu(), otr["Ub"]), "permutation matrices differ from the host pipeline"
        assert it_dev[:5] == it_ref[:5]


def test_cfg2_eval_dice_matches_host_on_trained_checkpoint(trained):
    """The eval half at full size on the trained checkpoint, TEST.DICE_THRES 0.9 as the reference sets it (config.py:14):
    Dice / E / S means of the device pipeline within 1e-3 relative of the host pipeline (BASELINE north_star: Dice within
    1e-3 of the reference path), same number of kept masks."""
    from ttdg_mgm_amd.evaluation import DiceEvaluator
    cfg, cpu, gpu, batches = trained
    batch = batches[1]
    gpu.eval(), cpu.eval()
    dd = [it["dataset_dict"] for it in batch]
    evg, evc = DiceEvaluator("trained_ds", cfg.TEST.DICE_THRES, dataset_dicts=dd), DiceEvaluator("trained_ds", cfg.TEST.DICE_THRES, dataset_dicts=dd)
    with torch.no_grad():
        evg.process(batch, gpu(batch))
        with _host_backend():
            evc.process(batch, cpu(batch))
    rg, rc = evg.evaluate(), evc.evaluate()
    print("device", rg, len(evg.dice_scores), "host", rc, len(evc.dice_scores))
    assert len(evc.dice_scores) >= 2 * len(batch) // 2 and rc["Dice Coefficient"] > 80.0, "the trained detector must segment the synthetic fundus"
    assert len(evg.dice_scores) == len(evc.dice_scores)
    for k in rg:
        assert abs(rg[k] - rc[k]) <= 1e-3 * abs(rc[k]), (k, rg[k], rc[k])
    gpu.train(), cpu.train()


def test_cfg5_bf16_backbone_vs_fp32_on_trained_checkpoint(trained):
    """cfg-5's precision split on a network whose features mean something: bf16 autocast for the backbone only, every
    matching operator in fp32.  Same detections-free node selection (teacher-forced boxes), node features within 2e-2
    relative (Frobenius) error of the node features against the fp32 backbone reported and bounded, fp32 matching tensors, loss of the
    same size."""
    cfg, cpu, gpu, batches = trained
    batch = batches[0]
    m32 = copy.deepcopy(gpu)
    m16 = copy.deepcopy(gpu)
    m16.autocast_backbone = True
    out = {}
    for name, m in (("f32", m32), ("bf16", m16)):
        m.train()
        m.teacher_forced = True
        m.multi_matching_unsup.eval()
        m.multi_matching_unsup.keep_trace = True
        with torch.no_grad():
            loss, _, _, _ = m(batch, branch="TTT")
        out[name] = (float(loss), m.multi_matching_unsup.last)
    (l32, t32), (l16, t16) = out["f32"], out["bf16"]
    assert t16["X"].dtype == torch.float32 and t16["Wds"].dtype == torch.float32
    assert t16["sizes"] == t32["sizes"]
    rel = float(torch.linalg.norm(t16["X"] - t32["X"]) / torch.linalg.norm(t32["X"]))
    # the free-running losses are not comparable when the two solves return different permutations (a discrete output): the
    # bf16 features are scored against the fp32 run's pseudo-labels instead
    mm = m16.multi_matching_unsup
    with torch.no_grad():
        l16f = float(mm(list(torch.split(t16["X"], t16["sizes"])), [torch.ones(n, dtype=torch.int64, device="cuda:0") for n in t16["sizes"]],
                        m16.multi_matching_sup.U, forced_U=t32["Ub"]))
    print("cfg-5 on the trained checkpoint: node features bf16 vs fp32 backbone, relative Frobenius error %.3e; loss %.5f (free-running %.5f) vs %.5f"
          % (rel, l16f, l16, l32))
    # bf16 keeps 8 significant bits through ~50 convolutions: measured 0.17 - 0.27 on round-2 checkpoints (the fit is not
    # bit-reproducible: vendor convolutions), 0.30 on random init
    assert rel <= 0.4, rel
    assert abs(l16f - l32) <= 0.5 * abs(l32) + 1e-3
