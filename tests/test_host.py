"""CPU tests of the host logic around the kernels: config reader, synthetic data + loader mapping, Dice / E- / S-
measure against goldens from the reference's own numpy functions, trainer bookkeeping (family means), model
construction / state-dict contract."""
import math

import numpy as np
import pytest
import torch

import cases
from ttdg_mgm_amd import data, synth
from ttdg_mgm_amd.config import add_ateacher_config, get_cfg
from ttdg_mgm_amd.evaluation import DiceEvaluator, dice_coefficient, enhanced_align, structure_measure


def test_config_reads_the_reference_yaml_keys(tmp_path):
    cfg = add_ateacher_config(get_cfg())
    cfg.merge_from_file("configs/test_segment.yaml")
    assert cfg.TEST.TTT is True and cfg.TEST.BATCH == 4 and cfg.TEST.DICE_THRES == 0.9
    assert cfg.SOLVER.BASE_LR == 0.005 and cfg.MODEL.ROI_HEADS.NUM_CLASSES == 2 and cfg.SEMISUPNET.Trainer == "baseline"
    cfg.merge_from_list(["TEST.BATCH", "2", "MODEL.DEVICE", "cpu"])
    assert cfg.TEST.BATCH == 2 and cfg.MODEL.DEVICE == "cpu"
    c2 = cfg.clone()
    c2.TEST.BATCH = 8
    assert cfg.TEST.BATCH == 2


@pytest.mark.parametrize("i", range(6))
def test_dice_e_s_measures_vs_reference_golden(golden, i):
    gold = golden("dice")
    p, g = (torch.from_numpy(m) for m in cases.dice_mask_pairs()[i])
    assert abs(dice_coefficient(p, g) - float(gold[f"c{i}_dice"])) <= 1e-9
    assert abs(enhanced_align(p, g) - float(gold[f"c{i}_ea"])) <= 1e-9
    assert abs(structure_measure(p, g) - float(gold[f"c{i}_sm"])) <= 1e-6   # the reference mixes float32/float64 here


def test_synthetic_images_are_deterministic_and_well_formed():
    a = synth.fundus_image(2007)
    b = synth.fundus_image(2007)
    assert torch.equal(a[0], b[0]) and a[0].dtype == torch.uint8 and a[0].shape == (3, 512, 512)
    img, boxes, classes, masks = a
    assert classes.tolist() == [0, 1] and masks.shape == (2, 512, 512)
    assert bool((masks[1] & ~masks[0]).sum() == 0)                       # cup inside disc
    for m, bx in zip(masks, boxes):
        ys, xs = torch.nonzero(m, as_tuple=True)
        assert [xs.min(), ys.min(), xs.max() + 1, ys.max() + 1] == bx.long().tolist()
    pi, pb, pc, pm = synth.polyp_image(5003, 384, 3)
    assert pi.shape == (3, 384, 384) and 1 <= len(pc) <= 3 and int(pc.max()) < 3


def test_loader_resizes_like_the_reference_test_mapper():
    cfg = get_cfg()
    cfg.TEST.BATCH = 3
    data.register_synthetic("host_ds", 4, size=128)
    batches = list(data.build_detection_test_loader(cfg, "host_ds"))
    assert [len(b) for b in batches] == [3, 1]
    it = batches[0][0]
    assert it["image"].shape == (3, 800, 800) and it["image"].dtype == torch.uint8 and (it["height"], it["width"]) == (128, 128)
    gt = torch.stack([a["bbox"] for a in it["dataset_dict"]["annotations"]]) * 800 / 128
    assert float((it["tf_boxes"] - gt).abs().max()) <= 2.0 * 800 / 128 + 1e-4       # +-2 px jitter in original pixels


def test_dice_evaluator_semantics():
    data.register_synthetic("host_ds2", 2, size=64)
    dd = data.dataset_dicts("host_ds2")
    ev = DiceEvaluator("host_ds2", 0.9)
    from ttdg_mgm_amd.modeling.structures import Boxes, Instances
    outs = []
    for d in dd:
        gm = torch.stack([a["mask"] for a in d["annotations"]])
        inst = Instances((64, 64), pred_boxes=Boxes(torch.zeros(3, 4)), scores=torch.tensor([0.95, 0.5, 0.99]),
                         pred_classes=torch.tensor([0, 1, 1]), pred_masks=torch.stack([gm[0], gm[1], ~gm[1]]))
        outs.append({"instances": inst})
    ev.process([{"image_id": d["image_id"]} for d in dd], outs)
    res = ev.evaluate()
    assert len(ev.dice_scores) == 4                                   # score 0.5 filtered out
    assert abs(ev.dice_scores[0] - 100.0) < 1e-3 and ev.dice_scores[1] < 50
    assert set(res) == {"Dice Coefficient", "Enhanced Alignment Metric", "Structural Similarity Metric"}
    ev.reset()
    assert math.isnan(ev.evaluate()["Dice Coefficient"])             # np.mean([]) in the reference


def test_model_state_dict_contract_and_frozen_stages():
    from ttdg_mgm_amd.engine import BaselineTrainer
    cfg = get_cfg()
    cfg.MODEL.DEVICE = "cpu"
    m = BaselineTrainer.build_model(cfg)
    sd = m.state_dict()
    for k in ("backbone.bottom_up.stem.conv1.weight", "backbone.bottom_up.stem.conv1.norm.running_var",
              "backbone.bottom_up.res5.2.conv3.norm.bias", "backbone.fpn_lateral2.weight", "backbone.fpn_output5.bias",
              "proposal_generator.rpn_head.anchor_deltas.weight", "roi_heads.box_head.fc1.weight",
              "roi_heads.box_predictor.cls_score.bias", "roi_heads.mask_head.mask_fcn4.weight", "roi_heads.mask_head.deconv.weight",
              "D_img.classifier.weight", "multi_matching_sup.U", "multi_matching_sup.Net_U.f2g.wq.weight",
              "multi_matching_unsup.node_affinity.fc_M.0.weight", "multi_matching_unsup.intra_domain_graph.layer_norm.bias"):
        assert k in sd, k
    assert sd["multi_matching_sup.U"].shape == (32, 256) and sd["roi_heads.box_predictor.bbox_pred.weight"].shape == (8, 1024)
    frozen = [n for n, p in m.named_parameters() if not p.requires_grad]
    assert frozen and all(n.startswith(("backbone.bottom_up.stem", "backbone.bottom_up.res2")) for n in frozen)
    opt = BaselineTrainer.build_optimizer(cfg, m)
    wds = {g["weight_decay"] for g in opt.param_groups}
    assert wds == {1e-4, 0.0} and all(len(g["params"]) == 1 for g in opt.param_groups)
    with pytest.raises(NotImplementedError):
        m.train()
        m([{"image": torch.zeros(3, 32, 32, dtype=torch.uint8)}], branch="supervised_source")


def test_family_means_like_the_reference_trainer():
    from collections import OrderedDict, defaultdict
    results = OrderedDict(a_1={"Dice Coefficient": 1.0, "Enhanced Alignment Metric": 2.0, "Structural Similarity Metric": 3.0},
                          a_2={"Dice Coefficient": 3.0, "Enhanced Alignment Metric": 4.0, "Structural Similarity Metric": 5.0})
    fam = defaultdict(lambda: defaultdict(list))
    for key, value in results.items():
        for k, v in value.items():
            fam[key.split('_')[0]][k].append(v)
    assert {k: sum(v) / len(v) for k, v in fam["a"].items()}["Dice Coefficient"] == 2.0


def test_dense_and_list_inference_agree_on_the_host_backend():
    """The padded (dense) eval inference and the list-of-Instances formulation are the same map: with the host backend
    (deterministic torch-CPU kernels + the oracle's detection helpers) the detections are bit-identical."""
    import torch
    from oracle import tta_cpu
    from ttdg_mgm_amd.modeling import detector, rcnn
    cfg, model, batches, det, name = tta_cpu._model_and_batches(1, 2, 128, True)
    saved = detector._backend
    detector._backend = tta_cpu._CpuBackend
    model.eval()
    try:
        with torch.no_grad():
            rcnn.DENSE_INFERENCE = True
            a = model(batches[0])
            rcnn.DENSE_INFERENCE = False
            b = model(batches[0])
    finally:
        rcnn.DENSE_INFERENCE = True
        detector._backend = saved
    for x, y in zip(a, b):
        ia, ib = x["instances"], y["instances"]
        assert len(ia) == len(ib) > 0
        assert torch.equal(ia.pred_boxes.tensor, ib.pred_boxes.tensor) and torch.equal(ia.scores, ib.scores)
        assert torch.equal(ia.pred_classes, ib.pred_classes) and torch.equal(ia.pred_masks, ib.pred_masks)


def test_synth_checkpoint_training_helpers():
    """tools/synth_checkpoint.py (measurement infrastructure): the differentiable ROIAlign agrees with the oracle's ROIAlign
    where the adaptive sampling ratio is 2, box deltas invert detectron2's apply_deltas, gradients reach the features."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth_checkpoint as sc
    from oracle import detection as od
    from ttdg_mgm_amd.modeling.detector import apply_deltas
    g = synth.gen(5150)
    f = synth.normal(g, (2, 6, 40, 40)).requires_grad_()
    # 7 x 7 bins of 1 < size <= 2 feature pixels -> the adaptive sampling ratio of ROIAlign is 2
    rois = torch.tensor([[0, 16.0, 20.0, 16 + 8 * 12.5, 20 + 8 * 9.0], [1, 40.0, 8.0, 40 + 8 * 10.0, 8 + 8 * 13.0]])
    got = sc.roi_align_torch([f], rois, (8,), 7, sr=2, canonical_size=1e-3, min_level=2, canonical_level=2)
    ref = od.roi_align(f.detach(), rois, 1.0 / 8, 7)
    assert float((got.detach() - ref).abs().max()) <= 1e-5
    got.square().sum().backward()
    assert float(f.grad.abs().sum()) > 0
    src = torch.tensor([[10.0, 12.0, 50.0, 70.0], [5.0, 5.0, 9.0, 30.0]])
    tgt = torch.tensor([[12.0, 10.0, 55.0, 61.0], [4.0, 6.0, 12.0, 28.0]])
    w = (10.0, 10.0, 5.0, 5.0)
    assert float((apply_deltas(sc.get_deltas(src, tgt, w), src, w) - tgt).abs().max()) <= 1e-4
    assert abs(float(sc.box_iou(src[:1], src[:1])) - 1.0) <= 1e-6


def test_streaming_loader_equals_resident_loader():
    """TestLoader(resident=False): batch-by-batch decode / resize with a bounded prefetch thread gives exactly the items of
    the resident loader, on every pass, and the lazy DiceEvaluator reads the ground truth each item carries."""
    cfg = get_cfg()
    cfg.TEST.BATCH = 3
    cfg.INPUT.MIN_SIZE_TEST = 96
    data.register_synthetic("stream_ds", 7, size=64, id_offset=100)
    res = data.build_detection_test_loader(cfg, "stream_ds", 0, 1, None, resident=True)
    stm = data.build_detection_test_loader(cfg, "stream_ds", 0, 1, None, resident=False)
    assert len(res) == len(stm) == 3 and stm.items is None
    for _ in range(2):                                  # the reference iterates the loader twice per dataset
        a, b = list(res), list(stm)
        assert [len(x) for x in a] == [len(x) for x in b] == [3, 3, 1]
        for ba, bb in zip(a, b):
            for ia, ib in zip(ba, bb):
                assert ia["image_id"] == ib["image_id"] >= 100 and torch.equal(ia["image"], ib["image"]) and torch.equal(ia["tf_boxes"], ib["tf_boxes"])
    from ttdg_mgm_amd.modeling.structures import Boxes, Instances
    ev = DiceEvaluator("stream_ds", 0.9, lazy=True)
    assert ev.dataset_dicts is None
    batch = next(iter(stm))
    outs = []
    for it in batch:
        gm = torch.stack([a["mask"] for a in it["dataset_dict"]["annotations"]])
        outs.append({"instances": Instances((64, 64), pred_boxes=Boxes(torch.zeros(2, 4)), scores=torch.tensor([0.95, 0.99]),
                                            pred_classes=torch.tensor([0, 1]), pred_masks=gm)})
    ev.process(batch, outs)
    assert abs(ev.evaluate()["Dice Coefficient"] - 100.0) < 1e-3 and len(ev.dice_scores) == 6


def test_config_rejects_unsupported_architecture_keys(tmp_path):
    cfg = get_cfg()
    p = tmp_path / "bad.yaml"
    p.write_text("MODEL:\n  RESNETS:\n    DEPTH: 101\n")
    with pytest.raises(ValueError):
        cfg.merge_from_file(str(p))
    good = tmp_path / "good.yaml"
    good.write_text("MODEL:\n  RESNETS:\n    DEPTH: 50\n    OUT_FEATURES: [\"res2\", \"res3\", \"res4\", \"res5\"]\n  ROI_BOX_HEAD:\n    NUM_FC: 2\nTEST:\n  BATCH: 4\n")
    cfg.merge_from_file(str(good))
    assert cfg.TEST.BATCH == 4 and cfg.TEST.EVAL_STU is False          # add_ateacher_config default (reference config.py:11)
    assert get_cfg().TEST.BATCH == 1                                    # reference default (config.py:16)


def test_measures_from_quadrant_counts_equal_the_elementwise_measures(golden):
    """The closed forms the device evaluator uses (twelve quadrant counts per mask pair) against the element-by-element
    measures pinned to the reference's numpy functions: the six golden pairs and random / degenerate masks."""
    from ttdg_mgm_amd.evaluation import (centroid_cuts, dice_tensor, enhanced_align_tensor, measures_from_counts, quadrant_counts,
                                         structure_measure_tensor)
    gold = golden("dice")
    pairs = [(torch.from_numpy(p), torch.from_numpy(g)) for p, g in cases.dice_mask_pairs()]
    g = synth.gen(77)
    for shape, dens in (((40, 56), (0.3, 0.5)), ((33, 17), (0.05, 0.9)), ((8, 8), (0.5, 0.5)), ((64, 64), (0.0, 0.4)), ((21, 30), (1.0, 0.2))):
        pairs.append((torch.from_numpy(g.uniform(size=shape) < dens[0]), torch.from_numpy(g.uniform(size=shape) < dens[1])))
    for i, (p, gt) in enumerate(pairs):
        ys, xs = torch.nonzero(gt, as_tuple=True)
        cen = (ys.double().mean().item(), xs.double().mean().item()) if ys.numel() else (float("nan"), float("nan"))
        cy, cx = centroid_cuts(cen)
        H, W = p.shape
        cnt = torch.tensor([quadrant_counts(p, gt, cy, cx)], dtype=torch.int32)
        d, e, s = measures_from_counts(cnt, H, W, [cy], [cx])[0].tolist()
        want = (float(dice_tensor(p, gt)), float(enhanced_align_tensor(p, gt)), float(structure_measure_tensor(p, gt, centroid=cen)))
        for got, ref in zip((d, e, s), want):
            assert (math.isnan(got) and math.isnan(ref)) or abs(got - ref) <= 1e-10, (i, (d, e, s), want)
        if i < 6:
            assert abs(d - float(gold[f"c{i}_dice"])) <= 1e-9 and abs(e - float(gold[f"c{i}_ea"])) <= 1e-9 and abs(s - float(gold[f"c{i}_sm"])) <= 1e-6


def test_sampler_box_tables_match_the_per_image_formulation():
    """GModule.build_graph.padded_tables (one concatenation + one scatter) against the slice assignment per image it replaced,
    for equal and ragged detection counts, float64 / int64 inputs included."""
    from ttdg_mgm_amd.GModule.build_graph import padded_tables
    g = torch.Generator().manual_seed(3)
    for lens in ((2, 2, 2, 2), (3, 1, 2), (1,), (4, 2)):
        bc = [(torch.rand(n, 4, generator=g, dtype=torch.float64) * 100, torch.randint(0, 3, (n,), generator=g)) for n in lens]
        kmax = max(lens)
        boxes, classes = padded_tables(bc, list(lens), kmax, torch.device("cpu"))
        ref_b = torch.zeros(len(lens), kmax, 4)
        ref_c = torch.zeros(len(lens), kmax, dtype=torch.int32)
        for k, (b, c) in enumerate(bc):
            ref_b[k, :len(b)] = b.to(torch.float32)
            ref_c[k, :len(c)] = c.to(torch.int32)
        assert boxes.dtype == torch.float32 and classes.dtype == torch.int32
        assert torch.equal(boxes, ref_b) and torch.equal(classes, ref_c)


def _blk_projector_model(V, n, tau, iters=20):
    """float32 numpy model of the ARITHMETIC of csrc/gagm.hip:sk_wave_project_blk (no lanes: the butterflies are plain sums):
    potentials f / g / dummy as the state, one exponential per entry per sweep pair (the column sweep reuses the row sweep's
    exponentials through the reciprocal of the row sums), previous potentials as stabilisers, the exact max-subtracted pair
    first and whenever a sum leaves [2^-60, 1e6].  V: (n, 32) block; rows of the oriented problem are the universe slots when
    n > 32."""
    import numpy as np
    f32 = np.float32
    LOG2E = f32(1.4426950408889634)
    tr = n > 32
    L = (V.T if tr else V).astype(f32) * f32(LOG2E / f32(tau))                # (r, c), r <= c
    r, c = L.shape
    mult = c - r
    D = f32(-100.0) * LOG2E
    f, g, fd = np.zeros(r, f32), np.zeros(c, f32), f32(0)
    lo_ok, hi_ok = f32(8.6736174e-19), f32(1.0e6)
    with np.errstate(over="ignore", under="ignore", divide="ignore", invalid="ignore"):
        for it in range(0, iters, 2):
            exact = it == 0
            if not exact:
                e = np.exp2((L - f[:, None]) - g[None, :]).astype(f32)
                s = e.sum(1, dtype=f32)
                ed = np.exp2((D - fd) - g).astype(f32) if mult else None
                sd = ed.sum(dtype=f32) if mult else f32(1)
                cs = (e * (f32(1) / s)[:, None]).sum(0, dtype=f32) + (ed * (f32(mult) / sd) if mult else f32(0))
                allv = np.concatenate([s, cs, [sd]])
                if np.all(allv >= lo_ok) and np.all(allv <= hi_ok):
                    f, g = f + np.log2(s).astype(f32), g + np.log2(cs).astype(f32)
                    fd = fd + np.log2(sd).astype(f32) if mult else fd
                else:
                    exact = True
            if exact:
                t = L - g[None, :]
                m = t.max(1)
                f = (m + np.log2(np.exp2(t - m[:, None]).astype(f32).sum(1, dtype=f32))).astype(f32)
                if mult:
                    dm = (-g).max()
                    fd = f32(D + dm + np.log2(np.exp2(-g - dm).astype(f32).sum(dtype=f32)))
                t = L - f[:, None]
                td = f32(D - fd) if mult else f32(-np.inf)
                m = np.maximum(t.max(0), td)
                acc = np.exp2(t - m[None, :]).astype(f32).sum(0, dtype=f32)
                if mult:
                    acc = acc + f32(mult) * np.exp2(td - m).astype(f32)
                g = (m + np.log2(acc)).astype(f32)
    U = np.exp2((L - f[:, None]) - g[None, :]).astype(f32)
    return U.T if tr else U


@pytest.mark.parametrize("n", [20, 31, 32, 38, 64])
@pytest.mark.parametrize("tau", [0.1, 0.0125, 0.00625])
def test_block_projector_arithmetic_is_the_log_sinkhorn(n, tau):
    """The arithmetic of the solver's block-layout projector - reusing the row sweep's exponentials in the column sweep, the
    dummy rows as one replicated row, stabilising with the previous potentials - restated in numpy float32 and compared with
    the oracle's log-Sinkhorn (float64) on one graph block: the 1e-4 bar of the device tests holds for the algorithm itself."""
    from oracle import gmodule as og
    g = torch.Generator().manual_seed(100 + n)
    V = torch.rand(n, 32, generator=g)
    ref = og._project_sinkhorn(V.double(), [n], 32, tau, 20).numpy()
    got = _blk_projector_model(V.numpy(), n, tau)
    assert float(abs(got - ref).max()) <= 1e-4


def test_reference_yaml_decodes_like_yacs(tmp_path):
    """The reference yamls write tuples as Python literals (`TEST: ("REFUGE_train", ...)`), which yaml.safe_load returns as
    a str: yacs decodes them, so must merge_from_file; command-line overrides go through the same architecture guard."""
    cfg = get_cfg()
    p = tmp_path / "t.yaml"
    p.write_text('DATASETS:\n  TEST: ("A_train", "B_test")\nTEST:\n  BATCH: 4\nSOLVER:\n  BASE_LR: "0.01"\nOUTPUT_DIR: "true"\nMODEL:\n  WEIGHTS: "null"\n')
    cfg.merge_from_file(str(p))
    assert cfg.DATASETS.TEST == ["A_train", "B_test"] and cfg.TEST.BATCH == 4 and cfg.SOLVER.BASE_LR == 0.01
    # a QUOTED yaml string that is not a Python literal stays a string in a file merge (yacs keeps it when literal_eval fails) ...
    assert cfg.OUTPUT_DIR == "true" and cfg.MODEL.WEIGHTS == "null"
    # ... the yaml-scalar reading is for command-line values only
    cfg.merge_from_list(["TEST.TTT", "true"])
    assert cfg.TEST.TTT is True
    with pytest.raises(ValueError):
        cfg.merge_from_list(["MODEL.MASK_ON", "False"])


def test_streaming_loader_is_bounded_and_releases_its_producer():
    """(1) inference_on_dataset consumes a streaming loader lazily: never more than prefetch + 2 batches alive (queue +
    the one being produced + the one being evaluated); (2) a consumer that stops early leaves no blocked producer thread."""
    import gc
    import threading
    import weakref
    from ttdg_mgm_amd.engine import trainer as tr
    cfg = get_cfg()
    cfg.TEST.BATCH = 2
    cfg.INPUT.MIN_SIZE_TEST = 64
    data.register_synthetic("bounded_ds", 24, size=64, id_offset=300)
    stm = data.build_detection_test_loader(cfg, "bounded_ds", 0, 1, None, resident=False)
    alive, peak = [], [0]
    orig = stm._load_batch

    def tracked(lo, hi, stream):
        items, ev = orig(lo, hi, stream)
        alive.append(weakref.ref(items[0]["image"]))
        return items, ev
    stm._load_batch = tracked

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward(self, batch):
            gc.collect()
            peak[0] = max(peak[0], sum(r() is not None for r in alive))
            return [None] * len(batch)

    class Ev:
        def reset(self): pass
        def process(self, b, o): pass
        def evaluate(self): return {}
    tr.inference_on_dataset(Model(), stm, Ev())
    assert len(alive) == 12 and peak[0] <= stm.prefetch + 2, peak
    n0 = threading.active_count()
    it = iter(stm)
    next(it)
    it.close()                                  # MIN_BATCH_NUM / an exception in the consumer
    for _ in range(100):
        if threading.active_count() <= n0:
            break
        import time
        time.sleep(0.05)
    assert threading.active_count() <= n0, "the producer thread is still blocked in q.put"


def test_disk_stream_with_worker_processes_equals_the_source_dataset(tmp_path):
    """data/disk.py: a dataset pre-rendered to disk and read back by 2 persistent worker PROCESSES (the reference's loader
    topology, data/build.py:148-153) yields exactly the items of the in-memory dataset, pass after pass, through the streaming
    TestLoader (host resize here: no GPU), and on a sub-range set per pass."""
    from ttdg_mgm_amd.data import disk
    cfg = get_cfg()
    cfg.TEST.BATCH = 3
    cfg.INPUT.MIN_SIZE_TEST = 96
    data.register_synthetic("disk_src", 7, size=64, id_offset=500)
    data.register_disk("disk_ds", str(tmp_path / "stream"), source="disk_src", workers=2)
    assert data.dataset_size("disk_ds") == 7
    a = data.dataset_dicts("disk_src")
    b = data.dataset_dicts("disk_ds")
    for x, y in zip(a, b):
        assert x["image_id"] == y["image_id"] and torch.equal(x["image"], y["image"]) and x["seed"] == y["seed"]
        for p, q in zip(x["annotations"], y["annotations"]):
            assert torch.equal(p["mask"], q["mask"]) and torch.equal(p["bbox"], q["bbox"]) and p["category_id"] == q["category_id"]
    res = data.build_detection_test_loader(cfg, "disk_src", 0, 1, None, resident=True)
    stm = data.build_detection_test_loader(cfg, "disk_ds", 0, 1, None, resident=False)
    stm.start_workers()
    for _ in range(2):
        ra, rb = list(res), list(stm)
        assert [len(x) for x in ra] == [len(x) for x in rb] == [3, 3, 1]
        for ba, bb in zip(ra, rb):
            for ia, ib in zip(ba, bb):
                assert ia["image_id"] == ib["image_id"] and torch.equal(ia["image"], ib["image"]) and torch.equal(ia["tf_boxes"], ib["tf_boxes"])
    ds = disk.DiskStream(str(tmp_path / "stream"), 7, 2, workers=2)
    assert [[d["image_id"] for d in disk.expand(batch, ds.ring)] for batch in ds.epoch(2, 7)] == [[502, 503], [504, 505], [506]]
    assert [[d["image_id"] for d in disk.expand(batch, ds.ring)] for batch in ds.epoch(0, 2)] == [[500, 501]]


def test_disk_stream_ring_slots(tmp_path):
    """The shared ring (data/disk.py::_Ring; page-locking switched off: no GPU here): worker processes write every batch into
    its slot, the main process sees exactly the dataset's pixels and masks as views of the ring, slot after slot, over more
    batches than there are slots (reuse), and on a ragged last batch."""
    from ttdg_mgm_amd.data import disk
    data.register_synthetic("ring_src", 23, size=32, id_offset=700)
    root = str(tmp_path / "ring")
    disk.prerender("ring_src", root)
    src = data.dataset_dicts("ring_src")
    ds = disk.DiskStream(root, 23, 2, workers=2, prefetch=1, ring=True, register=False)
    assert ds.ring is not None and ds.ring.slots == 6 and not ds.ring.pinned
    for _ in range(2):
        seen = 0
        for batch in ds.epoch(0, 23):
            assert batch["slot"] is not None and batch["images"] is None            # nothing but metadata crossed the process boundary
            for d in disk.expand(batch, ds.ring):
                ref = src[d["image_id"] - 700]
                assert torch.equal(d["image"], ref["image"]) and d["seed"] == ref["seed"]
                for a, b in zip(d["annotations"], ref["annotations"]):
                    assert torch.equal(a["mask"], b["mask"]) and torch.equal(a["bbox"], b["bbox"]) and a["category_id"] == b["category_id"]
                seen += 1
        assert seen == 23


def test_shipped_miopen_db_is_staged_to_a_private_writable_copy(monkeypatch, tmp_path):
    """The package ships MIOpen find-db / perf-db records for the bench shapes (ttdg-mgm_amd/miopen_db) and points
    MIOPEN_USER_DB_PATH at a COPY under the user's cache directory (MIOpen writes to its user db path; ranks may race: atomic
    renames, directory named after the content; 0700 and owned by this user - ADVICE r3: not a predictable world-writable /tmp
    path).  An explicit MIOPEN_USER_DB_PATH or TTDG_MIOPEN_DB=0 leaves the environment alone."""
    import os
    import stat
    import ttdg_mgm_amd as pkg
    files = sorted(f for f in os.listdir(pkg.MIOPEN_DB) if f.endswith(".txt"))
    assert len(files) == 2 and any(f.endswith(".ufdb.txt") for f in files) and any(f.endswith(".udb.txt") for f in files)
    assert all(f.startswith("gfx950") for f in files)                     # keyed by device: ignored on anything else
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path))
    dst = pkg._stage_miopen_db()
    assert dst.startswith(str(tmp_path / "ttdg_mgm_amd")) and "miopen_db_" in os.path.basename(dst) and sorted(os.listdir(dst)) == files
    for d in (dst, os.path.dirname(dst)):
        st = os.stat(d)
        assert st.st_uid == os.getuid() and stat.S_IMODE(st.st_mode) == 0o700
    for f in files:
        assert open(os.path.join(dst, f), "rb").read() == open(os.path.join(pkg.MIOPEN_DB, f), "rb").read()
    with open(os.path.join(dst, files[0]), "a") as fh:                     # MIOpen appends records: a second staging keeps them
        fh.write("x=y\n")
    assert pkg._stage_miopen_db() == dst and open(os.path.join(dst, files[0])).read().endswith("x=y\n")
    assert not [f for f in os.listdir(dst) if f.endswith(".tmp")]
    os.chmod(os.path.dirname(dst), 0o777)                                  # someone loosened the parent: tightened again
    assert pkg._stage_miopen_db() == dst and stat.S_IMODE(os.stat(os.path.dirname(dst)).st_mode) == 0o700
    # the switches
    for env, want in (({"TTDG_MIOPEN_DB": "0"}, None), ({"MIOPEN_USER_DB_PATH": "/somewhere/else"}, "/somewhere/else"), ({}, dst)):
        env = dict(env)
        pkg._configure_miopen_db(env)
        assert env.get("MIOPEN_USER_DB_PATH") == want


def test_layout_helpers():
    """ops.is_channels_last / like_layout: what the layout dispatch of the epilogue, gather and fold wrappers rests on."""
    from ttdg_mgm_amd import ops
    CL = torch.channels_last
    a = torch.zeros(2, 8, 5, 6)
    assert not ops.is_channels_last(a) and ops.is_channels_last(a.contiguous(memory_format=CL))
    assert not ops.is_channels_last(torch.zeros(2, 8, 1, 1).contiguous(memory_format=CL))        # both layouts at once: NCHW code paths apply
    assert not ops.is_channels_last(torch.zeros(2, 1, 5, 6).contiguous(memory_format=CL))
    assert not ops.is_channels_last(torch.zeros(4, 6)) and not ops.is_channels_last(a[:, ::2])
    b = torch.arange(2 * 8 * 5 * 6, dtype=torch.float32).view(2, 8, 5, 6)
    for ref in (a, a.contiguous(memory_format=CL)):
        for src in (b, b.contiguous(memory_format=CL), b.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)):
            out = ops.like_layout(src, ref)
            assert torch.equal(out, b) and out.stride() == ref.stride()
    assert ops.like_layout(b, a) is b and ops.like_layout(b.contiguous(memory_format=CL), a.contiguous(memory_format=CL)).is_contiguous(memory_format=CL)


def test_mode_s_flattens_gradients_in_the_parameters_storage_order():
    """engine/sync_universe.py: a channels-last filter gradient enters the all-reduce buffer as a VIEW in storage order (round 3
    went through reshape(-1): a transposing copy in, an NCHW-strided gradient out, a second copy inside the SGD step) and comes
    back with the parameter's own strides; plain tensors are untouched; the order depends on the parameter only."""
    import torch
    from ttdg_mgm_amd.engine import sync_universe as su
    p = torch.randn(6, 4, 3, 3).contiguous(memory_format=torch.channels_last)
    g = torch.randn(6, 4, 3, 3).contiguous(memory_format=torch.channels_last)
    f = su.storage_flat(g, p)
    assert f.data_ptr() == g.data_ptr() and f.is_contiguous() and f.numel() == g.numel()          # a view, no copy
    back = su.storage_unflat(f.clone(), p)
    assert back.stride() == p.stride() and torch.equal(back, g)
    # an NCHW-contiguous gradient of a channels-last parameter lands in the SAME order (every rank composes the same buffer)
    assert torch.equal(su.storage_flat(g.contiguous(), p), f)
    q = torch.randn(5, 7)
    assert su.storage_flat(q, q).data_ptr() == q.data_ptr() and torch.equal(su.storage_unflat(q.reshape(-1), q), q)
    one = torch.randn(8, 1, 1, 1)              # both layouts at once: stays the plain path
    assert su.storage_flat(one, one).data_ptr() == one.data_ptr()


def test_evaluator_centroids_batched_equals_the_per_mask_form():
    """ground-truth centroids of a streamed item are computed in ONE batched reduction (one host read per image, not one
    per annotation); same exact integer sums as the per-mask form (reference dice_metric.py:196-200), NaN for an empty map."""
    import math
    import torch
    from ttdg_mgm_amd.evaluation import DiceEvaluator
    g = torch.Generator().manual_seed(3)
    masks = [torch.rand(37, 53, generator=g) > t for t in (0.5, 0.9, 0.999, 2.0)]
    cens = DiceEvaluator._centroids(masks)
    for m, (cy, cx) in zip(masks, cens):
        n = int(m.sum())
        if n == 0:
            assert math.isnan(cy) and math.isnan(cx)
            continue
        ys, xs = torch.nonzero(m, as_tuple=True)
        assert cy == float(ys.sum()) / n and cx == float(xs.sum()) / n
    assert DiceEvaluator._centroids([]) == [] and DiceEvaluator._centroid(masks[0]) == cens[0]
