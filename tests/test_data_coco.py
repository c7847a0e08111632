"""CPU tests of the loader / config / checkpoint shim: COCO-json datasets without detectron2 or
pycocotools, checkpoint formats, and the train_net.py command line."""
import json
import os
import pickle
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rle_string(counts):
    """Independent statement of COCO's rleToString (the inverse of data/coco.py::rle_counts_from_string)."""
    s = ""
    for i, x in enumerate(counts):
        if i > 2:
            x -= counts[i - 2]
        more = True
        while more:
            c = x & 0x1f
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            s += chr(c + 48)
    return s


def mask_counts(mask):
    flat = np.asarray(mask, dtype=bool).T.reshape(-1)
    counts, val, run = [], False, 0
    for v in flat:
        if v == val:
            run += 1
        else:
            counts.append(run)
            val, run = v, 1
    counts.append(run)
    return counts


def write_dataset(tmp, n=3, size=(40, 56)):
    from PIL import Image
    h, w = size
    g = np.random.default_rng(5)
    images, anns, masks = [], [], {}
    aid = 1
    for i in range(n):
        arr = g.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        Image.fromarray(arr).save(os.path.join(tmp, "im%d.png" % i))
        images.append(dict(id=10 + i, file_name="im%d.png" % i, height=h, width=w))
        # annotation 1: axis-aligned rectangle as a polygon (category 7)
        x0, y0, x1, y1 = 5 + i, 6, 25 + i, 30
        anns.append(dict(id=aid, image_id=10 + i, category_id=7, bbox=[x0, y0, x1 - x0, y1 - y0], iscrowd=0,
                         segmentation=[[x0, y0, x1, y0, x1, y1, x0, y1]]))
        aid += 1
        # annotation 2: random blob as RLE (category 3), compressed on even images, uncompressed on odd ones
        m = np.zeros((h, w), bool)
        m[10:20 + i, 30:50] = g.random((10 + i, 20)) > 0.3
        masks[10 + i] = m
        counts = mask_counts(m)
        seg = dict(size=[h, w], counts=rle_string(counts) if i % 2 == 0 else counts)
        ys, xs = np.nonzero(m)
        anns.append(dict(id=aid, image_id=10 + i, category_id=3, bbox=[int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)],
                         iscrowd=0, segmentation=seg))
        aid += 1
    jf = os.path.join(tmp, "ann.json")
    with open(jf, "w") as f:
        json.dump(dict(images=images, annotations=anns, categories=[dict(id=7, name="cup"), dict(id=3, name="disc")]), f)
    return jf, masks


def test_rle_string_round_trip():
    from ttdg_mgm_amd.data import coco
    g = np.random.default_rng(1)
    for _ in range(50):
        h, w = int(g.integers(1, 40)), int(g.integers(1, 40))
        m = g.random((h, w)) > g.random()
        counts = mask_counts(m)
        assert coco.rle_counts_from_string(rle_string(counts)) == counts
        assert np.array_equal(coco.rle_to_mask(counts, h, w), m)
    big = [0, 5000, 3, 70000, 1, 2]                      # multi-group values, negative and positive deltas
    assert coco.rle_counts_from_string(rle_string(big)) == big
    with pytest.raises(ValueError):
        coco.rle_to_mask([3, 2], 2, 2)


def test_coco_json_dataset(tmp_path):
    from ttdg_mgm_amd import data
    from ttdg_mgm_amd.config import get_cfg
    jf, masks = write_dataset(str(tmp_path))
    data.register_coco_instances("coco_tmp", {}, jf, str(tmp_path))
    assert data.dataset_size("coco_tmp") == 3
    dd = data.dataset_dicts("coco_tmp")
    assert [d["image_id"] for d in dd] == [10, 11, 12]
    d0 = dd[0]
    assert d0["image"].dtype == torch.uint8 and tuple(d0["image"].shape) == (3, 40, 56)
    a_poly, a_rle = d0["annotations"]
    assert a_poly["category_id"] == 1 and a_rle["category_id"] == 0          # sorted category ids 3, 7 -> 0, 1
    assert a_poly["bbox"].tolist() == [5.0, 6.0, 25.0, 30.0]                  # XYWH -> XYXY
    pm = a_poly["mask"].numpy()
    inner = np.zeros_like(pm)
    inner[7:30, 6:25] = True
    assert pm[inner].all() and pm.sum() <= 26 * 22 and not pm[:5].any() and not pm[:, 28:].any()
    for d in dd:                                                              # RLE (both encodings) is exact
        assert np.array_equal(d["annotations"][1]["mask"].numpy(), masks[d["image_id"]])
    # BGR order flips the channels
    data.register_coco_instances("coco_tmp_bgr", {}, jf, str(tmp_path), "BGR")
    assert torch.equal(data.dataset_dicts("coco_tmp_bgr")[0]["image"], d0["image"].flip(0))
    # InferenceSampler shards + the test mapper
    cfg = get_cfg()
    cfg.TEST.BATCH = 2
    cfg.INPUT.MIN_SIZE_TEST = 80
    l0 = data.build_detection_test_loader(cfg, "coco_tmp", 0, 2)
    l1 = data.build_detection_test_loader(cfg, "coco_tmp", 1, 2)
    assert [x["image_id"] for b in l0 for x in b] == [10, 11] and [x["image_id"] for b in l1 for x in b] == [12]
    item = next(iter(l0))[0]
    assert tuple(item["image"].shape) == (3, 80, 112) and item["height"] == 40 and item["width"] == 56
    assert torch.allclose(item["tf_boxes"][0], torch.tensor([10.0, 12.0, 50.0, 60.0]))     # ground-truth boxes, resized frame
    with pytest.raises(KeyError):
        data.dataset_size("never_registered")


def test_checkpoint_formats(tmp_path):
    from ttdg_mgm_amd.engine.checkpoint import load_weights
    torch.manual_seed(0)
    src = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    sd = {k: v.clone() for k, v in src.state_dict().items()}

    def fresh():
        return torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))

    def same(m):
        return all(torch.equal(v, sd[k]) for k, v in m.state_dict().items())

    p1, p2, p3, p4 = (str(tmp_path / n) for n in ("plain.pth", "wrapped.pth", "d2.pkl", "ens.pth"))
    torch.save(sd, p1)
    torch.save({"model": {"module." + k: v for k, v in sd.items()}, "iteration": 7}, p2)
    with open(p3, "wb") as f:
        pickle.dump({"model": {k: v.numpy() for k, v in sd.items()}, "__author__": "x"}, f)
    ens = {"modelStudent." + k: v for k, v in sd.items()}
    ens.update({"modelTeacher." + k: v + 1 for k, v in sd.items()})
    torch.save({"model": ens}, p4)
    for p in (p1, p2):
        m = fresh()
        assert load_weights(m, p) == ([], []) and same(m)
    with pytest.raises(ValueError):                 # a pickle executes code from the file: refused unless trusted
        load_weights(fresh(), p3)
    m = fresh()
    assert load_weights(m, p3, trusted=True) == ([], []) and same(m)
    m = fresh()                                      # reference default TEST.EVAL_STU = False: the teacher half (config.py:11)
    load_weights(m, p4)
    assert torch.equal(m[0].weight, sd["0.weight"] + 1)
    m = fresh()
    assert load_weights(m, p4, prefer_student=True) == ([], []) and same(m)
    torch.save({"model": {"wrong.prefix." + k: v for k, v in sd.items()}}, p2)
    with pytest.raises(ValueError):                 # nothing matches: never a silent random-init run
        load_weights(fresh(), p2)
    m = fresh()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    assert load_weights(m, "") == ([], []) and all(torch.equal(v, before[k]) for k, v in m.state_dict().items())
    part = {k: v for k, v in sd.items() if k.startswith("0.")}
    part["extra.weight"] = torch.zeros(1)
    torch.save(part, p1)
    missing, unexpected = load_weights(fresh(), p1)
    assert missing == ["1.weight", "1.bias"] and unexpected == ["extra.weight"]
    torch.save({"0.weight": torch.zeros(5, 5)}, p1)
    with pytest.raises(ValueError):
        load_weights(fresh(), p1)


def test_train_net_command_line():
    sys.path.insert(0, ROOT)
    import train_net
    args = train_net.argument_parser().parse_args(["--eval-only", "--config-file", "configs/test_segment.yaml", "--num-gpus", "2",
                                                   "--register-synthetic", "s1", "6", "TEST.BATCH", "2", "OUTPUT_DIR", "out"])
    assert args.eval_only and args.num_gpus == 2 and args.register_synthetic == [["s1", "6"]]
    assert args.opts == ["TEST.BATCH", "2", "OUTPUT_DIR", "out"]
    with pytest.raises(NotImplementedError):
        train_net.main(["--config-file", "configs/test_segment.yaml"])
    with pytest.raises(NotImplementedError):
        train_net.main(["--eval-only", "--num-machines", "2"])


def test_train_net_setup_merges_config(tmp_path):
    sys.path.insert(0, ROOT)
    import train_net
    from ttdg_mgm_amd import data
    jf, _ = write_dataset(str(tmp_path))
    out = str(tmp_path / "out")
    args = train_net.argument_parser().parse_args(["--eval-only", "--config-file", os.path.join(ROOT, "configs", "test_segment.yaml"),
                                                   "--register-coco", "cli_ds", jf, str(tmp_path), "DATASETS.TEST", "('cli_ds',)",
                                                   "TEST.BATCH", "2", "SOLVER.BASE_LR", "0.001", "OUTPUT_DIR", out])
    cfg = train_net.setup(args)
    assert list(cfg.DATASETS.TEST) == ["cli_ds"] and cfg.TEST.BATCH == 2 and cfg.SOLVER.BASE_LR == 0.001 and cfg.TEST.TTT is True
    assert os.path.isdir(out) and data.dataset_size("cli_ds") == 3
