"""Test datasets (synthetic, or COCO-json via data/coco.py) + test loader (reference data/build.py:122-154:
InferenceSampler shard per rank, BatchSampler(TEST.BATCH, drop_last=False), trivial collate)."""
import numpy as np
import torch
import torch.nn.functional as F

from .. import synth

_REGISTRY = {}


def register_synthetic(name, num_images, size=512, cfg_id=2, kind="fundus", num_cls=2):
    _REGISTRY[name] = dict(n=num_images, size=size, cfg_id=cfg_id, kind=kind, num_cls=num_cls)


def register_coco_instances(name, metadata, json_file, image_root, input_format="RGB"):
    """Same call as detectron2's ``register_coco_instances`` (reference data/datasets/builtin.py:9-10); ``metadata`` is
    accepted for signature compatibility.  The json is parsed at registration, images are decoded when a shard is built."""
    from . import coco
    _REGISTRY[name] = dict(kind="coco", records=coco.load_coco_json(json_file, image_root), fmt=input_format,
                           json_file=json_file, image_root=image_root, metadata=dict(metadata or {}))
    _REGISTRY[name]["n"] = len(_REGISTRY[name]["records"])


def dataset_size(name):
    if name not in _REGISTRY:
        raise KeyError("Dataset '{}' is not registered! Available datasets are: {}".format(name, ", ".join(sorted(_REGISTRY))))
    return _REGISTRY[name]["n"]


def dataset_dicts(name, start=0, stop=None):
    """List of dicts: image (3,H,W) uint8, height, width, image_id, annotations [{bbox xyxy, category_id, mask bool}].
    ``start``/``stop`` generate only that index range (a rank's shard): images are synthesised on demand."""
    spec = _REGISTRY[name]
    if spec["kind"] == "coco":
        from . import coco
        return [coco.materialise(r, spec["fmt"]) for r in spec["records"][start:spec["n"] if stop is None else min(stop, spec["n"])]]
    out = []
    for i in range(start, spec["n"] if stop is None else min(stop, spec["n"])):
        seed = 1000 * spec["cfg_id"] + i
        gen = synth.fundus_image if spec["kind"] == "fundus" else synth.polyp_image
        img, boxes, classes, masks = gen(seed, spec["size"]) if spec["kind"] == "fundus" else gen(seed, spec["size"], spec["num_cls"])
        anns = [dict(bbox=boxes[k], category_id=int(classes[k]), mask=masks[k]) for k in range(len(classes))]
        out.append(dict(image=img, height=int(img.shape[1]), width=int(img.shape[2]), image_id=i, annotations=anns, seed=seed))
    return out


def map_for_test(d, min_size=800, max_size=1333):
    """DatasetMapper(is_train=False) equivalent: resize the shorter edge to ``min_size`` (bilinear) and carry the
    teacher-forced detections (GT boxes jittered +-2 px, in resized coordinates)."""
    h, w = d["height"], d["width"]
    s = min(min_size / min(h, w), max_size / max(h, w))
    nh, nw = int(round(h * s)), int(round(w * s))
    img = F.interpolate(d["image"][None].float(), size=(nh, nw), mode="bilinear", align_corners=False)[0]
    boxes = torch.stack([a["bbox"] for a in d["annotations"]]) if d["annotations"] else torch.zeros(0, 4)
    # teacher-forced detections: jittered ground-truth boxes for the seeded synthetic images, the plain ground truth otherwise
    tf = (synth.jitter_boxes(d["seed"] + 500000, boxes) if "seed" in d else boxes) * torch.tensor([nw / w, nh / h, nw / w, nh / h])
    return dict(image=img.round().clamp(0, 255).to(torch.uint8), height=h, width=w, image_id=d["image_id"],
                tf_boxes=tf, tf_classes=torch.tensor([a["category_id"] for a in d["annotations"]], dtype=torch.int64),
                dataset_dict=d)


class TestLoader:
    """Iterable over lists of mapped dicts; rank r sees the contiguous shard detectron2's InferenceSampler gives it."""

    def __init__(self, name, batch, rank=0, world=1, device=None, min_size=800, max_size=1333):
        n = dataset_size(name)
        shard = (n - 1) // world + 1 if n else 0           # detectron2 InferenceSampler [3P]: contiguous, unpadded
        self.dataset_dicts = dataset_dicts(name, shard * rank, min(shard * (rank + 1), n))
        self.items = [map_for_test(d, min_size, max_size) for d in self.dataset_dicts]
        if device is not None:                                   # keep inputs resident in HBM (bench)
            for it in self.items:
                it["image"] = it["image"].to(device)
        self.batch = batch

    def __len__(self):
        return (len(self.items) + self.batch - 1) // self.batch

    def __iter__(self):
        for i in range(0, len(self.items), self.batch):
            yield self.items[i:i + self.batch]


def build_detection_test_loader(cfg, dataset_name, rank=0, world=1, device=None):
    return TestLoader(dataset_name, cfg.TEST.BATCH, rank, world, device,
                      cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST)
