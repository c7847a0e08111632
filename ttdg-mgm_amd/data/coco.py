"""COCO-json test datasets without detectron2 / pycocotools.

Stands in for ``register_coco_instances`` + ``load_coco_json`` [3P detectron2] as the reference's builtin datasets use them
(adapteacher/data/datasets/builtin.py:9-10) and for the ``pycocotools.mask`` calls of ``DiceEvaluator.convert_to_binary_mask``
(adapteacher/evaluation/dice_metric.py:94-107): polygons, uncompressed RLE and compressed RLE -> boolean masks.
RLE decoding is exact (column-major runs; the compressed string is COCO's 5-bit-group / delta coding); polygons are
rasterised by PIL, whose boundary rule differs from pycocotools' upsampled edge walk by at most the boundary pixels
(pycocotools is absent here: unpinned)."""
import json
import os

import numpy as np
import torch


def rle_counts_from_string(s):
    """COCO compressed RLE string -> run lengths (pycocotools rleFrString): 6-bit characters offset by 48, five payload bits
    each, bit 5 = continuation, sign extension from the last group, runs after the second stored as deltas to counts[i-2]."""
    if isinstance(s, bytes):
        s = s.decode("ascii")
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_to_mask(counts, height, width):
    """Run lengths (alternating 0-runs and 1-runs, column-major) -> (height, width) bool."""
    flat = np.zeros(height * width, dtype=bool)
    pos, val = 0, False
    for c in counts:
        if val:
            flat[pos:pos + c] = True
        pos += c
        val = not val
    if pos != height * width:
        raise ValueError("RLE covers {} pixels, the mask has {}".format(pos, height * width))
    return flat.reshape(width, height).T


def polygons_to_mask(polygons, height, width):
    from PIL import Image, ImageDraw
    img = Image.new("1", (width, height), 0)
    draw = ImageDraw.Draw(img)
    for poly in polygons:
        if len(poly) >= 6:
            draw.polygon([(float(poly[i]), float(poly[i + 1])) for i in range(0, len(poly) - 1, 2)], outline=1, fill=1)
    return np.array(img, dtype=bool)


def segmentation_to_mask(segmentation, height, width):
    """dice_metric.py:94-107: list = polygons (merged), dict with list counts = uncompressed RLE, else compressed RLE."""
    if isinstance(segmentation, list):
        return polygons_to_mask(segmentation, height, width)
    h, w = segmentation.get("size", (height, width))
    counts = segmentation["counts"]
    if not isinstance(counts, list):
        counts = rle_counts_from_string(counts)
    return rle_to_mask(counts, h, w)


def load_coco_json(json_file, image_root):
    """-> list of records (file_name, height, width, image_id, annotations with xyxy ``bbox`` tensors, contiguous 0-based
    ``category_id`` in sorted-category-id order as detectron2 maps them, and the raw ``segmentation``).  Crowd annotations
    keep their flag; images without annotations are kept (the test loader does not filter)."""
    with open(json_file) as f:
        coco = json.load(f)
    cat_ids = sorted(c["id"] for c in coco.get("categories", []))
    cat_map = {cid: i for i, cid in enumerate(cat_ids)}
    by_image = {}
    for a in coco.get("annotations", []):
        by_image.setdefault(a["image_id"], []).append(a)
    records = []
    for img in sorted(coco["images"], key=lambda r: r["id"]):
        anns = []
        for a in by_image.get(img["id"], []):
            if a.get("ignore", 0):
                continue
            x, y, w, h = a["bbox"]
            anns.append(dict(bbox=torch.tensor([x, y, x + w, y + h], dtype=torch.float32), category_id=cat_map[a["category_id"]],
                             iscrowd=int(a.get("iscrowd", 0)), segmentation=a.get("segmentation")))
        records.append(dict(file_name=os.path.join(image_root, img["file_name"]), height=int(img["height"]), width=int(img["width"]),
                            image_id=img["id"], annotations=anns))
    return records


def read_image(path, fmt="RGB"):
    """-> (3, H, W) uint8 in ``fmt`` channel order (detectron2 ``read_image``: "BGR" flips the decoded RGB)."""
    from PIL import Image
    with Image.open(path) as im:
        arr = np.array(im.convert("RGB"))
    if fmt == "BGR":
        arr = arr[:, :, ::-1]
    elif fmt != "RGB":
        raise ValueError("unsupported INPUT.FORMAT {!r}".format(fmt))
    return torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).contiguous()


def materialise(record, fmt="RGB"):
    """Record -> the dict the loader / evaluator consume: decoded image and boolean ground-truth masks."""
    img = read_image(record["file_name"], fmt)
    if (int(img.shape[1]), int(img.shape[2])) != (record["height"], record["width"]):
        raise ValueError("{}: image is {}x{}, the json says {}x{}".format(record["file_name"], img.shape[1], img.shape[2],
                                                                         record["height"], record["width"]))
    anns = []
    for a in record["annotations"]:
        seg = a["segmentation"]
        mask = segmentation_to_mask(seg, record["height"], record["width"]) if seg else np.zeros((record["height"], record["width"]), bool)
        anns.append(dict(bbox=a["bbox"], category_id=a["category_id"], iscrowd=a["iscrowd"], mask=torch.from_numpy(np.ascontiguousarray(mask))))
    return dict(image=img, height=record["height"], width=record["width"], image_id=record["image_id"], annotations=anns,
                file_name=record["file_name"])
