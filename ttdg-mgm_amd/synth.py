"""Repo-owned deterministic synthetic inputs (numpy PCG64), shared by the golden
generator, the parity tests, smoke() and bench.py.  SURVEY.md §8d.

Nothing here depends on the reference or on the oracle.
"""
import numpy as np
import torch

DIM = 256
HID = 512
UNIV = 32

# state-dict layout of MGM3_unsup (reference multi_graph_matching.py:462-466,
# utils/affinity.py:21-29, utils/attentions.py:50-58); order is fixed so that one
# seed always yields the same tensors.
MGM3_PARAM_SHAPES = (
    ("node_affinity.fc_M.0.weight", (HID, HID)),
    ("node_affinity.fc_M.0.bias", (HID,)),
    ("node_affinity.fc_M.2.weight", (1, HID)),
    ("node_affinity.fc_M.2.bias", (1,)),
    ("node_affinity.project_sr.weight", (DIM, DIM)),
    ("node_affinity.project_tg.weight", (DIM, DIM)),
    ("intra_domain_graph.linear_k.weight", (DIM, DIM)),
    ("intra_domain_graph.linear_k.bias", (DIM,)),
    ("intra_domain_graph.linear_v.weight", (DIM, DIM)),
    ("intra_domain_graph.linear_v.bias", (DIM,)),
    ("intra_domain_graph.linear_q.weight", (DIM, DIM)),
    ("intra_domain_graph.linear_q.bias", (DIM,)),
    ("intra_domain_graph.linear_final.weight", (DIM, DIM)),
    ("intra_domain_graph.linear_final.bias", (DIM,)),
    ("intra_domain_graph.layer_norm.weight", (DIM,)),
    ("intra_domain_graph.layer_norm.bias", (DIM,)),
)


def gen(seed):
    return np.random.Generator(np.random.PCG64(int(seed)))


def normal(g, shape, scale=1.0, shift=0.0):
    return torch.from_numpy((g.standard_normal(shape) * scale + shift).astype(np.float32))


def mgm3_params(seed, std=0.05):
    """Weights 're-drawn at std 0.05 so outputs are not ~0' (SURVEY.md §8c)."""
    g = gen(seed)
    out = {}
    for name, shape in MGM3_PARAM_SHAPES:
        if name.endswith("layer_norm.weight"):
            out[name] = torch.ones(shape)
            g.standard_normal(shape)  # keep the stream position independent of the branch
        else:
            out[name] = normal(g, shape, std)
    return out


def _affinity_shapes(prefix):
    return ((prefix + "fc_M.0.weight", (HID, HID)), (prefix + "fc_M.0.bias", (HID,)), (prefix + "fc_M.2.weight", (1, HID)),
            (prefix + "fc_M.2.bias", (1,)), (prefix + "project_sr.weight", (DIM, DIM)), (prefix + "project_tg.weight", (DIM, DIM)))


# state dict of U_sup(num_cls, 32) (multi_graph_matching.py:77-88, 119-134; utils/graph_network.py:95-99)
USUP_PARAM_SHAPES = (
    (("Net_U.f2g.wq.weight", (DIM, DIM)), ("Net_U.f2g.wq.bias", (DIM,)), ("Net_U.f2g.wk.weight", (DIM, DIM)), ("Net_U.f2g.wk.bias", (DIM,)))
    + tuple(("Net_U.g_gene.linear_%s.%s" % (n, w), (DIM, DIM) if w == "weight" else (DIM,))
            for n in ("k", "v", "q", "final") for w in ("weight", "bias"))
    + (("Net_U.g_gene.layer_norm.weight", (DIM,)), ("Net_U.g_gene.layer_norm.bias", (DIM,)),
       ("Net_U.adapt.weight", (DIM, DIM)), ("Net_U.adapt.bias", (DIM,)))
    + _affinity_shapes("Net_U.affinity_layer.") + _affinity_shapes("node_affinity."))


def usup_params(seed, std=0.05):
    """U_sup state dict: ``U`` as the reference initialises it (:124), every other weight at std 0.05."""
    g = gen(seed)
    out = {"U": normal(g, (UNIV, DIM), 1.0, 1.0 / UNIV)}
    for name, shape in USUP_PARAM_SHAPES:
        if name.endswith("layer_norm.weight"):
            out[name] = torch.ones(shape)
            g.standard_normal(shape)
        else:
            out[name] = normal(g, shape, std)
    return out


def universe(seed):
    """U_sup.U init: randn + 1/univ_size (multi_graph_matching.py:124)."""
    return normal(gen(seed), (UNIV, DIM), 1.0, 1.0 / UNIV)


def node_sets(seed, sizes, scale=0.1, num_cls=2):
    """Operator-level graphs: ``nodes[g] = randn(n_g, 256)*scale``, labels in {1..num_cls}."""
    g = gen(seed)
    nodes = [normal(g, (n, DIM), scale) for n in sizes]
    labels = [torch.from_numpy(g.integers(1, num_cls + 1, size=n).astype(np.int64)) for n in sizes]
    return nodes, labels


# ----------------------------------------------------------------------------- images
def fundus_image(seed, size=512):
    """Fundus-shaped uint8 RGB image with an optic 'disc' ellipse (class 0) and a
    brighter concentric 'cup' (class 1).  Returns image (3,H,W) uint8, boxes (2,4)
    xyxy float32, classes (2,) int64, masks (2,H,W) bool."""
    g = gen(seed)
    H = W = int(size)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    cy, cx = H * 0.5, W * 0.5
    rr = np.sqrt(((yy - cy) / (0.5 * H)) ** 2 + ((xx - cx) / (0.5 * W)) ** 2)
    img = np.zeros((3, H, W), np.float32)
    img[0] = 150 - 70 * rr
    img[1] = 60 - 35 * rr
    img[2] = 30 - 15 * rr
    img *= (rr < 0.98)
    dcy = g.normal(0.5 * H, 0.05 * H)
    dcx = g.normal(0.5 * W, 0.05 * W)
    da, db = g.uniform(0.13, 0.19, size=2) * H
    frac = g.uniform(0.45, 0.6)
    disc = ((yy - dcy) / da) ** 2 + ((xx - dcx) / db) ** 2 <= 1.0
    cup = ((yy - dcy) / (da * frac)) ** 2 + ((xx - dcx) / (db * frac)) ** 2 <= 1.0
    img[:, disc] += np.array([70, 90, 50], np.float32)[:, None]
    img[:, cup] += np.array([30, 60, 70], np.float32)[:, None]
    img += g.normal(0, 8, size=img.shape).astype(np.float32)
    img = np.clip(img, 0, 255).astype(np.uint8)
    masks = np.stack([disc, cup])
    boxes = []
    for m in masks:
        ys, xs = np.nonzero(m)
        boxes.append([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1])
    return (torch.from_numpy(img), torch.tensor(boxes, dtype=torch.float32),
            torch.tensor([0, 1], dtype=torch.int64), torch.from_numpy(masks))


def polyp_image(seed, size=384, num_cls=3):
    """Polyp-shaped pinkish texture with 1-3 blobs drawn from ``num_cls`` classes."""
    g = gen(seed)
    H = W = int(size)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.empty((3, H, W), np.float32)
    img[0], img[1], img[2] = 190, 120, 130
    img += 12 * np.sin(xx / 17.0)[None] + 9 * np.cos(yy / 23.0)[None]
    k = int(g.integers(1, 4))
    masks, boxes, classes = [], [], []
    for _ in range(k):
        cy, cx = g.uniform(0.25, 0.75, size=2) * H
        a, b = g.uniform(0.08, 0.2, size=2) * H
        m = ((yy - cy) / a) ** 2 + ((xx - cx) / b) ** 2 <= 1.0
        c = int(g.integers(0, num_cls))
        img[:, m] += np.array([25, -20 + 15 * c, -25 + 10 * c], np.float32)[:, None]
        ys, xs = np.nonzero(m)
        masks.append(m)
        boxes.append([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1])
        classes.append(c)
    img += g.normal(0, 6, size=img.shape).astype(np.float32)
    img = np.clip(img, 0, 255).astype(np.uint8)
    return (torch.from_numpy(img), torch.tensor(boxes, dtype=torch.float32),
            torch.tensor(classes, dtype=torch.int64), torch.from_numpy(np.stack(masks)))


def jitter_boxes(seed, boxes, px=2.0):
    """'Teacher-forced' detections: GT boxes jittered by +-px."""
    g = gen(seed)
    return boxes + torch.from_numpy(g.uniform(-px, px, size=tuple(boxes.shape)).astype(np.float32))


def fpn_pyramid(seed, batch, size, scale=0.1):
    """Synthetic 5-level NCHW pyramid for an image of ``size`` (strides 4..64, p6 = ceil)."""
    g = gen(seed)
    feats, s = [], size
    hw = [-(-size // 4)]
    for _ in range(4):
        hw.append(-(-hw[-1] // 2))
    for h in hw:
        feats.append(normal(g, (batch, DIM, h, h), scale))
    return feats
