"""PrototypeComputation — mirror of reference build_graph.py:11-250 (graph-node sampler).

Same constructor, same ``__call__(features, targets) -> (nodes, labels) | (None, None)``; targets are
detectron2-style Instances (``len(t)``, ``t.pred_boxes.tensor`` / ``t.gt_boxes.tensor``,
``t.pred_classes`` / ``t.gt_classes``).  Label assignment, ordered strided selection and the feature
gather (+ its scatter backward) are HIP kernels (csrc/sampler.hip); one small D2H copy returns the node
counts the host needs to shape the graphs."""
import torch

from .. import ops

INF = 100000000


def padded_tables(boxes_classes, lens, kmax, dev):
    """[(boxes (n_k, 4), classes (n_k,))] -> zero-padded (B, kmax, 4) float32 / (B, kmax) int32 tables on ``dev`` in a handful of
    launches: one concatenation each and, for ragged counts, one scatter (a slice assignment per image is a 25 us
    device-to-device memcpy each: 0.4 ms per adaptation step)."""
    B = len(boxes_classes)
    allb = torch.cat([b.detach().reshape(-1, 4) for b, _ in boxes_classes]).to(dev, torch.float32)
    allc = torch.cat([c.detach().reshape(-1) for _, c in boxes_classes]).to(dev, torch.int32)
    if all(n == kmax for n in lens):
        return allb.view(B, kmax, 4).contiguous(), allc.view(B, kmax).contiguous()
    pos = torch.tensor([k * kmax + j for k, n in enumerate(lens) for j in range(n)], dtype=torch.int64).to(dev, non_blocking=True)
    boxes = torch.zeros(B * kmax, 4, device=dev, dtype=torch.float32).index_copy_(0, pos, allb).view(B, kmax, 4)
    classes = torch.zeros(B * kmax, device=dev, dtype=torch.int32).index_copy_(0, pos, allc).view(B, kmax)
    return boxes, classes


class PrototypeComputation(object):
    def __init__(self, num_cls, sample_dist):
        self.num_class = num_cls
        self.num_nodes_per_class = sample_dist
        self.bg_ratio = 8
        self.strides = [4, 8, 16, 32, 64]
        self.object_sizes_of_interest = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, INF]]

    @staticmethod
    def _boxes_classes(t):
        fields = getattr(t, "_fields", {})
        if 'gt_boxes' in fields:
            return t.gt_boxes.tensor, t.gt_classes
        return t.pred_boxes.tensor, t.pred_classes

    def __call__(self, features, targets):
        if not any(len(t) for t in targets):
            return None, None
        dev = features[0].device
        nl = len(features)
        shapes = [(int(f.shape[-2]), int(f.shape[-1])) for f in features]
        lv = ops.levels_desc(shapes, self.strides[:nl], self.object_sizes_of_interest[:nl])
        npts = sum(h * w for h, w in shapes)
        # the reference skips box-less images when it builds labels (:79) but pairs label list entry k with
        # feature image k (:173-181): reproduce that pairing
        with_boxes = [t for t in targets if len(t)]
        B = len(with_boxes)
        kmax = max(len(t) for t in with_boxes)
        lens = [len(t) for t in with_boxes]
        boxes, classes = padded_tables([self._boxes_classes(t) for t in with_boxes], lens, kmax, dev)
        nbox = torch.tensor(lens, dtype=torch.int32).to(dev, non_blocking=True)
        labels = ops.node_labels(boxes, classes, nbox, lv, npts)
        cap = nl * (2 * self.num_nodes_per_class - 1)
        sel_idx, sel_lab, count = ops.node_select(labels, lv, self.num_nodes_per_class, cap)
        counts = count.tolist()                                   # the one host sync of the sampler
        img = torch.cat([torch.full((c,), k, dtype=torch.int32) for k, c in enumerate(counts)]).to(dev, non_blocking=True)
        flat = torch.cat([torch.arange(c, dtype=torch.int64) + k * cap for k, c in enumerate(counts)]).to(dev, non_blocking=True)
        pid = sel_idx.view(-1)[flat].contiguous()
        lab = sel_lab.view(-1)[flat].to(torch.int64)
        # dense fp32 maps in either layout go to the gather as they are (the backbone hands over channels-last maps; all levels share it)
        cl = all(f.dim() == 4 and f.is_contiguous(memory_format=torch.channels_last) for f in features)
        feats = [f if (f.dtype == torch.float32 and (cl or f.is_contiguous())) else f.float().contiguous() for f in features]
        rows = ops.NodeGatherFn.apply(img, pid, *feats)
        nodes = list(torch.split(rows, counts, dim=0))
        labs = list(torch.split(lab, counts, dim=0))
        return nodes, labs
