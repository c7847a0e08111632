"""DiceEvaluator — mirror of reference evaluation/dice_metric.py:13-240 on bitmap ground truth (pycocotools is
absent), reductions on the device: per predicted mask with score >= thres, best Dice / E-measure / S-measure over
the same-class GT masks, x100, mean over all kept masks.  Like the reference it is rank-local; ``gather_scores``
adds the all-gather the reference omits (SURVEY.md §8e Mode R)."""
import numpy as np
import torch


def dice_coefficient(p, g):
    inter = (p & g).sum().double()
    return (2 * inter / (p.sum().double() + g.sum().double() + 1e-6)).item()


def enhanced_align(pred, gt):
    """dice_metric.py:110-143 (E-measure, IJCAI 2018) for boolean maps."""
    p = pred.double()
    th = min(2 * p.mean().item(), 1.0)
    fm = (p >= th).double()
    g = gt.double()
    if g.sum() == 0:
        em = 1.0 - fm
    elif (1 - g).sum() == 0:
        em = fm
    else:
        af, ag = fm - fm.mean(), g - g.mean()
        al = 2.0 * (ag * af) / (ag * ag + af * af + 1e-8)
        em = (al + 1) ** 2 / 4
    return (em.sum() / (g.numel() - 1 + 1e-8)).item()


def _ssim(a, b):
    b = b.double()
    n = a.numel()
    if n == 0:
        return float("nan")
    x, y = a.mean(), b.mean()
    sx, sy = a.var(unbiased=False), b.var(unbiased=False)
    sxy = ((a - x) * (b - y)).sum() / (n - 1) if n > 1 else torch.tensor(float("nan"), dtype=torch.float64)
    alpha, beta = 4 * x * y * sxy, (x * x + y * y) * (sx + sy)
    if alpha != 0:
        return (alpha / (beta + 1e-8)).item()
    return 1.0 if beta == 0 else 0.0


def _s_object(v, m):
    sel = v[m]
    x, s = sel.mean(), sel.std(unbiased=False)
    return (2 * x / (x * x + 1 + s + 1e-8)).item()


def structure_measure(pred, gt, alpha=0.5):
    """dice_metric.py:147-240 (S-measure, ICCV 2017) for boolean maps."""
    p, g = pred.double(), gt > 0.5
    y = g.double().mean().item()
    if y == 0:
        return 1 - p.mean().item()
    if y == 1:
        return p.mean().item()
    gd = g.double()
    obj = y * _s_object(p * gd, g) + (1 - y) * _s_object((1 - p) * (1 - gd), ~g)
    ys, xs = torch.nonzero(g, as_tuple=True)
    cy, cx = int(round(ys.double().mean().item())) + 1, int(round(xs.double().mean().item())) + 1
    h, w = g.shape
    area = h * w
    reg = 0.0
    for (r0, r1, c0, c1) in ((0, cy, 0, cx), (0, cy, cx, w), (cy, h, 0, cx), (cy, h, cx, w)):
        wgt = (r1 - r0) * (c1 - c0) / area
        if wgt > 0:
            reg += wgt * _ssim(p[r0:r1, c0:c1], gd[r0:r1, c0:c1])
    return alpha * obj + (1 - alpha) * reg


class DiceEvaluator:
    def __init__(self, dataset_name, thres, dataset_dicts=None):
        from ..data import dataset_dicts as _dd
        self.dataset_name = dataset_name
        self.dataset_dicts = dataset_dicts if dataset_dicts is not None else _dd(dataset_name)
        self._by_id = {d["image_id"]: d for d in self.dataset_dicts}
        self.score_threshold = thres
        self.reset()

    def reset(self):
        self.dice_scores, self.ea_scores, self.sm_scores = [], [], []

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            anns = self._by_id[inp["image_id"]]["annotations"]
            inst = out["instances"]
            keep = inst.scores >= self.score_threshold
            masks, classes = inst.pred_masks[keep], inst.pred_classes[keep]
            dev = masks.device
            gts = [(a["category_id"], a["mask"].to(dev)) for a in anns]
            for pc, pm in zip(classes.tolist(), masks):
                bd = be = bs = 0
                for gc, gm in gts:
                    if pc == gc:
                        bd = max(bd, dice_coefficient(pm, gm))
                        be = max(be, enhanced_align(pm, gm))
                        bs = max(bs, structure_measure(pm, gm))
                self.dice_scores.append(bd * 100)
                self.ea_scores.append(be * 100)
                self.sm_scores.append(bs * 100)

    def gather_scores(self):
        """All-gather of the per-rank score lists (the reference reports rank-local means)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, (self.dice_scores, self.ea_scores, self.sm_scores))
        self.dice_scores = [x for p in parts for x in p[0]]
        self.ea_scores = [x for p in parts for x in p[1]]
        self.sm_scores = [x for p in parts for x in p[2]]

    def evaluate(self):
        mean = lambda v: float(np.mean(v)) if len(v) else float("nan")
        return {"Dice Coefficient": mean(self.dice_scores), "Enhanced Alignment Metric": mean(self.ea_scores),
                "Structural Similarity Metric": mean(self.sm_scores)}
