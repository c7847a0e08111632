"""DiceEvaluator — mirror of reference evaluation/dice_metric.py:13-240 on bitmap ground truth (pycocotools is
absent), reductions on the device: per predicted mask with score >= thres, best Dice / E-measure / S-measure over
the same-class GT masks, x100, mean over all kept masks.  Like the reference it is rank-local; ``gather_scores``
adds the all-gather the reference omits (SURVEY.md §8e Mode R)."""
import numpy as np
import torch


# All three measures are written without host synchronisation (no .item(), no Python branch on a device value):
# every data-dependent case of the reference becomes a torch.where, so a whole pass enqueues and syncs once.
def _t(x, like):
    return torch.as_tensor(x, dtype=torch.float64, device=like.device)


def dice_tensor(p, g):
    inter = (p & g).sum().double()
    return 2 * inter / (p.sum().double() + g.sum().double() + 1e-6)


def enhanced_align_tensor(pred, gt):
    """dice_metric.py:110-143 (E-measure, IJCAI 2018) for boolean maps."""
    p, g = pred.double(), gt.double()
    th = torch.clamp(2 * p.mean(), max=1.0)
    fm = (p >= th).double()
    af, ag = fm - fm.mean(), g - g.mean()
    general = ((2.0 * (ag * af) / (ag * ag + af * af + 1e-8)) + 1) ** 2 / 4
    em = torch.where(g.sum() == 0, 1.0 - fm, torch.where((1 - g).sum() == 0, fm, general))
    return em.sum() / (g.numel() - 1 + 1e-8)


def _ssim_t(a, b):
    n = a.numel()
    if n == 0:
        return _t(float("nan"), a)
    x, y = a.mean(), b.mean()
    sx, sy = a.var(unbiased=False), b.var(unbiased=False)
    sxy = ((a - x) * (b - y)).sum() / (n - 1) if n > 1 else _t(float("nan"), a)
    alpha, beta = 4 * x * y * sxy, (x * x + y * y) * (sx + sy)
    return torch.where(alpha != 0, alpha / (beta + 1e-8), torch.where(beta == 0, _t(1.0, a), _t(0.0, a)))


def _s_object_t(v, m):
    w = m.double()
    cnt = w.sum()
    x = (v * w).sum() / cnt
    sig = torch.sqrt((((v - x) ** 2) * w).sum() / cnt)
    return 2 * x / (x * x + 1 + sig + 1e-8)


def structure_measure_tensor(pred, gt, alpha=0.5, centroid=None):
    """dice_metric.py:147-240 (S-measure, ICCV 2017) for boolean maps.  ``centroid`` = (cy, cx) of the GT (a host
    constant of the dataset; computed here with one sync when not supplied)."""
    p, g = pred.double(), gt > 0.5
    gd = g.double()
    y = gd.mean()
    obj = y * _s_object_t(p * gd, g) + (1 - y) * _s_object_t((1 - p) * (1 - gd), ~g)
    if centroid is None:
        ys, xs = torch.nonzero(g, as_tuple=True)
        centroid = (ys.double().mean().item(), xs.double().mean().item()) if ys.numel() else (0.0, 0.0)
    h, w = g.shape
    if centroid[0] != centroid[0]:            # NaN centroid of an empty GT: the general branch is not selected below
        cy, cx = 1, 1
    else:
        cy, cx = int(round(centroid[0])) + 1, int(round(centroid[1])) + 1
    reg = _t(0.0, p)
    for (r0, r1, c0, c1) in ((0, cy, 0, cx), (0, cy, cx, w), (cy, h, 0, cx), (cy, h, cx, w)):
        wgt = (r1 - r0) * (c1 - c0) / (h * w)
        if wgt > 0:
            reg = reg + wgt * _ssim_t(p[r0:r1, c0:c1], gd[r0:r1, c0:c1])
    general = alpha * obj + (1 - alpha) * reg
    return torch.where(y == 0, 1 - p.mean(), torch.where(y == 1, p.mean(), general))


def dice_coefficient(p, g):
    return dice_tensor(p, g).item()


def enhanced_align(pred, gt):
    return enhanced_align_tensor(pred, gt).item()


def structure_measure(pred, gt, alpha=0.5):
    return structure_measure_tensor(pred, gt, alpha).item()


class DiceEvaluator:
    GT_CACHE_IMAGES = 256        # device copies of the ground-truth masks kept between passes (bounded: oldest dropped)

    def __init__(self, dataset_name, thres, dataset_dicts=None, lazy=False):
        """``dataset_dicts``: ground truth of the local shard.  ``lazy=True``: hold none - every input of ``process`` carries
        its own ``dataset_dict`` (the streaming TestLoader), so a large dataset is never resident."""
        from ..data import dataset_dicts as _dd
        self.dataset_name = dataset_name
        if lazy:
            self.dataset_dicts, self._by_id = None, {}
        else:
            self.dataset_dicts = dataset_dicts if dataset_dicts is not None else _dd(dataset_name)
            self._by_id = {d["image_id"]: d for d in self.dataset_dicts}
        self.score_threshold = thres
        self._gt_cache = {}
        self.reset()

    def reset(self):
        self.dice_scores, self.ea_scores, self.sm_scores = [], [], []
        self._pending = []

    def _gt(self, image_id, dev, record=None):
        key = (image_id, str(dev))
        if key not in self._gt_cache:
            out = []
            if record is None:
                record = self._by_id[image_id]
            while len(self._gt_cache) >= self.GT_CACHE_IMAGES:
                self._gt_cache.pop(next(iter(self._gt_cache)))
            for a in record["annotations"]:
                m = a["mask"]
                ys, xs = torch.nonzero(m, as_tuple=True)
                cen = (ys.double().mean().item(), xs.double().mean().item()) if ys.numel() else (float("nan"), float("nan"))
                out.append((a["category_id"], m.to(dev), cen))
            self._gt_cache[key] = out
        return self._gt_cache[key]

    def process(self, inputs, outputs):
        """Enqueues everything on the device; ONE host read per batch (which predictions pass the score threshold and their
        classes), none when nothing passes."""
        insts = [out["instances"] for out in outputs]
        lens = [len(i) for i in insts]
        if sum(lens) == 0:
            return
        scores = torch.cat([i.scores for i in insts])
        classes = torch.cat([i.pred_classes for i in insts])
        sel = torch.where(scores >= self.score_threshold, classes, torch.full_like(classes, -1)).tolist()      # the one read
        start = 0
        for inp, inst, n in zip(inputs, insts, lens):
            mine = sel[start:start + n]
            start += n
            kept = [k for k, c in enumerate(mine) if c >= 0]
            if not kept:
                continue
            gts = self._gt(inp["image_id"], inst.pred_masks.device, inp.get("dataset_dict"))
            zero = torch.zeros((), dtype=torch.float64, device=inst.pred_masks.device)
            for k in kept:
                pc, pm = mine[k], inst.pred_masks[k]
                bd = be = bs = zero
                for gc, gm, cen in gts:
                    if pc == gc:
                        bd = torch.maximum(bd, dice_tensor(pm, gm))
                        be = torch.maximum(be, enhanced_align_tensor(pm, gm))
                        bs = torch.maximum(bs, structure_measure_tensor(pm, gm, centroid=cen))
                self._pending.append(torch.stack((bd, be, bs)) * 100)

    def _flush(self):
        if self._pending:
            vals = torch.stack(self._pending).cpu().tolist()
            self._pending = []
            for d, e, s in vals:
                self.dice_scores.append(d), self.ea_scores.append(e), self.sm_scores.append(s)

    def gather_scores(self):
        """All-gather of the per-rank score lists (the reference reports rank-local means, dice_metric.py:80-92): sizes
        first, then one padded all_gather of a (max_len, 3) tensor - RCCL on the GPUs (backend nccl), gloo on CPU."""
        import torch.distributed as dist
        self._flush()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        n = torch.tensor([len(self.dice_scores)], dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n)
        sizes = [int(x.item()) for x in sizes]
        mx = max(max(sizes), 1)
        mine = torch.full((mx, 3), float("nan"), dtype=torch.float64, device=dev)
        if sizes[dist.get_rank()]:
            mine[:len(self.dice_scores)] = torch.tensor(list(zip(self.dice_scores, self.ea_scores, self.sm_scores)), dtype=torch.float64, device=dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        rows = torch.cat([p[:k] for p, k in zip(parts, sizes)]).cpu().tolist()
        self.dice_scores, self.ea_scores, self.sm_scores = [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows]

    def evaluate(self):
        self._flush()
        mean = lambda v: float(np.mean(v)) if len(v) else float("nan")
        return {"Dice Coefficient": mean(self.dice_scores), "Enhanced Alignment Metric": mean(self.ea_scores),
                "Structural Similarity Metric": mean(self.sm_scores)}
