"""CPU restatements of the two detection helpers (TEST INFRASTRUCTURE, see oracle/__init__.py): greedy NMS and
ROIAlign (aligned=True, adaptive sampling).  They stand in for torchvision.ops.nms / detectron2 ROIAlignV2 [3P,
absent]; used to check csrc/detection.hip and by the CPU baseline (oracle/tta_cpu.py)."""
import numpy as np
import torch


def nms(boxes, scores, thr, group=None):
    N = boxes.shape[0]
    if N == 0:
        return torch.empty(0, dtype=torch.int64)
    order = torch.argsort(scores, descending=True)
    b = boxes[order].double().numpy()
    g = np.zeros(N, np.int64) if group is None else group[order].numpy()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    removed = np.zeros(N, bool)
    keep = []
    for i in range(N):
        if removed[i]:
            continue
        keep.append(i)
        j = np.arange(i + 1, N)
        iw = np.minimum(b[i, 2], b[j, 2]) - np.maximum(b[i, 0], b[j, 0])
        ih = np.minimum(b[i, 3], b[j, 3]) - np.maximum(b[i, 1], b[j, 1])
        inter = np.clip(iw, 0, None) * np.clip(ih, 0, None)
        hit = (inter > thr * (area[i] + area[j] - inter)) & (g[j] == g[i])
        removed[j[hit]] = True
    return order[torch.tensor(keep, dtype=torch.int64)]


def roi_align(feat, rois, scale, P):
    """feat (B,C,H,W), rois (R,5) -> (R,C,P,P).  Loop over ROIs, vectorised over the sampling grid."""
    feat = feat.detach().float()
    B, C, H, W = feat.shape
    out = feat.new_zeros((rois.shape[0], C, P, P))
    for r in range(rois.shape[0]):
        b = int(rois[r, 0])
        x1, y1 = float(rois[r, 1]) * scale - 0.5, float(rois[r, 2]) * scale - 0.5
        rw, rh = float(rois[r, 3]) * scale - 0.5 - x1, float(rois[r, 4]) * scale - 0.5 - y1
        gh, gw = max(1, int(np.ceil(rh / P))), max(1, int(np.ceil(rw / P)))
        ys = y1 + (torch.arange(P)[:, None] + (torch.arange(gh)[None] + 0.5) / gh) * (rh / P)    # (P, gh)
        xs = x1 + (torch.arange(P)[:, None] + (torch.arange(gw)[None] + 0.5) / gw) * (rw / P)    # (P, gw)
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        vy, vx = (ys >= -1) & (ys <= H), (xs >= -1) & (xs <= W)
        yc, xc = ys.clamp(min=0), xs.clamp(min=0)
        y0, x0 = yc.floor().long(), xc.floor().long()
        ty, tx = y0 >= H - 1, x0 >= W - 1
        y0, x0 = torch.where(ty, torch.full_like(y0, H - 1), y0), torch.where(tx, torch.full_like(x0, W - 1), x0)
        y1i, x1i = torch.where(ty, y0, y0 + 1), torch.where(tx, x0, x0 + 1)
        yc, xc = torch.where(ty, y0.float(), yc), torch.where(tx, x0.float(), xc)
        ly, lx = yc - y0, xc - x0
        f = feat[b]
        v = (f[:, y0][:, :, x0] * ((1 - ly)[:, None] * (1 - lx)[None]) + f[:, y0][:, :, x1i] * ((1 - ly)[:, None] * lx[None])
             + f[:, y1i][:, :, x0] * (ly[:, None] * (1 - lx)[None]) + f[:, y1i][:, :, x1i] * (ly[:, None] * lx[None]))
        v = v * (vy[:, None] & vx[None]).float()
        out[r] = v.reshape(C, P, gh, P, gw).sum(dim=(2, 4)) / (gh * gw)
    return out
