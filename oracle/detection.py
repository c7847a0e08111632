"""ctypes wrappers over oracle/detection.c (TEST INFRASTRUCTURE, see oracle/__init__.py): CPU ROIAlign
(aligned=True, adaptive sampling) and greedy NMS, standing in for detectron2 ROIAlignV2 / torchvision nms [3P,
absent].  They check csrc/detection.hip and serve bench.py's CPU baseline."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_det.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "detection.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = ctypes.CDLL(_SO)
    return _lib


def nms(boxes, scores, thr, group=None):
    N = boxes.shape[0]
    if N == 0:
        return torch.empty(0, dtype=torch.int64)
    order = torch.argsort(scores.detach(), descending=True)
    b = boxes.detach().float()[order].contiguous()
    g = (torch.zeros(N, dtype=torch.int32) if group is None else group[order].to(torch.int32)).contiguous()
    keep = torch.empty(N, dtype=torch.int32)
    n = _load().ttdg_oracle_nms(ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(g.data_ptr()), N, ctypes.c_float(thr),
                                ctypes.c_void_p(keep.data_ptr()))
    return order[keep[:n].long()]


def roi_align(feat, rois, scale, P):
    feat = feat.detach().float().contiguous()
    rois = rois.detach().float().contiguous()
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = torch.empty(R, C, P, P)
    if R:
        _load().ttdg_oracle_roi_align(ctypes.c_void_p(feat.data_ptr()), B, C, H, W, ctypes.c_void_p(rois.data_ptr()), R,
                                      ctypes.c_float(scale), P, ctypes.c_void_p(out.data_ptr()))
    return out
