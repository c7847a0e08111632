"""Checkpoint reading for the eval-only flow — stands in for ``DetectionCheckpointer(...).resume_or_load(cfg.MODEL.WEIGHTS)``
[3P detectron2] as train_net.py:58-60 uses it: ``.pth`` (torch.save of a state dict, or of ``{"model": state_dict, ...}``) and
detectron2's ``.pkl`` (``{"model": {name: ndarray}}``).  Teacher/student ensembles (``modelStudent.`` / ``modelTeacher.`` key
prefixes, ts_ensemble.py) load the half ``TEST.EVAL_STU`` selects.  Shape mismatches are errors; missing / unexpected keys are
reported like detectron2 does and returned."""
import logging
import pickle

import numpy as np
import torch


def read_state_dict(path):
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
    else:
        data = torch.load(path, map_location="cpu", weights_only=False)
    sd = data["model"] if isinstance(data, dict) and "model" in data and isinstance(data["model"], dict) else data
    out = {}
    for k, v in sd.items():
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        if torch.is_tensor(v):
            out[k[len("module."):] if k.startswith("module.") else k] = v
    return out


def load_weights(model, path, prefer_student=True):
    """Returns (missing_keys, unexpected_keys).  ``path`` == "" leaves the random initialisation in place."""
    log = logging.getLogger(__name__)
    if not path:
        log.info("No checkpoint given (MODEL.WEIGHTS is empty): the model keeps its random initialisation")
        return [], []
    sd = read_state_dict(path)
    halves = ("modelStudent.", "modelTeacher.") if prefer_student else ("modelTeacher.", "modelStudent.")
    for pre in halves:
        if any(k.startswith(pre) for k in sd):
            sd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
            break
    own = model.state_dict()
    for k, v in sd.items():
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            raise ValueError("checkpoint tensor {} has shape {}, the model expects {}".format(k, tuple(v.shape), tuple(own[k].shape)))
    res = model.load_state_dict(sd, strict=False)
    if res.missing_keys:
        log.warning("Keys of the model missing in the checkpoint: %s", ", ".join(res.missing_keys))
    if res.unexpected_keys:
        log.warning("Checkpoint keys the model does not use: %s", ", ".join(res.unexpected_keys))
    return list(res.missing_keys), list(res.unexpected_keys)
