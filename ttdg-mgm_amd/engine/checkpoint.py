"""Checkpoint reading for the eval-only flow — stands in for ``DetectionCheckpointer(...).resume_or_load(cfg.MODEL.WEIGHTS)``
[3P detectron2] as train_net.py:58-60 uses it: ``.pth`` (torch.save of a state dict, or of ``{"model": state_dict, ...}``) and
detectron2's ``.pkl`` (``{"model": {name: ndarray}}``).  Teacher/student ensembles (``modelStudent.`` / ``modelTeacher.`` key
prefixes, ts_ensemble.py) load the half ``TEST.EVAL_STU`` selects.  Shape mismatches are errors; missing / unexpected keys are
reported like detectron2 does and returned."""
import logging
import pickle

import numpy as np
import torch


def read_state_dict(path, trusted=False):
    """``.pth`` files are read with ``weights_only=True`` (tensors and plain containers only).  Formats that can only be read
    by unpickling arbitrary objects - detectron2's ``.pkl`` model-zoo files, or a ``.pth`` that carries non-tensor objects
    such as a pickled config - execute code from the file and are refused unless the caller says the file is ``trusted``
    (train_net.py: ``TTDG_TRUST_CHECKPOINT=1``)."""
    if path.endswith(".pkl"):
        if not trusted:
            raise ValueError("%s is a pickle: loading it executes code from the file; set TTDG_TRUST_CHECKPOINT=1 if you trust it" % path)
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
    else:
        try:
            data = torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as e:
            if not trusted:
                raise ValueError("%s holds more than tensors (%s); set TTDG_TRUST_CHECKPOINT=1 to unpickle it" % (path, e)) from e
            data = torch.load(path, map_location="cpu", weights_only=False)
    sd = data["model"] if isinstance(data, dict) and "model" in data and isinstance(data["model"], dict) else data
    out = {}
    for k, v in sd.items():
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        if torch.is_tensor(v):
            out[k[len("module."):] if k.startswith("module.") else k] = v
    return out


def load_weights(model, path, prefer_student=False, trusted=None):
    """Returns (missing_keys, unexpected_keys).  ``path`` == "" leaves the random initialisation in place.  A checkpoint
    none of whose keys match the model (a wrong prefix, a different architecture) is an error, not a random-init run."""
    import os
    if trusted is None:
        trusted = os.environ.get("TTDG_TRUST_CHECKPOINT", "0") == "1"
    log = logging.getLogger(__name__)
    if not path:
        log.info("No checkpoint given (MODEL.WEIGHTS is empty): the model keeps its random initialisation")
        return [], []
    sd = read_state_dict(path, trusted)
    halves = ("modelStudent.", "modelTeacher.") if prefer_student else ("modelTeacher.", "modelStudent.")
    for pre in halves:
        if any(k.startswith(pre) for k in sd):
            sd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
            break
    own = model.state_dict()
    for k, v in sd.items():
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            raise ValueError("checkpoint tensor {} has shape {}, the model expects {}".format(k, tuple(v.shape), tuple(own[k].shape)))
    matched = [k for k in sd if k in own]
    if not matched:
        raise ValueError("no tensor of checkpoint {} matches a model key (first checkpoint keys: {})".format(path, ", ".join(list(sd)[:4])))
    if any(k.startswith("backbone.") for k in own) and not any(k.startswith("backbone.") for k in matched):
        raise ValueError("checkpoint {} holds no backbone.* tensor the model knows".format(path))
    res = model.load_state_dict(sd, strict=False)
    if res.missing_keys:
        log.warning("Keys of the model missing in the checkpoint: %s", ", ".join(res.missing_keys))
    if res.unexpected_keys:
        log.warning("Checkpoint keys the model does not use: %s", ", ".join(res.unexpected_keys))
    return list(res.missing_keys), list(res.unexpected_keys)
