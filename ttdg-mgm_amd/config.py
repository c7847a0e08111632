"""Config for the TTA path without yacs/detectron2: the keys ``test_segment.yaml`` and
``add_ateacher_config`` (reference config.py:5-64) define that the eval-only flow actually reads, with the
detectron2 defaults it silently relies on."""
import copy

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_dict(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else v

    def merge_from_file(self, path):
        with open(path) as f:
            d = yaml.safe_load(f) or {}
        d.pop("_BASE_", None)          # Base-RCNN-FPN.yaml values are the defaults below
        self.merge_from_dict(d)

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0
        for k, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = _decode(v) if isinstance(v, str) else v


def _decode(v):
    """Command-line values as yacs reads them: a Python literal when it parses as one ("('a',)", "0.001", "True"), else
    yaml scalars ("true", "null"), else the string itself."""
    import ast
    try:
        out = ast.literal_eval(v)
        return list(out) if isinstance(out, tuple) else out
    except (ValueError, SyntaxError):
        try:
            return yaml.safe_load(v)
        except yaml.YAMLError:
            return v


def _node(d):
    return CfgNode({k: _node(v) if isinstance(v, dict) else v for k, v in d.items()})


def get_cfg():
    return _node({
        "MODEL": {
            "META_ARCHITECTURE": "DAobjTwoStagePseudoLabGeneralizedRCNN", "DEVICE": "cuda", "WEIGHTS": "",
            "PIXEL_MEAN": [103.530, 116.280, 123.675], "PIXEL_STD": [1.0, 1.0, 1.0],
            "ROI_HEADS": {"NAME": "StandardROIHeadsPseudoLab", "NUM_CLASSES": 2, "SCORE_THRESH_TEST": 0.05,
                          "NMS_THRESH_TEST": 0.5},
            "PROPOSAL_GENERATOR": {"NAME": "PseudoLabRPN"},
        },
        "INPUT": {"FORMAT": "RGB", "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333},
        "DATASETS": {"TEST": []},
        "DATALOADER": {"NUM_WORKERS": 4},
        "SOLVER": {"BASE_LR": 0.005, "MOMENTUM": 0.9, "WEIGHT_DECAY": 1e-4, "WEIGHT_DECAY_NORM": 0.0, "IMG_PER_BATCH_LABEL": 8},
        "TEST": {"TTT": True, "BATCH": 4, "DICE_THRES": 0.9, "MIN_BATCH_NUM": None, "DETECTIONS_PER_IMAGE": 100,
                 "EVALUATOR": "COCOeval"},
        "SEMISUPNET": {"Trainer": "baseline", "BBOX_THRESHOLD": 0.8, "DIS_TYPE": "p2"},
        "OUTPUT_DIR": "./output",
    })


def add_ateacher_config(cfg):
    """The reference adds its keys to a detectron2 cfg here; ours already carries them (kept for call compatibility)."""
    return cfg
