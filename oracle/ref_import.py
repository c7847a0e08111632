"""Import the reference's GModule read-only (BUILD CONTAINER ONLY).

TEST INFRASTRUCTURE.  Used by tests/golden/make_golden.py to validate the
oracle restatement and to emit golden vectors.  /root/reference does not exist
on the GPU box; nothing in `-m gpu` tests, smoke() or bench.py calls this.

Recipe: register empty namespace packages whose
``__path__`` points into the reference tree so ``adapteacher/__init__.py``
(-> detectron2, absent) never executes, and inject our Sinkhorn spec under the
name ``pygmtools`` (absent third-party dependency).
"""
import os
import sys
import types

REF = os.environ.get("TTDG_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "adapteacher", "modeling", "GModule"))


def load():
    """Returns (multi_graph_matching module, build_graph module)."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    from . import sinkhorn_spec
    for name, rel in (("adapteacher", "adapteacher"),
                      ("adapteacher.modeling", "adapteacher/modeling"),
                      ("adapteacher.modeling.GModule", "adapteacher/modeling/GModule"),
                      ("adapteacher.modeling.GModule.utils", "adapteacher/modeling/GModule/utils")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF, rel)]
            sys.modules[name] = m
    pg = types.ModuleType("pygmtools")
    pg.sinkhorn = sinkhorn_spec.sinkhorn
    sys.modules["pygmtools"] = pg
    import importlib
    mgm = importlib.import_module("adapteacher.modeling.GModule.multi_graph_matching")
    bg = importlib.import_module("adapteacher.modeling.GModule.build_graph")
    return mgm, bg


def load_tree_sinkhorn():
    """The reference tree's own log-Sinkhorn: ``GModule.sinkhorn_iter`` of the (otherwise dead) graph_matching.py:788-840.
    It does not touch ``self``; returned as a plain function (log_alpha, n_iters, slack) -> log_alpha."""
    load()
    import importlib
    gm = importlib.import_module("adapteacher.modeling.GModule.graph_matching")
    fn = gm.GModule.sinkhorn_iter
    return lambda log_alpha, n_iters, slack=False: fn(None, log_alpha, n_iters=n_iters, slack=slack)


class FakeBoxes:
    def __init__(self, t):
        self.tensor = t


class FakeInstances:
    """Duck-typed detectron2 Instances: what build_graph.py:79-85 touches."""

    def __init__(self, boxes, classes):
        self.pred_boxes = FakeBoxes(boxes)
        self.pred_classes = classes
        self._fields = {"pred_boxes": self.pred_boxes, "pred_classes": classes}

    def __len__(self):
        return len(self.pred_classes)


def load_dice_metric():
    """Import the reference's evaluation/dice_metric.py (pure-numpy E-/S-measure + Dice loop) with stub modules for
    its absent third-party imports (detectron2.evaluation / detectron2.data / pycocotools)."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    import importlib.util
    stubs = {"detectron2": {}, "detectron2.evaluation": {"DatasetEvaluator": object},
             "detectron2.data": {"MetadataCatalog": None, "DatasetCatalog": None}, "pycocotools": {}, "pycocotools.mask": {}}
    for name, attrs in stubs.items():
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]
    spec = importlib.util.spec_from_file_location("_ref_dice_metric", os.path.join(REF, "adapteacher", "evaluation", "dice_metric.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
