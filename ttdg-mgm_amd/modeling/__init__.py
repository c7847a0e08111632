"""Torch-native Mask R-CNN (ResNet-50-FPN) stand-in for the un-vendored detectron2 pieces the TTA path calls
(SURVEY.md §8a A0 / §8f N1).  Registered under the reference's class names; conv/GEMM run on MIOpen/hipBLASLt
through PyTorch (vendor kernels, reported only), ROIAlign / NMS / node sampling / matching are our HIP kernels.
Parameter names follow detectron2's so that its checkpoints load."""
from .rcnn import DAobjTwoStagePseudoLabGeneralizedRCNN, build_model, calibrate_frozen_bn  # noqa: F401
from .structures import Boxes, ImageList, Instances  # noqa: F401
