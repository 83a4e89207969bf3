"""Host-side PyTorch-ROCm model glue for the configs in BASELINE.json (ResNet-50 + FPN +
rotated RetinaNet heads + R3Det refinement).  Plumbing around the hot-path ops: the convs run
in MIOpen, the rotated ops in libr3det_hip.so."""
from .detectors import R3Det, RRetinaNet, build_detector
from .heads import RRetinaHead, RRetinaRefineHead

__all__ = ['R3Det', 'RRetinaNet', 'RRetinaHead', 'RRetinaRefineHead', 'build_detector']
