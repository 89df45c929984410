"""dedark_yolo_amd -- MI355X-native (gfx950) Dedark-YOLO training/validation hot path.

Public surface mirrors the reference (`from ultralytics import YOLO`): YOLO(model).train()/val(), DetectionModel, the
yolov8.yaml module registry.  All device work goes through libdedark_yolo.so (hand-written HIP); see include/dedark_yolo.h.
"""
from . import ops  # noqa: F401
from .ops import get_compute_dtype, set_compute_dtype  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    if name == "YOLO":
        from .engine.model import YOLO
        return YOLO
    if name == "DetectionModel":
        from .nn.tasks import DetectionModel
        return DetectionModel
    raise AttributeError(name)
