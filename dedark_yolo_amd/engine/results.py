"""Minimal prediction containers with the attribute surface of the reference's ultralytics/engine/results.py:
`Results.boxes` -> `Boxes` with `.data` ([n,6] xyxy, conf, cls), `.xyxy`, `.conf`, `.cls`, `.xywh`, `.xyxyn`, `len()`; `Results`
also carries `orig_shape` and `names`.  Plotting / saving helpers are outside the hot path."""
import torch

from ..utils import ops


class Boxes:
    def __init__(self, data, orig_shape):
        if data.ndim == 1:
            data = data[None, :]
        assert data.shape[-1] == 6, "Boxes expects rows of (x1, y1, x2, y2, conf, cls)"
        self.data = data
        self.orig_shape = orig_shape

    @property
    def xyxy(self):
        return self.data[:, :4]

    @property
    def conf(self):
        return self.data[:, 4]

    @property
    def cls(self):
        return self.data[:, 5]

    @property
    def xywh(self):
        return ops.xyxy2xywh(self.xyxy)

    @property
    def xyxyn(self):
        h, w = self.orig_shape
        return self.xyxy / torch.tensor([w, h, w, h], dtype=self.data.dtype, device=self.data.device)

    def cpu(self):
        return Boxes(self.data.cpu(), self.orig_shape)

    def __len__(self):
        return self.data.shape[0]


class Results:
    def __init__(self, orig_shape, boxes, names=None, path=None):
        self.orig_shape = tuple(orig_shape)
        self.boxes = Boxes(boxes, self.orig_shape)
        self.names = names
        self.path = path

    def __len__(self):
        return len(self.boxes)
