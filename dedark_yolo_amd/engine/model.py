"""`YOLO` facade with the reference call surface (ultralytics/engine/model.py:29-416): YOLO(model), .train(**kw), .val(**kw),
.load(), .fuse(); detect task only (the other tasks are outside the hot path)."""
from pathlib import Path

import torch

from ..nn.tasks import DetectionModel
from .trainer import DetectionTrainer, get_cfg


class YOLO:
    def __init__(self, model="yolov8l.yaml", task=None):
        if task not in (None, "detect"):
            raise NotImplementedError("only the detect task is on the Dedark-YOLO hot path")
        self.task = "detect"
        self.trainer = None
        self.overrides = {}
        suffix = Path(str(model)).suffix
        if suffix == ".yaml":
            self._new(model)
        elif suffix in (".pt", ".pth"):
            self._load(model)
        else:
            raise FileNotFoundError(f"'{model}': expected a model .yaml or a state_dict checkpoint .pt")

    def _new(self, cfg):
        self.cfg = cfg
        self.model = DetectionModel(cfg)
        self.overrides["model"] = cfg

    def _load(self, weights):
        """reference nn/tasks.py:592-630,674-707 (attempt_load_one_weight): `ckpt.get('ema') or ckpt['model']`, cast to fp32.  Reads
        both this package's state_dict checkpoints and the reference's pickled-module last.pt / best.pt (utils/checkpoint.py)."""
        from ..utils.checkpoint import load_checkpoint
        ck = load_checkpoint(weights)
        cfg = ck.yaml
        if cfg is None:
            raise RuntimeError(f"{weights}: the checkpoint carries no model yaml")
        self.model = DetectionModel(cfg, nc=ck.nc)
        n = self.model.load(ck.state_dict)
        if n == 0:
            raise RuntimeError(f"{weights}: no tensor of the checkpoint matches the graph of its yaml")
        if isinstance(ck.names, dict) and len(ck.names) == len(self.model.names):
            self.model.names = {int(k): str(v) for k, v in ck.names.items()}
        self.ckpt = ck
        self.cfg = cfg
        self.overrides["model"] = cfg

    def __call__(self, source, **kw):
        return self.predict(source, **kw)

    def load(self, weights):
        self.model.load(weights)
        return self

    def fuse(self):
        self.model.fuse()
        return self

    def train(self, loader=None, **kwargs):
        """model.train(data=..., epochs=..., imgsz=..., batch=..., device=...) -- `loader` is any iterable of reference-schema batch
        dicts (the cv2 data pipeline is outside the hot path)."""
        ov = dict(self.overrides)
        ov.update(kwargs)
        self.trainer = DetectionTrainer(get_cfg(ov))
        if loader is None:
            raise ValueError("pass loader=<iterable of batch dicts>; dataset decoding/augmentation is outside the hot path")
        self.trainer.setup(self.model, total_iterations=len(loader) * self.trainer.args.epochs)
        return self.trainer.train(loader)

    def val(self, loader=None, **kwargs):
        from .validator import DetectionValidator
        ov = dict(self.overrides)
        ov.update(kwargs)
        v = DetectionValidator(get_cfg(ov))
        return v(self.model, loader)

    @torch.no_grad()
    def predict(self, source, conf=0.25, iou=0.7, max_det=300, agnostic_nms=False, orig_shapes=None, **kw):
        """reference engine/predictor.py stream_inference + DetectionPredictor.postprocess (models/yolo/detect/predict.py:12-38)
        for an already letter-boxed batch: `source` is a uint8 [B,3,H,W] RGB tensor (or float in [0,1]); returns one `Results`
        per image with boxes scaled back to `orig_shapes[i]` (default: the network input shape).  Image decoding / letter-boxing
        (cv2) is outside the hot path."""
        from ..utils import ops as uops
        from .results import Results
        from .validator import DetectionValidator
        dev = next(self.model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("predict() needs the model on a GPU (there is no CPU path)")
        was_training = self.model.training
        self.model.eval()
        if source.dtype == torch.uint8:
            v = DetectionValidator(get_cfg(dict(self.overrides)))
            v.device = dev
            img = v.preprocess(dict(img=source))["img"]
        else:
            img = source.to(dev).float()
        preds = self.model(img)
        dets = uops.non_max_suppression(preds, conf, iou, agnostic=agnostic_nms, max_det=max_det)
        H, W = img.shape[2:]
        out = []
        for i, d in enumerate(dets):
            shape = tuple(orig_shapes[i]) if orig_shapes is not None else (H, W)
            d = d.clone()
            uops.scale_boxes((H, W), d[:, :4], shape)          # also clips to the image, like predict.py:27
            out.append(Results(shape, d, names=self.model.names))
        self.model.train(was_training)
        return out

    def save(self, path):
        torch.save(dict(state_dict=self.model.state_dict(), yaml=self.model.yaml, nc=self.model.yaml["nc"]), path)
