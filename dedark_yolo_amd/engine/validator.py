"""Detect-task validator with the reference's structure (ultralytics/engine/validator.py:95-200 `__call__` loop,
ultralytics/models/yolo/detect/val.py:30-174): preprocess -> model (eval) -> batched HIP NMS -> per-image greedy matching at
10 IoU thresholds -> AP per class / mAP50 / mAP50-95 / fitness.

The device part (img/255, front-end, network, Detect decode, NMS) stays on the GPU; one D2H copy per batch brings the kept
detections to the host, where the matching and AP bookkeeping run in numpy exactly as in the reference (val.py:151-174 uses
numpy for the matching as well)."""
import numpy as np
import torch

from .. import ops as kops
from .._C import call
from ..ops import ptr, stream
from ..utils import ops
from ..utils.metrics import DetMetrics, box_iou


def match_predictions(detections, labels, iouv):
    """Which detections count as true positives at each IoU threshold: detections [N,6] (xyxy, conf, cls), labels [M,5] (cls, xyxy),
    host tensors -> bool [N, len(iouv)].  The rule of the reference's `_process_batch` (ultralytics/models/yolo/detect/val.py:151-174),
    stated directly instead of through its sort / unique calls: per threshold, among the (label, detection) pairs of equal class with
    IoU >= threshold,
      1. every detection keeps ONE label, the one it overlaps most (an exact IoU tie goes to the higher label index: the order the
         reference's reversed ascending sort leaves equal keys in);
      2. every label keeps ONE of the detections that chose it -- the one with the LOWEST detection index (its second `np.unique` runs on
         rows that the first one has re-ordered by detection index, so it is not the best-overlapping one).
    The detections that survive both steps are correct at that threshold."""
    iou = box_iou(labels[:, 1:], detections[:, :4]).numpy()
    n_lab, n_det = iou.shape
    correct = np.zeros((n_det, len(iouv)), dtype=bool)
    if n_lab == 0 or n_det == 0:
        return torch.from_numpy(correct)
    same_cls = (labels[:, 0:1] == detections[:, 5]).numpy()
    thr = np.asarray(iouv, dtype=iou.dtype)
    for k in range(len(thr)):
        cand = (iou >= thr[k]) & same_cls
        dets = np.flatnonzero(cand.any(0))
        if dets.size == 0:
            continue
        score = np.where(cand[:, dets], iou[:, dets], -1.0)
        chosen = n_lab - 1 - np.argmax(score[::-1], axis=0)          # step 1 (argmax of the flipped column: last maximum)
        keep = np.unique(chosen, return_index=True)[1]               # step 2: first = lowest detection index per label
        correct[dets[keep], k] = True
    return torch.from_numpy(correct)


class DetectionValidator:
    def __init__(self, args=None, dataloader=None):
        from .trainer import get_cfg
        self.args = args if args is not None else get_cfg()
        self.dataloader = dataloader
        self.iouv = torch.linspace(0.5, 0.95, 10)
        self.niou = self.iouv.numel()
        self.metrics = DetMetrics()
        self.device = None
        self.training = False
        # engine/validator.py:86-87: conf None -> 0.001; this fork's default.yaml:48 sets conf 0.25, which therefore applies
        self.conf = 0.001 if getattr(self.args, "conf", None) is None else self.args.conf

    # ------------------------------------------------------------------------------------------ per batch
    def preprocess(self, batch):
        """val.py:30-41: uint8 -> float / 255 on the device (no darkening at validation time)."""
        img = batch["img"]
        if img.dtype != torch.uint8:
            raise RuntimeError("validator expects the dataloader's uint8 image tensor")
        img = img.to(self.device, non_blocking=True).contiguous()
        out = torch.empty(img.shape, dtype=torch.float32, device=self.device)
        acc = torch.zeros(1, dtype=torch.float64, device=self.device)
        call("dy_preprocess_batch", ptr(img), ptr(out), None, 1.0, 0, 0, ptr(acc), img.numel(), stream())
        batch["img"] = out
        return batch

    def postprocess(self, preds):
        a = self.args
        return ops.non_max_suppression(preds, self.conf, a.iou, multi_label=True, agnostic=bool(getattr(a, "single_cls", False)),
                                       max_det=a.max_det)

    def init_metrics(self, model):
        self.nc = model.model[-1].nc
        self.names = getattr(model, "names", None) or {i: str(i) for i in range(self.nc)}
        self.metrics.names = self.names
        self.seen = 0
        self.stats = []

    def update_metrics(self, preds, batch):
        """val.py:72-116 on host copies."""
        bi = batch["batch_idx"].cpu()
        cls_all, box_all = batch["cls"].cpu().float(), batch["bboxes"].cpu().float()
        height, width = batch["img"].shape[2:]
        for si, pred in enumerate(preds):
            pred = pred.cpu()
            idx = bi == si
            cls, bbox = cls_all[idx], box_all[idx]
            nl, npr = cls.shape[0], pred.shape[0]
            shape = batch["ori_shape"][si] if "ori_shape" in batch else (height, width)
            ratio_pad = batch["ratio_pad"][si] if "ratio_pad" in batch else None
            correct = torch.zeros(npr, self.niou, dtype=torch.bool)
            self.seen += 1
            if npr == 0:
                if nl:
                    self.stats.append((correct, torch.zeros(0), torch.zeros(0), cls.squeeze(-1)))
                continue
            if getattr(self.args, "single_cls", False):
                pred[:, 5] = 0
            predn = pred.clone()
            ops.scale_boxes((height, width), predn[:, :4], shape, ratio_pad=ratio_pad)
            if nl:
                tbox = ops.xywh2xyxy(bbox) * torch.tensor((width, height, width, height), dtype=torch.float32)
                ops.scale_boxes((height, width), tbox, shape, ratio_pad=ratio_pad)
                correct = match_predictions(predn, torch.cat((cls, tbox), 1), self.iouv)
            self.stats.append((correct, pred[:, 4], pred[:, 5], cls.squeeze(-1)))

    def get_stats(self):
        """val.py:123-129"""
        if not self.stats:
            self.nt_per_class = np.zeros(self.nc, dtype=int)
            return self.metrics.results_dict
        stats = [torch.cat(x, 0).numpy() for x in zip(*self.stats)]
        if len(stats) and stats[0].any():
            self.metrics.process(*stats)
        self.nt_per_class = np.bincount(stats[-1].astype(int), minlength=self.nc)
        return self.metrics.results_dict

    # ------------------------------------------------------------------------------------------ loop
    @torch.no_grad()
    def __call__(self, model, dataloader=None, dtype=None):
        """engine/validator.py:95-200 (model given directly; the trainer passes its EMA weights loaded into `model`)."""
        loader = dataloader if dataloader is not None else self.dataloader
        if loader is None:
            raise ValueError("pass an iterable of reference-schema batch dicts; dataset decoding is outside the hot path")
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("DetectionValidator needs the model on a GPU (there is no CPU path)")
        if dtype is not None:
            kops.set_compute_dtype(dtype)
        was_training = model.training
        model.eval()
        self.init_metrics(model)
        for batch in loader:
            batch = self.preprocess(dict(batch))
            preds = model(batch["img"])
            preds = self.postprocess(preds)
            self.update_metrics(preds, batch)
        stats = self.get_stats()
        model.train(was_training)
        return {k: float(v) for k, v in stats.items()}
