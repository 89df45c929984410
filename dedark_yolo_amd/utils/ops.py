"""Box utilities and batched NMS with the call surface of the reference's ultralytics/utils/ops.py
(xywh2xyxy :374-389, xyxy2xywh :357-371, clip_boxes :281-297, scale_boxes :95-125, non_max_suppression :144-278).

non_max_suppression runs the whole batch in three HIP launches (candidate keys -> segmented radix sort -> greedy scan,
csrc/nms.hip) instead of the reference's per-image Python loop around torchvision.ops.nms; there is no CPU fallback."""
import ctypes as C

import torch

from .._C import call
from ..ops import ptr, stream


def xywh2xyxy(x):
    y = torch.empty_like(x)
    dw, dh = x[..., 2] / 2, x[..., 3] / 2
    y[..., 0] = x[..., 0] - dw
    y[..., 1] = x[..., 1] - dh
    y[..., 2] = x[..., 0] + dw
    y[..., 3] = x[..., 1] + dh
    return y


def xyxy2xywh(x):
    y = torch.empty_like(x)
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def clip_boxes(boxes, shape):
    """in place, like the reference (ops.py:281-297)"""
    boxes[..., 0].clamp_(0, shape[1])
    boxes[..., 1].clamp_(0, shape[0])
    boxes[..., 2].clamp_(0, shape[1])
    boxes[..., 3].clamp_(0, shape[0])
    return boxes


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None, padding=True):
    """Rescale xyxy boxes from the network input shape to the original image shape, in place (ops.py:95-125)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1), round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    if padding:
        boxes[..., [0, 2]] -= pad[0]
        boxes[..., [1, 3]] -= pad[1]
    boxes[..., :4] /= gain
    return clip_boxes(boxes, img0_shape)


_ws = {}


def _workspace(key, nbytes, device):
    t = _ws.get(key)
    if t is None or t.numel() < nbytes or t.device != device:
        t = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        _ws[key] = t
    return t


def nms_batched(pred, conf_thres, iou_thres, multi_label, agnostic, max_det, max_nms, max_wh, return_indices=False):
    """pred [B, 4+nc, A] f32 (Detect eval output) -> (out [B,max_det,6], counts [B] int32[, keep_idx [B,max_det] int64]).
    keep_idx = anchor*nc + cls of every kept row (the integer output parity tests compare bit-exactly)."""
    if pred.dtype != torch.float32 or not pred.is_cuda:
        raise RuntimeError("nms_batched expects the f32 device tensor produced by Detect in eval mode")
    pred = pred.contiguous()
    B, no, A = pred.shape
    nc = no - 4
    dev = pred.device
    cap = A * nc if (multi_label and nc > 1) else A
    st = stream()
    keys = torch.empty((2, B, cap), dtype=torch.int64, device=dev)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    call("dy_nms_candidates", ptr(pred), B, nc, A, float(conf_thres), int(bool(multi_label)), ptr(keys[0]), ptr(counts), cap, st)
    nbytes = C.c_size_t(0)
    call("dy_nms_sort", ptr(keys[0]), ptr(keys[1]), ptr(counts), B, cap, None, C.addressof(nbytes), st)
    ws = _workspace("sort", nbytes.value, dev)
    nbytes = C.c_size_t(ws.numel())
    call("dy_nms_sort", ptr(keys[0]), ptr(keys[1]), ptr(counts), B, cap, ptr(ws), C.addressof(nbytes), st)
    n_eff = min(cap, max_nms)
    boxes_ws = torch.empty((B, n_eff, 4), dtype=torch.float32, device=dev)
    dead_ws = torch.empty((B, n_eff), dtype=torch.uint8, device=dev)
    out = torch.zeros((B, max_det, 6), dtype=torch.float32, device=dev)
    keep = torch.full((B, max_det), -1, dtype=torch.int64, device=dev)
    ocnt = torch.empty(B, dtype=torch.int32, device=dev)
    call("dy_nms_greedy", ptr(pred), ptr(keys[1]), ptr(counts), B, nc, A, cap, float(iou_thres), n_eff, max_det, float(max_wh),
         int(bool(agnostic)), ptr(boxes_ws), ptr(dead_ws), ptr(out), ptr(keep), ptr(ocnt), st)
    return (out, ocnt, keep) if return_indices else (out, ocnt)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, labels=(),
                        max_det=300, nc=0, max_time_img=0.05, max_nms=30000, max_wh=7680):
    """Reference signature (ops.py:144-156).  Returns a list with one [n,6] tensor (xyxy, conf, cls) per image.
    `max_time_img` is accepted and ignored: the wall-clock break of the reference (:274-276) makes its output depend on
    machine load."""
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if classes is not None or (labels is not None and len(labels)):
        raise NotImplementedError("class filtering / hybrid autolabelling are outside the Dedark-YOLO hot path")
    if nc not in (0, prediction.shape[1] - 4):
        raise NotImplementedError("mask coefficients (segment task) are outside the Dedark-YOLO hot path")
    out, cnt = nms_batched(prediction, conf_thres, iou_thres, multi_label, agnostic, max_det, max_nms, max_wh)
    cnt = cnt.tolist()
    return [out[i, :cnt[i]] for i in range(len(cnt))]
