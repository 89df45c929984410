"""Oracle: validation post-processing and mAP.  TEST INFRASTRUCTURE.

`nms` restates torchvision.ops.nms's published CPU kernel in the input dtype (f32 on the reference's call path); torchvision
is NOT vendored by the reference, its version is unpinned and it is absent from the image: **parity unpinned** for the greedy
stage (no build to run against); everything else here is golden-pinned.
"""
import numpy as np
import torch


def xywh2xyxy(x):
    """U/utils/ops.py:374-389."""
    y = torch.empty_like(x)
    half_w, half_h = x[..., 2] / 2, x[..., 3] / 2
    y[..., 0] = x[..., 0] - half_w
    y[..., 1] = x[..., 1] - half_h
    y[..., 2] = x[..., 0] + half_w
    y[..., 3] = x[..., 1] + half_h
    return y


def xyxy2xywh(x):
    """U/utils/ops.py:357-371."""
    y = torch.empty_like(x)
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def scale_boxes(shape1, boxes, shape0):
    """U/utils/ops.py:95-125 (ratio_pad=None, padding=True) + clip_boxes (:281-297). In place, like the reference."""
    gain = min(shape1[0] / shape0[0], shape1[1] / shape0[1])
    pad_x = round((shape1[1] - shape0[1] * gain) / 2 - 0.1)
    pad_y = round((shape1[0] - shape0[0] * gain) / 2 - 0.1)
    boxes[..., [0, 2]] -= pad_x
    boxes[..., [1, 3]] -= pad_y
    boxes[..., :4] /= gain
    boxes[..., 0].clamp_(0, shape0[1])
    boxes[..., 1].clamp_(0, shape0[0])
    boxes[..., 2].clamp_(0, shape0[1])
    boxes[..., 3].clamp_(0, shape0[0])
    return boxes


def box_iou(b1, b2, eps=1e-7):
    """U/utils/metrics.py:52-72: pairwise IoU [N,M] of xyxy boxes."""
    a1, a2 = b1[:, None, :2], b1[:, None, 2:]
    c1, c2 = b2[None, :, :2], b2[None, :, 2:]
    inter = (torch.minimum(a2, c2) - torch.maximum(a1, c1)).clamp(min=0).prod(2)
    return inter / ((a2 - a1).prod(2) + (c2 - c1).prod(2) - inter + eps)


def nms(boxes, scores, thr):
    """Greedy NMS with the arithmetic of torchvision.ops.nms's CPU kernel (`nms_kernel_impl<scalar_t>`, torchvision/csrc/ops/cpu/
    nms_kernel.cpp, called from U/utils/ops.py:261 with f32 boxes that already carry the cls * 7680 offset).  torchvision is not
    vendored by the reference and absent here, so this follows the PUBLISHED kernel, operation by operation, in the INPUT dtype:
      areas = (x2 - x1) * (y2 - y1)                      (tensor op, scalar_t)
      order = scores.sort(stable, descending)
      for each live i in order: keep it; for every later live j:
          w = max(0, min(ix2, x2[j]) - max(ix1, x1[j])); h likewise; inter = w * h        (scalar_t, no FMA)
          ovr = inter / (iarea + areas[j] - inter)        (scalar_t, left to right)
          suppressed[j] = ovr > iou_threshold             (the only widening: scalar_t quotient vs the double threshold)
    Returns kept indices in score order (int64).  **Unpinned** in the sense of SURVEY 8(c): no torchvision build exists here to run
    against; the arithmetic above is what the published source states."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    order = torch.argsort(scores, descending=True, stable=True)
    ft = np.float64 if boxes.dtype == torch.float64 else np.float32
    b = boxes[order].detach().cpu().numpy().astype(ft, copy=False)
    x1, y1, x2, y2 = (np.ascontiguousarray(b[:, k]) for k in range(4))
    area = (x2 - x1) * (y2 - y1)
    thr = np.float64(thr)
    zero = ft(0)
    dead = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            w = np.maximum(zero, np.minimum(x2[i], x2[i + 1:]) - np.maximum(x1[i], x1[i + 1:]))
            h = np.maximum(zero, np.minimum(y2[i], y2[i + 1:]) - np.maximum(y1[i], y1[i + 1:]))
            inter = w * h
            with np.errstate(invalid="ignore", divide="ignore"):
                ovr = inter / ((area[i] + area[i + 1:]) - inter)
            assert ovr.dtype == ft
            dead[i + 1:] |= ovr.astype(np.float64) > thr
    return order[torch.tensor(keep, dtype=torch.int64)]


def non_max_suppression(pred, conf_thres=0.25, iou_thres=0.7, multi_label=True, max_det=300, max_nms=30000, max_wh=7680):
    """U/utils/ops.py:144-278 for the detect validator's call (U/models/yolo/detect/val.py:62-70):
    pred [B, 4+nc, A] (xywh px, sigmoid scores) -> list of [n,6] (xyxy, conf, cls).  The wall-clock break
    (:274-276) is a nondeterminism hazard and is not restated."""
    bs, nc = pred.shape[0], pred.shape[1] - 4
    cand = pred[:, 4:].amax(1) > conf_thres
    pred = pred.transpose(-1, -2).clone()
    pred[..., :4] = xywh2xyxy(pred[..., :4])
    multi_label = multi_label and nc > 1
    out = [torch.zeros((0, 6))] * bs
    for xi in range(bs):
        x = pred[xi][cand[xi]]
        if not x.shape[0]:
            continue
        box, cls = x[:, :4], x[:, 4:]
        if multi_label:
            i, j = torch.where(cls > conf_thres)
            x = torch.cat((box[i], x[i, 4 + j, None], j[:, None].float()), 1)
        else:
            conf, j = cls.max(1, keepdim=True)
            x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
        if not x.shape[0]:
            continue
        if x.shape[0] > max_nms:
            x = x[x[:, 4].argsort(descending=True)[:max_nms]]
        off = x[:, 5:6] * max_wh
        keep = nms(x[:, :4] + off, x[:, 4], iou_thres)[:max_det]
        out[xi] = x[keep]
    return out


def match_predictions(detections, labels, iouv):
    """DetectionValidator._process_batch (U/models/yolo/detect/val.py:151-174): detections [N,6] (xyxy,conf,cls),
    labels [M,5] (cls,xyxy) -> correct [N, len(iouv)] bool; greedy one-to-one by IoU per threshold."""
    iou = box_iou(labels[:, 1:], detections[:, :4])
    correct = np.zeros((detections.shape[0], len(iouv)), dtype=bool)
    same = labels[:, 0:1] == detections[:, 5]
    for t, thr in enumerate(iouv):
        li, di = torch.where((iou >= thr) & same)
        if li.numel():
            m = torch.cat((torch.stack((li, di), 1).float(), iou[li, di][:, None]), 1).numpy()
            if li.numel() > 1:
                m = m[m[:, 2].argsort()[::-1]]
                m = m[np.unique(m[:, 1], return_index=True)[1]]
                m = m[np.unique(m[:, 0], return_index=True)[1]]
            correct[m[:, 1].astype(int), t] = True
    return torch.from_numpy(correct)


def compute_ap(recall, precision):
    """U/utils/metrics.py:418-448: 101-point interpolated AP of the precision envelope."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    trapz = getattr(np, "trapezoid", None) or np.trapz
    return trapz(np.interp(x, mrec, mpre), x), mpre, mrec


def _smooth(y, f=0.05):
    """U/utils/metrics.py:320-325."""
    nf = round(len(y) * f * 2) // 2 + 1
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """U/utils/metrics.py:451-554 without plotting. Returns dict(tp, fp, p, r, f1, ap, classes)."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, nt = np.unique(target_cls, return_counts=True)
    nc = classes.shape[0]
    px = np.linspace(0, 1, 1000)
    ap, p, r = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        n_l, n_p = nt[ci], sel.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[sel]).cumsum(0)
        tpc = tp[sel].cumsum(0)
        recall = tpc / (n_l + eps)
        r[ci] = np.interp(-px, -conf[sel], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p[ci] = np.interp(-px, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]
    f1 = 2 * p * r / (p + r + eps)
    i = _smooth(f1.mean(0), 0.1).argmax()
    p, r, f1 = p[:, i], r[:, i], f1[:, i]
    tpn = (r * nt).round()
    fpn = (tpn / (p + eps) - tpn).round()
    return dict(tp=tpn, fp=fpn, p=p, r=r, f1=f1, ap=ap, classes=classes.astype(int))


def fitness(ap):
    """Metric.fitness (U/utils/metrics.py:~640): 0.1*mAP50 + 0.9*mAP50-95 over classes."""
    return 0.1 * ap[:, 0].mean() + 0.9 * ap.mean()
