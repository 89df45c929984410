"""Oracle: RcoveryDetectionLoss + TaskAlignedAssigner.  TEST INFRASTRUCTURE.

Functional fp32 restatement on CPU tensors.  Integer outputs (target_gt_idx, fg_mask) rely on the same
`torch.topk` / `argmax` CPU tie-breaking as the reference (U/utils/tal.py:180, :45, :53).
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from .model import REG_MAX, make_anchors

TOPK, ALPHA, BETA, TAL_EPS = 10, 0.5, 6.0, 1e-9     # U/utils/loss.py:120 ; tal.py:72


def ciou(b1, b2, eps=1e-7):
    """bbox_iou(xywh=False, CIoU=True) (U/utils/metrics.py:75-128). b1, b2: [...,4] xyxy -> [...,1].
    Quirk kept: eps is added to h only (metrics.py:102-103); alpha is computed under no_grad (:122-123)."""
    x1, y1, x2, y2 = b1.unbind(-1)
    X1, Y1, X2, Y2 = b2.unbind(-1)
    w1, h1 = x2 - x1, y2 - y1 + eps
    w2, h2 = X2 - X1, Y2 - Y1 + eps
    inter = (torch.minimum(x2, X2) - torch.maximum(x1, X1)).clamp(min=0) * \
            (torch.minimum(y2, Y2) - torch.maximum(y1, Y1)).clamp(min=0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.maximum(x2, X2) - torch.minimum(x1, X1)
    chh = torch.maximum(y2, Y2) - torch.minimum(y1, Y1)
    c2 = cw ** 2 + chh ** 2 + eps
    rho2 = ((X1 + X2 - x1 - x2) ** 2 + (Y1 + Y2 - y1 - y2) ** 2) / 4
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
    with torch.no_grad():
        a = v / (v - iou + (1 + eps))
    return (iou - (rho2 / c2 + v * a)).unsqueeze(-1)


def bbox_iou(b1, b2, xywh=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """Every mode of bbox_iou (U/utils/metrics.py:75-128); [...,4] x [...,4] -> [...,1].  xywh: w, h as given (no eps, :95-99);
    xyxy: eps on h only (:100-104).  Flag precedence CIoU > DIoU > GIoU > IoU (:113-127)."""
    if xywh:
        (x, y, w1, h1), (X, Y, w2, h2) = b1.unbind(-1), b2.unbind(-1)
        x1, x2, y1, y2 = x - w1 / 2, x + w1 / 2, y - h1 / 2, y + h1 / 2
        X1, X2, Y1, Y2 = X - w2 / 2, X + w2 / 2, Y - h2 / 2, Y + h2 / 2
    else:
        x1, y1, x2, y2 = b1.unbind(-1)
        X1, Y1, X2, Y2 = b2.unbind(-1)
        w1, h1 = x2 - x1, y2 - y1 + eps
        w2, h2 = X2 - X1, Y2 - Y1 + eps
    inter = (torch.minimum(x2, X2) - torch.maximum(x1, X1)).clamp(min=0) * \
            (torch.minimum(y2, Y2) - torch.maximum(y1, Y1)).clamp(min=0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if not (CIoU or DIoU or GIoU):
        return iou.unsqueeze(-1)
    cw = torch.maximum(x2, X2) - torch.minimum(x1, X1)
    chh = torch.maximum(y2, Y2) - torch.minimum(y1, Y1)
    if CIoU or DIoU:
        c2 = cw ** 2 + chh ** 2 + eps
        rho2 = ((X1 + X2 - x1 - x2) ** 2 + (Y1 + Y2 - y1 - y2) ** 2) / 4
        if CIoU:
            v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
            with torch.no_grad():
                a = v / (v - iou + (1 + eps))
            return (iou - (rho2 / c2 + v * a)).unsqueeze(-1)
        return (iou - rho2 / c2).unsqueeze(-1)
    c_area = cw * chh + eps
    return (iou - (c_area - union) / c_area).unsqueeze(-1)


def pad_targets(batch_idx, cls, bboxes, bsz, img_wh):
    """v8DetectionLoss.preprocess (loss.py:124-139): group rows per image (order kept), zero-pad to the max
    count, scale normalised xywh by (w,h,w,h) and convert to xyxy.  Returns [B, n_max, 5] (cls, x1,y1,x2,y2)."""
    n = batch_idx.numel()
    if n == 0:
        return torch.zeros(bsz, 0, 5)
    bi = batch_idx.view(-1).long()
    counts = torch.bincount(bi, minlength=bsz)
    out = torch.zeros(bsz, int(counts.max()), 5)
    rows = torch.cat((cls.view(-1, 1).float(), bboxes.float()), 1)
    for j in range(bsz):
        sel = rows[bi == j]
        out[j, :sel.shape[0]] = sel
    w, h = img_wh
    xywh = out[..., 1:5] * torch.tensor([w, h, w, h], dtype=torch.float32)
    xy, half = xywh[..., :2], xywh[..., 2:] / 2                      # xywh2xyxy (U/utils/ops.py:374-389)
    out[..., 1:5] = torch.cat((xy - half, xy + half), -1)
    return out


def tal_assign(scores, boxes, anchors, gt_labels, gt_boxes, mask_gt, nc):
    """TaskAlignedAssigner.forward (tal.py:83-127), topk=10, alpha=.5, beta=6.

    scores [B,A,nc] (sigmoid), boxes [B,A,4] xyxy px, anchors [A,2] px, gt_labels [B,n,1], gt_boxes [B,n,4],
    mask_gt [B,n,1] -> target_labels[B,A] i64, target_bboxes[B,A,4], target_scores[B,A,nc], fg_mask[B,A] bool,
    target_gt_idx[B,A] i64.
    """
    B, A, _ = scores.shape
    n = gt_boxes.shape[1]
    if n == 0:                                                          # tal.py:106-110
        return (torch.full((B, A), nc, dtype=torch.float32), torch.zeros(B, A, 4), torch.zeros(B, A, nc),
                torch.zeros(B, A, dtype=torch.bool), torch.zeros(B, A, dtype=torch.int64))
    # anchor centre strictly inside gt (tal.py:12-28)
    lt, rb = gt_boxes[:, :, None, :2], gt_boxes[:, :, None, 2:]
    d = torch.cat((anchors[None, None] - lt, rb - anchors[None, None]), -1)      # [B,n,A,4]
    in_gt = (d.amin(-1) > TAL_EPS).float()
    valid = (in_gt * mask_gt).bool()                                             # [B,n,A]
    # metric (tal.py:141-160)
    lab = gt_labels.squeeze(-1).long()                                           # [B,n]
    sc = scores.permute(0, 2, 1)                                                 # [B,nc,A]
    gt_sc = torch.gather(sc, 1, lab.clamp(0, nc - 1)[:, :, None].expand(-1, -1, A))
    bbox_scores = torch.where(valid, gt_sc, torch.zeros(()))
    ov = ciou(gt_boxes[:, :, None, :].expand(-1, -1, A, -1), boxes[:, None].expand(-1, n, -1, -1)).squeeze(-1)
    overlaps = torch.where(valid, ov.clamp(min=0), torch.zeros(()))
    align = bbox_scores.pow(ALPHA) * overlaps.pow(BETA)
    # top-k per gt (tal.py:162-196)
    _, idx = torch.topk(align, TOPK, dim=-1, largest=True)
    idx = idx.masked_fill(~mask_gt.bool().expand(-1, -1, TOPK), 0)
    cnt = torch.zeros(B, n, A, dtype=torch.int32)
    cnt.scatter_add_(-1, idx, torch.ones_like(idx, dtype=torch.int32))
    cnt = cnt.masked_fill(cnt > 1, 0)
    mask_pos = cnt.float() * in_gt * mask_gt
    # anchors claimed by several gts keep the one with max CIoU (tal.py:31-56)
    fg = mask_pos.sum(-2)
    if fg.max() > 1:
        multi = (fg[:, None] > 1).expand(-1, n, -1)
        best = overlaps.argmax(1)
        onehot = torch.zeros_like(mask_pos).scatter_(1, best[:, None], 1.0)
        mask_pos = torch.where(multi, onehot, mask_pos)
        fg = mask_pos.sum(-2)
    gt_idx = mask_pos.argmax(-2)                                                 # [B,A]
    # targets (tal.py:198-243)
    flat = gt_idx + torch.arange(B)[:, None] * n
    t_labels = gt_labels.long().flatten()[flat].clamp(min=0)
    t_boxes = gt_boxes.reshape(-1, 4)[flat]
    t_scores = F.one_hot(t_labels, nc).float() * (fg > 0)[..., None]
    # normalise (tal.py:120-125)
    align = align * mask_pos
    pos_align = align.amax(-1, keepdim=True)
    pos_ov = (overlaps * mask_pos).amax(-1, keepdim=True)
    norm = (align * pos_ov / (pos_align + TAL_EPS)).amax(-2).unsqueeze(-1)
    return t_labels, t_boxes, t_scores * norm, fg.bool(), gt_idx


def decode_boxes(pred_dist, anchors):
    """bbox_decode (loss.py:141-146): softmax over 16 bins . arange -> ltrb -> xyxy (grid units)."""
    b, a, _ = pred_dist.shape
    d = pred_dist.view(b, a, 4, REG_MAX).softmax(3).matmul(torch.arange(REG_MAX, dtype=pred_dist.dtype))
    return torch.cat((anchors - d[..., :2], anchors + d[..., 2:]), -1)


def dfl_loss(pred, target):
    """BboxLoss._df_loss (loss.py:75-84). pred [N*4... ,16] logits rows, target [N,4] in [0, 14.99]."""
    tl = target.long()
    tr = tl + 1
    wl = tr - target
    wr = 1 - wl
    left = F.cross_entropy(pred, tl.view(-1), reduction="none").view(tl.shape)
    right = F.cross_entropy(pred, tr.view(-1), reduction="none").view(tl.shape)
    return (left * wl + right * wr).mean(-1, keepdim=True)


def detection_loss(maps, batch, strides, nc, hyp, details=False):
    """v8DetectionLoss.__call__ (loss.py:148-193).  maps: 3x[B, 64+nc, h, w]; batch: dict(batch_idx, cls, bboxes).
    Returns (loss_sum * B, loss_items[3].detach()); with details=True also the assigner outputs."""
    B = maps[0].shape[0]
    no = 4 * REG_MAX + nc
    cat = torch.cat([m.view(B, no, -1) for m in maps], 2)
    pred_dist = cat[:, :4 * REG_MAX].permute(0, 2, 1).contiguous()
    pred_scores = cat[:, 4 * REG_MAX:].permute(0, 2, 1).contiguous()
    img_h, img_w = maps[0].shape[2] * strides[0], maps[0].shape[3] * strides[0]
    anchors, stride_t = make_anchors([m.shape[2:] for m in maps], strides)

    tg = pad_targets(batch["batch_idx"], batch["cls"], batch["bboxes"], B, (img_w, img_h))
    gt_labels, gt_boxes = tg[..., :1], tg[..., 1:]
    mask_gt = (gt_boxes.sum(2, keepdim=True) > 0).float()

    pred_boxes = decode_boxes(pred_dist, anchors)
    with torch.no_grad():
        t_labels, t_boxes, t_scores, fg, gt_idx = tal_assign(
            pred_scores.detach().sigmoid(), pred_boxes.detach() * stride_t, anchors * stride_t,
            gt_labels, gt_boxes, mask_gt, nc)
    tss = max(t_scores.sum(), 1)

    loss = torch.zeros(3)
    loss[1] = F.binary_cross_entropy_with_logits(pred_scores, t_scores, reduction="none").sum() / tss
    if fg.sum():
        t_boxes = t_boxes / stride_t
        weight = t_scores.sum(-1)[fg].unsqueeze(-1)
        iou = ciou(pred_boxes[fg], t_boxes[fg])
        loss[0] = ((1.0 - iou) * weight).sum() / tss
        ltrb = torch.cat((anchors - t_boxes[..., :2], t_boxes[..., 2:] - anchors), -1).clamp(0, REG_MAX - 1 - 0.01)
        l_dfl = dfl_loss(pred_dist[fg].view(-1, REG_MAX), ltrb[fg]) * weight
        loss[2] = l_dfl.sum() / tss
    loss = loss * torch.tensor([hyp.box, hyp.cls, hyp.dfl])
    out = (loss.sum() * B, loss.detach())
    if details:
        return out + (dict(target_gt_idx=gt_idx, fg_mask=fg, target_scores=t_scores, target_bboxes=t_boxes,
                           target_labels=t_labels, pred_boxes=pred_boxes.detach()),)
    return out


def recovery_detection_loss(maps, batch, strides, nc, hyp):
    """RcoveryDetectionLoss.__call__ (loss.py:393-416): adds lrl * recovery to the total and to the cls item."""
    loss, items = detection_loss(maps, batch, strides, nc, hyp)
    rec = batch.get("recovery_loss_batch")
    box, cls, dfl = items
    if rec is not None:
        rec = rec.mean() if rec.ndim > 0 else rec
        cls = cls + hyp.lrl * rec
        loss = loss + hyp.lrl * rec
    return loss, torch.stack([box.detach(), cls.detach(), dfl.detach()])


def default_hyp():
    """box/cls/dfl/lrl gains of U/cfg/default.yaml:92-95."""
    return SimpleNamespace(box=7.5, cls=0.5, dfl=1.5, lrl=2.0)


def preprocess_batch(img_u8, dark_param=15.0, lowlight=True, dedark=True):
    """Tensor part of DetectionTrainer.preprocess_batch (U/models/yolo/detect/train.py:70-111).
    The numpy dark-channel branch (:81-97) is dead downstream and reads uninitialised memory; it is
    deliberately not restated.  Returns (img, clean_img, recovery_loss)."""
    clean = img_u8.float() / 255
    if dedark and lowlight:
        clean = torch.pow(clean, dark_param)
        img = clean
    elif lowlight:
        img = torch.pow(clean, dark_param)
    else:
        img = clean
    return img, clean, F.mse_loss(img, clean)
