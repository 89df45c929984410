"""Detection criterion on HIP kernels.

Same classes / call signatures as the reference ultralytics/utils/loss.py (v8DetectionLoss :103-193,
RcoveryDetectionLoss :388-416) and ultralytics/utils/tal.py (TaskAlignedAssigner :59-243).  The whole criterion --
target grouping, DFL decode, task-aligned assignment, BCE / CIoU / DFL sums and the gradient wrt the three Detect maps --
runs in libdedark_yolo.so; there are no host syncs on the way (n_max is taken from the CPU-side batch_idx when available).
"""
import ctypes as C
import weakref

import torch

from .. import ops
from .._C import call
from ..ops import ld_of, ptr, stream

REG_MAX = 16


_n_max_cache = {}          # id(tensor object) -> (weakref to it, _version, bsz, n_max)


def n_max_of(batch_idx, bsz):
    """Largest number of boxes in one image (host value), the reference's `counts.max()` (loss.py:130-132).  batch_idx normally
    lives on the CPU (dataloader), where this is a host-side bincount; DevicePrefetcher / preprocess_batch compute it BEFORE the
    upload and hand it over as batch['n_max'].  A batch_idx that only exists on the device costs one synchronisation per tensor
    OBJECT: the value is remembered for that object alone (weak reference + in-place version counter), so a resident batch that is
    reused stays sync-free while a fresh upload -- which the caching allocator usually puts at the previous batch's address -- can
    never inherit another batch's value (too small an n_max would drop ground-truth boxes in dy_loss_prepare_targets)."""
    if batch_idx.numel() == 0:
        return 0
    if not batch_idx.is_cuda:
        return int(torch.bincount(batch_idx.detach().view(-1).long(), minlength=bsz).max())
    hit = _n_max_cache.get(id(batch_idx))
    if hit is not None and hit[0]() is batch_idx and hit[1] == batch_idx._version and hit[2] == bsz:
        return hit[3]
    n = int(torch.bincount(batch_idx.detach().view(-1).long(), minlength=bsz).max().item())
    if len(_n_max_cache) > 64:
        for k in [k for k, v in _n_max_cache.items() if v[0]() is None]:
            del _n_max_cache[k]
        if len(_n_max_cache) > 64:
            _n_max_cache.clear()
    _n_max_cache[id(batch_idx)] = (weakref.ref(batch_idx), batch_idx._version, bsz, n)
    return n


_n_max_of = n_max_of


class _Assignment:
    __slots__ = ("pred_boxes", "gt", "counts", "target_gt_idx", "fg_mask", "norm", "target_label", "target_box", "n_max")


def assign(maps, strides, nc, batch_idx, cls, bboxes, n_max=None, frozen=None):
    """prepare targets + decode + task-aligned assignment; returns an _Assignment of device tensors.
    `frozen` (test hook, v8DetectionLoss.frozen_assignment): an _Assignment of an earlier call on the same batch whose discrete
    outcome (tal.py:84-132: target_gt_idx, fg_mask, target labels / boxes and the normalised target score) is reused instead of
    running the assigner; the predicted boxes are still decoded from THESE maps."""
    B = maps[0].shape[0]
    dev = maps[0].device
    A = sum(m.shape[2] * m.shape[3] for m in maps)
    st = stream()
    img_h, img_w = maps[0].shape[2] * strides[0], maps[0].shape[3] * strides[0]
    if n_max is None:
        n_max = _n_max_of(batch_idx, B)
    n_t = int(batch_idx.numel())
    f32 = torch.float32
    bi = batch_idx.to(dev, f32).contiguous().view(-1)
    cl = cls.to(dev, f32).contiguous().view(-1)
    bb = bboxes.to(dev, f32).contiguous().view(-1, 4)
    a = _Assignment()
    a.n_max = n_max
    a.gt = torch.empty((B, max(n_max, 1), 5), dtype=f32, device=dev)
    a.counts = torch.empty(B, dtype=torch.int32, device=dev)
    call("dy_loss_prepare_targets", ptr(bi) if n_t else None, ptr(cl) if n_t else None, ptr(bb) if n_t else None, n_t, B,
         max(n_max, 1), float(img_w), float(img_h), ptr(a.gt), ptr(a.counts), st)
    dm = ops.det_maps(maps, strides, nc)
    a.pred_boxes = torch.empty((B, A, 4), dtype=f32, device=dev)
    call("dy_loss_decode", C.byref(dm), ptr(a.pred_boxes), st)
    if frozen is not None:
        if frozen.fg_mask.shape != (B, A) or frozen.n_max != n_max:
            raise ValueError("frozen assignment belongs to another batch / anchor grid")
        a.target_gt_idx, a.fg_mask, a.norm = frozen.target_gt_idx, frozen.fg_mask, frozen.norm
        a.target_label, a.target_box = frozen.target_label, frozen.target_box
        return a
    a.target_gt_idx = torch.empty((B, A), dtype=torch.int32, device=dev)
    a.fg_mask = torch.empty((B, A), dtype=torch.uint8, device=dev)
    a.norm = torch.empty((B, A), dtype=f32, device=dev)
    a.target_label = torch.empty((B, A), dtype=torch.int32, device=dev)
    a.target_box = torch.empty((B, A, 4), dtype=f32, device=dev)
    R = B * max(n_max, 1) * A
    work_f = torch.empty(2 * R + 2 * B * max(n_max, 1), dtype=f32, device=dev)
    work_i = torch.empty(R, dtype=torch.int32, device=dev)
    work_b = torch.empty(R, dtype=torch.uint8, device=dev)
    call("dy_tal_assign", C.byref(dm), ptr(a.pred_boxes), ptr(a.gt), ptr(a.counts), n_max, ptr(work_f), ptr(work_i),
         ptr(work_b), ptr(a.target_gt_idx), ptr(a.fg_mask), ptr(a.norm), ptr(a.target_label), ptr(a.target_box), st)
    return a


class _DetLossFn(torch.autograd.Function):
    """loss, loss_items = f(map0, map1, map2); backward writes d(loss)/d(maps) with one kernel."""

    @staticmethod
    def forward(ctx, crit, batch, n_maps, *maps):
        maps = [ops.as_nhwc(m) for m in maps]
        B = maps[0].shape[0]
        dev = maps[0].device
        st = stream()
        strides = crit.strides_as_floats()[:n_maps]
        a = assign(maps, strides, crit.nc, batch["batch_idx"], batch["cls"], batch["bboxes"], batch.get("n_max"),
                   frozen=crit.frozen_assignment)
        dm = ops.det_maps(maps, strides, crit.nc)
        acc = torch.zeros(4, dtype=torch.float64, device=dev)
        call("dy_loss_fwd", C.byref(dm), ptr(a.pred_boxes), ptr(a.fg_mask), ptr(a.norm), ptr(a.target_label), ptr(a.target_box),
             ptr(acc), st)
        rec = batch.get("recovery_loss_batch") if crit.use_recovery else None
        if rec is not None:
            rec = rec.detach().to(dev, torch.float32).reshape(-1)
            rec = rec.mean().reshape(1) if rec.numel() > 1 else rec
        out = torch.empty(4, dtype=torch.float32, device=dev)
        call("dy_loss_finish", ptr(acc), ptr(rec), float(crit.hyp.box), float(crit.hyp.cls), float(crit.hyp.dfl),
             float(getattr(crit.hyp, "lrl", 0.0)), B, ptr(out[0:1]), ptr(out[1:4]), st)
        ctx.crit, ctx.maps, ctx.assign, ctx.acc, ctx.strides = crit, maps, a, acc, strides
        crit.last_assignment = a
        if crit.keep_maps:
            crit.last_maps = maps
        loss, items = out[0], out[1:4]
        ctx.mark_non_differentiable(items)
        return loss, items

    @staticmethod
    def backward(ctx, gloss, _gitems):
        crit, maps, a = ctx.crit, ctx.maps, ctx.assign
        dev = maps[0].device
        dt = maps[0].dtype
        ve = ops.vec_elems(dt)
        dm = ops.det_maps(maps, ctx.strides, crit.nc)
        width = 4 * REG_MAX + ops.round_up(crit.nc, ve)
        dbufs = [ops.empty_nhwc(m.shape[0], width, m.shape[2], m.shape[3], dt, dev) for m in maps]
        arr_p = (C.c_void_p * 3)(*[d.data_ptr() for d in dbufs] + [None] * (3 - len(dbufs)))
        arr_l = (C.c_int64 * 3)(*[ld_of(d) for d in dbufs] + [0] * (3 - len(dbufs)))
        g = gloss.detach().to(torch.float32).reshape(1).contiguous()
        call("dy_loss_bwd", C.byref(dm), arr_p, arr_l, ptr(a.pred_boxes), ptr(a.fg_mask), ptr(a.norm), ptr(a.target_label),
             ptr(a.target_box), ptr(ctx.acc), ptr(g), float(crit.hyp.box), float(crit.hyp.cls), float(crit.hyp.dfl), stream())
        ops.emu_round(*dbufs)
        no = 4 * REG_MAX + crit.nc
        return (None, None, None, *[d[:, :no] for d in dbufs])


class _DFL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_dist, target):
        n = target.shape[0]
        out = torch.empty((n, 1), dtype=torch.float32, device=pred_dist.device)
        grad = torch.empty_like(pred_dist) if pred_dist.requires_grad else None
        call("dy_dfl_loss", ptr(pred_dist), ptr(target), n, ptr(out), ptr(grad), stream())
        ctx.grad = grad
        return out

    @staticmethod
    def backward(ctx, g):
        return ctx.grad * g.repeat_interleave(4, 0), None


class BboxLoss:
    """The stand-alone pieces of the reference's BboxLoss (ultralytics/utils/loss.py:54-84); the training step itself computes both
    terms inside the fused dy_loss_fwd / dy_loss_bwd kernels with the same device functions (csrc/dy_lossmath.h)."""

    def __init__(self, reg_max, use_dfl=False):
        self.reg_max, self.use_dfl = reg_max, use_dfl

    @staticmethod
    def _df_loss(pred_dist, target):
        """pred_dist [n*4, 16] logits, target [n, 4] in [0, 15) -> [n, 1] (mean over the sides of the two-bin cross entropy)."""
        if pred_dist.device.type != "cuda":
            raise RuntimeError("_df_loss needs device tensors (there is no CPU path)")
        if pred_dist.shape[-1] != REG_MAX or pred_dist.shape[0] != target.numel():
            raise ValueError(f"_df_loss: expected pred_dist [n*4, {REG_MAX}] and target [n, 4]")
        return _DFL.apply(pred_dist.float().contiguous(), target.float().contiguous().detach())


class v8DetectionLoss:
    """reference loss.py:103-193. `model.args` must carry .box/.cls/.dfl (and .lrl for the recovery variant)."""
    use_recovery = False

    def __init__(self, model):
        m = model.model[-1]
        self.hyp = model.args
        self.stride = m.stride
        self.nc = m.nc
        self.no = m.no
        self.reg_max = m.reg_max
        self.device = next(model.parameters()).device
        self.use_dfl = m.reg_max > 1
        self.assigner = TaskAlignedAssigner(topk=10, num_classes=self.nc, alpha=0.5, beta=6.0)
        self.last_assignment = None
        self.frozen_assignment = None        # test hooks: reuse an earlier assignment / keep the Detect maps of the last call
        self.keep_maps, self.last_maps = False, None

    def strides_as_floats(self):
        """Detect.stride as host floats, read back once (a per-step float(tensor) is a device synchronisation)."""
        key = (id(self.stride), self.stride._version)
        if getattr(self, "_stride_key", None) != key:
            self._stride_host = [float(s) for s in self.stride.detach().cpu()]
            self._stride_key = key
        return self._stride_host

    def __call__(self, preds, batch):
        feats = preds[1] if isinstance(preds, tuple) else preds
        loss, items = _DetLossFn.apply(self, batch, len(feats), *feats)
        return loss, items


class RcoveryDetectionLoss(v8DetectionLoss):
    """reference loss.py:388-416: adds lrl * recovery_loss_batch to the total and to the cls item."""
    use_recovery = True

    def __init__(self, model):
        super().__init__(model)
        self.recovery_weight = self.hyp.lrl


class TaskAlignedAssigner:
    """reference tal.py:59-243 (topk must be 10, alpha 0.5, beta 6.0: the constants compiled into the kernel).

    `forward(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt)` is the reference's own call (tal.py:84-132) on the
    HIP assigner (dy_tal_assign_decoded); the criterion itself goes from the raw Detect maps (`assign_from_maps`, no decoded
    [B,A,nc] score tensor is ever materialised there)."""

    def __init__(self, topk=13, num_classes=80, alpha=1.0, beta=6.0, eps=1e-9):
        if (topk, alpha, beta) != (10, 0.5, 6.0):
            raise NotImplementedError("HIP TaskAlignedAssigner is built for topk=10, alpha=0.5, beta=6.0 (loss.py:120)")
        self.topk, self.num_classes, self.alpha, self.beta, self.eps = topk, num_classes, alpha, beta, eps
        self.bg_idx = num_classes

    def assign_from_maps(self, maps, strides, batch_idx, cls, bboxes, n_max=None):
        return assign([ops.as_nhwc(m) for m in maps], [float(s) for s in strides], self.num_classes, batch_idx, cls, bboxes, n_max)

    @torch.no_grad()
    def forward(self, pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt):
        """Returns (target_labels [B,A] int64, target_bboxes [B,A,4], target_scores [B,A,nc], fg_mask [B,A] bool,
        target_gt_idx [B,A] int64) like tal.py:84-132."""
        ops.require_gpu(pd_scores)
        dev = pd_scores.device
        f32 = torch.float32
        B, A, nc = pd_scores.shape
        n = gt_bboxes.shape[1]
        if n == 0:                                            # tal.py:106-110
            return (torch.full((B, A), float(self.bg_idx), dtype=pd_scores.dtype, device=dev), torch.zeros_like(pd_bboxes),
                    torch.zeros_like(pd_scores), torch.zeros((B, A), dtype=pd_scores.dtype, device=dev),
                    torch.zeros((B, A), dtype=pd_scores.dtype, device=dev))
        sc = pd_scores.detach().to(dev, f32).contiguous()
        bx = pd_bboxes.detach().to(dev, f32).contiguous()
        an = anc_points.detach().to(dev, f32).contiguous()
        mk = mask_gt.to(dev, f32).reshape(B, n, 1)
        gt = torch.cat((gt_labels.to(dev, f32).reshape(B, n, 1), gt_bboxes.to(dev, f32) * mk), 2).contiguous()    # masked rows: zero box
        counts = torch.full((B,), n, dtype=torch.int32, device=dev)
        st = stream()
        tgi = torch.empty((B, A), dtype=torch.int32, device=dev)
        fg = torch.empty((B, A), dtype=torch.uint8, device=dev)
        norm = torch.empty((B, A), dtype=f32, device=dev)
        tl = torch.empty((B, A), dtype=torch.int32, device=dev)
        tb = torch.empty((B, A, 4), dtype=f32, device=dev)
        R = B * n * A
        work_f = torch.empty(2 * R + 2 * B * n, dtype=f32, device=dev)
        work_i = torch.empty(R, dtype=torch.int32, device=dev)
        work_b = torch.empty(R, dtype=torch.uint8, device=dev)
        call("dy_tal_assign_decoded", ptr(sc), ptr(bx), ptr(an), ptr(gt), ptr(counts), B, A, nc, n, ptr(work_f), ptr(work_i), ptr(work_b),
             ptr(tgi), ptr(fg), ptr(norm), ptr(tl), ptr(tb), st)
        fgb = fg.bool()
        labels = tl.long().clamp_(min=0)
        scores = torch.zeros((B, A, nc), dtype=f32, device=dev)
        scores.scatter_(2, labels.clamp(max=nc - 1).unsqueeze(-1), (norm * fgb).unsqueeze(-1))
        return labels, tb, scores, fgb, tgi.long()

    __call__ = forward
