"""Detection metrics behind the reference's names (ultralytics/utils/metrics.py: box_iou :52-72, compute_ap :418-448,
ap_per_class :451-554 without the plots, Metric :557-688, DetMetrics :691-801).  Host-side numpy bookkeeping that runs once per
validation epoch (the GPU side ends at the matched-prediction table of engine/validator.py).

Organisation (own design, checked against the reference's values in tests/golden/g5_ap.npz and KA3 of SURVEY.md):
  * `class_curves` sorts the detections once (confidence, then a stable grouping by class), builds the cumulative TP / FP counts of
    ALL classes and IoU thresholds with one segmented cumulative sum, and samples precision / recall per class on a fixed grid;
  * `envelope_area` integrates the monotone precision envelope on the 101-point grid for every IoU threshold of a class at once;
  * `BoxSummary` stores the per-class arrays and derives every reported number from one table of reductions.
"""
import numpy as np
import torch

_CONF_GRID = np.linspace(0.0, 1.0, 1000)       # confidence axis of the P / R / F1 curves
_REC_GRID = np.linspace(0.0, 1.0, 101)         # recall axis of the AP integral (COCO-style 101 points)
_trapezoid = getattr(np, "trapezoid", None) or np.trapz


def box_iou(box1, box2, eps=1e-7):
    """Pairwise IoU [N, M] of xyxy boxes."""
    lo = torch.maximum(box1[:, None, :2], box2[None, :, :2])
    hi = torch.minimum(box1[:, None, 2:], box2[None, :, 2:])
    inter = (hi - lo).clamp(min=0).prod(-1)
    area1 = (box1[:, 2:] - box1[:, :2]).prod(-1)[:, None]
    area2 = (box2[:, 2:] - box2[:, :2]).prod(-1)[None, :]
    return inter / (area1 + area2 - inter + eps)


class _IoUPairs(torch.autograd.Function):
    """IoU / GIoU / DIoU / CIoU of n box pairs through dy_bbox_iou; gradient wrt the first boxes (the second ones are targets)."""

    @staticmethod
    def forward(ctx, b1, b2, xywh, kind, eps):
        from .._C import call
        from ..ops import ptr, stream
        n = b1.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=b1.device)
        grad = torch.empty((n, 4), dtype=torch.float32, device=b1.device) if b1.requires_grad else None
        if kind == 3 and not xywh and eps == 1e-7:
            call("dy_bbox_ciou", ptr(b1), ptr(b2), n, ptr(out), ptr(grad), stream())        # the entry the training path's kernels share
        else:
            call("dy_bbox_iou", ptr(b1), ptr(b2), n, int(bool(xywh)), kind, float(eps), ptr(out), ptr(grad), stream())
        ctx.grad = grad
        return out

    @staticmethod
    def backward(ctx, g):
        return ctx.grad * g[:, None], None, None, None, None


def bbox_iou(box1, box2, xywh=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """Reference signature and flag precedence (ultralytics/utils/metrics.py:75-128): CIoU over DIoU over GIoU over plain IoU,
    (cx, cy, w, h) or xyxy boxes, broadcastable leading shapes [..., 4] -> [..., 1].  The training path calls
    (xywh=False, CIoU=True) (loss.py:71, tal.py:158).  Device tensors only; the gradient flows to box1 (box2 is a target)."""
    if box1.device.type != "cuda":
        raise RuntimeError("bbox_iou needs device tensors (there is no CPU path)")
    kind = 3 if CIoU else 2 if DIoU else 1 if GIoU else 0
    box1, box2 = torch.broadcast_tensors(box1, box2)
    shape = box1.shape[:-1]
    b1 = box1.reshape(-1, 4).float().contiguous()
    b2 = box2.reshape(-1, 4).float().contiguous().detach()
    return _IoUPairs.apply(b1, b2, bool(xywh), kind, float(eps)).reshape(*shape, 1)


def box_filter(y, frac=0.05):
    """Moving average over ~frac of the samples (odd window, edge values repeated): the smoothing applied to the mean F1 curve."""
    taps = round(len(y) * frac * 2) // 2 + 1
    half = taps // 2
    padded = np.concatenate((np.full(half, y[0]), y, np.full(half, y[-1])))
    return np.convolve(padded, np.full(taps, 1.0 / taps), mode="valid")


smooth = box_filter


def envelope_area(recall, precision):
    """AP of one class for every IoU threshold at once.  recall / precision: [n, T] cumulative curves (detections by falling
    confidence).  Adds the (0, 1) and (1, 0) end points, replaces precision by its running maximum from the right (the envelope) and
    integrates it over the 101 recall samples with the trapezoid rule.  Returns (ap [T], envelope [n+2, T], recall [n+2, T])."""
    recall = np.atleast_2d(recall.T).T if recall.ndim == 1 else recall
    precision = np.atleast_2d(precision.T).T if precision.ndim == 1 else precision
    T = recall.shape[1]
    rec = np.vstack((np.zeros((1, T)), recall, np.ones((1, T))))
    env = np.vstack((np.ones((1, T)), precision, np.zeros((1, T))))
    env = np.maximum.accumulate(env[::-1], axis=0)[::-1]
    ap = np.array([_trapezoid(np.interp(_REC_GRID, rec[:, t], env[:, t]), _REC_GRID) for t in range(T)])
    return ap, env, rec


def compute_ap(recall, precision):
    """Reference signature (metrics.py:418-448) for one curve: (ap, precision envelope, recall with end points)."""
    ap, env, rec = envelope_area(np.asarray(recall, dtype=np.float64)[:, None], np.asarray(precision, dtype=np.float64)[:, None])
    return ap[0], env[:, 0], rec[:, 0]


def class_curves(tp, conf, pred_cls, target_cls, eps=1e-16):
    """tp [n, T] bool/0-1 (prediction matched at IoU threshold t), conf [n], pred_cls [n], target_cls [m].
    Returns dict(classes, n_labels, ap [C, T], p_curve [C, 1000], r_curve [C, 1000]) over the classes that have labels."""
    by_conf = np.argsort(-conf)
    tp, conf, pred_cls = np.asarray(tp)[by_conf].astype(np.float64), conf[by_conf], pred_cls[by_conf]
    classes, n_labels = np.unique(target_cls, return_counts=True)
    C, T = classes.shape[0], tp.shape[1]
    ap = np.zeros((C, T))
    p_curve, r_curve = np.zeros((C, _CONF_GRID.size)), np.zeros((C, _CONF_GRID.size))
    if tp.shape[0]:
        # one segmented cumulative sum for all classes: group the confidence-sorted rows by class (stable), cumsum, subtract the
        # running total at each segment start
        grp = np.argsort(pred_cls, kind="stable")
        g_cls, g_tp, g_conf = pred_cls[grp], tp[grp], conf[grp]
        starts = np.concatenate(([0], np.flatnonzero(g_cls[1:] != g_cls[:-1]) + 1))
        ends = np.concatenate((starts[1:], [g_cls.shape[0]]))
        tot_tp = np.cumsum(g_tp, axis=0)
        tot_fp = np.cumsum(1.0 - g_tp, axis=0)
        seg_of = {g_cls[s]: (s, e) for s, e in zip(starts, ends)}
        for ci, c in enumerate(classes):
            if c not in seg_of or n_labels[ci] == 0:
                continue
            s, e = seg_of[c]
            base_tp = tot_tp[s - 1] if s else 0.0
            base_fp = tot_fp[s - 1] if s else 0.0
            tpc, fpc = tot_tp[s:e] - base_tp, tot_fp[s:e] - base_fp
            recall = tpc / (n_labels[ci] + eps)
            precision = tpc / (tpc + fpc)
            neg_conf = -g_conf[s:e]
            r_curve[ci] = np.interp(-_CONF_GRID, neg_conf, recall[:, 0], left=0)       # curves are reported at IoU 0.5
            p_curve[ci] = np.interp(-_CONF_GRID, neg_conf, precision[:, 0], left=1)
            ap[ci] = envelope_area(recall, precision)[0]
    return dict(classes=classes.astype(int), n_labels=n_labels, ap=ap, p_curve=p_curve, r_curve=r_curve)


def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, on_plot=None, save_dir=None, names=(), eps=1e-16, prefix=""):
    """Reference return order (tp, fp, p, r, f1, ap, unique_classes); plotting arguments are accepted and ignored.  P / R / F1 are
    read at the confidence that maximises the smoothed mean F1 over classes."""
    cur = class_curves(tp, conf, pred_cls, target_cls, eps)
    p, r = cur["p_curve"], cur["r_curve"]
    f1 = 2 * p * r / (p + r + eps)
    best = int(box_filter(f1.mean(0), 0.1).argmax())
    p, r, f1 = p[:, best], r[:, best], f1[:, best]
    n_tp = (r * cur["n_labels"]).round()
    n_fp = (n_tp / (p + eps) - n_tp).round()
    return n_tp, n_fp, p, r, f1, cur["ap"], cur["classes"]


class BoxSummary:
    """Per-class precision / recall / F1 / AP table and everything derived from it (the reference's `Metric`)."""

    # name -> reduction over (p, r, all_ap); all_ap is [classes, 10 IoU thresholds 0.50:0.05:0.95]
    _REDUCE = dict(
        ap50=lambda s: s.all_ap[:, 0], ap=lambda s: s.all_ap.mean(1),
        mp=lambda s: float(np.mean(s.p)), mr=lambda s: float(np.mean(s.r)),
        map50=lambda s: float(s.all_ap[:, 0].mean()), map75=lambda s: float(s.all_ap[:, 5].mean()), map=lambda s: float(s.all_ap.mean()))
    _EMPTY = dict(ap50=[], ap=[], mp=0.0, mr=0.0, map50=0.0, map75=0.0, map=0.0)
    FITNESS_WEIGHTS = np.array([0.0, 0.0, 0.1, 0.9])          # P, R, mAP@0.5, mAP@0.5:0.95

    def __init__(self):
        self.p, self.r, self.f1, self.all_ap, self.ap_class_index, self.nc = [], [], [], [], [], 0

    def __getattr__(self, name):
        red = type(self)._REDUCE.get(name)
        if red is None:
            raise AttributeError(name)
        return red(self) if len(self.all_ap) else type(self)._EMPTY[name]

    def update(self, results):
        self.p, self.r, self.f1, self.all_ap, self.ap_class_index = results

    def mean_results(self):
        return [self.mp, self.mr, self.map50, self.map]

    def class_result(self, i):
        return self.p[i], self.r[i], self.ap50[i], self.ap[i]

    @property
    def maps(self):
        """mAP@0.5:0.95 of every class index 0..nc-1 (classes without labels get the overall mean)."""
        out = np.full(self.nc, self.map, dtype=np.float64)
        if len(self.ap_class_index):
            out[np.asarray(self.ap_class_index, dtype=int)] = self.ap
        return out

    def fitness(self):
        return float(np.dot(np.array(self.mean_results()), self.FITNESS_WEIGHTS))


Metric = BoxSummary


class DetMetrics:
    """The validator's metric object (reference DetMetrics :691-801): `.process(tp, conf, pred_cls, target_cls)` then
    `.results_dict`, `.fitness`, `.maps`, `.mean_results()`, `.class_result(i)`, `.ap_class_index`, `.keys`, `.speed`."""
    keys = ["metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)"]

    def __init__(self, save_dir=None, plot=False, on_plot=None, names=()):
        self.save_dir, self.plot, self.on_plot, self.names = save_dir, plot, on_plot, names
        self.box = BoxSummary()
        self.speed = dict(preprocess=0.0, inference=0.0, loss=0.0, postprocess=0.0)

    def process(self, tp, conf, pred_cls, target_cls):
        self.box.nc = len(self.names)
        self.box.update(ap_per_class(tp, conf, pred_cls, target_cls, names=self.names)[2:])

    def __getattr__(self, name):            # mean_results / class_result / maps / ap_class_index live on the box table
        if name in ("mean_results", "class_result", "maps", "ap_class_index"):
            return getattr(self.box, name)
        raise AttributeError(name)

    @property
    def fitness(self):
        return self.box.fitness()

    @property
    def results_dict(self):
        return dict(zip(self.keys + ["fitness"], self.box.mean_results() + [self.fitness]))
