#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE (/root/reference) in the build container.

Fixtures are data only: seeds / small inputs and the reference's outputs.  No reference source travels.
cv2, easydict and torchvision are absent from the image and are imported by the reference only for names
(SURVEY.md 8(c)); the stand-ins below hold no arithmetic.  Parameters are produced by
`oracle.model.rng_fill` (numpy PCG64 keyed by sorted state_dict names), so tests can rebuild identical weights
without storing them.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import types
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ONLY = sys.argv[1:]
sys.argv = [sys.argv[0]]                       # filter_cfg.py parses argv at import
sys.dont_write_bytecode = True
os.environ.setdefault("YOLO_CONFIG_DIR", "/tmp/yolo_cfg_golden")
os.makedirs(os.environ["YOLO_CONFIG_DIR"], exist_ok=True)


def _install_import_stubs():
    class _Names(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return 0
    cv2 = _Names("cv2")
    cv2.setNumThreads = lambda *a, **k: None
    cv2.imshow = lambda *a, **k: None
    sys.modules["cv2"] = cv2
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.0.0"
    tv.ops = types.ModuleType("torchvision.ops")
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules.update({"torchvision": tv, "torchvision.ops": tv.ops, "torchvision.transforms": tv.transforms})


_install_import_stubs()
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from oracle.model import rng_fill  # noqa: E402
from ultralytics.nn.modules import (C2f, SPPF, Conv, Detect, AsffTribeLevel, AsffDoubLevel, AsffDetect, RFBblock, lowlight_recovery)  # noqa: E402
from ultralytics.nn.modules.filter_cfg import cfg as filter_cfg  # noqa: E402
from ultralytics.nn.tasks import DetectionModel, yaml_model_load  # noqa: E402
from ultralytics.utils.loss import BboxLoss  # noqa: E402
from ultralytics.utils.metrics import bbox_iou, box_iou, compute_ap, ap_per_class  # noqa: E402
from ultralytics.utils.tal import TaskAlignedAssigner, make_anchors, bbox2dist  # noqa: E402
from ultralytics.utils import ops as uops  # noqa: E402

torch.set_num_threads(8)
HYP = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5, lrl=2.0)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rnd(seed, *shape, lo=0.0, hi=1.0):
    g = np.random.default_rng(seed)
    return T((lo + (hi - lo) * g.random(shape, dtype=np.float32)).astype(np.float32))


def set_bn(m):
    for x in m.modules():
        if isinstance(x, nn.BatchNorm2d):
            x.eps, x.momentum = 1e-3, 0.03           # what initialize_weights does inside DetectionModel
    return m


def fill(m, seed):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(rng_fill(shapes, seed), strict=True)
    return m


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  keys={sorted(out)}")


# ------------------------------------------------------------------ G1: front-end
def g_frontend():
    m = fill(lowlight_recovery(3, 3), 101).train()
    x = rnd(1, 2, 3, 64, 96).pow(2.0).requires_grad_(True)
    wgt = rnd(2, 2, 3, 64, 96, lo=-1, hi=1)
    out = m(x)
    (out * wgt).sum().backward()
    r = torch.nn.functional.interpolate(x.detach(), size=(256, 256), mode="bilinear", align_corners=False)
    feat = m.extractor(r)
    # each filter stage for the same feat
    stages, img = [], x.detach().clone()
    A = torch.ones(2, 3) * 0.8
    IcA = torch.ones(2, 1, 64, 96) * 0.5
    for f in filter_cfg.filters:
        img, _ = f(img, feat.detach(), A, IcA)
        stages.append(img.clone())
    g = {k: p.grad for k, p in m.named_parameters()}
    # non-default A / IcA (eval call signature, tasks.py:107-110)
    A2, IcA2 = rnd(3, 2, 3, lo=0.5, hi=1.0), rnd(4, 2, 1, 64, 96, lo=0.0, hi=1.0)
    with torch.no_grad():
        out2 = m.eval()(x.detach(), A2, IcA2)
    save("g1_frontend", x=x, wgt=wgt, out=out, feat=feat, s1=stages[0][..., ::3, ::3], s2=stages[1][..., ::3, ::3],
         s3=stages[2][..., ::3, ::3], s4=stages[3][..., ::3, ::3], s5=stages[4][..., ::3, ::3], dx=x.grad, d_fc2_w=g["extractor.fc2.weight"], d_fc2_b=g["extractor.fc2.bias"],
         d_fc1_b=g["extractor.fc1.bias"], d_c0_w=g["extractor.conv_layers.0.conv_block.0.weight"],
         d_c4_b=g["extractor.conv_layers.4.conv_block.0.bias"], A2=A2, IcA2=IcA2, out2=out2, seed=101)
    # KA1-style fixed-feature chain on the survey's analytic image
    c, h, w = torch.meshgrid(torch.arange(3), torch.arange(16), torch.arange(20), indexing="ij")
    xk = (((7 * c + 3 * h + 5 * w) % 23).float() / 23)[None]
    fk = torch.linspace(-1, 1, 15)[None]
    img, sums = xk.clone(), []
    for f in filter_cfg.filters:
        img, _ = f(img, fk, torch.ones(1, 3) * 0.8, torch.ones(1, 1, 16, 20) * 0.5)
        sums.append(img.clone())
    save("g1_ka1", x=xk, feat=fk, s1=sums[0], s2=sums[1], s3=sums[2], s4=sums[3], s5=sums[4])


# ------------------------------------------------------------------ G2: blocks
def run_block(name, m, xs, seed, train=True, listin=False):
    m = fill(set_bn(m), seed)
    m.train(train)
    xs = [x.clone().requires_grad_(True) for x in xs]
    y = m(list(xs) if listin else xs[0])
    ys = y if isinstance(y, (list, tuple)) else [y]
    tot = 0
    for i, t in enumerate(ys):
        tot = tot + (t * rnd(900 + i, *t.shape, lo=-1, hi=1)).sum()
    tot.backward()
    arrs = {f"x{i}": x for i, x in enumerate(xs)}
    arrs.update({f"dx{i}": x.grad for i, x in enumerate(xs)})
    arrs.update({f"y{i}": t for i, t in enumerate(ys)})
    sd = m.state_dict()
    for k, p in m.named_parameters():
        if p.grad is not None and p.numel() <= 4096:
            arrs["g:" + k] = p.grad
        elif p.grad is not None:
            arrs["gn:" + k] = p.grad.norm()
    for k, v in sd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            arrs["b:" + k] = v
    save(name, seed=seed, **arrs)


def g_blocks():
    run_block("g2_conv_s2", Conv(16, 32, 3, 2), [rnd(10, 2, 16, 12, 20, lo=-1, hi=1)], 201)
    run_block("g2_conv_1x1", Conv(24, 16, 1, 1), [rnd(11, 2, 24, 9, 7, lo=-1, hi=1)], 202)
    run_block("g2_c2f_sc", C2f(32, 32, 2, True), [rnd(12, 2, 32, 10, 12, lo=-1, hi=1)], 203)
    run_block("g2_c2f_nosc", C2f(48, 32, 1, False), [rnd(13, 2, 48, 8, 8, lo=-1, hi=1)], 204)
    run_block("g2_sppf", SPPF(32, 32, 5), [rnd(14, 2, 32, 9, 11, lo=-1, hi=1)], 205)
    run_block("g2_rfb", RFBblock(32), [rnd(15, 2, 32, 10, 9, lo=-1, hi=1)], 206)
    asff_in = [rnd(16, 1, 512, 2, 3, lo=-1, hi=1), rnd(17, 1, 512, 4, 6, lo=-1, hi=1), rnd(18, 1, 256, 8, 12, lo=-1, hi=1)]
    for lv in range(3):
        run_block(f"g2_asff{lv}", AsffTribeLevel(lv), asff_in, 210 + lv, listin=True)
    # Detect: train maps and eval decode
    det = Detect(5, (16, 32, 32))
    det.stride = torch.tensor([8., 16., 32.])
    din = [rnd(19, 2, 16, 8, 8, lo=-1, hi=1), rnd(20, 2, 32, 4, 4, lo=-1, hi=1), rnd(21, 2, 32, 2, 2, lo=-1, hi=1)]
    run_block("g2_detect_train", det, din, 220, listin=True)
    det2 = Detect(5, (16, 32, 32))
    det2.stride = torch.tensor([8., 16., 32.])
    det2 = fill(set_bn(det2), 220).eval()
    with torch.no_grad():
        y, maps = det2([t.clone() for t in din])
    save("g2_detect_eval", seed=220, x0=din[0], x1=din[1], x2=din[2], y=y, m0=maps[0], m1=maps[1], m2=maps[2])


# ------------------------------------------------------------------ G3: whole models
def make_batch(seed, B, S, nbox):
    g = np.random.default_rng(seed)
    img = T(g.random((B, 3, S, S), dtype=np.float32))
    bi, cls, bb = [], [], []
    for b in range(B):
        for _ in range(nbox[b]):
            bi.append(b)
            cls.append(int(g.integers(0, 20)))
            cx, cy = g.uniform(0.25, 0.75, 2)
            w, h = g.uniform(0.15, 0.5, 2)
            bb.append([cx, cy, w, h])
    return dict(img=img, batch_idx=torch.tensor(bi, dtype=torch.float32), cls=torch.tensor(cls, dtype=torch.float32).view(-1, 1),
                bboxes=torch.tensor(bb, dtype=torch.float32).view(-1, 4))


def run_model(name, yaml_name, scale, seed, S, B, nbox, add_scale=None, lowlight_front=False):
    d = yaml_model_load(yaml_name)
    if add_scale:
        d["scales"][scale] = add_scale
    d["scale"] = scale
    if lowlight_front:
        # C2-style graph: stock yolov8ori with lowlight_recovery inserted as layer 0 (every absolute `from` shifts by 1)
        def sh(f):
            return f if f < 0 else f + 1
        rows = d["backbone"] + d["head"]
        rows = [[[sh(j) for j in f] if isinstance(f, list) else sh(f), n, m, a] for f, n, m, a in rows]
        d["backbone"] = [[-1, 1, "lowlight_recovery", [3]]] + rows[:len(d["backbone"])]
        d["head"] = rows[len(d["backbone"]) - 1:]
    m = DetectionModel(d, ch=3, nc=20, verbose=False)
    m.args = HYP
    fill(m, seed)
    m.train()
    batch = make_batch(seed + 1, B, S, nbox)
    batch["img"] = batch["img"].pow(3.0)
    batch["recovery_loss_batch"] = torch.tensor(0.0123)
    loss, items = m(batch)
    loss.backward()
    arrs = dict(loss=loss, items=items, seed=seed, S=S, B=B, nbox=np.array(nbox), scale_def=np.array(add_scale or [0]))
    named = dict(m.named_parameters())
    picks = [k for k in named if (k.endswith("conv.weight") or k.endswith("fc2.weight") or k.endswith(".2.bias"))
             and named[k].grad is not None]
    for k in picks[:6] + picks[-6:]:
        arrs["gn:" + k] = named[k].grad.norm()
    k0 = [k for k in named if k.endswith("bn.weight")]
    for k in (k0[0], k0[len(k0) // 2], k0[-1]):
        arrs["g:" + k] = named[k].grad
    sd = m.state_dict()
    rm = [k for k in sd if k.endswith("running_var")]
    for k in (rm[0], rm[-1]):
        arrs["b:" + k] = sd[k]
    m.eval()
    with torch.no_grad():
        y, maps = m(batch["img"])
    arrs.update(y=y[:, :, ::7], ysum=y.sum(), m2=maps[2])
    save(name, **arrs)


def g_models():
    run_model("g3_ori_tiny", "yolov8nori.yaml", "t", 301, 64, 2, [3, 1], add_scale=[0.33, 0.125, 1024])
    run_model("g3_ll_tiny", "yolov8nori.yaml", "t", 302, 64, 2, [2, 4], add_scale=[0.33, 0.125, 1024], lowlight_front=True)
    run_model("g3_repo_l", "yolov8l.yaml", "l", 303, 64, 2, [2, 3])


# ------------------------------------------------------------------ G4: assigner
def g_assigner():
    nc, B, n = 6, 3, 4
    hw = [(8, 8), (4, 4), (2, 2)]
    feats = [torch.zeros(1, 1, h, w) for h, w in hw]
    pts, st = make_anchors(feats, torch.tensor([8., 16., 32.]), 0.5)
    anc = pts * st
    A = anc.shape[0]
    g = np.random.default_rng(41)
    scores = T(g.random((B, A, nc), dtype=np.float32))
    ctr = anc[None].repeat(B, 1, 1)
    wh = T(g.uniform(4, 30, (B, A, 2)).astype(np.float32))
    jit = T(g.uniform(-6, 6, (B, A, 2)).astype(np.float32))
    boxes = torch.cat((ctr + jit - wh / 2, ctr + jit + wh / 2), -1)
    gt = torch.zeros(B, n, 4)
    lab = torch.zeros(B, n, 1)
    gt[0, 0] = torch.tensor([4., 4., 40., 44.]); lab[0, 0] = 2
    gt[0, 1] = torch.tensor([20., 10., 60., 50.]); lab[0, 1] = 5
    gt[0, 2] = torch.tensor([0., 0., 64., 64.]); lab[0, 2] = 0
    gt[1, 0] = torch.tensor([30., 30., 34., 35.]); lab[1, 0] = 1      # tiny: < 10 positive anchors -> zero-metric ties
    gt[1, 1] = torch.tensor([8., 8., 56., 56.]); lab[1, 1] = 3
    gt[1, 2] = torch.tensor([8., 8., 56., 56.]); lab[1, 2] = 3         # duplicate gt -> exact overlap ties
    gt[1, 3] = torch.tensor([40., 2., 63., 30.]); lab[1, 3] = 4
    # image 2: no gt at all
    mask = (gt.sum(2, keepdim=True) > 0).float()
    # force some exact-zero CIoU anchors inside gt: far-away predicted boxes
    boxes[0, :20] = torch.tensor([200., 200., 210., 210.])
    boxes[1, 5:9] = boxes[1, 4:5]                                       # identical predictions -> metric ties
    scores[1, 5:9] = scores[1, 4:5]
    asg = TaskAlignedAssigner(topk=10, num_classes=nc, alpha=0.5, beta=6.0)
    tl, tb, ts, fg, gi = asg(scores, boxes, anc, lab, gt, mask)
    save("g4_assigner", scores=scores, boxes=boxes, anc=anc, gt=gt, lab=lab, mask=mask, target_labels=tl, target_bboxes=tb,
         target_scores=ts, fg_mask=fg, target_gt_idx=gi, nc=nc)
    # second, larger random case at the real anchor count for 128x128
    hw = [(16, 16), (8, 8), (4, 4)]
    feats = [torch.zeros(1, 1, h, w) for h, w in hw]
    pts, st = make_anchors(feats, torch.tensor([8., 16., 32.]), 0.5)
    anc = pts * st
    A = anc.shape[0]
    nc, B, n = 20, 4, 7
    scores = T(g.random((B, A, nc), dtype=np.float32)) * 0.3
    wh = T(g.uniform(6, 70, (B, A, 2)).astype(np.float32))
    jit = T(g.uniform(-8, 8, (B, A, 2)).astype(np.float32))
    ctr = anc[None].repeat(B, 1, 1)
    boxes = torch.cat((ctr + jit - wh / 2, ctr + jit + wh / 2), -1)
    gt = torch.zeros(B, n, 4)
    lab = torch.zeros(B, n, 1)
    cnt = [7, 3, 0, 5]
    for b in range(B):
        for j in range(cnt[b]):
            c = g.uniform(25, 100, 2)
            s = g.uniform(10, 60, 2)
            gt[b, j] = torch.tensor([c[0] - s[0] / 2, c[1] - s[1] / 2, c[0] + s[0] / 2, c[1] + s[1] / 2], dtype=torch.float32)
            lab[b, j] = float(g.integers(0, nc))
    mask = (gt.sum(2, keepdim=True) > 0).float()
    asg = TaskAlignedAssigner(topk=10, num_classes=nc, alpha=0.5, beta=6.0)
    tl, tb, ts, fg, gi = asg(scores, boxes, anc, lab, gt, mask)
    save("g4_assigner_b", scores=scores, boxes=boxes, anc=anc, gt=gt, lab=lab, mask=mask, target_labels=tl, target_bboxes=tb,
         target_scores=ts, fg_mask=fg, target_gt_idx=gi, nc=nc)


# ------------------------------------------------------------------ G5: iou / dfl / ap
def g_small():
    g = np.random.default_rng(51)
    c1, s1 = g.uniform(10, 90, (64, 2)), g.uniform(1, 60, (64, 2))
    c2, s2 = c1 + g.uniform(-20, 20, (64, 2)), g.uniform(1, 60, (64, 2))
    b1 = T(np.concatenate((c1 - s1 / 2, c1 + s1 / 2), 1).astype(np.float32)).requires_grad_(True)
    b2 = T(np.concatenate((c2 - s2 / 2, c2 + s2 / 2), 1).astype(np.float32))
    v = bbox_iou(b1, b2, xywh=False, CIoU=True)
    v.sum().backward()
    iou = bbox_iou(b1.detach(), b2, xywh=False)
    pw = box_iou(b1.detach()[:8], b2[:12])
    pd = T(g.normal(0, 1.5, (40, 16)).astype(np.float32)).requires_grad_(True)
    tg = T(g.uniform(0, 14.99, (10, 4)).astype(np.float32))
    dl = BboxLoss._df_loss(pd, tg)
    dl.sum().backward()
    b2d = bbox2dist(T(np.array([[2.5, 3.5]], np.float32)), T(np.array([[0.7, 1.2, 6.9, 9.4]], np.float32)), 15)
    rec = np.array([.1, .2, .2, .4, .5, .5, .8])
    prec = np.array([1, 1, .67, .75, .8, .67, .6])
    ap, mpre, mrec = compute_ap(rec, prec)
    save("g5_small", b1=b1, b2=b2, ciou=v, dciou_db1=b1.grad, iou=iou, pairwise=pw, dfl_pred=pd, dfl_tgt=tg, dfl=dl,
         dfl_grad=pd.grad, b2d=b2d, ap=ap)
    # ap_per_class on synthetic detections
    n = 300
    tp = g.random((n, 10)) < np.linspace(0.8, 0.2, 10)[None]
    tp = np.logical_and.accumulate(tp, 1)
    conf = g.random(n)
    pcls = g.integers(0, 4, n)
    tcls = g.integers(0, 4, 120)
    res = ap_per_class(tp, conf, pcls, tcls, plot=False, names={i: str(i) for i in range(4)})
    tpc, fpc, p, r, f1, ap, uc = res[:7]
    save("g5_ap", tp=tp, conf=conf, pred_cls=pcls, target_cls=tcls, p=p, r=r, f1=f1, ap=ap, unique=uc, tpc=tpc, fpc=fpc)


def g_iou_modes():
    """bbox_iou in every mode the public helper has (metrics.py:75-128): xywh / xyxy x IoU / GIoU / DIoU / CIoU, values and the
    gradient wrt box1, on overlapping, disjoint, nested and identical pairs (g5_iou_modes.npz)."""
    g = np.random.default_rng(52)
    n = 96
    c1, s1 = g.uniform(10, 90, (n, 2)), g.uniform(1, 60, (n, 2))
    c2, s2 = c1 + g.uniform(-25, 25, (n, 2)), g.uniform(1, 60, (n, 2))
    c2[:8] = c1[:8] + 200.0                                  # disjoint
    c2[8:16], s2[8:16] = c1[8:16], s1[8:16] * 0.4            # nested, same centre
    c2[16:20], s2[16:20] = c1[16:20], s1[16:20]              # identical
    out = {}
    for xywh in (True, False):
        if xywh:
            a, b = np.concatenate((c1, s1), 1), np.concatenate((c2, s2), 1)
        else:
            a, b = np.concatenate((c1 - s1 / 2, c1 + s1 / 2), 1), np.concatenate((c2 - s2 / 2, c2 + s2 / 2), 1)
        tag = "xywh" if xywh else "xyxy"
        out[f"{tag}_b1"], out[f"{tag}_b2"] = a.astype(np.float32), b.astype(np.float32)
        for kind, kw in (("iou", {}), ("giou", dict(GIoU=True)), ("diou", dict(DIoU=True)), ("ciou", dict(CIoU=True))):
            b1 = T(a.astype(np.float32)).requires_grad_(True)
            v = bbox_iou(b1, T(b.astype(np.float32)), xywh=xywh, **kw)
            v.sum().backward()
            out[f"{tag}_{kind}"], out[f"{tag}_{kind}_grad"] = v.detach(), b1.grad
    save("g5_iou_modes", **out)


# ------------------------------------------------------------------ G6: val-side pure-torch pieces (no torchvision)
def g_val():
    g = np.random.default_rng(61)
    xywh = T(g.uniform(0, 100, (50, 4)).astype(np.float32))
    xyxy = uops.xywh2xyxy(xywh)
    back = uops.xyxy2xywh(xyxy)
    boxes = xyxy.clone()
    sc = uops.scale_boxes((640, 640), boxes.clone(), (480, 360))
    save("g6_val", xywh=xywh, xyxy=xyxy, back=back, scaled=sc)
    # DetectionValidator._process_batch (greedy one-to-one matching at 10 IoU thresholds) + the pre-NMS candidate stage of
    # non_max_suppression are pure torch/numpy; the NMS call itself needs torchvision (absent) and stays unpinned.
    from types import SimpleNamespace
    from ultralytics.models.yolo.detect.val import DetectionValidator
    me = SimpleNamespace(iouv=torch.linspace(0.5, 0.95, 10))
    nl, nd = 14, 60
    c = g.uniform(60, 580, (nl, 2))
    s = g.uniform(30, 160, (nl, 2))
    lab_box = np.concatenate((c - s / 2, c + s / 2), 1)
    lab_cls = g.integers(0, 3, (nl, 1)).astype(np.float64)
    src = g.integers(0, nl, nd)
    det_box = lab_box[src] + g.normal(0, 9, (nd, 4))
    det_box[nd // 2:] += g.normal(0, 40, (nd - nd // 2, 4))
    det_cls = np.where(g.random(nd) < 0.8, lab_cls[src, 0], g.integers(0, 3, nd))
    det = T(np.concatenate((det_box, g.random((nd, 1)), det_cls[:, None]), 1).astype(np.float32))
    lab = T(np.concatenate((lab_cls, lab_box), 1).astype(np.float32))
    correct = DetectionValidator._process_batch(me, det, lab)
    save("g6_match", det=det, lab=lab, correct=correct)


# ------------------------------------------------------------------ G2b: registry variants
def g_variants():
    din = [rnd(31, 1, 512, 3, 4, lo=-1, hi=1), rnd(32, 1, 256, 6, 8, lo=-1, hi=1)]
    for lv in range(2):
        run_block(f"g2_asff2_{lv}", AsffDoubLevel(lv), din, 230 + lv, listin=True)
    det = AsffDetect(5, (16, 32, 32))
    det.stride = torch.tensor([8., 16., 32.])
    x = [rnd(33, 2, 16, 8, 8, lo=-1, hi=1), rnd(34, 2, 32, 4, 4, lo=-1, hi=1), rnd(35, 2, 32, 2, 2, lo=-1, hi=1)]
    run_block("g2_asffdetect_train", det, x, 240, listin=True)
    det2 = AsffDetect(5, (16, 32, 32))
    det2.stride = torch.tensor([8., 16., 32.])
    det2 = fill(set_bn(det2), 240).eval()
    with torch.no_grad():
        y, maps = det2([t.clone() for t in x])
    save("g2_asffdetect_eval", seed=240, x0=x[0], x1=x[1], x2=x[2], y=y, m0=maps[0], m1=maps[1], m2=maps[2])
    # SCConv (conv.py:420-440) alone and MFRU (block.py:164-217), whose two SCConvs / pwconv are each applied twice
    from ultralytics.nn.modules.block import MFRU
    from ultralytics.nn.modules.conv import SCConv
    run_block("g2_scconv", SCConv(64), [rnd(36, 2, 64, 7, 9, lo=-1, hi=1)], 250)
    mfru_in = [rnd(37, 2, 512, 2, 3, lo=-1, hi=1), rnd(38, 2, 512, 4, 6, lo=-1, hi=1), rnd(39, 2, 256, 8, 12, lo=-1, hi=1)]
    run_block("g2_mfru", MFRU(None), mfru_in, 251, listin=True)
    # the whole yolov8-3.yaml graph at scale l: key set and parameter count of the reference model (names only, no weights)
    d = yaml_model_load("yolov8-3.yaml")
    d["scale"] = "l"
    m = DetectionModel(d, ch=3, nc=20, verbose=False)
    keys = sorted(m.state_dict().keys())
    save("g2_yolov8_3_keys", n_params=sum(p.numel() for p in m.parameters()), keys=np.array(keys),
         shapes=np.array([str(tuple(m.state_dict()[k].shape)) for k in keys]))


# ------------------------------------------------------------------ G7: a checkpoint exactly as the reference trainer writes it
def g_ckpt():
    """last.pt of ultralytics/engine/trainer.py:408-433 for a tiny model: pickled half-precision DetectionModel objects under
    'model' and 'ema' (different weights), train_args, counters.  Data only (tensors + class NAMES); the reader under test is
    dedark_yolo_amd/utils/checkpoint.py, which must rebuild the state_dicts without the reference package."""
    from copy import deepcopy
    d = yaml_model_load("yolov8nori.yaml")
    d["scales"]["t"] = [0.33, 0.0625, 1024]
    d["scale"] = "t"
    m = DetectionModel(d, ch=3, nc=4, verbose=False)
    m.args = dict(box=7.5, cls=0.5, dfl=1.5, lrl=2.0, imgsz=64)
    fill(m, 701)
    ema = deepcopy(m)
    fill(ema, 702)
    ckpt = {"epoch": 3, "best_fitness": 0.4321, "model": deepcopy(m).half(), "ema": deepcopy(ema).half(), "updates": 77,
            "optimizer": None, "train_args": dict(model="yolov8nori.yaml", imgsz=64, batch=2, lowlight_FLAG=True, dedark_FLAG=True),
            "date": "2026-01-01T00:00:00", "version": "8.0.142"}
    path = os.path.join(HERE, "g7_ref_last.pt")
    torch.save(ckpt, path)
    sd_m, sd_e = m.state_dict(), ema.state_dict()
    keys = list(sd_m.keys())
    save("g7_ckpt", n_keys=len(keys), key_first=np.array(keys[0]), key_last=np.array(keys[-1]),
         sum_model=np.array([float(sd_m[k].half().float().double().sum()) for k in keys]),
         sum_ema=np.array([float(sd_e[k].half().float().double().sum()) for k in keys]), keys=np.array(keys))
    print(f"g7_ref_last.pt: {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------ G8: preprocess_batch (tensor part) and the pre-NMS stage
def g_pre():
    """DetectionTrainer.preprocess_batch (models/yolo/detect/train.py:70-111), tensor part: img/255, x^dark_param, mse -- for the
    three flag combinations.  The numpy dark-channel branch reads np.empty memory (train.py:65-67): its outputs (dedark_A, IcA) are
    not captured.  Pre-NMS stage of non_max_suppression (utils/ops.py:196-259): torchvision.ops.nms is replaced by a recorder that
    keeps what it is called with (boxes + class offsets, scores) and keeps everything, so the candidate filter, xywh2xyxy,
    multi-label expansion, max_nms cap and the class offset are pinned; the greedy suppression itself stays unpinned."""
    from ultralytics.models.yolo.detect.train import DetectionTrainer
    g = np.random.default_rng(81)
    img = T(g.integers(0, 256, (2, 3, 24, 32), dtype=np.uint8))
    out = {}
    for tag, low, ded in (("both", True, True), ("low", True, False), ("none", False, False)):
        # (the numpy dark-channel helpers call cv2, which is absent: arithmetic-free stand-ins; their outputs are not captured)
        me = SimpleNamespace(device=torch.device("cpu"), args=SimpleNamespace(dedark_FLAG=ded, lowlight_FLAG=low, dark_param=3.5),
                             DarkChannel=lambda im: np.zeros(im.shape[:2]), AtmLight=lambda im, dk: np.zeros((1, 3)),
                             DarkIcA=lambda im, A: np.zeros(im.shape[:2]))
        b = DetectionTrainer.preprocess_batch(me, dict(img=img.clone()))
        out[f"{tag}_img"], out[f"{tag}_clean"], out[f"{tag}_rec"] = b["img"], b["clean_img"], b["recovery_loss_batch"]
    save("g8_preprocess", u8=img, dark_param=3.5, **out)

    import torchvision
    calls = []

    def rec_nms(boxes, scores, iou):
        calls.append((boxes.clone(), scores.clone(), float(iou)))
        return torch.arange(boxes.shape[0])
    torchvision.ops.nms = rec_nms
    uops.torchvision = torchvision
    nc, A, B = 5, 300, 2
    pred = torch.zeros(B, 4 + nc, A)
    pred[:, 0:2] = T(g.uniform(40, 600, (B, 2, A)).astype(np.float32))
    pred[:, 2:4] = T(g.uniform(10, 200, (B, 2, A)).astype(np.float32))
    pred[:, 4:] = T((g.random((B, nc, A)) ** 3).astype(np.float32))
    arrs = dict(pred=pred)
    for tag, kw in (("ml", dict(multi_label=True)), ("sl", dict(multi_label=False)), ("cap", dict(multi_label=True, max_nms=50)),
                    ("agn", dict(multi_label=True, agnostic=True))):
        calls.clear()
        outs = uops.non_max_suppression(pred.clone(), conf_thres=0.3, iou_thres=0.6, max_det=1000, **kw)
        for i, (bx, sc, iou) in enumerate(calls):
            arrs[f"{tag}_boxes{i}"], arrs[f"{tag}_scores{i}"] = bx, sc
        for i, o in enumerate(outs):
            arrs[f"{tag}_out{i}"] = o
        arrs[f"{tag}_ncalls"] = len(calls)
    save("g9_prenms", **arrs)


# ------------------------------------------------------------------ G10: the state a freshly constructed reference model starts from
def g_initbuf():
    """DetectionModel.__init__ (nn/tasks.py:284-292) probes the strides with TWO train-mode forward passes of zeros(1, ch, 256, 256)
    before initialize_weights() sets eps / momentum: the BatchNorm running statistics of a new reference model are therefore not
    (0, 1) but two momentum-0.1, eps-1e-5 updates on the zero image (num_batches_tracked = 2).  Captured for a tiny plain graph:
    every parameter (torch's default init under a fixed seed) and every buffer right after construction."""
    d = yaml_model_load("yolov8nori.yaml")
    d["scales"]["t"] = [0.33, 0.0625, 1024]
    d["scale"] = "t"
    torch.manual_seed(1234)
    m = DetectionModel(d, ch=3, nc=4, verbose=False)
    sd = m.state_dict()
    bn = [k for k in sd if k.endswith("num_batches_tracked")]
    assert bn and all(int(sd[k]) == 2 for k in bn)
    save("g10_initbuf", **{k: v for k, v in sd.items()})
    print("g10_initbuf:", len(sd), "tensors,", sum(v.numel() for v in sd.values()), "elements")


if __name__ == "__main__":
    which = sys.argv[1:] or None
    todo = dict(frontend=g_frontend, blocks=g_blocks, models=g_models, assigner=g_assigner, small=g_small, val=g_val, ckpt=g_ckpt,
                pre=g_pre, variants=g_variants, initbuf=g_initbuf, iou_modes=g_iou_modes)
    for k, fn in todo.items():
        if not ONLY or k in ONLY:
            fn()
