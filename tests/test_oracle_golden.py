"""CPU: pin the oracle (our restatement) against golden vectors captured from the reference
(tests/golden/make_golden.py) and the known-answer values of SURVEY.md Appendix A."""
import numpy as np
import pytest
import torch

from oracle import frontend as ofe
from oracle import loss as oloss
from oracle import model as om
from oracle import val as oval
from util import close, gold, load_yaml, make_batch, rnd


# ------------------------------------------------------------------ front-end
def test_ka1_filter_chain_known_answers():
    """SURVEY Appendix A KA1 table (values quoted from the reference run)."""
    c, h, w = torch.meshgrid(torch.arange(3), torch.arange(16), torch.arange(20), indexing="ij")
    x = (((7 * c + 3 * h + 5 * w) % 23).float() / 23)[None]
    feat = torch.linspace(-1, 1, 15)[None]
    out, st, p = ofe.filter_chain(x, feat, stages=True)
    assert abs(float(x.sum()) - 458.913055) < 1e-3
    assert abs(float(p["omega"]) - 0.2072826) < 1e-6
    close(p["wb"][0], torch.tensor([1.2354821, 0.9091753, 0.9543331]), 1e-6, 1e-6, "wb")
    assert abs(float(p["gamma"]) - 0.6414791) < 1e-6
    assert abs(float(p["alpha"]) - 0.6947827) < 1e-6
    assert abs(float(p["lam"]) - 4.4039855) < 1e-5
    sums = [423.17493, 437.10339, 533.00995, 513.07953, 518.42413]
    v000 = [-0.0924999, -0.1142819, 0.0027170, 0.0019966, -2.6847267]
    v257 = [0.7805978, 0.7449504, 0.8278894, 0.8423490, 2.3631115]
    for s, sm, a, b in zip(st, sums, v000, v257):
        assert abs(float(s.sum()) - sm) < 2e-3
        assert abs(float(s[0, 0, 0, 0]) - a) < 2e-5
        assert abs(float(s[0, 2, 5, 7]) - b) < 2e-5
    k = ofe.gaussian_taps()
    close(k[[0, 1, 2, 12]], torch.tensor([0.0045345617, 0.0071830819, 0.0109323747, 0.0807799324]), 1e-6, 1e-9, "taps")
    g = gold("g1_ka1")
    for i, s in enumerate(st):
        close(s, g[f"s{i + 1}"], 1e-5, 1e-6, f"ka1 stage {i + 1}")


def test_frontend_golden_forward_backward():
    g = gold("g1_frontend")
    shapes = {k[len("model.0."):]: v for k, v in om.param_shapes([dict(i=0, kind="lowlight_recovery")]).items()}
    sd = om.rng_fill(shapes, int(g["seed"]))
    for v in sd.values():
        v.requires_grad_(v.is_floating_point())
    x = g["x"].clone().requires_grad_(True)
    out, feat, st, _ = ofe.lowlight_recovery(sd, "", x, stages=True)
    close(feat, g["feat"], 1e-4, 1e-5, "feat")
    for i, s in enumerate(st):
        close(s[..., ::3, ::3], g[f"s{i + 1}"], 1e-4, 1e-4, f"stage {i + 1}")
    close(out, g["out"], 1e-4, 1e-4, "out")
    (out * g["wgt"]).sum().backward()
    close(x.grad, g["dx"], 1e-3, 1e-3, "dx")
    close(sd["extractor.fc2.weight"].grad, g["d_fc2_w"], 1e-3, 1e-2, "d fc2.w")
    close(sd["extractor.fc2.bias"].grad, g["d_fc2_b"], 1e-3, 1e-2, "d fc2.b")
    close(sd["extractor.fc1.bias"].grad, g["d_fc1_b"], 1e-3, 1e-2, "d fc1.b")
    close(sd["extractor.conv_layers.0.conv_block.0.weight"].grad, g["d_c0_w"], 2e-3, 2e-2, "d conv0.w")
    close(sd["extractor.conv_layers.4.conv_block.0.bias"].grad, g["d_c4_b"], 2e-3, 2e-2, "d conv4.b")
    with torch.no_grad():
        out2 = ofe.lowlight_recovery(sd, "", g["x"], g["A2"], g["IcA2"])
    close(out2, g["out2"], 1e-4, 1e-4, "out (A, IcA given)")


# ------------------------------------------------------------------ blocks
def _block(name, fn, shapes, nin=1):
    g = gold(name)
    sd = om.rng_fill(shapes, int(g["seed"]))
    for k, v in sd.items():
        v.requires_grad_(v.is_floating_point() and v.ndim > 0 and "running_" not in k)
    xs = [g[f"x{i}"].clone().requires_grad_(True) for i in range(nin)]
    y = fn(sd, xs)
    ys = y if isinstance(y, (list, tuple)) else [y]
    tot = 0
    for i, t in enumerate(ys):
        close(t, g[f"y{i}"], 1e-4, 1e-4, f"{name} y{i}")
        tot = tot + (t * rnd(900 + i, *t.shape, lo=-1, hi=1)).sum()
    tot.backward()
    for i, x in enumerate(xs):
        close(x.grad, g[f"dx{i}"], 1e-3, 1e-3, f"{name} dx{i}")
    for k, v in g.items():
        if k.startswith("g:"):
            close(sd[k[2:]].grad, v, 2e-3, 2e-3, f"{name} {k}")
        elif k.startswith("gn:"):
            close(sd[k[3:]].grad.norm(), v, 1e-3, 1e-4, f"{name} {k}")
        elif k.startswith("b:"):
            close(sd[k[2:]].detach(), v, 1e-4, 1e-5, f"{name} {k}")


def _shapes(spec):
    spec = dict(i=0, **spec)
    return {k[len("model.0."):]: v for k, v in om.param_shapes([spec]).items()}


def test_block_goldens():
    _block("g2_conv_s2", lambda sd, x: om.conv_bn_silu(sd, "", x[0], 3, 2, True), _shapes(dict(kind="Conv", c1=16, c2=32, k=3)))
    _block("g2_conv_1x1", lambda sd, x: om.conv_bn_silu(sd, "", x[0], 1, 1, True), _shapes(dict(kind="Conv", c1=24, c2=16, k=1)))
    _block("g2_c2f_sc", lambda sd, x: om.c2f(sd, "", x[0], 2, True, True), _shapes(dict(kind="C2f", c1=32, c2=32, n=2)))
    _block("g2_c2f_nosc", lambda sd, x: om.c2f(sd, "", x[0], 1, False, True), _shapes(dict(kind="C2f", c1=48, c2=32, n=1)))
    _block("g2_sppf", lambda sd, x: om.sppf(sd, "", x[0], 5, True), _shapes(dict(kind="SPPF", c1=32, c2=32)))
    _block("g2_rfb", lambda sd, x: om.rfb(sd, "", x[0]), _shapes(dict(kind="RFBblock", c1=32)))


@pytest.mark.parametrize("level", [0, 1, 2])
def test_asff_golden(level):
    _block(f"g2_asff{level}", lambda sd, x: om.asff(sd, "", x, level, True),
           _shapes(dict(kind="AsffTribeLevel", level=level)), nin=3)


@pytest.mark.parametrize("level", [0, 1])
def test_asff_two_level_golden(level):
    _block(f"g2_asff2_{level}", lambda sd, x: om.asff2(sd, "", x, level, True), _shapes(dict(kind="AsffDoubLevel", level=level)), nin=2)


def test_scconv_and_mfru_goldens():
    """SCConv (conv.py:420-440) alone, and MFRU (block.py:164-217) whose SCConvs / pwconv are applied twice each (shared weights)."""
    _block("g2_scconv", lambda sd, x: om.scconv(sd, "", x[0]), _shapes(dict(kind="SCConv", c=64)))
    _block("g2_mfru", lambda sd, x: om.mfru(sd, "", x), _shapes(dict(kind="MFRU")), nin=3)


def test_yolov8_3_graph_keys_and_parameter_count():
    """cfg/models/v8/yolov8-3.yaml at scale l: state_dict key set, shapes and parameter count of the reference model."""
    g = gold("g2_yolov8_3_keys")
    plan, save = om.build_plan(load_yaml("yolov8-3.yaml"), scale="l", nc=20)
    shapes = om.param_shapes(plan)
    want = {str(k): str(v) for k, v in zip(g["keys"], g["shapes"])}
    assert set(shapes) == set(want), sorted(set(shapes) ^ set(want))[:6]
    assert all(str(tuple(shapes[k])) == want[k] for k in shapes)
    n = sum(int(np.prod(v)) for k, v in shapes.items() if "running_" not in k and "num_batches" not in k and ".dfl." not in k)
    assert n == int(g["n_params"]) - 16          # the DFL conv weight (16 constants) is a frozen parameter of the reference model


def test_asff_detect_goldens():
    sh = _shapes(dict(kind="AsffDetect", nc=5, ch=[16, 32, 32]))
    _block("g2_asffdetect_train", lambda sd, x: om.asff_detect(sd, "", x, 5, [8., 16., 32.], True), sh, nin=3)
    g = gold("g2_asffdetect_eval")
    sd = om.rng_fill(sh, int(g["seed"]))
    with torch.no_grad():
        y, maps = om.asff_detect(sd, "", [g["x0"], g["x1"], g["x2"]], 5, [8., 16., 32.], False)
    close(y, g["y"], 1e-4, 1e-4, "asffdetect eval y")
    for i, m in enumerate(maps):
        close(m, g[f"m{i}"], 1e-4, 1e-4, f"asffdetect eval map{i}")


def test_detect_goldens():
    sh = _shapes(dict(kind="Detect", nc=5, ch=[16, 32, 32]))
    _block("g2_detect_train", lambda sd, x: om.detect(sd, "", x, 5, [8., 16., 32.], True), sh, nin=3)
    g = gold("g2_detect_eval")
    sd = om.rng_fill(sh, int(g["seed"]))
    with torch.no_grad():
        y, maps = om.detect(sd, "", [g["x0"], g["x1"], g["x2"]], 5, [8., 16., 32.], False)
    close(y, g["y"], 1e-4, 1e-4, "detect eval y")
    for i, m in enumerate(maps):
        close(m, g[f"m{i}"], 1e-4, 1e-4, f"detect eval map{i}")


# ------------------------------------------------------------------ whole models
def _model(name, yaml_name, scale):
    g = gold(name)
    cfg = load_yaml(yaml_name)
    if g["scale_def"].numel() == 3:
        cfg["scales"][scale] = [float(v) for v in g["scale_def"]]
    plan, save = om.build_plan(cfg, scale=scale, nc=20)
    sd = om.rng_fill(om.param_shapes(plan), int(g["seed"]))
    for k, v in sd.items():
        v.requires_grad_(v.is_floating_point() and v.ndim > 0 and ".dfl." not in k and "running_" not in k)
    B, S = int(g["B"]), int(g["S"])
    batch = make_batch(int(g["seed"]) + 1, B, S, [int(v) for v in g["nbox"]])
    batch["img"] = batch["img"].pow(3.0)
    batch["recovery_loss_batch"] = torch.tensor(0.0123)
    maps = om.forward(plan, save, sd, batch["img"], True)
    strides = [float(S // m.shape[2]) for m in maps]
    loss, items = oloss.recovery_detection_loss(maps, batch, strides, 20, oloss.default_hyp())
    close(loss, g["loss"], 1e-4, 1e-4, f"{name} loss")
    close(items, g["items"], 1e-4, 1e-4, f"{name} loss_items")
    loss.backward()
    for k, v in g.items():
        if k.startswith("gn:"):
            close(sd[k[3:]].grad.norm(), v, 2e-3, 1e-4, f"{name} {k}")
        elif k.startswith("g:"):
            close(sd[k[2:]].grad, v, 5e-3, 2e-3, f"{name} {k}")
        elif k.startswith("b:"):
            close(sd[k[2:]].detach(), v, 1e-4, 1e-5, f"{name} {k}")
    with torch.no_grad():
        y, maps = om.forward(plan, save, sd, batch["img"], False)
    close(y[:, :, ::7], g["y"], 1e-3, 1e-3, f"{name} eval y")
    close(y.sum(), g["ysum"], 1e-4, 1e-2, f"{name} eval ysum")
    close(maps[2], g["m2"], 1e-3, 1e-3, f"{name} eval map2")


def test_model_ori_tiny():
    _model("g3_ori_tiny", "yolov8ori.yaml", "t")


def test_model_lowlight_tiny():
    _model("g3_ll_tiny", "yolov8-lowlight.yaml", "t")


def test_model_repo_l():
    _model("g3_repo_l", "yolov8.yaml", "l")


def test_param_counts_match_survey():
    """SURVEY 3.2 / BASELINE.md: 51,776,780 params for repo yolov8.yaml@L nc=20; 3,014,748 for yolov8ori@n."""
    for yml, sc, n in (("yolov8.yaml", "l", 51776780), ("yolov8ori.yaml", "n", 3014748)):
        plan, _ = om.build_plan(load_yaml(yml), scale=sc, nc=20)
        tot = sum(int(np.prod(s)) for k, s in om.param_shapes(plan).items()
                  if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
        assert tot == n, (yml, tot)


# ------------------------------------------------------------------ assigner (integer outputs bit-exact)
@pytest.mark.parametrize("name", ["g4_assigner", "g4_assigner_b"])
def test_assigner_golden(name):
    g = gold(name)
    tl, tb, ts, fg, gi = oloss.tal_assign(g["scores"], g["boxes"], g["anc"], g["lab"], g["gt"], g["mask"], int(g["nc"]))
    assert torch.equal(gi, g["target_gt_idx"])
    assert torch.equal(fg, g["fg_mask"].bool())
    assert torch.equal(tl.long(), g["target_labels"].long())
    close(tb, g["target_bboxes"], 0, 0, "target_bboxes")
    close(ts, g["target_scores"], 1e-6, 1e-7, "target_scores")


# ------------------------------------------------------------------ small functions (KA2-KA5)
def test_small_known_answers():
    b1 = torch.tensor([[10., 20, 110, 220], [0, 0, 10, 10], [5, 5, 15, 15]])
    b2 = torch.tensor([[30., 40, 100, 260], [0, 0, 10, 10], [20, 20, 30, 40]])
    close(oloss.ciou(b1, b2).squeeze(-1), torch.tensor([0.53873754, 1.0, -0.33952731]), 1e-6, 1e-6, "KA2 ciou")
    close(oval.box_iou(b1, b2).diag(), torch.tensor([0.55263156, 1.0, 0.0]), 1e-6, 1e-7, "KA2 iou")
    ap, _, _ = oval.compute_ap(np.array([.1, .2, .2, .4, .5, .5, .8]), np.array([1, 1, .67, .75, .8, .67, .6]))
    assert abs(ap - 0.68885) < 1e-8
    anchors = torch.tensor([[2.5, 3.5]])
    box = torch.tensor([[0.7, 1.2, 6.9, 9.4]])
    ltrb = torch.cat((anchors - box[:, :2], box[:, 2:] - anchors), -1).clamp(0, 14.99)
    close(ltrb, torch.tensor([[1.8, 2.3, 4.4, 5.9]]), 1e-6, 1e-6, "KA4 bbox2dist")
    close(oloss.dfl_loss(torch.linspace(-2, 2, 64).view(4, 16), ltrb), torch.tensor([[3.0626757]]), 1e-6, 1e-6, "KA4 dfl")
    pts, st = om.make_anchors([(2, 3), (1, 2)], [8, 16])
    assert pts.tolist() == [[.5, .5], [1.5, .5], [2.5, .5], [.5, 1.5], [1.5, 1.5], [2.5, 1.5], [.5, .5], [1.5, .5]]
    assert st.view(-1).tolist() == [8, 8, 8, 8, 8, 8, 16, 16]


def test_small_goldens():
    g = gold("g5_small")
    b1 = g["b1"].clone().requires_grad_(True)
    v = oloss.ciou(b1, g["b2"])
    close(v, g["ciou"], 1e-5, 1e-6, "ciou")
    v.sum().backward()
    close(b1.grad, g["dciou_db1"], 1e-4, 1e-6, "dciou")
    close(oval.box_iou(g["b1"][:8], g["b2"][:12]), g["pairwise"], 1e-6, 1e-7, "pairwise iou")
    pd = g["dfl_pred"].clone().requires_grad_(True)
    dl = oloss.dfl_loss(pd, g["dfl_tgt"])
    close(dl, g["dfl"], 1e-5, 1e-6, "dfl")
    dl.sum().backward()
    close(pd.grad, g["dfl_grad"], 1e-5, 1e-6, "dfl grad")
    assert abs(float(g["ap"]) - 0.68885) < 1e-8


def test_ap_per_class_golden():
    g = gold("g5_ap")
    r = oval.ap_per_class(g["tp"].numpy().astype(bool), g["conf"].numpy(), g["pred_cls"].numpy(), g["target_cls"].numpy())
    close(r["ap"], g["ap"], 1e-9, 1e-12, "ap")
    close(r["p"], g["p"], 1e-9, 1e-12, "p")
    close(r["r"], g["r"], 1e-9, 1e-12, "r")
    close(r["f1"], g["f1"], 1e-9, 1e-12, "f1")
    close(r["tp"], g["tpc"], 0, 0.5, "tp count")
    close(r["fp"], g["fpc"], 0, 0.5, "fp count")


def test_val_box_helpers_golden():
    g = gold("g6_val")
    close(oval.xywh2xyxy(g["xywh"]), g["xyxy"], 0, 0, "xywh2xyxy")
    close(oval.xyxy2xywh(g["xyxy"]), g["back"], 0, 0, "xyxy2xywh")
    close(oval.scale_boxes((640, 640), g["xyxy"].clone(), (480, 360)), g["scaled"], 0, 1e-6, "scale_boxes")


def test_bbox_iou_every_mode_golden():
    """bbox_iou(xywh / xyxy, IoU / GIoU / DIoU / CIoU) values and d/d box1 captured from the reference (g5_iou_modes.npz)."""
    from oracle import loss as oloss
    g = gold("g5_iou_modes")
    for tag, xywh in (("xywh", True), ("xyxy", False)):
        for kind, kw in (("iou", {}), ("giou", dict(GIoU=True)), ("diou", dict(DIoU=True)), ("ciou", dict(CIoU=True))):
            b1 = g[f"{tag}_b1"].clone().requires_grad_(True)
            v = oloss.bbox_iou(b1, g[f"{tag}_b2"], xywh=xywh, **kw)
            v.sum().backward()
            close(v.detach(), g[f"{tag}_{kind}"], 1e-6, 1e-7, f"{tag} {kind}")
            close(b1.grad, g[f"{tag}_{kind}_grad"], 1e-5, 1e-7, f"{tag} {kind} grad")


def test_nms_unpinned_semantics():
    """torchvision is absent: check the restated greedy NMS on a hand case (kept order = score order)."""
    boxes = torch.tensor([[0., 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95])
    assert oval.nms(boxes, scores, 0.5).tolist() == [3, 2]
    assert oval.nms(boxes, scores, 0.99).tolist() == [3, 0, 1, 2]


def test_nms_arithmetic_is_f32_like_torchvision():
    """g11_nms_borderline.npz (tests/golden/make_nms_borderline.py): 24 (keeper, victim) pairs per image whose IoU on the class-offset
    f32 boxes lies on opposite sides of 0.7 in f32 and in f64 arithmetic, 12 each way.  torchvision's CPU kernel computes in the
    input dtype (f32, U/utils/ops.py:261), so the oracle must suppress exactly the `n32` victims -- and an f64 evaluation of the
    same boxes must suppress the OTHER 12 (the fixture separates the two rules)."""
    g = gold("g11_nms_borderline")
    for k in (0, 1):
        pred, n32, n64 = g[f"pred{k}"], int(g[f"n32_{k}"]), int(g[f"n64_{k}"])
        pairs = pred.shape[2] // 2
        got = oval.non_max_suppression(pred, 0.25, 0.7)[0]
        assert got.shape[0] == 2 * pairs - n32
        orig = oval.nms
        try:
            oval.nms = lambda b, s, t: orig(b.double(), s, t)
            got64 = oval.non_max_suppression(pred, 0.25, 0.7)[0]
        finally:
            oval.nms = orig
        assert got64.shape[0] == 2 * pairs - n64
        a, b = set(map(tuple, got.tolist())), set(map(tuple, got64.tolist()))
        assert len(a - b) == n64 and len(b - a) == n32


def test_match_predictions_golden():
    """DetectionValidator._process_batch (val.py:151-174) captured from the reference."""
    g = gold("g6_match")
    got = oval.match_predictions(g["det"], g["lab"], torch.linspace(0.5, 0.95, 10))
    assert torch.equal(got, g["correct"].bool())


def test_preprocess_batch_golden():
    """DetectionTrainer.preprocess_batch, tensor part (detect/train.py:70-111) captured from the reference for the three flag
    combinations (g8_preprocess.npz)."""
    from oracle import loss as oloss
    g = gold("g8_preprocess")
    for tag, low, ded in (("both", True, True), ("low", True, False), ("none", False, False)):
        img, clean, rec = oloss.preprocess_batch(g["u8"], float(g["dark_param"]), low, ded)
        close(img, g[f"{tag}_img"], 0, 1e-7, tag + " img")
        close(clean, g[f"{tag}_clean"], 0, 1e-7, tag + " clean")
        close(rec, g[f"{tag}_rec"], 1e-6, 1e-9, tag + " recovery")


def test_pre_nms_stage_golden(monkeypatch):
    """Everything non_max_suppression does BEFORE torchvision.ops.nms (utils/ops.py:196-259: candidate filter, xywh2xyxy,
    multi-label expansion, max_nms cap, class offsets) captured from the reference with a recording stand-in for the nms call that
    keeps every box (g9_prenms.npz); the oracle must hand ITS nms the same boxes / scores and produce the same rows."""
    g = gold("g9_prenms")
    calls = []

    def rec(boxes, scores, thr):
        calls.append((boxes.clone(), scores.clone()))
        return torch.arange(boxes.shape[0])
    monkeypatch.setattr(oval, "nms", rec)
    for tag, kw in (("ml", dict(multi_label=True)), ("sl", dict(multi_label=False)), ("cap", dict(multi_label=True, max_nms=50)),
                    ("agn", dict(multi_label=True, max_wh=0))):
        calls.clear()
        out = oval.non_max_suppression(g["pred"].clone(), 0.3, 0.6, max_det=1000, **kw)
        assert len(calls) == int(g[f"{tag}_ncalls"]) == 2
        for i in range(2):
            assert torch.equal(calls[i][0], g[f"{tag}_boxes{i}"]), (tag, i, "boxes handed to nms")
            assert torch.equal(calls[i][1], g[f"{tag}_scores{i}"]), (tag, i, "scores handed to nms")
            assert torch.equal(out[i], g[f"{tag}_out{i}"]), (tag, i, "rows")


def test_reference_initial_bn_buffers():
    """A new reference model does not start from BatchNorm buffers (0, 1): its constructor runs two train-mode forward passes of a
    zero image with torch's default eps / momentum (tasks.py:284-292).  g10_initbuf.npz holds the parameters and buffers of a
    freshly constructed reference model; the oracle must reproduce the buffers from the parameters alone."""
    g = gold("g10_initbuf")
    cfg = load_yaml("yolov8ori.yaml")
    cfg["scales"]["t"] = [0.33, 0.0625, 1024]
    plan, save = om.build_plan(cfg, scale="t", nc=4)
    shapes = om.param_shapes(plan)
    assert set(shapes) == set(g), sorted(set(shapes) ^ set(g))[:6]
    sd = {}
    for k, v in g.items():
        if k.endswith("running_mean"):
            sd[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            sd[k] = torch.ones_like(v)
        elif k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros_like(v)
        else:
            sd[k] = v.clone()
    om.reference_initial_buffers(plan, save, sd)
    for k, v in g.items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v) == 2, k
        elif "running_" in k:
            close(sd[k], v, 1e-5, 1e-7, k)
