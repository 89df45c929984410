"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against
  * golden vectors captured from the reference (tests/golden/*.npz), and
  * the CPU oracle on the same seeded inputs.
fp32 path tolerances: forward 1e-4 (rel+abs), gradients 2e-3; loss / loss_items 1e-4 (BASELINE.json north_star);
integer assigner outputs bit-exact.  bf16 path: documented looser bounds vs the fp32 oracle."""
import numpy as np
import pytest
import torch

from util import close, gold, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fp32():
    import dedark_yolo_amd as dy
    dy.set_compute_dtype(torch.float32)
    yield
    dy.set_compute_dtype(torch.float32)


def _run_block(name, module, nin=1, listin=False, prefix="", tol_y=1e-4, tol_g=2e-3):
    from parity_helpers import load_sd, set_bn
    from oracle import model as om
    g = gold(name)
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = om.rng_fill(shapes, int(g["seed"]))
    load_sd(set_bn(module), sd)
    module = module.cuda().train()
    xs = [g[f"x{i}"].clone().cuda().requires_grad_(True) for i in range(nin)]
    y = module(list(xs) if listin else xs[0])
    ys = y if isinstance(y, (list, tuple)) else [y]
    tot = 0
    for i, t in enumerate(ys):
        close(t.detach().float().cpu(), g[f"y{i}"], tol_y, tol_y, f"{name} y{i}")
        tot = tot + (t.float() * rnd(900 + i, *t.shape, lo=-1, hi=1).cuda()).sum()
    tot.backward()
    torch.cuda.synchronize()
    for i, x in enumerate(xs):
        close(x.grad.float().cpu(), g[f"dx{i}"], tol_g, tol_g, f"{name} dx{i}")
    named = dict(module.named_parameters())
    msd = module.state_dict()
    for k, v in g.items():
        if k.startswith("g:"):
            close(named[k[2:]].grad.cpu(), v, tol_g, tol_g, f"{name} {k}")
        elif k.startswith("gn:"):
            close(named[k[3:]].grad.norm().cpu(), v, tol_g, 1e-4, f"{name} {k}")
        elif k.startswith("b:"):
            close(msd[k[2:]].cpu(), v, 1e-4, 1e-5, f"{name} {k}")


def test_conv_3x3_s2_golden():
    from dedark_yolo_amd.nn.modules import Conv
    _run_block("g2_conv_s2", Conv(16, 32, 3, 2))


def test_conv_1x1_golden():
    from dedark_yolo_amd.nn.modules import Conv
    _run_block("g2_conv_1x1", Conv(24, 16, 1, 1))


def test_c2f_shortcut_golden():
    from dedark_yolo_amd.nn.modules import C2f
    _run_block("g2_c2f_sc", C2f(32, 32, 2, True))


def test_c2f_noshortcut_golden():
    from dedark_yolo_amd.nn.modules import C2f
    _run_block("g2_c2f_nosc", C2f(48, 32, 1, False))


def test_sppf_golden():
    from dedark_yolo_amd.nn.modules import SPPF
    _run_block("g2_sppf", SPPF(32, 32, 5))


def test_rfb_golden():
    from dedark_yolo_amd.nn.modules import RFBblock
    _run_block("g2_rfb", RFBblock(32))


@pytest.mark.parametrize("level", [0, 1, 2])
def test_asff_golden(level):
    from dedark_yolo_amd.nn.modules import AsffTribeLevel
    _run_block(f"g2_asff{level}", AsffTribeLevel(level), nin=3, listin=True)


@pytest.mark.parametrize("level", [0, 1])
def test_asff_two_level_golden(level):
    from dedark_yolo_amd.nn.modules import AsffDoubLevel
    _run_block(f"g2_asff2_{level}", AsffDoubLevel(level), nin=2, listin=True)


def test_scconv_golden():
    """SCConv (reference conv.py:420-440): group norm with unbiased std, gate, cross reconstruction, grouped 3x3, pooled softmax."""
    from dedark_yolo_amd.nn.modules import SCConv
    _run_block("g2_scconv", SCConv(64))


def test_mfru_golden():
    """MFRU (reference block.py:164-217): scconv512 + pwconv applied to P5 and P4, scconv256 to P3 and to the fused map -- the
    parameter gradients are sums over both uses."""
    from dedark_yolo_amd.nn.modules import MFRU
    _run_block("g2_mfru", MFRU(None), nin=3, listin=True)


def test_yolov8_3_graph_vs_oracle_and_direct_gradient_placement():
    """cfg/models/v8/yolov8-3.yaml at scale l (MFRU + RFB + ASFF): one training step against the oracle, then the same step under
    the trainer -- whose kernels write gradients straight into the flat buffer, except for MFRU's shared parameters, which are
    summed on the tape and copied in -- must leave the same gradients."""
    from dedark_yolo_amd.engine.trainer import DetectionTrainer, get_cfg
    from parity_helpers import build_models, make_batch, model_parity_case
    r = model_parity_case("yolov8-3.yaml", "l", None, 77, 128, 2, [2, 3])
    print("yolov8-3", {k: r[k] for k in ("loss", "oracle_loss", "median_grad_rel", "worst5")})
    assert abs(r["loss"] - r["oracle_loss"]) <= 1e-4 * abs(r["oracle_loss"])
    assert r["grad_finite"] and r["n_nograd"] == 0 and r["median_grad_rel"] < 1e-2
    model, _ = build_models("yolov8-3.yaml", "l", None, 77)
    batch = make_batch(78, 2, 128, [2, 3])
    gb = dict(batch)
    gb["img"] = batch["img"].pow(3.0).cuda()
    gb["recovery_loss_batch"] = torch.tensor(0.0123, device="cuda")
    model.train()
    loss, _ = model(dict(gb))
    loss.backward()
    torch.cuda.synchronize()
    want = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    model2, _ = build_models("yolov8-3.yaml", "l", None, 77)
    tr = DetectionTrainer(get_cfg(dict(model="yolov8-3", dtype="fp32", optimizer="SGD", batch=64)))
    tr.setup(model2)
    loss2, _ = tr.model(dict(gb))
    loss2.backward()
    from dedark_yolo_amd import ops
    ops.wgrad_join()
    torch.cuda.synchronize()
    assert abs(float(loss2) - float(loss)) <= 1e-5 * abs(float(loss))
    worst = 0.0
    for k, p in tr.model.named_parameters():
        if k in want:
            den = float(want[k].norm())
            e = float((p.grad - want[k]).norm()) / den if den > 1e-8 else float((p.grad - want[k]).norm())
            worst = max(worst, e)
            if "scconv" in k or "pwconv" in k:
                assert e <= 1e-3, (k, e)
    assert worst <= 5e-2, worst


def test_asff_detect_goldens():
    from dedark_yolo_amd.nn.modules import AsffDetect
    from parity_helpers import load_sd, set_bn
    from oracle import model as om
    d = AsffDetect(5, (16, 32, 32))
    d.stride = torch.tensor([8., 16., 32.])
    _run_block("g2_asffdetect_train", d, nin=3, listin=True)
    g = gold("g2_asffdetect_eval")
    d = AsffDetect(5, (16, 32, 32))
    d.stride = torch.tensor([8., 16., 32.])
    load_sd(set_bn(d), om.rng_fill({k: tuple(v.shape) for k, v in d.state_dict().items()}, int(g["seed"])))
    d = d.cuda().eval()
    with torch.no_grad():
        y, maps = d([g["x0"].cuda(), g["x1"].cuda(), g["x2"].cuda()])
    close(y.cpu(), g["y"], 1e-4, 1e-4, "asffdetect eval y")
    for i, m in enumerate(maps):
        close(m.float().cpu(), g[f"m{i}"], 1e-4, 1e-4, f"asffdetect eval map{i}")


def test_detect_train_golden():
    from dedark_yolo_amd.nn.modules import Detect
    d = Detect(5, (16, 32, 32))
    d.stride = torch.tensor([8., 16., 32.])
    _run_block("g2_detect_train", d, nin=3, listin=True)


def test_detect_eval_golden():
    from dedark_yolo_amd.nn.modules import Detect
    from parity_helpers import load_sd, set_bn
    from oracle import model as om
    g = gold("g2_detect_eval")
    d = Detect(5, (16, 32, 32))
    d.stride = torch.tensor([8., 16., 32.])
    sd = om.rng_fill({k: tuple(v.shape) for k, v in d.state_dict().items()}, int(g["seed"]))
    load_sd(set_bn(d), sd)
    d = d.cuda().eval()
    with torch.no_grad():
        y, maps = d([g["x0"].cuda(), g["x1"].cuda(), g["x2"].cuda()])
    close(y.cpu(), g["y"], 1e-4, 1e-4, "detect eval y")
    for i, m in enumerate(maps):
        close(m.float().cpu(), g[f"m{i}"], 1e-4, 1e-4, f"detect eval map{i}")


def test_frontend_golden():
    from dedark_yolo_amd.nn.modules import lowlight_recovery
    from parity_helpers import load_sd
    from oracle import model as om
    g = gold("g1_frontend")
    m = lowlight_recovery(3, 3)
    sd = om.rng_fill({k: tuple(v.shape) for k, v in m.state_dict().items()}, int(g["seed"]))
    load_sd(m, sd)
    m = m.cuda().train()
    x = g["x"].clone().cuda().requires_grad_(True)
    out = m(x)
    close(out.float().cpu(), g["out"], 1e-4, 2e-4, "front-end out")
    (out.float() * g["wgt"].cuda()).sum().backward()
    torch.cuda.synchronize()
    close(x.grad.cpu(), g["dx"], 2e-3, 2e-3, "front-end dx")
    named = dict(m.named_parameters())
    close(named["extractor.fc2.weight"].grad.cpu(), g["d_fc2_w"], 2e-3, 2e-2, "d fc2.w")
    close(named["extractor.fc2.bias"].grad.cpu(), g["d_fc2_b"], 2e-3, 2e-2, "d fc2.b")
    close(named["extractor.fc1.bias"].grad.cpu(), g["d_fc1_b"], 2e-3, 2e-2, "d fc1.b")
    close(named["extractor.conv_layers.0.conv_block.0.weight"].grad.cpu(), g["d_c0_w"], 3e-3, 3e-2, "d conv0.w")
    close(named["extractor.conv_layers.4.conv_block.0.bias"].grad.cpu(), g["d_c4_b"], 3e-3, 3e-2, "d conv4.b")
    m.eval()
    with torch.no_grad():
        out2 = m(g["x"].cuda(), g["A2"].cuda(), g["IcA2"].cuda())
    close(out2.float().cpu(), g["out2"], 1e-4, 2e-4, "front-end out (A, IcA)")


def test_ka1_filter_chain_on_gpu():
    """SURVEY Appendix A KA1 through the C-ABI: fixed features, analytic image."""
    import ctypes  # noqa: F401
    from dedark_yolo_amd._C import call
    from dedark_yolo_amd.ops import ptr, stream
    g = gold("g1_ka1")
    x = g["x"].cuda().contiguous()
    feat = torch.zeros(1, 16, device="cuda")
    feat[:, :15] = g["feat"].cuda()
    params = torch.empty(1, 8, device="cuda")
    call("dy_filter_params_fwd", ptr(feat), 16, ptr(params), 1, stream())
    p = params.cpu()[0]
    close(p[:7], torch.tensor([0.2072826, 1.2354821, 0.9091753, 0.9543331, 0.6414791, 0.6947827, 4.4039855]), 1e-5, 1e-6, "KA1 params")
    s4 = torch.empty_like(x)
    call("dy_filters_pointwise_fwd", ptr(x), ptr(params), None, None, ptr(s4), 1, 16, 20, 0, stream())
    close(s4.cpu(), g["s4"], 1e-5, 1e-5, "KA1 contrast stage")
    s4_fast = torch.empty_like(x)                   # throughput-mode transcendentals (v_log/v_exp/v_rcp): ~2e-6 relative
    call("dy_filters_pointwise_fwd", ptr(x), ptr(params), None, None, ptr(s4_fast), 1, 16, 20, 1, stream())
    close(s4_fast.cpu(), g["s4"], 2e-5, 2e-5, "KA1 contrast stage, fast math")
    out = torch.empty_like(x)
    call("dy_usm_fwd", ptr(s4), ptr(params), ptr(out), None, None, 1, 16, 20, 0, stream())
    close(out.cpu(), g["s5"], 2e-5, 2e-5, "KA1 usm stage")
    assert abs(float(out.sum()) - 518.42413) < 5e-3


@pytest.mark.parametrize("name,yml,scale", [("g3_ori_tiny", "yolov8ori.yaml", "t"), ("g3_ll_tiny", "yolov8-lowlight.yaml", "t")])
def test_model_tiny_golden_and_oracle(name, yml, scale):
    """Whole training step vs the golden loss of the reference AND vs oracle gradients for every parameter."""
    from parity_helpers import model_parity_case
    g = gold(name)
    r = model_parity_case(yml, scale, [float(v) for v in g["scale_def"]], int(g["seed"]), int(g["S"]), int(g["B"]),
                          [int(v) for v in g["nbox"]])
    print(name, r)
    close(r["loss"], g["loss"], 1e-4, 1e-4, f"{name} loss vs reference golden")
    close(torch.tensor(r["items"]), g["items"], 1e-4, 1e-4, f"{name} items vs reference golden")
    assert r["grad_finite"] and r["n_nograd"] == 0
    assert r["worst_grad_rel"] < 5e-3, (r["worst_grad_key"], r["worst_grad_rel"])
    assert r["worst_running_stat_abs"] < 1e-4


def test_model_repo_l_golden():
    from parity_helpers import model_parity_case
    g = gold("g3_repo_l")
    r = model_parity_case("yolov8.yaml", "l", None, int(g["seed"]), int(g["S"]), int(g["B"]), [int(v) for v in g["nbox"]],
                          fp64=True)
    print("repo_l", r)
    close(r["loss"], g["loss"], 1e-4, 1e-4, "repo-L loss vs reference golden")
    close(torch.tensor(r["items"]), g["items"], 1e-4, 1e-4, "repo-L items vs reference golden")
    assert r["grad_finite"] and r["n_nograd"] == 0
    # 126 BatchNorm layers over 2x2..8x8 maps with B=2 (8..128 samples per channel) make this case chaotic in fp32: the CPU
    # oracle itself is 0.6 % off its own float64 run.  Only sanity-bound the gradients here; gradient parity for this graph is
    # judged on the better-conditioned case below.
    assert r["worst_grad_rel"] < 0.5, r["worst5"]


def test_model_repo_l_gradients_vs_fp64_oracle():
    """Repo yolov8.yaml@L at 128x128, B=4: every parameter gradient of the HIP fp32 path against a float64 oracle run, with
    the fp32 oracle's own error as the yardstick."""
    from parity_helpers import model_parity_case
    r = model_parity_case("yolov8.yaml", "l", None, 404, 128, 4, [3, 2, 5, 1], fp64=True)
    print("repo_l_128", r)
    assert abs(r["loss"] - r["oracle_loss"]) <= 1e-4 * abs(r["oracle_loss"])
    assert r["grad_finite"] and r["n_nograd"] == 0
    # yardstick from the same run: the fp32 CPU oracle is itself up to 8 % off its float64 twin on the ASFF weight branch
    assert r["median_grad_rel"] < 1e-2, r["median_grad_rel"]
    assert r["worst_grad_rel"] < 0.1, r["worst5"]


def _assign_case(B, S, nbox, seed, tie=False):
    """HIP assigner vs oracle (torch.topk on CPU) on identical Detect maps; 192x192 images give A = 756 >= 640, the regime
    where torch.topk uses std::partial_sort, whose tie order the kernel emulates."""
    from dedark_yolo_amd.utils.loss import assign
    from oracle import loss as oloss
    from oracle import model as om
    from util import make_batch
    nc = 20
    gsz = [S // 8, S // 16, S // 32]
    gen = np.random.default_rng(seed)
    maps = [torch.from_numpy(gen.normal(0, 1.0, (B, 64 + nc, h, h)).astype(np.float32)) for h in gsz]
    if tie:
        for m in maps:                     # identical logits on a row of cells -> exactly equal metrics
            m[:, :, 1, :] = m[:, :, 1, :1]
            m[0, 64:] = -120.0             # sigmoid underflows to 0 -> zero-metric ties inside the boxes of image 0
    batch = make_batch(seed, B, S, nbox)
    strides = [8.0, 16.0, 32.0]
    _, _, det = oloss.detection_loss(maps, batch, strides, nc, oloss.default_hyp(), details=True)
    gm = [m.cuda().contiguous(memory_format=torch.channels_last) for m in maps]
    a = assign(gm, strides, nc, batch["batch_idx"], batch["cls"], batch["bboxes"])
    torch.cuda.synchronize()
    assert torch.equal(a.fg_mask.cpu().bool(), det["fg_mask"]), "fg_mask differs"
    assert torch.equal(a.target_gt_idx.cpu().long(), det["target_gt_idx"]), "target_gt_idx differs"
    ts = det["target_scores"]
    close(a.norm.cpu(), ts.sum(-1), 1e-4, 1e-6, "target score per anchor")
    fg = det["fg_mask"]
    assert torch.equal(a.target_label.cpu().long()[fg], det["target_labels"][fg])


def test_assigner_bit_exact_random():
    _assign_case(3, 192, [5, 0, 8], 11)


def test_assigner_bit_exact_ties():
    _assign_case(2, 192, [4, 6], 12, tie=True)


def test_loss_forward_backward_vs_oracle():
    """Criterion alone on random maps: loss/items 1e-4, d loss / d maps 1e-3 vs oracle autograd."""
    from types import SimpleNamespace
    from dedark_yolo_amd.utils.loss import RcoveryDetectionLoss
    from oracle import loss as oloss
    from util import make_batch
    nc, B, S = 20, 3, 192
    gen = np.random.default_rng(5)
    maps = [torch.from_numpy(gen.normal(0, 1.0, (B, 64 + nc, S // s, S // s)).astype(np.float32)) for s in (8, 16, 32)]
    batch = make_batch(21, B, S, [3, 7, 1])
    batch["recovery_loss_batch"] = torch.tensor(0.05)
    om_ = [m.clone().requires_grad_(True) for m in maps]
    ol, oi = oloss.recovery_detection_loss(om_, batch, [8.0, 16.0, 32.0], nc, oloss.default_hyp())
    ol.backward()
    det = SimpleNamespace(stride=torch.tensor([8.0, 16.0, 32.0]), nc=nc, no=64 + nc, reg_max=16)
    holder = SimpleNamespace(args=oloss.default_hyp(), model=[det], parameters=lambda: iter([torch.zeros(1, device="cuda")]))
    crit = RcoveryDetectionLoss(holder)
    gm = [m.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) for m in maps]
    gb = dict(batch)
    gb["recovery_loss_batch"] = batch["recovery_loss_batch"].cuda()
    loss, items = crit(gm, gb)
    loss.backward()
    torch.cuda.synchronize()
    close(loss.cpu(), ol.detach(), 1e-4, 1e-4, "loss")
    close(items.cpu(), oi, 1e-4, 1e-4, "loss_items")
    for i in range(3):
        close(gm[i].grad.cpu(), om_[i].grad, 2e-3, 1e-5, f"d loss / d map{i}")


def test_bf16_step_close_to_fp32_oracle():
    """bf16 throughput path (no reference equivalent: the reference AMP is fp16 and off by default): loss within 3 % of the
    fp32 oracle on the tiny model, gradients finite."""
    from parity_helpers import model_parity_case
    r = model_parity_case("yolov8-lowlight.yaml", "t", [0.33, 0.125, 1024], 302, 64, 2, [2, 4], dtype=torch.bfloat16)
    print("bf16", r)
    assert r["grad_finite"] and r["n_nograd"] == 0
    assert abs(r["loss"] - r["oracle_loss"]) < 0.03 * abs(r["oracle_loss"])


def test_full_size_properties():
    """BASELINE config C2 size (YOLOv8n + lowlight_recovery, 640x640, bf16) with size-independent checks: finite loss and
    gradients, eval decode shape, conv linearity conv(a)+conv(b) == conv(a+b) on the f32 path."""
    import dedark_yolo_amd as dy
    from dedark_yolo_amd.nn.modules import Conv
    from parity_helpers import HYP
    from dedark_yolo_amd.nn.tasks import DetectionModel
    from util import make_batch
    dy.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    model = DetectionModel("yolov8n-lowlight.yaml", nc=20).cuda().train()
    model.args = HYP
    B = 4
    batch = make_batch(3, B, 640, [3, 1, 5, 2])
    gb = dict(batch)
    gb["img"] = batch["img"].pow(5.0).cuda()
    gb["recovery_loss_batch"] = torch.tensor(0.0, device="cuda")
    loss, items = model(gb)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and torch.isfinite(items).all()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    model.eval()
    with torch.no_grad():
        y, maps = model(gb["img"])
    assert y.shape == (B, 24, 8400) and torch.isfinite(y).all()
    dy.set_compute_dtype(torch.float32)
    conv = Conv(32, 64, 3, 1).cuda().eval()
    a, b = torch.randn(2, 32, 40, 40, device="cuda"), torch.randn(2, 32, 40, 40, device="cuda")
    conv.act = torch.nn.Identity()
    conv._act = 0
    conv.bn.running_mean.zero_()          # pure linear map
    with torch.no_grad():
        close((conv(a) + conv(b)).cpu(), conv(a + b).cpu(), 1e-4, 1e-4, "conv linearity")


@pytest.mark.parametrize("case", ["one_image_empty", "no_boxes"])
def test_loss_with_empty_targets_vs_oracle(case):
    """Ragged / empty label sets (reference loss.py:125-140 preprocess handles images without boxes; with no boxes at all only
    the classification term and the recovery term remain)."""
    from parity_helpers import build_models
    from oracle import loss as oloss
    from oracle import model as om
    model, (plan, save, sd) = build_models("yolov8-lowlight.yaml", "t", [0.33, 0.125, 1024], 5)
    S, B = 64, 3
    img = rnd(41, B, 3, S, S)
    if case == "one_image_empty":
        bi, cls, bb = [0, 0, 2], [1, 3, 5], [[.5, .5, .3, .4], [.3, .6, .2, .2], [.6, .4, .5, .5]]
    else:
        bi, cls, bb = [], [], []
    batch = dict(img=img.clone(), batch_idx=torch.tensor(bi, dtype=torch.float32), cls=torch.tensor(cls, dtype=torch.float32).view(-1, 1),
                 bboxes=torch.tensor(bb, dtype=torch.float32).view(-1, 4), recovery_loss_batch=torch.tensor(0.01))
    gb = {k: v.cuda() for k, v in batch.items()}
    model.train()
    loss, items = model(gb)
    loss.backward()
    torch.cuda.synchronize()
    maps = om.forward(plan, save, {k: v.detach().clone() for k, v in sd.items()}, batch["img"], True)
    strides = [float(S // m.shape[2]) for m in maps]
    ol, oi = oloss.recovery_detection_loss(maps, batch, strides, 20, oloss.default_hyp())
    close(loss.detach().cpu(), ol.detach(), 1e-4, 1e-4, f"{case} loss")
    close(items.detach().cpu(), oi.detach(), 1e-4, 1e-4, f"{case} items")
    assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    if case == "no_boxes":
        assert float(items[0]) == 0.0 and float(items[2]) == 0.0


def test_model_gradients_with_branch_streams_and_autograd_gradients():
    """The Detect levels on branch streams while the gradients go back through autograd (no trainer, no flat buffer): the three
    levels then run their weight gradients on three streams at once -- every one needs its own split-K workspace (a shared one
    produced garbage in model.26.cv2.1.0.conv.weight).  Same bounds as the single-stream case above."""
    from dedark_yolo_amd import ops
    from parity_helpers import model_parity_case
    ops.enable_branch_streams(True)
    try:
        r = model_parity_case("yolov8.yaml", "l", None, 404, 128, 4, [3, 2, 5, 1], fp64=True)
    finally:
        ops.enable_branch_streams(False)
    assert abs(r["loss"] - r["oracle_loss"]) <= 1e-4 * abs(r["oracle_loss"])
    assert r["grad_finite"] and r["n_nograd"] == 0
    assert r["median_grad_rel"] < 1e-2 and r["worst_grad_rel"] < 0.1, r["worst5"]


@pytest.mark.parametrize("name", ["g4_assigner", "g4_assigner_b"])
def test_assigner_forward_on_reference_vectors(name):
    """TaskAlignedAssigner.forward(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt) -- the reference's own call
    (tal.py:84-132) -- through dy_tal_assign_decoded on the inputs captured from the reference (incl. the crafted zero-metric ties).
    84 / 336 anchors per row: torch.topk takes std::nth_element there, which the kernel reproduces (integer outputs bit-exact)."""
    from dedark_yolo_amd.utils.loss import TaskAlignedAssigner
    g = gold(name)
    nc = int(g["nc"])
    asg = TaskAlignedAssigner(topk=10, num_classes=nc, alpha=0.5, beta=6.0)
    tl, tb, ts, fg, gi = asg(g["scores"].cuda(), g["boxes"].cuda(), g["anc"].cuda(), g["lab"].cuda(), g["gt"].cuda(), g["mask"].cuda())
    torch.cuda.synchronize()
    assert torch.equal(fg.cpu(), g["fg_mask"].bool()), "fg_mask"
    assert torch.equal(gi.cpu(), g["target_gt_idx"].long()), "target_gt_idx"
    assert torch.equal(tl.cpu(), g["target_labels"].long()), "target_labels"
    assert torch.equal(tb.cpu(), g["target_bboxes"]), "target_bboxes"
    close(ts.cpu(), g["target_scores"], 1e-5, 1e-6, "target_scores")


def test_assigner_small_rows_vs_oracle_with_many_ties():
    """Rows of 100 .. 600 anchors (below torch's partial_sort switch at 64 * k) with few positive candidates per ground truth, so that
    the top-10 are filled up with zero-metric anchors: the integer outputs must still equal the oracle's (= torch.topk's) choice."""
    from oracle import loss as oloss
    from dedark_yolo_amd.utils.loss import TaskAlignedAssigner
    g = np.random.default_rng(123)
    for A_side, n in ((10, 3), (15, 5), (20, 6), (24, 8)):
        A, B, nc = A_side * A_side, 3, 7
        st = 8.0
        ys, xs = np.meshgrid(np.arange(A_side) + 0.5, np.arange(A_side) + 0.5, indexing="ij")
        anc = torch.tensor(np.stack((xs.ravel(), ys.ravel()), 1) * st, dtype=torch.float32)
        c = g.uniform(2 * st, (A_side - 2) * st, (B, n, 2))
        wh = g.uniform(1.2 * st, 3.5 * st, (B, n, 2))                      # small boxes: 1 .. 12 cells inside each
        gtb = torch.tensor(np.concatenate((c - wh / 2, c + wh / 2), 2), dtype=torch.float32)
        lab = torch.tensor(g.integers(0, nc, (B, n, 1)), dtype=torch.float32)
        mask = torch.ones(B, n, 1)
        mask[1, n - 1] = 0
        gtb[1, n - 1] = 0
        pc = anc[None].repeat(B, 1, 1) + torch.tensor(g.normal(0, 3, (B, A, 2)), dtype=torch.float32)
        pwh = torch.tensor(g.uniform(1.0 * st, 4.0 * st, (B, A, 2)), dtype=torch.float32)
        boxes = torch.cat((pc - pwh / 2, pc + pwh / 2), 2)
        scores = torch.tensor(g.random((B, A, nc)), dtype=torch.float32)
        want = oloss.tal_assign(scores, boxes, anc, lab, gtb, mask, nc)
        asg = TaskAlignedAssigner(topk=10, num_classes=nc, alpha=0.5, beta=6.0)
        tl, tb, ts, fg, gi = asg(scores.cuda(), boxes.cuda(), anc.cuda(), lab.cuda(), gtb.cuda(), mask.cuda())
        torch.cuda.synchronize()
        assert torch.equal(fg.cpu(), want[3]), (A, "fg_mask")
        assert torch.equal(gi.cpu(), want[4]), (A, "target_gt_idx")
        assert torch.equal(tl.cpu(), want[0].long()), (A, "target_labels")
        close(ts.cpu(), want[2], 1e-5, 1e-6, f"A={A} target_scores")


def test_graph_backward_is_one_autograd_node_and_matches_the_per_module_path(monkeypatch):
    """nn/tasks.py:_GraphFn -- the training graph behind ONE autograd node (fan-out gradients added by the consuming Conv's data
    gradient or dy_copy2d) against one autograd.Function per module (ATen adds): same loss, gradients equal up to the
    order of the fan-out sums (fp32: 2e-5 of each tensor's norm, 1e-4 for the front-end), and the graph path is the one that ran."""
    from dedark_yolo_amd.nn import tasks
    from parity_helpers import model_parity_case
    calls = []
    orig = tasks.GraphPlan.backward_train
    monkeypatch.setattr(tasks.GraphPlan, "backward_train", lambda self, st, gouts, xn: (calls.append(1), orig(self, st, gouts, xn))[1])
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(tasks, "_GRAPH_BACKWARD", mode)
        n0 = len(calls)
        res[mode] = model_parity_case("yolov8.yaml", "l", None, 505, 128, 2, [2, 3], with_oracle=False, keep_grads=True)
        assert (len(calls) > n0) == mode
    a, b = res[True], res[False]
    assert abs(a["loss"] - b["loss"]) <= 1e-6 * abs(b["loss"])
    assert a["grads"].keys() == b["grads"].keys() and len(a["grads"]) > 300
    for k in a["grads"]:
        d = float((a["grads"][k] - b["grads"][k]).norm()) / max(float(b["grads"][k].norm()), 1e-30)
        assert d < (1e-4 if k.startswith("model.0.") else 2e-5), (k, d)      # the front-end's gradient has passed through the whole network


@pytest.mark.parametrize("tag,name,nc", [("ori_t", "yolov8ori.yaml", 4), ("ll_t", "yolov8-lowlight.yaml", 20)])
def test_product_eval_equals_the_reference_running_our_checkpoint(tag, name, nc):
    """tests/golden/g12_ckpt_interop.npz: the REFERENCE's eval output after loading a last.pt written by this package
    (attempt_load_one_weight, ultralytics/nn/tasks.py:674-707; produced in the build container by make_ckpt_interop.py).  The product
    on the GPU, with the same fp16-rounded EMA weights and the same seeded input, must give the same decoded predictions."""
    import dedark_yolo_amd as dy
    from dedark_yolo_amd.nn.tasks import DetectionModel
    from oracle import model as om
    from parity_helpers import load_sd
    from util import gold, load_yaml, rnd
    g = gold("g12_ckpt_interop")
    cfg = load_yaml(name)
    cfg["scales"]["t"] = [0.33, 0.0625, 1024]
    cfg["scale"] = "t"
    dy.set_compute_dtype(torch.float32)
    model = DetectionModel(cfg, nc=nc)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    ema = om.rng_fill(shapes, 1202)                                   # what the script stored as the checkpoint's `ema`
    load_sd(model, {k: (v.half().float() if v.is_floating_point() else v) for k, v in ema.items()})
    model = model.cuda().eval()
    x = rnd(int(g[f"{tag}_x_seed"]), 2, 3, 64, 64).pow(2.0)
    with torch.no_grad():
        y = model(x.cuda())
    y = y[0] if isinstance(y, (list, tuple)) else y
    want = g[f"{tag}_y"]
    err = float((y.float().cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-30)
    assert y.shape == want.shape and err <= 1e-4, err
