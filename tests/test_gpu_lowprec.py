"""Low-precision (bf16 / fp16) throughput paths against the fp32 goldens of the reference, and full-size property tests.

The reference has no bf16 path (its AMP is fp16 autocast, engine/trainer.py:221,330,340), so there is no golden in these dtypes:
the yardstick is the reference's fp32 golden (per-block vectors of tests/golden/g2_*.npz, the L graph against the fp32 oracle), and
the bound is stated per quantity as a relative L2 error |a - ref|_2 / |ref|_2 (measured values of round 2 in brackets):
                                               bf16 (8-bit mantissa)              fp16 (11-bit mantissa)
    block forward                              <= 2e-2   [3e-3 .. 1e-2]           <= 3e-3   [4e-4 .. 1.4e-3]
    Conv / C2f / RFB / Detect: dx, dparam      <= 3e-2   [4e-3 .. 1.5e-2]         <= 6e-3   [4e-4 .. 3.2e-3]
    SPPF / ASFF: dx, dparam                    <= 0.30   [6e-2 .. 0.17]           <= 0.15   [1e-3 .. 9e-2]
      (ill-conditioned on purpose: max-pool argmax routing flips on roundings, and the ASFF level-weight gradient is a
       sum over channels of dout * (x_level - out) that cancels to a few per cent of its terms; fp32 meets 2e-3 on the same vectors)
    L graph (front-end + ASFF neck) at 256x256, B=4: Detect maps and the gradient DIRECTION of a smooth functional of the maps
    (cosine to the fp32 oracle's), training loss within 5 %; see test_l_graph_low_precision_vs_fp32_oracle for why the
    criterion's own gradients are not comparable in 16 bit.
Full-size cases (BASELINE configs[0] C1 and configs[2] C3 shapes) use properties that need no oracle: finiteness, invariance of
the training loss under a permutation of the batch, independence of an eval prediction from the other images of its batch."""
import numpy as np
import pytest
import torch

from util import gold, make_batch, rnd

pytestmark = pytest.mark.gpu

LOWP = [torch.bfloat16, torch.float16]


@pytest.fixture(autouse=True)
def _restore_dtype():
    import dedark_yolo_amd as dy
    yield
    dy.set_compute_dtype(torch.float32)


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _supported(dtype):
    import dedark_yolo_amd as dy
    try:
        dy.set_compute_dtype(dtype)
    except (ValueError, NotImplementedError) as e:
        pytest.skip(str(e))


def _block(name, module, dtype, nin=1, listin=False):
    """The fp32 golden of a block, replayed in `dtype`: returns the worst relative L2 error of (outputs, input grads, param grads)."""
    from parity_helpers import load_sd, set_bn
    from oracle import model as om
    g = gold(name)
    sd = om.rng_fill({k: tuple(v.shape) for k, v in module.state_dict().items()}, int(g["seed"]))
    load_sd(set_bn(module), sd)
    module = module.cuda().train()
    xs = [g[f"x{i}"].clone().cuda().requires_grad_(True) for i in range(nin)]
    y = module(list(xs) if listin else xs[0])
    ys = y if isinstance(y, (list, tuple)) else [y]
    e_y, tot = 0.0, 0
    for i, t in enumerate(ys):
        e_y = max(e_y, _rel(t.detach().float(), g[f"y{i}"]))
        tot = tot + (t.float() * rnd(900 + i, *t.shape, lo=-1, hi=1).cuda()).sum()
    tot.backward()
    torch.cuda.synchronize()
    e_dx = max(_rel(x.grad.float(), g[f"dx{i}"]) for i, x in enumerate(xs))
    named = dict(module.named_parameters())
    e_dp = max([_rel(named[k[2:]].grad, v) for k, v in g.items() if k.startswith("g:")] or [0.0])
    return e_y, e_dx, e_dp


def _blocks():
    from dedark_yolo_amd.nn import modules as M

    def detect():
        d = M.Detect(5, (16, 32, 32))
        d.stride = torch.tensor([8., 16., 32.])
        return d

    def asffdetect():
        d = M.AsffDetect(5, (16, 32, 32))
        d.stride = torch.tensor([8., 16., 32.])
        return d
    return {
        "g2_conv_s2": (lambda: M.Conv(16, 32, 3, 2), 1, False), "g2_conv_1x1": (lambda: M.Conv(24, 16, 1, 1), 1, False),
        "g2_c2f_sc": (lambda: M.C2f(32, 32, 2, True), 1, False), "g2_c2f_nosc": (lambda: M.C2f(48, 32, 1, False), 1, False),
        "g2_sppf": (lambda: M.SPPF(32, 32, 5), 1, False), "g2_rfb": (lambda: M.RFBblock(32), 1, False),
        "g2_asff0": (lambda: M.AsffTribeLevel(0), 3, True), "g2_asff1": (lambda: M.AsffTribeLevel(1), 3, True),
        "g2_asff2": (lambda: M.AsffTribeLevel(2), 3, True), "g2_asff2_0": (lambda: M.AsffDoubLevel(0), 2, True),
        "g2_detect_train": (detect, 3, True), "g2_asffdetect_train": (asffdetect, 3, True),
        "g2_scconv": (lambda: M.SCConv(64), 1, False), "g2_mfru": (lambda: M.MFRU(None), 3, True),
    }


@pytest.mark.parametrize("dtype", LOWP, ids=["bf16", "fp16"])
@pytest.mark.parametrize("name", ["g2_conv_s2", "g2_conv_1x1", "g2_c2f_sc", "g2_c2f_nosc", "g2_sppf", "g2_rfb", "g2_asff0", "g2_asff1",
                                  "g2_asff2", "g2_asff2_0", "g2_detect_train", "g2_asffdetect_train", "g2_scconv", "g2_mfru"])
def test_block_goldens_low_precision(name, dtype):
    _supported(dtype)
    make, nin, listin = _blocks()[name]
    e_y, e_dx, e_dp = _block(name, make(), dtype, nin, listin)
    print(f"{name} {dtype}: forward {e_y:.2e}  dx {e_dx:.2e}  dparam {e_dp:.2e}")
    ill = "sppf" in name or ("asff" in name and "detect" not in name) or "scconv" in name or "mfru" in name      # + hard gates of SRU
    b_y, b_g = (2e-2, 0.30 if ill else 3e-2) if dtype == torch.bfloat16 else (3e-3, 0.15 if ill else 6e-3)
    assert e_y <= b_y and e_dx <= b_g and e_dp <= b_g, (e_y, e_dx, e_dp)


@pytest.mark.parametrize("dtype", LOWP, ids=["bf16", "fp16"])
def test_frontend_golden_low_precision(dtype):
    """lowlight_recovery: the extractor convs run in `dtype`, the filter chain in fp32 on the fp32 image."""
    _supported(dtype)
    from dedark_yolo_amd.nn.modules import lowlight_recovery
    from parity_helpers import load_sd
    from oracle import model as om
    g = gold("g1_frontend")
    m = lowlight_recovery(3, 3)
    load_sd(m, om.rng_fill({k: tuple(v.shape) for k, v in m.state_dict().items()}, int(g["seed"])))
    m = m.cuda().train()
    x = g["x"].clone().cuda().requires_grad_(True)
    out = m(x)
    (out.float() * g["wgt"].cuda()).sum().backward()
    torch.cuda.synchronize()
    e_y, e_dx = _rel(out.float(), g["out"]), _rel(x.grad, g["dx"])
    print(f"front-end {dtype}: out {e_y:.2e} dx {e_dx:.2e}")
    assert e_y <= 2e-2 and e_dx <= 5e-2, (e_y, e_dx)


class _StorageEmulation:
    """The fp32 oracle with every conv / BatchNorm / activation output (and the conv weights) rounded to `dtype`: what an IDEAL
    implementation that merely STORES its tensors in 16 bit computes.  The yardstick for a graph that amplifies roundings."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        import torch.nn.functional as F
        from oracle import frontend as ofe
        from oracle import model as om
        q = lambda t: t.to(self.dtype).float()
        outer = self

        class Proxy:
            def __getattr__(self, name):
                fn = getattr(F, name)
                if name == "conv2d":
                    return lambda x, w, *a, **k: q(fn(x, q(w), *a, **k))
                if name in ("batch_norm", "silu", "leaky_relu"):
                    return lambda *a, **k: q(fn(*a, **k))
                return fn
        self.mods = (om, ofe)
        for m in self.mods:
            m.F = Proxy()
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        for m in self.mods:
            m.F = F


def _functional_case(dtype, S=256, B=4):
    """Maps and parameter gradients of sum_i <map_i, w_i> (fixed random w_i) for the repo L graph in train mode:
    product in `dtype`, fp32 oracle, and the oracle under 16-bit storage emulation."""
    from parity_helpers import build_models
    from oracle import model as om
    model, (plan, save, sd) = build_models("yolov8.yaml", "l", None, 404)
    img = make_batch(405, B, S, [1] * B)["img"].pow(2.0)
    model.train()
    maps = model(img.cuda())
    ws = [rnd(950 + i, *m.shape, lo=-1, hi=1) for i, m in enumerate(maps)]
    sum((m.float() * w.cuda()).sum() for m, w in zip(maps, ws)).backward()
    torch.cuda.synchronize()
    prod = ([m.detach().float().cpu() for m in maps], {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None})

    def oracle_run(emulate):
        sd_ = om.rng_fill(om.param_shapes(plan), 404)
        for k, v in sd_.items():
            v.requires_grad_(v.is_floating_point() and v.ndim > 0 and ".dfl." not in k and "running_" not in k)
        if emulate:
            with _StorageEmulation(dtype):
                om_maps = om.forward(plan, save, sd_, img, True)
                sum((m * w).sum() for m, w in zip(om_maps, ws)).backward()
        else:
            om_maps = om.forward(plan, save, sd_, img, True)
            sum((m * w).sum() for m, w in zip(om_maps, ws)).backward()
        return [m.detach() for m in om_maps], {k: v.grad.double() for k, v in sd_.items() if v.grad is not None}
    return prod, oracle_run(False), oracle_run(True)


def _direction(ga, gb):
    keys = [k for k in ga if k in gb and not k.startswith("model.0.")]
    trip = [(float((ga[k] * gb[k]).sum()), float(ga[k].norm()) ** 2, float(gb[k].norm()) ** 2) for k in keys]
    cos_all = sum(t[0] for t in trip) / max((sum(t[1] for t in trip) * sum(t[2] for t in trip)) ** 0.5, 1e-300)
    per = sorted(t[0] / max((t[1] * t[2]) ** 0.5, 1e-300) for t in trip)
    return cos_all, per[len(per) // 2]


@pytest.mark.parametrize("dtype", LOWP, ids=["bf16", "fp16"])
def test_l_graph_low_precision_vs_fp32_oracle(dtype):
    """Repo yolov8.yaml@L (front-end + ASFF neck), 256x256, B=4, train mode, in `dtype` against the fp32 oracle: the three Detect
    maps (relative L2) and the parameter gradients of a SMOOTH functional of the maps, sum_i <map_i, w_i>.

    Two facts shape the bounds.  (1) The criterion is no yardstick for 16-bit gradients: its task-aligned assignment is a discrete
    top-k over near-tied scores (random-init class logits all sit at the bias), one flipped rounding re-targets anchors and moves
    head gradients by tens of per cent (a batch permutation alone: 35 % in bf16 at C3 size) -- hence the smooth functional.
    (2) This random-init graph (100+ train-mode BatchNorm layers, B = 4) amplifies roundings by one to two orders of magnitude, so
    the bound is RELATIVE to an ideal 16-bit implementation: the fp32 oracle with every layer output rounded to `dtype`
    (_StorageEmulation; forward tensors only -- its backward runs in fp32, the product also stores gradients in 16 bit).
    Product error <= 1.5 x the emulation's error (+ 1e-2); gradient direction no worse than the emulation's by more than 0.15
    (measured bf16: maps 0.630 vs 0.624, cosine 0.16 vs 0.27 -- both essentially decorrelated from fp32 on this graph; fp16:
    0.115 vs 0.14, cosine 0.90 vs 0.9).  The training loss follows the same logic: maps that differ from fp32 by 60 % (bf16) put the
    criterion's discrete assignment somewhere else, so its distance from the fp32 oracle is bounded by 1.5 x the distance of the
    storage emulation's loss (+ 5 % of the oracle's value; fp16, whose maps stay within 12 %, is inside the 5 % alone)."""
    _supported(dtype)
    import dedark_yolo_amd as dy
    from parity_helpers import model_parity_case
    (pm, pg), (om_, og), (em, eg) = _functional_case(dtype)
    e_prod = max(_rel(a, b) for a, b in zip(pm, om_))
    e_emu = max(_rel(a, b) for a, b in zip(em, om_))
    c_prod, c_prod_med = _direction(pg, og)
    c_emu, c_emu_med = _direction(eg, og)
    print(f"L graph {dtype}: maps rel L2 vs fp32 oracle: product {e_prod:.3e}, 16-bit storage emulation {e_emu:.3e}; gradient "
          f"direction (cosine all / per-tensor median): product {c_prod:.4f} / {c_prod_med:.4f}, emulation {c_emu:.4f} / {c_emu_med:.4f}")
    assert e_prod <= 1.5 * e_emu + 1e-2, (e_prod, e_emu)
    assert c_prod >= c_emu - 0.15 and c_prod_med >= c_emu_med - 0.15, (c_prod, c_emu, c_prod_med, c_emu_med)
    dy.set_compute_dtype(dtype)
    r = model_parity_case("yolov8.yaml", "l", None, 404, 256, 4, [3, 2, 5, 1], dtype=dtype)
    print(f"   training loss {r['loss']:.4f} vs fp32 oracle {r['oracle_loss']:.4f}")
    assert r["grad_finite"] and r["n_nograd"] == 0
    # the same step on fp32 kernels with every stored tensor rounded to `dtype`
    from parity_helpers import build_models
    dy.set_compute_dtype(torch.float32)
    model, _ = build_models("yolov8.yaml", "l", None, 404)
    batch = make_batch(405, 4, 256, [3, 2, 5, 1])
    gb = dict(batch)
    gb["img"] = batch["img"].pow(3.0).cuda()
    gb["recovery_loss_batch"] = torch.tensor(0.0123, device="cuda")
    model.train()
    with _emulated_storage(model, dtype):
        l_emu = float(model(gb)[0])
    d_prod, d_emu = abs(r["loss"] - r["oracle_loss"]), abs(l_emu - r["oracle_loss"])
    print(f"   storage emulation {l_emu:.4f}: distance from the oracle {d_emu:.2f} (product {d_prod:.2f})")
    assert d_prod <= 1.5 * d_emu + 0.05 * abs(r["oracle_loss"]), (r["loss"], l_emu, r["oracle_loss"])


import contextlib


@contextlib.contextmanager
def _emulated_storage(model, dtype):
    """fp32 kernels + 16-bit storage: activations / gradients rounded by ops.set_storage_emulation, conv weights (what the 16-bit
    paths pack into the compute dtype; BatchNorm affine, biases and the regressor's fully connected layers stay fp32) rounded here."""
    import dedark_yolo_amd as dy
    from dedark_yolo_amd import ops
    if dtype is None:
        yield
        return
    prev = ops.get_compute_dtype()
    keep = {}
    try:
        dy.set_compute_dtype(torch.float32)
        ops.set_storage_emulation(dtype)
        with torch.no_grad():
            for k, p in model.named_parameters():
                if p.ndim == 4:
                    keep[k] = p.detach().clone()
                    p.copy_(p.to(dtype))
        ops.bump_weights_epoch()
        yield
    finally:
        ops.set_storage_emulation(None)
        with torch.no_grad():
            for k, p in model.named_parameters():
                if k in keep:
                    p.copy_(keep[k])
        ops.bump_weights_epoch()
        dy.set_compute_dtype(prev)


def _cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a * b).sum() / max(float(a.norm() * b.norm()), 1e-300))


def _permute_assignment(a, perm):
    """_Assignment of batch[perm]: every field is per image (row b of the new batch = row perm[b] of the old one)."""
    from dedark_yolo_amd.utils.loss import _Assignment
    p = perm.to(a.fg_mask.device)
    out = _Assignment()
    out.n_max = a.n_max
    for k in ("target_gt_idx", "fg_mask", "norm", "target_label", "target_box"):
        setattr(out, k, getattr(a, k).index_select(0, p).contiguous())
    return out


def _perm_batch(batch, perm):
    inv = {int(p): i for i, p in enumerate(perm)}
    out = dict(batch)
    out["img"] = batch["img"][perm]
    out["batch_idx"] = torch.tensor([inv[int(b)] for b in batch["batch_idx"]], dtype=torch.float32)
    order = torch.argsort(out["batch_idx"], stable=True)
    for k in ("batch_idx", "cls", "bboxes"):
        out[k] = (out[k] if k == "batch_idx" else batch[k])[order]
    return out


def _full_size(yaml_name, scale, dtype, B, nbox, loss_tol, eval_tol, S=640):
    import dedark_yolo_amd as dy
    from parity_helpers import HYP
    from dedark_yolo_amd.nn.tasks import DetectionModel
    from util import load_yaml
    dy.set_compute_dtype(dtype)
    torch.manual_seed(0)
    cfg = load_yaml(yaml_name)
    cfg["scale"] = scale
    model = DetectionModel(cfg, nc=20).cuda().train()
    model.args = HYP
    batch = make_batch(7, B, S, nbox)
    batch["img"] = batch["img"].pow(2.0)
    perm = torch.roll(torch.arange(B), 1)

    def step(b, emulate=None):
        gb = dict(b)
        gb["img"] = b["img"].cuda()
        gb["recovery_loss_batch"] = torch.tensor(0.01, device="cuda")
        for p in model.parameters():
            p.grad = None
        # momentum 0 semantics: running stats must not drift between the two evaluations
        st = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
        with _emulated_storage(model, emulate):
            loss, items = model(gb)
            loss.backward()
            torch.cuda.synchronize()
        model.load_state_dict(st, strict=False)
        grads = {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}
        return float(loss), items.float().cpu(), grads

    l0, i0, g0 = step(batch)
    a0 = model.criterion.last_assignment
    l1, i1, g1 = step(_perm_batch(batch, perm))
    assert np.isfinite(l0) and torch.isfinite(i0).all() and all(torch.isfinite(v).all() for v in g0.values())
    assert abs(l0 - l1) <= loss_tol * abs(l0), (l0, l1)
    head = [k for k in g0 if k.endswith(".2.weight") and (".cv2." in k or ".cv3." in k)]
    early = [k for k in g0 if g0[k].numel() >= 4096][:3]
    worst = max(_rel(g1[k], g0[k]) for k in head)
    worst_early = max(_rel(g1[k], g0[k]) for k in early)
    print(f"{yaml_name}@{scale} {dtype} B={B}: loss {l0:.4f} / permuted {l1:.4f}; gradient change under permutation, free assignment: "
          f"head {worst:.2e}, first backbone convs {worst_early:.2e}")
    if dtype == torch.float32:
        assert head and worst <= 20 * loss_tol and worst_early <= 2e-2, (worst, worst_early)
    # 16 bit: one flipped rounding re-orders near-tied scores in the task-aligned top-k and re-targets anchors, so the free-running
    # numbers above are reported only.  With the ASSIGNMENT FROZEN (tal.py:84-132's discrete outcome of the first evaluation, rows
    # permuted with the batch) what is left is kernel arithmetic, and that is held to a bound in every dtype.
    frozen = _permute_assignment(a0, perm)
    g0f = None
    model.criterion.frozen_assignment = frozen
    try:
        l2, i2, g2 = step(_perm_batch(batch, perm))
        if dtype != torch.float32:
            model.criterion.frozen_assignment = a0
            _, _, g0f = step(batch)
    finally:
        model.criterion.frozen_assignment = None
    worst_f = max(_rel(g2[k], g0[k]) for k in head)
    worst_early_f = max(_rel(g2[k], g0[k]) for k in early)
    big = [k for k in g0 if g0[k].numel() >= 64 and float(g0[k].norm()) > 0]
    cos_min = min(_cos(g2[k], g0[k]) for k in big)
    print(f"   frozen assignment: loss {l2:.4f}; gradient change under permutation: head {worst_f:.2e}, first backbone convs "
          f"{worst_early_f:.2e}, worst per-tensor cosine {cos_min:.5f}")
    assert abs(l2 - l0) <= loss_tol * abs(l0), (l0, l2)
    if dtype == torch.float32:
        assert worst_f <= 20 * loss_tol and worst_early_f <= 2e-2 and cos_min >= 0.99, (worst_f, worst_early_f, cos_min)
    else:
        # yardstick: the same two evaluations on the fp32 kernels with 16-bit STORAGE (ops.set_storage_emulation): how much an ideal
        # implementation of this storage format moves under a batch permutation (this random-init graph amplifies a rounding by
        # four to five orders of magnitude on the way back to the first layers: fp32 itself moves by 5e-3 there)
        model.criterion.frozen_assignment = a0
        try:
            _, _, e0 = step(batch, emulate=dtype)
            model.criterion.frozen_assignment = frozen
            _, _, e1 = step(_perm_batch(batch, perm), emulate=dtype)
        finally:
            model.criterion.frozen_assignment = None
            dy.set_compute_dtype(dtype)
        emu_f = max(_rel(e1[k], e0[k]) for k in head)
        emu_early = max(_rel(e1[k], e0[k]) for k in early)
        emu_cos = min(_cos(e1[k], e0[k]) for k in big)
        med = lambda v: sorted(v)[len(v) // 2]
        p_med, e_med = med([_rel(g2[k], g0[k]) for k in big]), med([_rel(e1[k], e0[k]) for k in big])
        p_cmed, e_cmed = med([_cos(g2[k], g0[k]) for k in big]), med([_cos(e1[k], e0[k]) for k in big])
        # run-to-run: the same evaluation twice (f64-atomic order of the BatchNorm sums is the only free variable of a step)
        model.criterion.frozen_assignment = a0
        try:
            _, _, r1 = step(batch)
        finally:
            model.criterion.frozen_assignment = None
        rr = med([_rel(r1[k], g0f[k]) for k in big]) if g0f is not None else float("nan")
        print(f"   16-bit storage emulation on the fp32 kernels, same two evaluations: head {emu_f:.2e}, first backbone convs {emu_early:.2e}, "
              f"worst per-tensor cosine {emu_cos:.5f}; over all {len(big)} tensors: median change product {p_med:.3e} / emulation {e_med:.3e}, "
              f"median cosine {p_cmed:.4f} / {e_cmed:.4f}; product run-to-run median change {rr:.3e}")
        # single tensors of this graph are chaotic (heavy-tailed) under any rounding change, so the bound is on the population: the
        # product's median change and median cosine against the emulation's
        assert p_med <= 2.0 * e_med + 2e-2 and p_cmed >= e_cmed - 0.1, (p_med, e_med, p_cmed, e_cmed)
        assert worst_f <= 2.0 * emu_f + 5e-2, (worst_f, emu_f)
    model.eval()
    with torch.no_grad():
        ya, _ = model(batch["img"].cuda())
        other = batch["img"].clone()
        other[1:] = other[1:].flip(0).pow(1.5)          # same image 0, different companions
        yb, _ = model(other.cuda())
    A = sum((S // s) ** 2 for s in (8, 16, 32))
    assert ya.shape == (B, 24, A) and torch.isfinite(ya).all()
    e = _rel(yb[0].float(), ya[0].float())
    print(f"   eval prediction of image 0 with other companions: rel L2 change {e:.2e}")
    assert e <= eval_tol, e


def test_full_size_c1_fp32_properties():
    """BASELINE configs[0] (C1): plain YOLOv8n graph (yolov8ori.yaml@n), 640x640, batch 4, fp32."""
    _full_size("yolov8ori.yaml", "n", torch.float32, 4, [3, 1, 5, 2], 1e-4, 1e-5)


def test_full_size_c3_bf16_properties():
    """BASELINE configs[2] (C3) graph: repo yolov8.yaml@L (front-end + ASFF neck), 640x640, bf16, batch 8."""
    _full_size("yolov8.yaml", "l", torch.bfloat16, 8, [3, 1, 5, 2, 4, 2, 1, 6], 1e-2, 1e-5)


def test_full_size_c5_fp16_properties():
    """BASELINE configs[4] (C5) graph and size: repo yolov8.yaml@L at 1280x1280 in fp16 (batch 2 here; bench.py runs batch 16)."""
    _supported(torch.float16)
    _full_size("yolov8.yaml", "l", torch.float16, 2, [3, 5], 1e-2, 1e-5, S=1280)


def test_ciou_and_dfl_entries_on_reference_vectors():
    """dy_bbox_ciou / dy_dfl_loss (the device functions the fused loss kernels inline) on the reference's own vectors
    (tests/golden/g5_small.npz: bbox_iou(CIoU=True) + backward, BboxLoss._df_loss + backward)."""
    from dedark_yolo_amd.utils.loss import BboxLoss
    from dedark_yolo_amd.utils.metrics import bbox_iou
    from util import close
    g = gold("g5_small")
    b1 = g["b1"].clone().cuda().requires_grad_(True)
    v = bbox_iou(b1, g["b2"].cuda(), xywh=False, CIoU=True)
    assert v.shape == tuple(g["ciou"].shape) or v.reshape(-1).shape == g["ciou"].reshape(-1).shape
    close(v.detach().cpu().reshape(-1), g["ciou"].reshape(-1), 1e-5, 1e-6, "ciou")
    v.sum().backward()
    close(b1.grad.cpu(), g["dciou_db1"], 1e-4, 1e-6, "d ciou / d b1")
    pd = g["dfl_pred"].clone().cuda().requires_grad_(True)
    dl = BboxLoss._df_loss(pd, g["dfl_tgt"].cuda())
    close(dl.detach().cpu(), g["dfl"], 1e-5, 1e-6, "dfl")
    dl.sum().backward()
    close(pd.grad.cpu(), g["dfl_grad"], 1e-5, 1e-6, "d dfl / d logits")
    # every other mode of the public helper (metrics.py:75-128) on the reference's vectors: value and gradient wrt box1
    m = gold("g5_iou_modes")
    for tag, xywh in (("xywh", True), ("xyxy", False)):
        for kind, kw in (("iou", {}), ("giou", dict(GIoU=True)), ("diou", dict(DIoU=True)), ("ciou", dict(CIoU=True))):
            b1 = m[f"{tag}_b1"].clone().cuda().requires_grad_(True)
            v = bbox_iou(b1, m[f"{tag}_b2"].cuda(), xywh=xywh, **kw)
            close(v.detach().cpu(), m[f"{tag}_{kind}"], 1e-5, 1e-6, f"{tag} {kind}")
            v.sum().backward()
            close(b1.grad.cpu(), m[f"{tag}_{kind}_grad"], 2e-4, 1e-6, f"{tag} {kind} grad")
    # broadcasting like the reference's (1, 4) against (n, 4)
    one = bbox_iou(m["xyxy_b1"][:1].cuda(), m["xyxy_b2"].cuda(), xywh=False, GIoU=True)
    assert one.shape == (m["xyxy_b2"].shape[0], 1)


@pytest.mark.parametrize("c,hw,B", [(128, 80, 16), (256, 40, 32), (64, 160, 8)])
def test_c2f_shortcut_on_large_tile_kernels_vs_fp32(c, hw, B):
    """C2f with shortcut Bottlenecks at sizes where the data gradients run on conv_v5 / conv_v4: the shortcut gradient is added in the
    epilogue of cv1's data gradient (dy_conv_desc.add_src) together with the accumulation into C2f's gradient buffer.  Reference: the
    same module in fp32 (generic kernels; there the addend is a separate dy_copy2d pass behind the same call).  bf16 bounds of the
    block table above."""
    import dedark_yolo_amd as dy
    from dedark_yolo_amd import _C
    from dedark_yolo_amd.nn.modules import C2f
    torch.manual_seed(c)
    m = C2f(c, c, 2, True).cuda().train()
    x0 = torch.randn(B, c, hw, hw, device="cuda")
    gy = torch.randn(B, c, hw, hw, device="cuda")

    def run(dtype, log=False):
        dy.set_compute_dtype(dtype)
        for p in m.parameters():
            p.grad = None
        for b in m.modules():
            if isinstance(b, torch.nn.BatchNorm2d):
                b.reset_running_stats()
        x = x0.clone().requires_grad_(True)
        if log:
            _C._prof = []
        y = m(x)
        y.backward(gy.to(y.dtype))
        torch.cuda.synchronize()
        rec, _C._prof = _C._prof, None
        return y.detach().float(), x.grad.float(), {k: p.grad.float().clone() for k, p in m.named_parameters()}, rec

    try:
        y1, dx1, g1, rec = run(torch.bfloat16, log=True)
        y0, dx0, g0, _ = run(torch.float32)
    finally:
        _C._prof = None
    kerns = {r[4] for r in rec if r[0] == "dy_conv2d_dgrad"}
    assert any(k.startswith("v4::") or k.startswith("v5::") for k in kerns), kerns
    assert "dy_copy2d" not in [r[0] for r in rec], "the shortcut gradient must not need a copy pass on this path"
    e_y, e_dx = _rel(y1, y0), _rel(dx1, dx0)
    e_dp = max(_rel(g1[k], g0[k]) for k in g0)
    print(f"C2f({c}) {hw}x{hw} B={B}: forward {e_y:.2e} dx {e_dx:.2e} dparam {e_dp:.2e}; dgrad kernels {sorted(kerns)}")
    assert e_y <= 2e-2 and e_dx <= 3e-2 and e_dp <= 3e-2, (e_y, e_dx, e_dp)


def _bench_workload(S, B, dtype, want_kernels, seed=11):
    """The bench.py workload itself (repo yolov8.yaml@L, S x S, batch B, `dtype`): one training evaluation in fp32 on the HIP path
    (pinned to the reference goldens by test_gpu_parity.py), one in `dtype` with the fp32 run's ASSIGNMENT FROZEN (the discrete
    outcome of tal.py:84-132 and its target scores: what loss.py:173-187 consumes), so that the comparison sees kernel arithmetic
    only.  Asserts through dy_last_kernel (the symbol each C-ABI entry reports) that the `dtype` run went through the kernels the
    benchmark's step is made of, then returns maps / gradients of both runs."""
    import dedark_yolo_amd as dy
    from dedark_yolo_amd import _C
    from parity_helpers import HYP
    from dedark_yolo_amd.nn.tasks import DetectionModel
    from util import load_yaml
    cfg = load_yaml("yolov8.yaml")
    cfg["scale"] = "l"
    dy.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    model = DetectionModel(cfg, nc=20).cuda().train()
    model.args = HYP
    g = np.random.default_rng(seed)
    batch = make_batch(seed, B, S, [int(n) for n in g.integers(1, 9, B)])
    batch["img"] = batch["img"].pow(2.0).cuda()
    batch["recovery_loss_batch"] = torch.tensor(0.01, device="cuda")
    bn0 = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}

    def run(dt, frozen, log=False, emulate=None):
        dy.set_compute_dtype(dt)
        model.load_state_dict(bn0, strict=False)
        for p in model.parameters():
            p.grad = None
        if not hasattr(model, "criterion"):
            model.criterion = model.init_criterion()
        crit = model.criterion
        crit.frozen_assignment, crit.keep_maps = frozen, True
        if log:
            _C._prof = []
        try:
            with _emulated_storage(model, emulate):
                loss, items = model(dict(batch))
                loss.backward()
                torch.cuda.synchronize()
        finally:
            rec, _C._prof = _C._prof, None
            crit.frozen_assignment = None
        maps = [m.detach().float()[:, :64 + 20].clone() for m in crit.last_maps]
        grads = {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}
        crit.last_maps, crit.keep_maps = None, False
        return float(loss), maps, grads, crit.last_assignment, rec

    l32, m32, g32, a32, _ = run(torch.float32, None)
    l32f, _, g32f, _, _ = run(torch.float32, a32)                  # the hook itself: freezing its own assignment changes nothing
    assert abs(l32f - l32) <= 1e-6 * abs(l32) and max(_rel(g32f[k], g32[k]) for k in g32) <= 1e-5
    lem, mem, gem, _, _ = run(torch.float32, a32, emulate=dtype)   # ideal 16-bit storage on the fp32 kernels
    l16, m16, g16, _, rec = run(dtype, a32, log=True)
    kerns = {r[4] for r in rec if r[4]}
    print(f"kernels of the {dtype} step: {sorted(kerns)}")
    for w in want_kernels:
        assert any(w in k for k in kerns), f"{w} did not run at B={B}, {S}x{S} ({sorted(kerns)})"
    dy.set_compute_dtype(torch.float32)
    return (l32, m32, g32), (lem, mem, gem), (l16, m16, g16)


def _report_and_bound(tag, ref, emu, got):
    """Product in 16 bit against the fp32 HIP path, with the 16-bit storage emulation as the yardstick (same frozen assignment in
    all three).  Bounds: Detect maps and the per-tensor gradient errors no worse than 1.5 x the emulation's (+ a floor); every tensor
    the emulation itself keeps at cosine >= 0.995 with fp32 must stay at >= 0.99 in the product (the well-conditioned part of the
    graph: on the rest fp32 is not a yardstick for ANY 16-bit implementation, see test_l_graph_low_precision_vs_fp32_oracle)."""
    (l32, m32, g32), (lem, mem, gem), (l16, m16, g16) = ref, emu, got
    e_maps, e_maps_emu = max(_rel(a, b) for a, b in zip(m16, m32)), max(_rel(a, b) for a, b in zip(mem, m32))
    keys = [k for k in g32 if k in g16 and g32[k].numel() >= 64 and float(g32[k].norm()) > 0 and not k.startswith("model.0.")]
    cos_p = {k: _cos(g16[k], g32[k]) for k in keys}
    cos_e = {k: _cos(gem[k], g32[k]) for k in keys}
    rel_p = sorted(_rel(g16[k], g32[k]) for k in keys)
    rel_e = sorted(_rel(gem[k], g32[k]) for k in keys)
    med = lambda v: sorted(v)[len(v) // 2]
    good = [k for k in keys if cos_e[k] >= 0.995]
    worst_good = min((cos_p[k], k) for k in good) if good else (1.0, "-")
    print(f"{tag}: loss fp32 {l32:.4f} / emulation {lem:.4f} / product {l16:.4f}; Detect maps rel L2: product {e_maps:.3e}, emulation "
          f"{e_maps_emu:.3e}; parameter gradients ({len(keys)} tensors) vs fp32: cosine median product {med(cos_p.values()):.4f} / emulation "
          f"{med(cos_e.values()):.4f}, worst {min(cos_p.values()):.4f} / {min(cos_e.values()):.4f}; rel L2 median {med(rel_p):.3e} / "
          f"{med(rel_e):.3e}, worst {rel_p[-1]:.3e} / {rel_e[-1]:.3e}; {len(good)} tensors where the emulation keeps cosine >= 0.995: "
          f"product worst {worst_good[0]:.5f} ({worst_good[1]})")
    assert np.isfinite(l16) and abs(l16 - l32) <= max(0.05 * abs(l32), 1.5 * abs(lem - l32)), (l16, lem, l32)
    assert e_maps <= 1.5 * e_maps_emu + 1e-2, (e_maps, e_maps_emu)
    assert med(rel_p) <= 1.5 * med(rel_e) + 2e-2 and rel_p[-1] <= 1.5 * rel_e[-1] + 5e-2, (med(rel_p), med(rel_e), rel_p[-1], rel_e[-1])
    assert med(cos_p.values()) >= med(cos_e.values()) - 0.1, (med(cos_p.values()), med(cos_e.values()))
    assert worst_good[0] >= 0.99, worst_good


def test_c3_bench_workload_routed_kernels_bf16_vs_fp32_frozen_assignment():
    """BASELINE configs[2] exactly as bench.py runs it: L graph, 640x640, batch 64, bf16.  At this batch the 256->256@40x40 layers
    are 400 tiles (conv_v4's >= 192-tile rule), the weight gradients take wgrad_v4 / wgrad_v3 and the 128- / 64-channel 3x3 layers
    the band kernels -- none of which the batch-8 property test reaches."""
    ref, emu, got = _bench_workload(640, 64, torch.bfloat16,
                               ("v4::conv_kernel", "wg4::wgrad_kernel", "wg3::wgrad_kernel", "v5::band_kernel<128", "v5::band_kernel<64",
                                "v5::conv_kernel"))
    _report_and_bound("C3 B=64 bf16 vs fp32 (frozen assignment)", ref, emu, got)


def test_c5_bench_workload_routed_kernels_fp16_vs_fp32_frozen_assignment():
    """BASELINE configs[4] as bench.py --imgsz 1280 --batch 16 --dtype fp16 runs it."""
    _supported(torch.float16)
    ref, emu, got = _bench_workload(1280, 16, torch.float16, ("v4::conv_kernel", "wg4::wgrad_kernel", "v5::band_kernel<128", "v5::band_kernel<64"))
    _report_and_bound("C5 B=16 fp16 vs fp32 (frozen assignment)", ref, emu, got)
