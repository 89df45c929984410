"""GPU parity tests of the caller contract around the network: preprocess_batch against the oracle,
and the fused clip + SGD(nesterov)/AdamW + EMA kernels (plus gradient accumulation) against torch.optim on the same
gradients -- the reference's optimizer_step is clip_grad_norm_(10.0) -> optimizer.step() -> ema.update()
(ultralytics/engine/trainer.py:459-467, build_optimizer :611-665, ModelEMA torch_utils.py:344-377)."""
import math

import numpy as np
import pytest
import torch

from util import load_yaml

pytestmark = pytest.mark.gpu


def _tiny_trainer(optimizer, batch=64, dtype="fp32", **over):
    import dedark_yolo_amd as dy
    from dedark_yolo_amd.engine.trainer import DetectionTrainer, get_cfg
    from dedark_yolo_amd.nn.tasks import DetectionModel
    dy.set_compute_dtype(torch.float32)
    cfgd = load_yaml("yolov8-lowlight.yaml")
    cfgd["scales"]["t"] = [0.33, 0.125, 1024]
    cfgd["scale"] = "t"
    torch.manual_seed(3)
    cfg = get_cfg(dict(model="tiny", dtype=dtype, optimizer=optimizer, batch=batch, lowlight_FLAG=True, dedark_FLAG=True, **over))
    tr = DetectionTrainer(cfg)
    tr.setup(DetectionModel(cfgd, nc=20))
    return tr


@pytest.mark.parametrize("low,ded", [(True, True), (True, False), (False, False)])
def test_preprocess_batch_vs_oracle(low, ded):
    from oracle import loss as oloss
    tr = _tiny_trainer("SGD")
    tr.args.lowlight_FLAG, tr.args.dedark_FLAG, tr.args.dark_param = low, ded, 7.3
    g = np.random.default_rng(0)
    img = torch.from_numpy(g.integers(0, 256, (3, 3, 40, 56), dtype=np.uint8))       # 20,160 pixels: vector body + scalar tail
    out = tr.preprocess_batch(dict(img=img.clone()))
    w_img, w_clean, w_rec = oloss.preprocess_batch(img, 7.3, low, ded)
    assert float((out["img"].cpu() - w_img).abs().max()) <= 1e-6
    assert float((out["clean_img"].cpu() - w_clean).abs().max()) <= 1e-6
    assert abs(float(out["recovery_loss_batch"]) - float(w_rec)) <= 1e-6 * max(1.0, float(w_rec))


def _reference_optimizer(tr, name, lr0, momentum, wd):
    flat = tr.flat
    refs, groups = [], {0: [], 1: [], 2: []}
    for p, off, n, gid in flat.slots:
        q = flat.p[off:off + n].detach().clone().view(p.shape).requires_grad_(True)
        refs.append((q, off, n))
        groups[gid].append(q)
    pg = [dict(params=groups[2], weight_decay=0.0), dict(params=groups[0], weight_decay=wd), dict(params=groups[1], weight_decay=0.0)]
    if name == "SGD":
        opt = torch.optim.SGD(pg, lr=lr0, momentum=momentum, nesterov=True)
    else:
        opt = torch.optim.AdamW(pg, lr=lr0, betas=(momentum, 0.999))
    return refs, opt


@pytest.mark.parametrize("name", ["SGD", "AdamW"])
def test_fused_optimizer_clip_ema_vs_torch(name):
    tr = _tiny_trainer(name)
    flat = tr.flat
    assert tr.accumulate == 1 and abs(tr.weight_decay - tr.args.weight_decay) < 1e-12
    refs, opt = _reference_optimizer(tr, name, tr.lr0, tr.momentum, tr.weight_decay)
    ema = flat.p.detach().clone()
    gen = torch.Generator(device="cuda").manual_seed(5)
    for step in range(1, 5):
        scale = [0.02, 3.0, 0.5, 40.0][step - 1]                    # steps 2 and 4 exceed max_norm = 10 -> clipping is active
        g = torch.randn(flat.n, generator=gen, device="cuda") * scale / math.sqrt(flat.n) * 10
        flat.g.copy_(g)
        lr = [tr.lr0 * 0.7, tr.lr0 * 0.9, tr.lr0 * 1.3] if step % 2 else [tr.lr0] * 3     # per-group lr (warm-up uses it)
        mom = 0.85 if step == 1 else tr.momentum
        tr.optimizer_step(lr, mom)
        # --- torch reference on the same gradient
        for q, off, n in refs:
            q.grad = g[off:off + n].view(q.shape).clone()
        torch.nn.utils.clip_grad_norm_([q for q, _, _ in refs], max_norm=10.0)
        for grp, l in zip(opt.param_groups, (lr[2], lr[0], lr[1])):                        # torch group order: bias, decayed, norm
            grp["lr"] = l
            if name == "SGD":
                grp["momentum"] = mom
            else:
                grp["betas"] = (mom, 0.999)
        opt.step()
        d = 0.9999 * (1 - math.exp(-step / 2000))
        for q, off, n in refs:
            ema[off:off + n].mul_(d).add_((1 - d) * q.detach().reshape(-1))
        torch.cuda.synchronize()
        for q, off, n in refs:
            got, want = flat.p[off:off + n], q.detach().reshape(-1)
            assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max())), (name, step)
            assert float((flat.ema[off:off + n] - ema[off:off + n]).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))


def test_gradient_accumulation_and_weight_decay_scaling():
    """batch 16 with nbs 64: accumulate = 4 and weight_decay * 16 * 4 / 64 (trainer.py:248-249); the optimizer sees the SUM of
    the accumulated gradients (the reference calls backward() repeatedly before optimizer.step())."""
    tr = _tiny_trainer("SGD", batch=16)
    assert tr.accumulate == 4 and abs(tr.weight_decay - tr.args.weight_decay * 16 * 4 / 64) < 1e-12
    tr2 = _tiny_trainer("SGD", batch=24)                                # round(64 / 24) = 3 -> wd * 24 * 3 / 64
    assert tr2.accumulate == 3 and abs(tr2.weight_decay - tr2.args.weight_decay * 72 / 64) < 1e-12
    flat = tr.flat
    refs, opt = _reference_optimizer(tr, "SGD", tr.lr0, tr.momentum, tr.weight_decay)
    gen = torch.Generator(device="cuda").manual_seed(9)
    gs = [torch.randn(flat.n, generator=gen, device="cuda") * 0.01 for _ in range(3)]
    for g in gs[:2]:
        flat.g.copy_(g)
        tr.accumulate_gradients()
    flat.g.copy_(gs[2])
    tr.optimizer_step([tr.lr0] * 3, tr.momentum)
    total = gs[0] + gs[1] + gs[2]
    for q, off, n in refs:
        q.grad = total[off:off + n].view(q.shape).clone()
    torch.nn.utils.clip_grad_norm_([q for q, _, _ in refs], max_norm=10.0)
    opt.step()
    torch.cuda.synchronize()
    for q, off, n in refs:
        want = q.detach().reshape(-1)
        assert float((flat.p[off:off + n] - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
    assert tr.acc_count == 0 and float(flat.g_acc.abs().max()) == 0.0


def test_train_loop_steps_optimizer_every_accumulate_batches():
    """train(): with batch 32 / nbs 64 and no warm-up the optimizer runs on every second batch (trainer.py:340-342)."""
    import bench
    tr = _tiny_trainer("SGD", batch=32, warmup_epochs=0.0, epochs=1, imgsz=64)
    batches = [bench.synth_batch(40 + i, 2, 64, 20, "cuda") for i in range(4)]
    for b in batches:
        b.pop("gamma"), b.pop("n_max", None)
    before = tr.updates
    hist = tr.train(batches, epochs=1)
    assert tr.updates - before == 2 and len(hist) == 1 and all(np.isfinite(hist[0]))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_training_reduces_the_loss_on_a_fixed_batch(dtype):
    """End-to-end sanity of the whole path (front-end, network, assigner, loss, backward, clip, SGD, EMA, weight re-pack):
    120 optimizer steps on one fixed synthetic batch must drive the loss down substantially in both compute dtypes (the first
    tens of steps are noisy: batch-4 BatchNorm and a moving assignment; measured 24 -> 7.4 in fp32)."""
    import bench
    import dedark_yolo_amd as dy
    tr = _tiny_trainer("SGD", batch=64, dtype=dtype)       # fp16 brings the dynamic loss scale with it
    try:
        b = bench.synth_batch(77, 4, 96, 20, "cuda")
        tr.args.dark_param = b.pop("gamma")
        b.pop("n_max", None)
        losses = []
        for _ in range(120):
            loss, _ = tr.train_step(dict(b), [0.01] * 3, 0.9)
            losses.append(float(loss))
        assert all(np.isfinite(losses)), losses
        first, last = float(np.mean(losses[:10])), float(np.mean(losses[-10:]))
        assert last < 0.6 * first, (first, last)
        if dtype == "fp16":
            scale, good, _ = [float(v) for v in tr.loss_scale]
            assert np.isfinite(scale) and 1.0 <= scale <= 65536.0 and good >= 1, (scale, good)
    finally:
        dy.set_compute_dtype(torch.float32)


def test_fp16_loss_scale_follows_gradscaler():
    """The device-side loss scale of the fp16 path against torch.cuda.amp.GradScaler's rules (the reference's AMP,
    ultralytics/engine/trainer.py:221,459-467): an inf in the scaled gradients skips the update (parameters and momentum
    untouched, EMA still updated), halves the scale and resets the counter; finite steps unscale before clipping and count up;
    `growth_interval` finite steps double the scale."""
    import dedark_yolo_amd as dy
    from dedark_yolo_amd._C import call
    from dedark_yolo_amd.ops import ptr, stream
    tr = _tiny_trainer("SGD", batch=64, dtype="fp16")
    try:
        f = tr.flat
        gen = torch.Generator(device="cuda").manual_seed(5)
        g_true = torch.randn(f.n, device="cuda", generator=gen) * 1e-3
        # 1) overflow
        f.g.copy_(g_true * 65536.0)
        f.g[7] = float("inf")
        p0, m0, e0 = f.p.clone(), f.m.clone(), f.ema.clone()
        tr.optimizer_step([0.01] * 3, 0.9)
        torch.cuda.synchronize()
        assert torch.equal(f.p, p0) and torch.equal(f.m, m0), "an overflowed step must not touch parameters / momentum"
        d = 0.9999 * (1 - math.exp(-tr.updates / 2000))
        assert torch.allclose(f.ema, d * e0 + (1 - d) * p0, rtol=1e-6, atol=1e-7)
        assert [float(v) for v in tr.loss_scale] == [32768.0, 0.0, 1.0]
        # 2) finite step at scale 32768 == the unscaled step of the plain entry on a twin state
        f.g.copy_(g_true * 32768.0)
        twin_p, twin_m = f.p.clone(), f.m.clone()
        ss = torch.zeros(1, dtype=torch.float64, device="cuda")
        call("dy_sumsq", ptr(g_true), f.n, ptr(ss), stream())
        call("dy_sgd_step", ptr(twin_p), ptr(g_true), ptr(twin_m), None, ptr(f.gid), 0.01, 0.01, 0.01, tr.weight_decay, 0.0, 0.0, 0.9, 1,
             0.0, ptr(ss), 10.0, 1.0, f.n, stream())
        tr.optimizer_step([0.01] * 3, 0.9)
        torch.cuda.synchronize()
        assert torch.allclose(f.p, twin_p, rtol=1e-5, atol=1e-8) and torch.allclose(f.m, twin_m, rtol=1e-5, atol=1e-8)
        assert [float(v) for v in tr.loss_scale] == [32768.0, 1.0, 1.0]
        # 3) growth after `interval` finite steps (the update entry alone, interval 3)
        st = torch.tensor([1024.0, 0.0, 0.0], device="cuda")
        for want in ([1024.0, 1.0, 0.0], [1024.0, 2.0, 0.0], [2048.0, 0.0, 0.0]):
            call("dy_loss_scale_update", ptr(st), ptr(ss), 2.0, 0.5, 3, stream())
            assert [float(v) for v in st] == want
        # 4) AdamW: the bias correction does not advance on a skipped step (torch: scaler.step() leaves optimizer.step() uncalled)
        n = 1024
        gen2 = torch.Generator(device="cuda").manual_seed(6)
        g1 = torch.randn(n, device="cuda", generator=gen2) * 1e-3
        ref_p = torch.nn.Parameter(torch.ones(n, device="cuda"))
        opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
        p = torch.ones(n, device="cuda")
        m1, m2 = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        ls = torch.tensor([4.0, 0.0, 0.0], device="cuda")
        ssq = torch.zeros(1, dtype=torch.float64, device="cuda")
        for host_step, overflow in ((1, True), (2, False), (3, True), (4, False)):      # the HOST counter advances every time
            gs = g1 * float(ls[0])
            if overflow:
                gs = gs.clone()
                gs[3] = float("nan")
            ssq.zero_()
            call("dy_sumsq", ptr(gs), n, ptr(ssq), stream())
            call("dy_adamw_step_scaled", ptr(p), ptr(gs), ptr(m1), ptr(m2), None, None, 1e-3, 1e-3, 1e-3, 0.0, 0.0, 0.0, 0.9, 0.999, 1e-8,
                 host_step, 0.0, ptr(ssq), 1e9, 1.0, ptr(ls), n, stream())
            call("dy_loss_scale_update", ptr(ls), ptr(ssq), 2.0, 0.5, 2000, stream())
            if not overflow:
                ref_p.grad = g1.clone()
                opt.step()
        torch.cuda.synchronize()
        assert float(ls[2]) == 2.0 and torch.allclose(p, ref_p.detach(), rtol=1e-6, atol=1e-7)
    finally:
        dy.set_compute_dtype(torch.float32)


def test_side_streams_do_not_change_the_step():
    """Weight gradients on the second stream + Detect levels on branch streams against everything on one stream: same start
    state, same batches, three optimizer steps.  The schedules differ only in WHEN kernels run; the per-channel f64 atomics make
    neither of them bit-reproducible, so the comparison is at f32 round-off (the gradients themselves are the same sums)."""
    import bench
    from dedark_yolo_amd import ops

    def run(on):
        tr = _tiny_trainer("SGD", batch=64)
        ops.enable_wgrad_stream(on)
        ops.enable_branch_streams(on)
        losses = []
        for i in range(3):
            b = bench.synth_batch(50 + i, 4, 96, 20, "cuda")
            tr.args.dark_param = b.pop("gamma")
            b.pop("n_max", None)
            loss, _ = tr.train_step(b, [0.01] * 3, 0.9)
            losses.append(float(loss))
        torch.cuda.synchronize()
        return losses, tr.flat.p.detach().clone(), tr.flat.g.detach().clone()

    try:
        l1, p1, g1 = run(True)
        import os
        if os.environ.get("DY_WGRAD_STREAM", "1") != "0":
            assert ops.wgrad_side_stream() is not None                  # the side stream really was in use
        l0, p0, g0 = run(False)
    finally:
        ops.enable_wgrad_stream(True)
        ops.enable_branch_streams(True)
    assert np.allclose(l1, l0, rtol=2e-5, atol=1e-5), (l1, l0)
    assert float((p1 - p0).abs().max()) <= 1e-5 * max(1.0, float(p0.abs().max()))
    assert float((g1 - g0).abs().max()) <= 2e-4 * max(1e-3, float(g0.abs().max()))


def test_checkpoint_save_and_resume(tmp_path):
    """save_model writes last.pt / best.pt in the reference's own format (pickled module objects under its class paths, torch.optim
    style optimizer state); resume_training restores
    parameters (through the reference's half-precision round trip), EMA, optimizer buffers and counters, and `YOLO(last.pt)`
    reads the same file."""
    import bench
    from dedark_yolo_amd.engine.model import YOLO
    tr = _tiny_trainer("SGD", batch=64)
    for i in range(3):
        b = bench.synth_batch(60 + i, 4, 96, 20, "cuda")
        tr.args.dark_param = b.pop("gamma")
        b.pop("n_max", None)
        tr.train_step(b, [0.01] * 3, 0.9)
    last = tr.save_model(str(tmp_path), epoch=4, fitness=0.25)
    from dedark_yolo_amd.utils.checkpoint import load_checkpoint, load_raw
    raw = load_raw(last)                                    # pickled module objects under the reference's class paths
    for k in ("epoch", "best_fitness", "model", "ema", "updates", "optimizer", "train_args", "date", "version"):
        assert k in raw, k
    assert type(raw["model"]).__name__ == "DetectionModel" and type(raw["model"])._dy_module == "ultralytics.nn.tasks"
    ck = load_checkpoint(last)
    assert ck.source == "reference-pickle" and ck.epoch == 4 and ck.best_fitness == 0.25 and (tmp_path / "best.pt").exists()
    assert set(ck.optimizer) == {"state", "param_groups"} and len(ck.optimizer["param_groups"]) == 3
    assert sum(len(g["params"]) for g in ck.optimizer["param_groups"]) == len(tr.flat.slots)
    msd = ck.model_sd
    assert list(msd) == list(tr.model.state_dict())
    tr2 = _tiny_trainer("SGD", batch=64)
    assert tr2.resume_training(last) == 5
    torch.cuda.synchronize()
    want_p = tr.flat.p.half().float()
    assert float((tr2.flat.p - want_p).abs().max()) == 0.0
    assert float((tr2.flat.ema - tr.flat.ema.half().float()).abs().max()) == 0.0
    assert float((tr2.flat.buf_flat - tr.flat.buf_flat.half().float()).abs().max()) == 0.0
    assert torch.equal(tr2.flat.m, tr.flat.m) and tr2.updates == tr.updates and tr2.step_count == tr.step_count
    b = bench.synth_batch(70, 4, 96, 20, "cuda")
    tr2.args.dark_param = b.pop("gamma")
    b.pop("n_max", None)
    loss, _ = tr2.train_step(b, [0.01] * 3, 0.9)
    assert np.isfinite(float(loss))
    y = YOLO(last)                                          # the predictor reads the same file (weights_only load)
    sd = y.model.state_dict()
    k0 = next(k for k in sd if k.endswith("conv.weight"))
    assert float((sd[k0].cpu() - ck.state_dict[k0]).abs().max()) == 0.0           # EMA weights first (nn/tasks.py:640,682)
    assert float((ck.state_dict[k0] - ck.model_sd[k0]).abs().max()) > 0.0


def test_train_loop_uploads_host_batches_through_the_prefetcher():
    """train() on host-resident batch dicts (pinned uint8 images + targets, what a dataloader yields) gives the same loss items as
    on device-resident copies: the one-ahead upload on the copy stream hands complete tensors to the compute stream."""
    import bench

    def run(host):
        tr = _tiny_trainer("SGD", batch=64, warmup_epochs=0.0, epochs=1, imgsz=64)
        batches = [bench.synth_batch(90 + i, 2, 64, 20, "cuda") for i in range(5)]
        for b in batches:
            b.pop("gamma"), b.pop("n_max", None)
        if host:
            batches = [{k: (v.cpu().pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]
        return tr.train(batches, epochs=2)

    h, d = run(True), run(False)
    # (two runs of the same ten optimizer steps differ at 1e-4..1e-3 relative: f64-atomic order + batch-2 BatchNorm; a batch that
    #  was consumed before its upload finished would be off by O(1))
    assert np.allclose(np.array(h), np.array(d), rtol=1e-2, atol=1e-5), (h, d)


def test_yolo_reads_a_reference_checkpoint_and_predicts():
    """YOLO('<reference last.pt>'): pickled reference modules -> state_dict -> HIP model; the EMA weights are the ones loaded
    (nn/tasks.py:640,682) and the predictor runs on them."""
    import os
    from dedark_yolo_amd.engine.model import YOLO
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    y = YOLO(os.path.join(root, "tests", "golden", "g7_ref_last.pt"))
    z = np.load(os.path.join(root, "tests", "golden", "g7_ckpt.npz"))
    sd = y.model.state_dict()
    keys = [str(k) for k in z["keys"]]
    got = np.array([float(sd[k].double().sum()) for k in keys if not k.endswith("num_batches_tracked")])
    want = np.array([s for k, s in zip(keys, z["sum_ema"]) if not k.endswith("num_batches_tracked")])
    assert np.abs(got - want).max() == 0.0
    y.model.cuda()
    img = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (2, 3, 64, 64), dtype=np.uint8)).cuda()
    res = y.predict(img, conf=0.001)
    assert len(res) == 2 and all(r.boxes.data.shape[1] == 6 for r in res)


def test_preprocess_batch_vs_reference_fixture():
    """HIP preprocess kernel against the values captured from the reference's DetectionTrainer.preprocess_batch (g8_preprocess.npz)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(root, "tests", "golden", "g8_preprocess.npz"))
    tr = _tiny_trainer("SGD")
    for tag, low, ded in (("both", True, True), ("low", True, False), ("none", False, False)):
        tr.args.lowlight_FLAG, tr.args.dedark_FLAG, tr.args.dark_param = low, ded, float(z["dark_param"])
        out = tr.preprocess_batch(dict(img=torch.from_numpy(z["u8"]).clone()))
        assert float((out["img"].cpu() - torch.from_numpy(z[f"{tag}_img"])).abs().max()) <= 1e-6, tag
        assert float((out["clean_img"].cpu() - torch.from_numpy(z[f"{tag}_clean"])).abs().max()) <= 1e-6, tag
        assert abs(float(out["recovery_loss_batch"]) - float(z[f"{tag}_rec"])) <= 1e-6, tag


def test_arena_growth_inside_branch_streams():
    """The statistics arena has to grow in the middle of a pass when it is too small (always in the first step).  Force that to
    happen inside Detect's branch streams and in the side-stream weight gradients (a 64-double arena: every request overflows) and
    compare with the single-stream run: a chunk that were zero-filled on a branch stream, or handed out before its fill was ordered
    for the other streams, would corrupt the BatchNorm batch statistics."""
    import bench
    from dedark_yolo_amd import ops

    def run(on):
        tr = _tiny_trainer("SGD", batch=64)
        ops.enable_wgrad_stream(on)
        ops.enable_branch_streams(on)
        out = []
        for i in range(2):
            ops.arena.buf = torch.zeros(64, dtype=torch.float64, device="cuda")      # too small for any layer
            ops.arena.off, ops.arena.used, ops.arena.chunks = 0, 0, 1
            b = bench.synth_batch(60 + i, 4, 96, 20, "cuda")
            tr.args.dark_param = b.pop("gamma")
            loss, _ = tr.train_step(b, [0.01] * 3, 0.9)
            out.append(float(loss))
        torch.cuda.synchronize()
        return out, tr.flat.buf_flat.detach().clone(), tr.flat.g.detach().clone()

    try:
        l1, b1, g1 = run(True)
        l0, b0, g0 = run(False)
    finally:
        ops.enable_wgrad_stream(True)
        ops.enable_branch_streams(True)
    assert np.allclose(l1, l0, rtol=2e-5, atol=1e-5), (l1, l0)
    assert float((b1 - b0).abs().max()) <= 1e-5 * max(1.0, float(b0.abs().max()))          # BatchNorm running statistics
    assert float((g1 - g0).abs().max()) <= 2e-4 * max(1e-3, float(g0.abs().max()))


def test_reference_initial_bn_buffers_on_gpu():
    """DetectionModel.reference_initial_buffers() against the buffers of a freshly constructed reference model
    (tests/golden/g10_initbuf.npz: its constructor's two zero-image probes, tasks.py:284-292)."""
    import dedark_yolo_amd as dy
    from dedark_yolo_amd.nn.tasks import DetectionModel
    from util import close, gold
    dy.set_compute_dtype(torch.float32)
    g = gold("g10_initbuf")
    cfg = load_yaml("yolov8ori.yaml")
    cfg["scales"]["t"] = [0.33, 0.0625, 1024]
    cfg["scale"] = "t"
    m = DetectionModel(cfg, nc=4)
    own = m.state_dict()
    assert set(own) == set(g)
    m.load_state_dict({k: (own[k] if ("running_" in k or "num_batches" in k) else g[k]) for k in own}, strict=True)
    m = m.cuda()
    eps0 = {id(b): (b.eps, b.momentum) for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d)}
    m.reference_initial_buffers()
    torch.cuda.synchronize()
    after = m.state_dict()
    for k, v in g.items():
        if k.endswith("num_batches_tracked"):
            assert int(after[k]) == 2, k
        elif "running_" in k:
            close(after[k].cpu(), v, 1e-4, 1e-6, k)
    assert all((b.eps, b.momentum) == eps0[id(b)] == (1e-3, 0.03) for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d))


def test_train_loop_validates_on_ema_weights_and_tracks_best(tmp_path):
    """The epoch contract of BaseTrainer._do_train (engine/trainer.py:366-380, 408-433): after every epoch validate the EMA weights
    in fp32 (validator.py:105-107), track fitness / best, write last.pt (+ best.pt).  Checked: the metrics are those of the
    EMA weights (a validator run on a model loaded from the checkpoint's `ema` gives the same numbers), the live parameters are
    back in place afterwards, the compute dtype is restored, and a worse epoch does not overwrite best.pt."""
    import bench
    import dedark_yolo_amd as dy
    from dedark_yolo_amd.engine.model import YOLO
    from dedark_yolo_amd.engine.validator import DetectionValidator
    tr = _tiny_trainer("SGD", batch=64, dtype="bf16", warmup_epochs=0.0, epochs=2, imgsz=64, conf=0.001, iou=0.7)
    try:
        batches = [bench.synth_batch(90 + i, 4, 64, 20, "cuda") for i in range(3)]
        for b in batches:
            b.pop("gamma"), b.pop("n_max", None)
        vb = bench.synth_batch(99, 4, 64, 20, "cpu")
        vb.pop("gamma"), vb.pop("n_max", None)
        vb["ori_shape"] = [(64, 64)] * 4
        hist = tr.train(batches, epochs=2, val_loader=[vb], save_dir=tmp_path)
        assert len(hist) == 2 and tr.fitness is not None and set(tr.metrics) >= {"metrics/mAP50(B)", "metrics/mAP50-95(B)", "fitness"}
        assert dy.get_compute_dtype() == torch.bfloat16 and tr.model.training
        last, best = tmp_path / "weights" / "last.pt", tmp_path / "weights" / "best.pt"
        assert last.exists() and best.exists()
        from dedark_yolo_amd.utils.checkpoint import load_checkpoint
        ck = load_checkpoint(str(last))                      # the reference's format: pickled module objects (restricted reader)
        assert ck.epoch == 1 and ck.best_fitness == tr.best_fitness >= tr.fitness
        # live parameters are not the EMA ones (they moved by SGD steps; the EMA barely moved) and were restored after validation
        live = tr.flat.p.detach().clone()
        assert float((live - tr.flat.ema).abs().max()) > 0
        m2, f2 = tr.validate([vb])
        assert torch.equal(tr.flat.p, live) and abs(f2 - tr.fitness) <= 1e-6
        # the same numbers from the checkpoint's EMA weights through the public loader
        dy.set_compute_dtype(torch.float32)
        y = YOLO(str(last))
        got = DetectionValidator(tr.args)(y.model.cuda(), [vb], dtype=torch.float32)
        assert abs(got["fitness"] - tr.fitness) <= 2e-3, (got["fitness"], tr.fitness)        # checkpoint weights are fp16
    finally:
        dy.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_multi_pack_equals_single_pack(dtype):
    """dy_pack_weights_multi (one launch re-packing every weight of the model after the optimizer step: LDS-tiled transposes, both
    layouts, all dtypes, 1x1 / 3x3 / the extractor's 8x8 window, ragged channel counts) against dy_pack_weight of each weight."""
    import bench
    from dedark_yolo_amd import ops
    from dedark_yolo_amd._C import call
    from dedark_yolo_amd.ops import ptr, stream
    tr = _tiny_trainer("SGD", batch=64, dtype=dtype)
    try:
        b = bench.synth_batch(31, 2, 64, 20, "cuda")
        tr.args.dark_param = b.pop("gamma")
        b.pop("n_max", None)
        tr.train_step(b, [0.01] * 3, 0.9)                    # creates the packed copies (both layouts) and runs the multi-pack
        torch.cuda.synchronize()
        ent = tr.pack_plan._collect(tr.model)
        assert len(ent) > 40
        seen = set()
        for w, (cout_pad, cin_pad, transposed, dt), out in ent:
            Co, Ci, KH, KW = w.shape
            ref = torch.empty_like(out)
            call("dy_pack_weight", ptr(w.detach().float().contiguous()), ptr(ref), Co, cout_pad, Ci, cin_pad, KH, KW, 1 if transposed else 0,
                 ops.dt_id(dt), stream())
            torch.cuda.synchronize()
            assert torch.equal(out.view(torch.int16 if out.element_size() == 2 else torch.int32),
                               ref.view(torch.int16 if ref.element_size() == 2 else torch.int32)), (tuple(w.shape), cout_pad, cin_pad, transposed)
            seen.add((KH, bool(transposed)))
        assert {(1, False), (3, False), (3, True), (8, False)} <= seen, seen
    finally:
        import dedark_yolo_amd as dy
        dy.set_compute_dtype(torch.float32)


def test_n_max_of_never_inherits_another_batch():
    """A fresh device-resident batch_idx normally lands at the address the previous one had (caching allocator) with the same
    element count and version 0: n_max must be recomputed for it (it sizes the ground-truth table of the criterion)."""
    from dedark_yolo_amd.utils.loss import n_max_of
    a = torch.tensor([0., 0, 1, 1, 2, 2], device="cuda")
    assert n_max_of(a, 3) == 2
    pa = a.data_ptr()
    del a
    b = torch.tensor([0., 0, 0, 0, 0, 1], device="cuda")
    assert b.data_ptr() == pa or True                   # usually the same block; the result must not depend on it
    assert n_max_of(b, 3) == 5
    assert n_max_of(b, 3) == 5                          # remembered for this object
    b[5] = 0                                            # in-place change bumps the version
    assert n_max_of(b, 3) == 6


def test_freezing_parameters_between_steps_is_noticed():
    """The graph's autograd node is fed the trainable parameters; their requires_grad flags are read every step (from a cached tuple of
    the Parameter objects), so freezing / unfreezing a layer after model.train() takes effect at the next backward."""
    import dedark_yolo_amd as dy
    from parity_helpers import build_models, make_batch
    dy.set_compute_dtype(torch.float32)
    model, _ = build_models("yolov8-lowlight.yaml", "t", (0.33, 0.125, 1024), 21)
    batch = make_batch(22, 2, 64, [2, 3])
    gb = dict(batch)
    gb["img"] = batch["img"].pow(3.0).cuda()
    gb["recovery_loss_batch"] = torch.tensor(0.01, device="cuda")
    model.train()
    w = model.model[1].conv.weight                     # the stem conv behind the front-end
    other = model.model[2].conv.weight

    def grads():
        model.zero_grad(set_to_none=True)
        loss, _ = model(dict(gb))
        loss.backward()
        torch.cuda.synchronize()
        return (None if w.grad is None else w.grad.clone()), other.grad.clone()
    g0, o0 = grads()
    assert g0 is not None and float(g0.abs().sum()) > 0
    w.requires_grad_(False)
    g1, o1 = grads()
    assert g1 is None and torch.allclose(o1, o0, rtol=1e-4, atol=1e-7)
    w.requires_grad_(True)
    g2, _ = grads()
    assert g2 is not None and torch.allclose(g2, g0, rtol=1e-4, atol=1e-7)


def test_deterministic_flag_selects_the_one_stream_schedule():
    """args.deterministic (reference cfg/default.yaml:23, default True): the one-stream schedule.  Every cross-block sum of the step is
    either added in a fixed order (weight-gradient slabs, LDS partials) or an f64 atomic sum of f32 partials (BatchNorm / loss / clip /
    filter-parameter sums: order-free to 2^-52 relative, so the f32 value derived from it repeats unless it sits within 1e-16 of a rounding
    boundary), so two trainers from the same start state over the same three batches end with IDENTICAL parameters, momentum and EMA;
    deterministic=False switches the side streams on."""
    import bench
    from dedark_yolo_amd import ops

    def run(det):
        tr = _tiny_trainer("SGD", batch=64, dtype="bf16", deterministic=det)
        assert ops.wgrad_stream_enabled() == (not det)
        for i in range(3):
            b = bench.synth_batch(40 + i, 4, 96, 20, "cuda")
            tr.args.dark_param = b.pop("gamma")
            b.pop("n_max", None)
            tr.train_step(b, [0.01] * 3, 0.9)
        torch.cuda.synchronize()
        return tr.flat.p.clone(), tr.flat.m.clone(), tr.flat.ema.clone()
    try:
        a, b = run(True), run(True)
        for x, y, what in zip(a, b, ("parameters", "momentum", "ema")):
            d = float((x - y).abs().max()) / max(float(x.abs().max()), 1e-30)
            print(f"deterministic=True, two runs: {what} differ by {d:.3e} of max|x| ({int((x != y).sum())} of {x.numel()} elements)")
            assert torch.equal(x, y), (what, d, int((x != y).sum()))
        run(False)
    finally:
        import dedark_yolo_amd as dy
        dy.set_compute_dtype(torch.float32)
