"""GPU parity tests of the validation side: batched HIP NMS against the oracle's restatement of
non_max_suppression on identical prediction tensors (kept (anchor, class) indices bit-exact, rows bit-exact), and the
validator's mAP against the oracle pipeline fed with the product's own predictions (|dmAP| <= 1e-4)."""
import numpy as np
import pytest
import torch

from util import load_yaml, rnd

pytestmark = pytest.mark.gpu


def _synthetic_pred(seed, B, nc, A, S=640.0, clusters=40, hot=0.02):
    """xywh px + class scores: boxes clustered around a few centres (so that suppression really happens), a small fraction
    of confident scores, the rest near zero."""
    g = np.random.default_rng(seed)
    cen = g.uniform(0.1 * S, 0.9 * S, (B, clusters, 2))
    siz = g.uniform(0.05 * S, 0.3 * S, (B, clusters, 2))
    which = g.integers(0, clusters, (B, A))
    bi = np.arange(B)[:, None]
    xy = cen[bi, which] + g.normal(0, 6.0, (B, A, 2))
    wh = siz[bi, which] * np.exp(g.normal(0, 0.08, (B, A, 2)))
    sc = g.random((B, A, nc)) * 0.05
    m = g.random((B, A, nc)) < hot
    sc = np.where(m, g.uniform(0.2, 0.99, (B, A, nc)), sc)
    pred = np.concatenate((xy, wh, sc), 2).transpose(0, 2, 1)
    return torch.from_numpy(np.ascontiguousarray(pred.astype(np.float32)))


def _check_nms(pred, conf, iou, multi_label=True, agnostic=False, max_det=300, max_nms=30000):
    from oracle import val as oval
    from dedark_yolo_amd.utils.ops import nms_batched
    want = oval.non_max_suppression(pred, conf, iou, multi_label=multi_label, max_det=max_det, max_nms=max_nms,
                                    max_wh=0 if agnostic else 7680)
    out, cnt, keep = nms_batched(pred.cuda(), conf, iou, multi_label, agnostic, max_det, max_nms, 7680, return_indices=True)
    torch.cuda.synchronize()
    out, cnt, keep = out.cpu(), cnt.cpu(), keep.cpu()
    B, no, A = pred.shape
    nc = no - 4
    total = 0
    for b in range(B):
        w = want[b]
        assert int(cnt[b]) == w.shape[0], f"image {b}: kept {int(cnt[b])} vs oracle {w.shape[0]}"
        got = out[b, :w.shape[0]]
        assert torch.equal(got, w), f"image {b}: kept rows differ"
        # integer output: (anchor, class) of every kept row, in order
        a = keep[b, :w.shape[0]] // nc
        j = keep[b, :w.shape[0]] % nc
        assert torch.equal(j.float(), w[:, 5])
        assert torch.equal(pred[b, 4 + j, a], w[:, 4])
        assert (keep[b, w.shape[0]:] == -1).all()
        total += w.shape[0]
    return total


def test_nms_bit_exact_random():
    pred = _synthetic_pred(1, 4, 20, 2100)
    assert _check_nms(pred, 0.25, 0.7) > 50
    assert _check_nms(pred, 0.25, 0.45) > 50
    assert _check_nms(pred, 0.001, 0.7, max_det=300) == 4 * 300          # val-style threshold: far more candidates than max_det


def test_nms_single_label_and_agnostic():
    pred = _synthetic_pred(2, 3, 5, 1344, hot=0.05)
    assert _check_nms(pred, 0.25, 0.6, multi_label=False) > 20
    assert _check_nms(pred, 0.25, 0.6, agnostic=True) > 10
    one = _synthetic_pred(3, 2, 1, 1344, hot=0.2)                         # nc == 1 turns multi_label off (ops.py:211)
    assert _check_nms(one, 0.3, 0.5) > 5


def test_nms_ties_empty_and_caps():
    pred = _synthetic_pred(4, 3, 4, 800, hot=0.1)
    pred[0, 4:] = 0.0                                                     # image without candidates
    pred[1, 4:] = torch.round(pred[1, 4:] * 8) / 8                        # heavy score ties -> order by candidate index
    pred[2, :4, 100:140] = pred[2, :4, 100:101]                           # identical boxes
    assert _check_nms(pred, 0.25, 0.7) > 10
    assert _check_nms(pred, 0.2, 0.7, max_det=7) <= 14                    # max_det truncation
    # max_nms smaller than the candidate count: only exact when the cut does not fall inside a tie -> unique scores
    p2 = _synthetic_pred(5, 2, 4, 800, hot=0.3)
    assert _check_nms(p2, 0.25, 0.7, max_nms=200) > 10


def test_nms_f32_arithmetic_on_borderline_pairs():
    """Pairs whose IoU crosses 0.7 differently in f32 and f64 (g11_nms_borderline.npz): the kernel must follow torchvision's f32
    rule, i.e. the oracle -- whose own f64 evaluation keeps a different set (tests/test_oracle_golden.py)."""
    from util import gold
    g = gold("g11_nms_borderline")
    for k in (0, 1):
        pred = g[f"pred{k}"]
        kept = _check_nms(pred, 0.25, 0.7)
        assert kept == pred.shape[2] - int(g[f"n32_{k}"])
    both = torch.cat((g["pred0"], g["pred1"]), 0)
    _check_nms(both, 0.25, 0.7)


def test_nms_full_size_batch():
    """BASELINE shapes: B=32, nc=20, A=8400 -- checked through size-independent properties (sorted scores, no surviving
    pair above the IoU threshold within a class, every survivor above conf)."""
    from dedark_yolo_amd.utils.ops import nms_batched
    from oracle import val as oval
    pred = _synthetic_pred(6, 32, 20, 8400)
    out, cnt = nms_batched(pred.cuda(), 0.25, 0.7, True, False, 300, 30000, 7680)
    out, cnt = out.cpu(), cnt.cpu()
    assert int(cnt.min()) > 0 and int(cnt.max()) <= 300
    for b in range(32):
        d = out[b, :int(cnt[b])]
        assert (d[:, 4] > 0.25).all() and (d[1:, 4] <= d[:-1, 4]).all()
        iou = oval.box_iou(d[:, :4], d[:, :4])
        same = d[:, 5:6] == d[:, 5]
        iou = torch.triu(iou * same, diagonal=1)
        assert float(iou.max()) <= 0.7 + 5e-3       # suppression runs on class-offset f32 boxes (ops.py:259): 1/64 px grid at cls 19
    # first image against the oracle exactly
    w = oval.non_max_suppression(pred[:1], 0.25, 0.7)[0]
    assert torch.equal(out[0, :w.shape[0]], w)


def test_validator_map_vs_oracle_on_product_predictions():
    """model (eval) -> Detect decode -> HIP NMS -> matching -> AP through the product validator, against the oracle's NMS +
    matching + AP on the SAME prediction tensor; then the eval forward itself against the oracle's forward."""
    import dedark_yolo_amd as dy
    from dedark_yolo_amd.engine.trainer import get_cfg
    from dedark_yolo_amd.engine.validator import DetectionValidator
    from dedark_yolo_amd.nn.tasks import DetectionModel
    from oracle import model as om
    from oracle import val as oval
    from parity_helpers import load_sd
    dy.set_compute_dtype(torch.float32)
    nc, S, B = 20, 128, 6
    cfgd = load_yaml("yolov8-lowlight.yaml")
    cfgd["scales"]["t"] = [0.33, 0.125, 1024]
    cfgd["scale"] = "t"
    model = DetectionModel(cfgd, nc=nc)
    plan, save = om.build_plan(cfgd, scale="t", nc=nc)
    sd = om.rng_fill(om.param_shapes(plan), 11)
    # confident heads: bias the class logits up so that NMS has something to do
    for k in sd:
        if ".cv3." in k and k.endswith("2.bias"):
            sd[k] = sd[k] + 4.0
    load_sd(model, sd)
    model = model.cuda().eval()
    g = np.random.default_rng(5)
    img = torch.from_numpy((g.random((B, 3, S, S)) * 255).astype(np.uint8))
    bi, cls, bb = [], [], []
    for b in range(B):
        for _ in range(3):
            bi.append(b)
            cls.append(int(g.integers(0, nc)))
            cx, cy, w, h = g.uniform(0.3, 0.7), g.uniform(0.3, 0.7), g.uniform(0.1, 0.5), g.uniform(0.1, 0.5)
            bb.append([cx, cy, w, h])
    batch = dict(img=img, batch_idx=torch.tensor(bi, dtype=torch.float32), cls=torch.tensor(cls, dtype=torch.float32).view(-1, 1),
                 bboxes=torch.tensor(bb, dtype=torch.float32), ori_shape=[(S, S)] * B)
    v = DetectionValidator(get_cfg(dict(conf=0.25, iou=0.7)))
    got = v(model, [batch])
    # oracle pipeline on the product's predictions
    with torch.no_grad():
        x = (img.float() / 255).cuda()
        y = model(x)[0].float().cpu()
        y_or = om.forward(plan, save, sd, img.float() / 255, False)[0]
    assert float((y - y_or).abs().max()) < 2e-3 * max(1.0, float(y_or.abs().max())), "eval forward vs oracle"
    dets = oval.non_max_suppression(y, 0.25, 0.7)
    stats = []
    iouv = torch.linspace(0.5, 0.95, 10)
    for si, d in enumerate(dets):
        idx = batch["batch_idx"] == si
        c, bx = batch["cls"][idx], batch["bboxes"][idx]
        tbox = oval.xywh2xyxy(bx) * torch.tensor((S, S, S, S), dtype=torch.float32)
        if d.shape[0] == 0:
            stats.append((torch.zeros(0, 10, dtype=torch.bool), torch.zeros(0), torch.zeros(0), c.squeeze(-1)))
            continue
        dn = d.clone()
        oval.scale_boxes((S, S), dn[:, :4], (S, S))
        oval.scale_boxes((S, S), tbox, (S, S))
        stats.append((oval.match_predictions(dn, torch.cat((c, tbox), 1), iouv), d[:, 4], d[:, 5], c.squeeze(-1)))
    tp, conf, pcls, tcls = [torch.cat(z, 0).numpy() for z in zip(*stats)]
    assert tp.shape[0] > 20, "test is vacuous without detections"
    r = oval.ap_per_class(tp, conf, pcls, tcls)
    want50, want = r["ap"][:, 0].mean(), r["ap"].mean()
    assert abs(got["metrics/mAP50(B)"] - want50) <= 1e-4 and abs(got["metrics/mAP50-95(B)"] - want) <= 1e-4
    assert abs(got["metrics/precision(B)"] - r["p"].mean()) <= 1e-4 and abs(got["metrics/recall(B)"] - r["r"].mean()) <= 1e-4
    assert abs(got["fitness"] - oval.fitness(r["ap"])) <= 1e-4


def test_bench_two_ranks_on_one_gpu_gloo():
    """Rehearsal of `bench.py --gpus 2` (the driver's multi-GPU launch line) on a one-GPU box: two ranks share cuda:0 and
    all-reduce through gloo.  Checks the launch contract (one JSON line from rank 0, n_gpus, global batch), that the bucketed
    all-reduce path runs from the backward hooks, and that the replicas stay bit-identical."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DY_SINGLE_DEVICE="1", DY_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4",
           "--imgsz", "128", "--no-cpu-baseline"]       # with the roofline legs: their extra steps all-reduce on every rank
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["scaling"] == "weak"
    assert out["replicas_in_sync"] is True
    assert np.isfinite(out["final_loss"]) and out["value"] > 0
    assert out["roofline"]["achieved"] > 0 and "cpu_baseline" not in out          # cpu_baseline is an N=1 field
    pr = out["per_rank"]                          # own step time of each rank and the all-reduce time its backward did not hide
    assert len(pr["ms_per_step"]) == 2 and all(0 < t <= out["ms_per_step"] * 1.05 for t in pr["ms_per_step"])
    assert len(pr["allreduce_exposed_ms_per_step"]) == 2 and all(0 <= t < out["ms_per_step"] for t in pr["allreduce_exposed_ms_per_step"])
    assert pr["buckets"] >= 1 and pr["gradient_bytes"] > 0


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (what a driver without its own launcher runs): bench.py starts a
    torch.distributed.run child before touching the GPU and exits with its code; one JSON line, two ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DY_SINGLE_DEVICE="1", DY_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--imgsz", "128",
           "--model", "yolov8n-lowlight.yaml", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["replicas_in_sync"] is True and out["rccl_ranks"] == 0


def test_predict_results_vs_oracle_nms():
    """YOLO.predict(): preprocess -> eval forward -> HIP NMS (predictor defaults: multi_label off) -> Results, against the oracle's
    NMS + scale_boxes on the same raw predictions."""
    import dedark_yolo_amd as dy
    from dedark_yolo_amd import YOLO
    from oracle import model as om
    from oracle import val as oval
    from parity_helpers import load_sd
    dy.set_compute_dtype(torch.float32)
    nc, S, B = 20, 128, 3
    y = YOLO("yolov8n-lowlight.yaml")
    shapes = {k: tuple(v.shape) for k, v in y.model.state_dict().items()}
    sd = om.rng_fill(shapes, 21)
    for k in sd:
        if ".cv3." in k and k.endswith("2.bias"):
            sd[k] = sd[k] + 3.0
    load_sd(y.model, sd)
    y.model.cuda()
    g = np.random.default_rng(9)
    img = torch.from_numpy((g.random((B, 3, S, S)) * 255).astype(np.uint8))
    res = y.predict(img, conf=0.3, iou=0.6, orig_shapes=[(S, S), (96, 120), (S, S)])
    with torch.no_grad():
        raw = y.model.eval()((img.float() / 255).cuda())[0].float().cpu()
    want = oval.non_max_suppression(raw, 0.3, 0.6, multi_label=False)
    assert len(res) == B and sum(len(r) for r in res) > 10
    for i, (r, w) in enumerate(zip(res, want)):
        w = w.clone()
        oval.scale_boxes((S, S), w[:, :4], r.orig_shape)
        got = r.boxes.data.cpu()
        assert got.shape == w.shape and torch.equal(got[:, 4:], w[:, 4:]), f"image {i}"         # same detections, same order
        # torch divides by the gain as a multiplication by its reciprocal on the GPU: 1 ulp on rescaled boxes
        assert torch.allclose(got[:, :4], w[:, :4], rtol=1e-6, atol=1e-4), f"image {i}"
        assert r.boxes.xyxy.shape == (len(r), 4) and r.boxes.conf.shape == (len(r),) and r.boxes.cls.shape == (len(r),)
        assert float(r.boxes.xyxyn.max()) <= 1.0 + 1e-6 if len(r) else True


def test_nms_candidate_stage_vs_reference_fixture():
    """The part of non_max_suppression in front of torchvision.ops.nms, captured from the reference (g9_prenms.npz: the nms call
    was replaced by a recorder that keeps every box).  With iou_thres = 1 nothing is suppressed here either, so the HIP pipeline
    must return exactly the reference's candidate rows (as a set: the reference keeps candidate order, the kernel score order)."""
    import os
    import numpy as np
    from dedark_yolo_amd.utils.ops import nms_batched
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(root, "tests", "golden", "g9_prenms.npz"))
    pred = torch.from_numpy(z["pred"]).cuda()

    def rows(t):
        a = t.numpy() if isinstance(t, torch.Tensor) else t
        return a[np.lexsort(a.T[::-1])]
    for tag, ml, agn, max_nms in (("ml", True, False, 30000), ("sl", False, False, 30000), ("cap", True, False, 50), ("agn", True, True, 30000)):
        out, cnt = nms_batched(pred, 0.3, 1.0, ml, agn, 2000, max_nms, 7680)
        torch.cuda.synchronize()
        for i in range(2):
            want = z[f"{tag}_out{i}"]
            assert int(cnt[i]) == want.shape[0], (tag, i, int(cnt[i]), want.shape[0])
            assert np.array_equal(rows(out[i, :want.shape[0]].cpu()), rows(want)), (tag, i)


def test_fuse_caches_the_batchnorm_fold():
    """BaseModel.fuse(): the eval-mode BatchNorm fold of every Conv is computed once and cached; outputs are unchanged, a training
    step invalidates the caches, the next eval forward rebuilds them (reference fuse_conv_and_bn, torch_utils.py:123-144)."""
    import dedark_yolo_amd as dy
    from dedark_yolo_amd import _C
    from dedark_yolo_amd.nn.tasks import DetectionModel
    from util import load_yaml
    dy.set_compute_dtype(torch.float32)
    cfg = load_yaml("yolov8ori.yaml")
    cfg["scales"]["t"] = [0.33, 0.125, 1024]
    cfg["scale"] = "t"
    torch.manual_seed(3)
    m = DetectionModel(cfg, nc=6).cuda().eval()
    for b in m.modules():
        if isinstance(b, torch.nn.BatchNorm2d):
            b.running_mean.uniform_(-0.2, 0.2)
            b.running_var.uniform_(0.5, 1.5)
    x = torch.rand(2, 3, 64, 64, device="cuda")
    assert not m.is_fused()
    with torch.no_grad():
        y0 = m(x)[0].clone()
    assert m.is_fused()                              # the first eval forward filled the caches
    calls = []
    orig = _C.call

    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)
    import dedark_yolo_amd.ops as ops
    ops.call = spy
    try:
        with torch.no_grad():
            y1 = m(x)[0]
    finally:
        ops.call = orig
    assert "dy_bn_fold_eval" not in calls and torch.equal(y0, y1)
    ops.bump_weights_epoch()                          # what an optimizer step does
    assert not m.is_fused()
    m.fuse()
    assert m.is_fused()
    with torch.no_grad():
        assert torch.equal(m(x)[0], y0)
