"""CPU tests of the host side: registry / state_dict drop-in contract, C-ABI library exports, layout helpers.
No kernel is launched here (there is no GPU in the build container)."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from util import ROOT, load_yaml

YAMLS = [("yolov8ori.yaml", "n"), ("yolov8-lowlight.yaml", "n"), ("yolov8.yaml", "l"), ("yolov8-RBF-ASFF.yaml", "l")]


@pytest.mark.parametrize("yml,scale", YAMLS)
def test_state_dict_names_match_reference_layout(yml, scale):
    """The product model must expose exactly the reference's state_dict keys/shapes (oracle.param_shapes is pinned to the
    reference by the golden tests: its rng_fill dict loads strictly into the reference modules)."""
    from dedark_yolo_amd.nn.tasks import DetectionModel
    from oracle import model as om
    cfg = load_yaml(yml)
    cfg["scale"] = scale
    plan, save = om.build_plan(cfg, scale=scale, nc=20)
    want = om.param_shapes(plan)
    model = DetectionModel(dict(cfg), nc=20)
    have = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert set(have) == set(want), (sorted(set(want) - set(have))[:5], sorted(set(have) - set(want))[:5])
    for k in want:
        assert tuple(want[k]) == have[k], k
    assert set(model.save) == set(save)
    assert [float(s) for s in model.stride] == [8.0, 16.0, 32.0]


def test_param_counts():
    from dedark_yolo_amd.nn.tasks import DetectionModel
    m = DetectionModel("yolov8l.yaml", nc=20)
    assert sum(p.numel() for p in m.parameters()) == 51776780
    m = DetectionModel("yolov8nori.yaml", nc=20)
    assert sum(p.numel() for p in m.parameters()) == 3014748


def test_repo_yaml_only_builds_at_scale_l_contract():
    """AsffTribeLevel hard-codes (512, 512, 256) (reference block.py:52): at scale n the graph is inconsistent."""
    from dedark_yolo_amd.nn.tasks import DetectionModel
    m = DetectionModel("yolov8n.yaml", nc=20)          # constructing is possible, running would raise on channel mismatch
    assert m.model[23].inter_dim == 512 and m.model[22].cv2.conv.out_channels != 512


def test_detect_bias_init_and_bn_constants():
    from dedark_yolo_amd.nn.tasks import DetectionModel
    import math
    m = DetectionModel("yolov8nori.yaml", nc=20)
    det = m.model[-1]
    assert torch.allclose(det.cv2[0][-1].bias, torch.ones(64))
    assert abs(float(det.cv3[1][-1].bias[0]) - math.log(5 / 20 / (640 / 16) ** 2)) < 1e-6
    bns = [x for x in m.modules() if isinstance(x, torch.nn.BatchNorm2d)]
    assert bns and all(b.eps == 1e-3 and b.momentum == 0.03 for b in bns)


def test_library_exports_every_declared_symbol():
    """include/dedark_yolo.h <-> libdedark_yolo.so <-> ctypes table agree (symbol names and argument counts)."""
    from dedark_yolo_amd import _C
    hdr = open(os.path.join(ROOT, "include", "dedark_yolo.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(dy_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    assert len(decls) > 30
    table = dict(_C._SIGS)
    for name, n in decls.items():
        if name in ("dy_last_error", "dy_last_kernel"):     # const char* accessors, bound by hand in _C.lib()
            continue
        assert name in table, f"{name} declared in the header but missing from the ctypes table"
        assert len(table[name]) == n, f"{name}: header has {n} args, ctypes table {len(table[name])}"
    assert set(table) <= set(decls), set(table) - set(decls)
    lib = _C.lib()                       # raises if the .so is missing or a symbol is not exported
    assert lib.dy_version() == 2
    out = subprocess.run(["nm", "-D", "--defined-only", _C.LIB_PATH], capture_output=True, text=True).stdout
    for name in decls:
        assert re.search(rf"\b{name}\b", out), f"{name} not exported"


def test_product_path_has_no_cpu_fallback():
    from dedark_yolo_amd.nn.modules import Conv
    c = Conv(8, 8, 3, 1)
    with pytest.raises(RuntimeError, match="GPU"):
        c(torch.zeros(1, 8, 8, 8))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "dedark_yolo_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_nhwc_view_helpers():
    from dedark_yolo_amd import ops
    t = torch.empty((2, 16, 5, 7), memory_format=torch.channels_last)
    assert ops.ld_of(t) == 16 and ops.ld_of(t[:, 4:12]) == 16
    with pytest.raises(RuntimeError):
        ops.ld_of(torch.empty(2, 16, 5, 7))
    z = ops.zeros_nhwc(1, 8, 3, 3, torch.float32, "cpu")
    assert ops.ld_of(z) == 8 and z.shape == (1, 8, 3, 3)
    assert ops.round_up(20, 8) == 24 and ops.vec_elems(torch.bfloat16) == 8


def test_yaml_loader_and_scale_guess():
    from dedark_yolo_amd.nn.tasks import guess_model_scale, yaml_model_load
    d = yaml_model_load("yolov8l.yaml")
    assert d["scale"] == "l" and d["backbone"][0][2] == "lowlight_recovery"
    assert guess_model_scale("yolov8n-lowlight.yaml") == "n"
    with pytest.raises(FileNotFoundError):
        yaml_model_load("yolov9q.yaml")


# ---------------------------------------------------------------------------------------------- validation host logic
def test_product_metrics_match_reference_goldens():
    """utils/metrics.py + engine/validator.match_predictions + utils/ops box helpers against vectors captured from the
    reference (g5_ap, g5_small, g6_val, g6_match) and against the oracle."""
    import numpy as np
    from util import close, gold
    from dedark_yolo_amd.engine.validator import match_predictions
    from dedark_yolo_amd.utils import metrics as M
    from dedark_yolo_amd.utils import ops as O
    g = gold("g5_ap")
    tp, fp, p, r, f1, ap, uc = M.ap_per_class(g["tp"].numpy().astype(bool), g["conf"].numpy(), g["pred_cls"].numpy(), g["target_cls"].numpy())
    close(ap, g["ap"], 1e-9, 1e-12, "ap")
    close(p, g["p"], 1e-9, 1e-12, "p")
    close(r, g["r"], 1e-9, 1e-12, "r")
    close(f1, g["f1"], 1e-9, 1e-12, "f1")
    assert np.array_equal(tp, g["tpc"].numpy()) and np.array_equal(fp, g["fpc"].numpy()) and np.array_equal(uc, g["unique"].numpy())
    ka3 = M.compute_ap(np.array([.1, .2, .2, .4, .5, .5, .8]), np.array([1, 1, .67, .75, .8, .67, .6]))[0]
    assert abs(ka3 - 0.68885) < 1e-9
    dm = M.DetMetrics(names={i: str(i) for i in range(4)})
    dm.process(g["tp"].numpy().astype(bool), g["conf"].numpy(), g["pred_cls"].numpy(), g["target_cls"].numpy())
    rd = dm.results_dict
    assert abs(rd["metrics/mAP50(B)"] - g["ap"].numpy()[:, 0].mean()) < 1e-12
    assert abs(rd["fitness"] - (0.1 * g["ap"].numpy()[:, 0].mean() + 0.9 * g["ap"].numpy().mean())) < 1e-12
    g = gold("g6_val")
    close(O.xywh2xyxy(g["xywh"]), g["xyxy"], 0, 0, "xywh2xyxy")
    close(O.xyxy2xywh(g["xyxy"]), g["back"], 0, 0, "xyxy2xywh")
    close(O.scale_boxes((640, 640), g["xyxy"].clone(), (480, 360)), g["scaled"], 0, 1e-6, "scale_boxes")
    g = gold("g6_match")
    assert torch.equal(match_predictions(g["det"], g["lab"], torch.linspace(0.5, 0.95, 10)), g["correct"].bool())
    s = gold("g5_small")
    close(M.box_iou(s["b1"][:8], s["b2"][:12]), s["pairwise"], 0, 1e-7, "box_iou")


def test_reference_checkpoint_reader():
    """tests/golden/g7_ref_last.pt was written by the reference's own classes exactly as trainer.save_model does
    (engine/trainer.py:408-433: pickled half-precision DetectionModel objects under 'model' and 'ema').  The restricted unpickler
    rebuilds both state_dicts without the reference package, 'ema' takes precedence (nn/tasks.py:640,682), DetectionModel.load takes it."""
    import numpy as np
    import torch
    from dedark_yolo_amd.utils.checkpoint import load_checkpoint
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ck = load_checkpoint(os.path.join(root, "tests", "golden", "g7_ref_last.pt"))
    z = np.load(os.path.join(root, "tests", "golden", "g7_ckpt.npz"))
    keys = [str(k) for k in z["keys"]]
    assert ck.source == "reference-pickle" and list(ck.state_dict) == keys and list(ck.model_sd) == keys
    se = np.array([float(ck.state_dict[k].double().sum()) for k in keys])
    sm = np.array([float(ck.model_sd[k].double().sum()) for k in keys])
    assert np.abs(se - z["sum_ema"]).max() == 0.0 and np.abs(sm - z["sum_model"]).max() == 0.0
    assert np.abs(z["sum_ema"] - z["sum_model"]).max() > 0.0            # the two really differ
    assert (ck.epoch, ck.updates, ck.nc) == (3, 77, 4) and abs(ck.best_fitness - 0.4321) < 1e-12
    assert ck.train_args["lowlight_FLAG"] is True and isinstance(ck.yaml, dict) and ck.yaml["scale"] == "t"
    assert all(v.dtype == torch.float32 for v in ck.state_dict.values() if v.is_floating_point())
    # the graph of the checkpoint's yaml builds here and takes every tensor
    from dedark_yolo_amd.nn.tasks import DetectionModel
    m = DetectionModel(ck.yaml, nc=ck.nc)
    assert m.load(ck.state_dict) == len(keys)
    sd = m.state_dict()
    assert all(torch.equal(sd[k], ck.state_dict[k]) for k in keys if k in sd and not k.endswith("num_batches_tracked"))


def test_checkpoint_reader_refuses_to_run_code(tmp_path):
    """A pickle that names an arbitrary callable must come back as an inert record, never be called."""
    import pickle
    import torch
    from dedark_yolo_amd.utils.checkpoint import load_checkpoint

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))
    f = tmp_path / "evil.pt"
    torch.save(dict(model=dict(w=torch.zeros(1)), junk=Evil()), f)
    ck = load_checkpoint(str(f))
    assert not (tmp_path / "pwned").exists() and "w" in ck.state_dict
    # gadgets INSIDE the packages a checkpoint legitimately names (torch.*, numpy.*): an exact allow-list, not a prefix rule
    import subprocess
    import numpy.testing._private.utils as npu
    import torch.utils.collect_env as tce
    import torch.hub

    def gadget(fn, *args):
        class G:
            def __reduce__(self):
                return (fn, args)
        return G()
    mark = tmp_path / "pwned2"
    for k, obj in enumerate((gadget(npu.runstring, "open(%r, 'w').write('x')" % str(mark), {}),
                             gadget(tce.run, "echo x > %s" % mark),
                             gadget(subprocess.check_call, ["touch", str(mark)]),
                             gadget(torch.hub.load, "x/y", "z"),
                             gadget(eval, "open(%r, 'w').write('x')" % str(mark)))):
        f = tmp_path / f"evil{k}.pt"
        torch.save(dict(model=dict(w=torch.ones(2)), junk=obj), f)
        ck = load_checkpoint(str(f))
        assert not mark.exists(), f"gadget {k} ran"
        assert torch.equal(ck.state_dict["w"], torch.ones(2))


def test_tile_walk_reciprocal_is_exact():
    """dy_common.h DyTileWalk: floor((x + 0.5) * (1 / d)) in float32 equals x // d for every row offset a conv tile can ask for
    (x < 2^16), every divisor an image can have, and for a reciprocal that is off by one ulp either way (v_rcp_f32)."""
    x = np.arange(0, 1 << 16, dtype=np.int64)
    xf = x.astype(np.float32) + np.float32(0.5)
    for d0 in range(1, 4097, 256):
        d = np.arange(d0, min(d0 + 256, 4097), dtype=np.int64)
        inv = (np.float32(1.0) / d.astype(np.float32)).astype(np.float32)
        want = x[None, :] // d[:, None]
        for rcp in (inv, np.nextafter(inv, np.float32(0)), np.nextafter(inv, np.float32(2))):
            got = (xf[None, :] * rcp[:, None]).astype(np.float32).astype(np.int64)
            assert np.array_equal(got, want), int(d0)


def _written_skeleton(o):
    """same walk as tests/golden/make_ckpt_interop.py::skeleton, on the writer's stand-in object tree"""
    import torch

    def plain(v):
        if isinstance(v, (bool, int, float, str, type(None))):
            return v
        if isinstance(v, torch.Tensor):
            return {"__tensor__": [list(v.shape), str(v.dtype).replace("torch.", "")]}
        if isinstance(v, (list, tuple)):
            return {"__seq__": type(v).__name__, "items": [plain(x) for x in v]}
        if isinstance(v, dict):
            return {"__dict__": type(v).__module__ + "." + type(v).__name__,
                    "items": {str(k): plain(x) for k, x in v.items() if k != "filters"}}
        return {"__object__": type(v).__module__ + "." + type(v).__name__}
    base = set(torch.nn.Module().__dict__.keys())
    d = o.__dict__
    return {"cls": type(o).__module__ + "." + type(o).__qualname__, "training": d["training"],
            "attrs": {k: plain(v) for k, v in d.items() if k not in base},
            "params": {k: (None if v is None else plain(v.data)) for k, v in d["_parameters"].items()},
            "buffers": {k: (None if v is None else plain(v)) for k, v in d["_buffers"].items()},
            "children": {k: (None if c is None else _written_skeleton(c)) for k, c in d["_modules"].items()}}


@pytest.mark.parametrize("tag,name,scale", [("repo_l", "yolov8.yaml", "l"), ("ori_n", "yolov8ori.yaml", "n"), ("v3_l", "yolov8-3.yaml", "l"),
                                            ("rbf_l", "yolov8-RBF-ASFF.yaml", "l")])
def test_reference_checkpoint_writer_layout_matches_the_reference(tag, name, scale):
    """What save_reference_checkpoint pickles for a graph must be, module by module, what the reference itself pickles for it
    (tests/golden/g12_ref_skeleton.json, captured from `deepcopy(model).half()` of the reference by make_ckpt_interop.py): class
    path, plain attributes and their values, parameter / buffer names, shapes and dtypes, children in order."""
    import json
    from util import load_yaml
    from dedark_yolo_amd.nn.tasks import DetectionModel
    from dedark_yolo_amd.utils.checkpoint import reference_module_object
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "g12_ref_skeleton.json")) as f:
        want = json.load(f)[tag]
    cfg = load_yaml(name)
    cfg["scale"] = scale
    got = _written_skeleton(reference_module_object(DetectionModel(cfg, nc=20), None, True, dict(box=7.5, cls=0.5, dfl=1.5, lrl=2.0)))
    bad = []

    def walk(a, b, path):
        if a["cls"] != b["cls"]:
            bad.append((path, "class", a["cls"], b["cls"]))
        for k in set(a["attrs"]) | set(b["attrs"]):
            if k not in ("yaml",) and a["attrs"].get(k, "<absent>") != b["attrs"].get(k, "<absent>"):
                bad.append((path, k, a["attrs"].get(k, "<absent>"), b["attrs"].get(k, "<absent>")))
        for f_ in ("params", "buffers"):
            if a[f_] != b[f_]:
                bad.append((path, f_, a[f_], b[f_]))
        if list(a["children"]) != list(b["children"]):
            bad.append((path, "children", list(a["children"]), list(b["children"])))
        for k, c in a["children"].items():
            if c is not None and b["children"].get(k) is not None:
                walk(c, b["children"][k], path + "." + k)
    walk(want, got, tag)
    assert not bad, bad[:10]
    wy, gy = want["attrs"]["yaml"]["items"], got["attrs"]["yaml"]["items"]
    for k in ("nc", "scale", "backbone", "head", "ch"):
        assert wy[k] == gy[k], k


def test_reference_checkpoint_writer_round_trip(tmp_path):
    """The file names the REFERENCE's classes (and none of this package's), torch.load(weights_only=True) refuses it like any pickled
    module object, and the restricted reader gets both state_dicts, counters and optimizer back."""
    import torch
    from util import load_yaml
    from dedark_yolo_amd.nn.tasks import DetectionModel
    from dedark_yolo_amd.utils.checkpoint import load_checkpoint, save_reference_checkpoint
    cfg = load_yaml("yolov8-lowlight.yaml")
    cfg["scales"]["t"] = [0.33, 0.0625, 1024]
    cfg["scale"] = "t"
    torch.manual_seed(3)
    m = DetectionModel(cfg, nc=20)
    sd = m.state_dict()
    ema = {k: (v + 0.25 if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    opt = dict(state={0: dict(momentum_buffer=torch.ones(3))}, param_groups=[dict(lr=0.01, params=[0])])
    path = save_reference_checkpoint(str(tmp_path / "last.pt"), m, ema_state=ema, epoch=7, best_fitness=0.5, updates=11, optimizer=opt,
                                     train_args=dict(imgsz=64, lrl=2.0), extra=dict(dy_state=dict(step_count=5)))
    import zipfile
    with zipfile.ZipFile(path) as z:
        pkl = z.read([n for n in z.namelist() if n.endswith("data.pkl")][0])
    for needle in (b"ultralytics.nn.tasks\nDetectionModel", b"ultralytics.nn.modules.conv\nConv", b"ultralytics.nn.modules.llie\nlowlight_recovery",
                   b"ultralytics.nn.modules.filtersB\nUsmFilter", b"easydict\nEasyDict", b"torch.nn.modules.conv\nConv2d"):
        assert needle in pkl, needle
    assert b"dedark_yolo_amd" not in pkl
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=True)
    ck = load_checkpoint(path)
    assert ck.source == "reference-pickle" and (ck.epoch, ck.updates, ck.nc) == (7, 11, 20) and ck.dy_state["step_count"] == 5
    assert list(ck.state_dict) == list(sd) and list(ck.model_sd) == list(sd)
    for k, v in sd.items():
        w = v.half().float() if v.is_floating_point() else v
        e = ema[k].half().float() if v.is_floating_point() else v
        assert torch.equal(ck.model_sd[k], w) and torch.equal(ck.state_dict[k], e), k
    assert torch.equal(ck.optimizer["state"][0]["momentum_buffer"], torch.ones(3))
    m2 = DetectionModel(ck.yaml, nc=ck.nc)
    assert m2.load(ck.state_dict) == len(sd)


def test_match_predictions_equals_the_references_sort_unique_formulation():
    """engine/validator.match_predictions states the matching rule directly; oracle/val.match_predictions keeps the reference's own
    argsort / np.unique sequence (pinned by g6_match).  Dense random scenes: many labels per detection and detections per label."""
    from dedark_yolo_amd.engine.validator import match_predictions
    from oracle import val as oval
    iouv = torch.linspace(0.5, 0.95, 10)
    g = np.random.default_rng(8)
    for trial in range(30):
        nl, nd = int(g.integers(0, 12)), int(g.integers(0, 60))
        c = g.uniform(50, 300, (nl, 2))
        s = g.uniform(20, 120, (nl, 2))
        lab = torch.tensor(np.concatenate((g.integers(0, 3, (nl, 1)), c - s / 2, c + s / 2), 1), dtype=torch.float32).reshape(nl, 5)
        pick = g.integers(0, max(nl, 1), nd)
        dc = (c[pick] if nl else np.zeros((nd, 2))) + g.normal(0, 6, (nd, 2))
        ds = (s[pick] if nl else np.ones((nd, 2))) * np.exp(g.normal(0, 0.1, (nd, 2)))
        det = torch.tensor(np.concatenate((dc - ds / 2, dc + ds / 2, g.random((nd, 1)), g.integers(0, 3, (nd, 1))), 1), dtype=torch.float32).reshape(nd, 6)
        a, b = match_predictions(det, lab, iouv), oval.match_predictions(det, lab, iouv)
        assert torch.equal(a, b), trial
