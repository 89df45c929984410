#!/usr/bin/env python3
"""Checkpoint interop with the REFERENCE, both ways, in the build container (needs /root/reference; cv2 / easydict / torchvision are
arithmetic-free stand-ins exactly as in make_golden.py).

  python tests/golden/make_ckpt_interop.py skeleton   -> tests/golden/g12_ref_skeleton.json
      what `deepcopy(model).half()` of the reference looks like when pickled (ultralytics/engine/trainer.py:408-433): per module
      its class path, plain attributes (values), tensor attributes (shape / dtype) and children in order -- for the repo graphs.
      DATA the writer under test (dedark_yolo_amd/utils/checkpoint.py:save_reference_checkpoint) is checked against on the CPU.

  python tests/golden/make_ckpt_interop.py load       -> tests/golden/g12_ckpt_interop.npz
      builds THIS package's models on the CPU (parameters from oracle.model.rng_fill, no GPU needed to write a checkpoint), writes
      last.pt through save_reference_checkpoint, loads it with the reference's own attempt_load_one_weight
      (ultralytics/nn/tasks.py:674-707: torch.load with the real classes, .float(), fuse(), eval()) and runs the reference's eval
      forward on a seeded image.  Asserts the reference state_dict equals ours (fp16-rounded) and its output equals the oracle's on
      the same weights; stores input seed + reference output so that the GPU test can hold the product's eval output against it.
"""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
MODE = sys.argv[1] if len(sys.argv) > 1 else "all"
sys.argv = [sys.argv[0]]                       # filter_cfg.py parses argv at import
sys.dont_write_bytecode = True
# the reference calls torch.load(file, map_location='cpu') (tasks.py:614), written for torch < 2.6 where that unpickles module objects;
# this image's torch 2.10 defaults to weights_only=True and would refuse the reference's OWN checkpoints too
os.environ["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = "1"
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

    class EasyDict(dict):                      # attribute-style dict; like the real package it keeps items and attributes in step
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            dict.__setitem__(self, k, v)
            self.__dict__[k] = v
        __setitem__ = __setattr__
    EasyDict.__module__ = "easydict"
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

from ultralytics.nn.tasks import DetectionModel, attempt_load_one_weight, yaml_model_load  # noqa: E402

torch.set_num_threads(8)
TINY = [0.33, 0.0625, 1024]


def _plain(v, depth=0):
    if isinstance(v, (bool, int, float, str, type(None))):
        return v
    if isinstance(v, torch.Tensor):
        return {"__tensor__": [list(v.shape), str(v.dtype).replace("torch.", "")]}
    if isinstance(v, (list, tuple)):
        return {"__seq__": type(v).__name__, "items": [_plain(x, depth + 1) for x in v]} if depth < 6 else "..."
    if isinstance(v, dict):
        return {"__dict__": type(v).__module__ + "." + type(v).__name__,
                "items": {str(k): _plain(x, depth + 1) for k, x in v.items() if not isinstance(x, nn.Module) and not (
                    isinstance(x, list) and x and isinstance(x[0], nn.Module))}} if depth < 6 else "..."
    return {"__object__": type(v).__module__ + "." + type(v).__name__}


def skeleton(m):
    """class path, plain / tensor attributes of the instance __dict__ (nn.Module bookkeeping left out), children in order."""
    base = set(nn.Module().__dict__.keys())
    d = m.__dict__
    attrs = {k: _plain(v) for k, v in d.items() if k not in base}
    params = {k: (None if v is None else _plain(v.data)) for k, v in d["_parameters"].items()}
    bufs = {k: (None if v is None else _plain(v)) for k, v in d["_buffers"].items()}
    return {"cls": type(m).__module__ + "." + type(m).__name__, "training": m.training, "attrs": attrs, "params": params, "buffers": bufs,
            "children": {k: (None if c is None else skeleton(c)) for k, c in d["_modules"].items()}}


def do_skeleton():
    from copy import deepcopy
    out = {}
    for tag, name, scale, nc in (("repo_l", "yolov8.yaml", "l", 20), ("ori_n", "yolov8ori.yaml", "n", 20), ("v3_l", "yolov8-3.yaml", "l", 20),
                                 ("rbf_l", "yolov8-RBF-ASFF.yaml", "l", 20)):
        d = yaml_model_load(name)
        d["scale"] = scale
        m = DetectionModel(d, ch=3, nc=nc, verbose=False)
        m.args = dict(box=7.5, cls=0.5, dfl=1.5, lrl=2.0)
        out[tag] = skeleton(deepcopy(m).half())
        print(tag, sum(p.numel() for p in m.parameters()), "params")
    path = os.path.join(HERE, "g12_ref_skeleton.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"), sort_keys=False)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def do_load():
    import tempfile
    from oracle import model as om
    from dedark_yolo_amd.nn.tasks import DetectionModel as OurModel
    from dedark_yolo_amd.utils.checkpoint import save_reference_checkpoint
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import load_yaml, rnd
    res = {}
    for tag, name, scale, sdef, nc, S in (("ori_t", "yolov8ori.yaml", "t", TINY, 4, 64), ("ll_t", "yolov8-lowlight.yaml", "t", TINY, 20, 64),
                                          ("repo_l", "yolov8.yaml", "l", None, 20, 64)):
        cfg = load_yaml(name)
        if sdef is not None:
            cfg["scales"][scale] = sdef
        cfg["scale"] = scale
        ours = OurModel(cfg, nc=nc)
        shapes = {k: tuple(v.shape) for k, v in ours.state_dict().items()}
        sd = om.rng_fill(shapes, 1201)
        ema_sd = om.rng_fill(shapes, 1202)
        ours.load_state_dict(sd, strict=True)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "last.pt")
            save_reference_checkpoint(path, ours, ema_state=ema_sd, epoch=2, best_fitness=0.5, updates=9,
                                      train_args=dict(model=name, imgsz=S, batch=2, lowlight_FLAG=True, dedark_FLAG=True, lrl=2.0))
            size = os.path.getsize(path)
            ref, ck = attempt_load_one_weight(path, fuse=False)          # `ema` first (tasks.py:682), .float(), eval()
            ref_fused, _ = attempt_load_one_weight(path, fuse=True)
            raw = torch.load(path, map_location="cpu", weights_only=False)
        rsd = ref.state_dict()
        assert list(rsd.keys()) == list(ours.state_dict().keys()), (tag, "state_dict key order")
        for k, v in rsd.items():
            w = ema_sd[k]
            w = w.half().float() if w.is_floating_point() else w
            assert torch.equal(v, w), (tag, k)
        msd = raw["model"].float().state_dict()
        for k, v in msd.items():
            w = sd[k].half().float() if sd[k].is_floating_point() else sd[k]
            assert torch.equal(v, w), (tag, "model", k)
        assert raw["epoch"] == 2 and raw["updates"] == 9 and raw["train_args"]["imgsz"] == S
        x = rnd(1203, 2, 3, S, S).pow(2.0)
        with torch.no_grad():
            y = ref(x)
            yf = ref_fused(x)
        y, yf = (y[0] if isinstance(y, (list, tuple)) else y), (yf[0] if isinstance(yf, (list, tuple)) else yf)
        # the oracle (pinned to the reference by make_golden.py's fixtures) on the checkpoint's fp16-rounded EMA weights
        plan, save = om.build_plan(cfg, scale=scale, nc=nc)
        osd = {k: (v.half().float() if v.is_floating_point() else v.clone()) for k, v in ema_sd.items()}
        with torch.no_grad():
            yo = om.forward(plan, save, osd, x, False)
        yo = yo[0] if isinstance(yo, (list, tuple)) else yo
        err = float((yo - y).abs().max()) / max(float(y.abs().max()), 1e-30)
        assert err <= 1e-5, (tag, "oracle vs reference-on-our-checkpoint", err)
        res[f"{tag}_x_seed"] = np.array(1203)
        res[f"{tag}_y"] = y.numpy()
        res[f"{tag}_y_fused"] = yf.numpy()
        print(f"{tag}: checkpoint {size / 1e6:.1f} MB loads in the reference ({type(ref).__module__}.{type(ref).__name__}); eval output "
              f"{tuple(y.shape)}; fused vs unfused max |d| {float((y - yf).abs().max()):.3e}; oracle on the same weights: rel max err {err:.2e}")
    path = os.path.join(HERE, "g12_ckpt_interop.npz")
    np.savez_compressed(path, **res)
    print("wrote", path)


if __name__ == "__main__":
    if MODE in ("skeleton", "all"):
        do_skeleton()
    if MODE in ("load", "all"):
        do_load()
