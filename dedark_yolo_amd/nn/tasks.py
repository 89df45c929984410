"""Model graph / registry: yaml -> module list, layer-graph executor, criterion dispatch.

Mirrors the reference ultralytics/nn/tasks.py: BaseModel (:29-253), DetectionModel (:257-344), parse_model (:803-921),
yaml_model_load (:924-946), guess_model_scale (:950-965) -- same names, arguments and `model.<i>.` state_dict prefixes.
"""
import contextlib
import math
import re
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn
import yaml

from .. import ops
from .modules import (AsffTribeLevel, C2f, Concat, Conv, Detect, RFBblock, SPPF, Upsample, lowlight_recovery)

CFG_DIR = Path(__file__).resolve().parent.parent / "cfg" / "models" / "v8"

_REGISTRY = dict(Conv=Conv, C2f=C2f, SPPF=SPPF, Concat=Concat, Detect=Detect, AsffTribeLevel=AsffTribeLevel,
                 RFBblock=RFBblock, lowlight_recovery=lowlight_recovery)
_REGISTRY["nn.Upsample"] = Upsample


def make_divisible(x, divisor):
    """reference ultralytics/utils/ops.py:128-142."""
    return math.ceil(x / divisor) * divisor


def guess_model_scale(model_path):
    """'yolov8l.yaml' -> 'l' (reference tasks.py:950-965)."""
    with contextlib.suppress(AttributeError):
        return re.search(r"yolov\d+([nslmx])", Path(model_path).stem).group(1)
    return ""


def yaml_model_load(path):
    """'yolov8l.yaml' -> dict of yolov8.yaml with d['scale']='l' (reference tasks.py:924-946)."""
    path = Path(path)
    unified = re.sub(r"(\d+)([nslmx])(.+)?$", r"\1\3", str(path))
    for cand in (Path(unified), path, CFG_DIR / Path(unified).name, CFG_DIR / path.name):
        if cand.is_file():
            with open(cand) as f:
                d = yaml.safe_load(f)
            break
    else:
        raise FileNotFoundError(f"model yaml '{path}' not found (searched {CFG_DIR})")
    d["scale"] = guess_model_scale(path)
    d["yaml_file"] = str(path)
    return d


def parse_model(d, ch, verbose=False):
    """yaml dict -> (nn.Sequential, save list); channel rules of the reference parse_model (tasks.py:803-921)."""
    import ast
    max_channels = float("inf")
    nc, scales = d.get("nc"), d.get("scales")
    depth, width = d.get("depth_multiple", 1.0), d.get("width_multiple", 1.0)
    if scales:
        scale = d.get("scale") or tuple(scales.keys())[0]
        depth, width, max_channels = scales[scale]
    ch = [ch]
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, mname, args) in enumerate(d["backbone"] + d["head"]):
        if mname not in _REGISTRY:
            raise NotImplementedError(f"module '{mname}' is outside the Dedark-YOLO hot path (SURVEY.md 8)")
        m = _REGISTRY[mname]
        args = list(args)
        for j, a in enumerate(args):
            if isinstance(a, str):
                if a == "nc":
                    args[j] = nc
                else:
                    with contextlib.suppress(ValueError, SyntaxError):
                        args[j] = ast.literal_eval(a)
        n = n_ = max(round(n * depth), 1) if n > 1 else n
        if m in (Conv, C2f, SPPF):
            c1, c2 = ch[f], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_channels) * width, 8)
            args = [c1, c2, *args[1:]]
            if m is C2f:
                args.insert(2, n)
                n = 1
        elif m is Concat:
            c2 = sum(ch[x] for x in f)
        elif m is lowlight_recovery:
            c2 = args[0]
        elif m is AsffTribeLevel:
            c2 = 512 if args[0] in (0, 1) else 256
        elif m is Detect:
            args.append([ch[x] for x in f])
        else:
            c2 = ch[f]
        m_ = nn.Sequential(*(m(*args) for _ in range(n))) if n > 1 else m(*args)
        m_.np = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type = i, f, mname
        if verbose:
            print(f"{i:>3}{str(f):>20}{n_:>3}{m_.np:10.0f}  {mname:<45}{str(args):<30}")
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)


def initialize_weights(model):
    """reference ultralytics/utils/torch_utils.py:257-267."""
    for m in model.modules():
        if type(m) is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03


class BaseModel(nn.Module):
    """reference tasks.py:29-253."""

    def __init__(self):
        super().__init__()
        self.current_dedark_A = None
        self.current_IcA = None

    def forward(self, x, *args, **kwargs):
        if isinstance(x, dict):
            return self.loss(x, *args, **kwargs)
        return self.predict(x, *args, **kwargs)

    def predict(self, x, profile=False, visualize=False, augment=False):
        return self._predict_once(x)

    def _predict_once(self, x, profile=False, visualize=False):
        if isinstance(x, dict):
            img = x.get("img", x.get("clean_img", None))
            self.current_dedark_A = x.get("dedark_A", None)
            self.current_IcA = x.get("IcA", None)
            x = img
        else:
            self.current_dedark_A = None
            self.current_IcA = None
        ops.arena.reset()
        y = []
        for m in self.model:
            if m.f != -1:
                x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
            if isinstance(m, lowlight_recovery) and not self.training:
                x = m(x, self.current_dedark_A, self.current_IcA)
            else:
                x = m(x)
            y.append(x if m.i in self.save else None)
        return x

    def loss(self, batch, preds=None):
        if not hasattr(self, "criterion"):
            self.criterion = self.init_criterion()
        preds = self._predict_once(batch) if preds is None else preds
        return self.criterion(preds, batch)

    def init_criterion(self):
        raise NotImplementedError

    def fuse(self, verbose=True):
        """The HIP conv folds BatchNorm into its epilogue at call time in eval mode; nothing to rewrite."""
        return self

    def is_fused(self, thresh=10):
        return False

    def load(self, weights, verbose=True):
        """reference tasks.py:222-234: intersect by name and shape, non-strict load."""
        model = weights["model"] if isinstance(weights, dict) and "model" in weights else weights
        csd = model.float().state_dict() if hasattr(model, "state_dict") else model
        own = self.state_dict()
        csd = {k: v for k, v in csd.items() if k in own and own[k].shape == v.shape}
        self.load_state_dict(csd, strict=False)
        ops.bump_weights_epoch()
        return len(csd)

    def state_dict(self, *args, **kwargs):
        ops.flush_bn_counters()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        ops.bump_weights_epoch()
        return r

    def _apply(self, fn):
        self = super()._apply(fn)
        m = self.model[-1]
        if isinstance(m, Detect):
            m.stride = fn(m.stride)
            m.anchors = fn(m.anchors)
            m.strides = fn(m.strides)
        ops.bump_weights_epoch()
        return self


class DetectionModel(BaseModel):
    """YOLOv8 detection model (reference tasks.py:257-344)."""

    def __init__(self, cfg="yolov8n.yaml", ch=3, nc=None, verbose=False):
        super().__init__()
        self.yaml = cfg if isinstance(cfg, dict) else yaml_model_load(cfg)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            self.yaml["nc"] = nc
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=ch, verbose=verbose)
        self.names = {i: f"{i}" for i in range(self.yaml["nc"])}
        self.inplace = self.yaml.get("inplace", True)
        m = self.model[-1]
        if isinstance(m, Detect):
            # The reference probes strides with two train-mode forwards of zeros(1,ch,256,256) (tasks.py:284-292); here they
            # follow from the graph (no GPU needed at construction). BatchNorm buffers therefore start at their nn defaults.
            m.stride = torch.tensor(self._graph_strides())
            self.stride = m.stride
            m.bias_init()
        else:
            self.stride = torch.Tensor([32])
        initialize_weights(self)

    def _graph_strides(self):
        s = []
        for L in self.model:
            f = L.f
            prev = (s[L.i - 1] if L.i > 0 else 1) if f == -1 else None
            if isinstance(L, Conv):
                base = prev if f == -1 else s[f]
                s.append(base * L.conv.stride[0])
            elif isinstance(L, Upsample):
                s.append((prev if f == -1 else s[f]) / L.scale_factor)
            elif isinstance(L, Concat):
                s.append(s[L.i - 1] if f[0] == -1 else s[f[0]])
            elif isinstance(L, AsffTribeLevel):
                s.append(s[f[L.level]])
            elif isinstance(L, Detect):
                return [float(s[j]) for j in f]
            else:
                s.append(prev if f == -1 else s[f])
        raise RuntimeError("no Detect layer")

    def init_criterion(self):
        from ..utils.loss import RcoveryDetectionLoss
        return RcoveryDetectionLoss(self)
