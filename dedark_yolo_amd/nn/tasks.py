"""Model graph / registry: yaml -> module list, layer-graph executor, criterion dispatch.

Mirrors the reference ultralytics/nn/tasks.py: BaseModel (:29-253), DetectionModel (:257-344), parse_model (:803-921),
yaml_model_load (:924-946), guess_model_scale (:950-965) -- same names, arguments and `model.<i>.` state_dict prefixes.
"""
import contextlib
import math
import os
import re
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn
import yaml

from .. import ops
from .modules import (AsffDetect, AsffDoubLevel, AsffTribeLevel, C2f, Concat, Conv, Detect, DyModule, MFRU, RFBblock, SPPF, Tape,
                      Upsample, lowlight_recovery)

# One autograd node for the whole layer graph (training): the plan walks its nodes forwards with one Tape per module and backwards in
# reverse, adding the gradients of a multi-consumer output itself -- no autograd.Function per yaml node, no ATen `add` for the fan-outs.
# DY_GRAPH_BACKWARD=0 keeps one autograd.Function per module (modules.py:_ModuleFn).
_GRAPH_BACKWARD = os.environ.get("DY_GRAPH_BACKWARD", "1") != "0"

CFG_DIR = Path(__file__).resolve().parent.parent / "cfg" / "models" / "v8"

_REGISTRY = dict(Conv=Conv, C2f=C2f, SPPF=SPPF, Concat=Concat, Detect=Detect, AsffDetect=AsffDetect, AsffTribeLevel=AsffTribeLevel,
                 AsffDoubLevel=AsffDoubLevel, MFRU=MFRU, RFBblock=RFBblock, lowlight_recovery=lowlight_recovery)
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


# ------------------------------------------------------------------------------------------------ yaml -> graph
# The yaml grammar ([from, repeats, module, args] rows, `scales`, 'nc' substitution) and the channel rules per module kind are
# the drop-in contract (reference parse_model, tasks.py:803-921).  Here each module kind registers ONE rule that turns a row into
# (constructor arguments, output channels, module repeats), and the rows are compiled ONCE into a flat execution plan
# (GraphPlan: per node its source nodes and the node after which its output is dead), which forward passes just walk.
class _Row:
    __slots__ = ("index", "src", "repeats", "kind", "args", "ch_in", "nc", "width", "max_ch")


def _scaled_width(c, row):
    return make_divisible(min(c, row.max_ch) * row.width, 8)


def _rule_conv_like(row):                 # Conv, SPPF: [c_out, ...] -> (c_in, scaled c_out, ...)
    c2 = row.args[0] if row.args[0] == row.nc else _scaled_width(row.args[0], row)
    return [row.ch_in[0], c2, *row.args[1:]], c2, row.repeats


def _rule_c2f(row):                       # the row's repeat count becomes the number of inner Bottlenecks
    args, c2, _ = _rule_conv_like(row)
    return [args[0], args[1], row.repeats, *args[2:]], c2, 1


_RULES = {
    Conv: _rule_conv_like, SPPF: _rule_conv_like, C2f: _rule_c2f,
    Concat: lambda r: (r.args, sum(r.ch_in), r.repeats),
    lowlight_recovery: lambda r: (r.args, r.args[0], r.repeats),
    AsffTribeLevel: lambda r: (r.args, 512 if r.args[0] in (0, 1) else 256, r.repeats),       # tasks.py:892-896
    AsffDoubLevel: lambda r: (r.args, 512 if r.args[0] == 0 else 256, r.repeats),
    MFRU: lambda r: (r.args, 256, r.repeats),                                                  # tasks.py:890-891
    Detect: lambda r: ([*r.args, list(r.ch_in)], r.ch_in[0], r.repeats),
    AsffDetect: lambda r: ([*r.args, list(r.ch_in)], r.ch_in[0], r.repeats),
}
_PASS_THROUGH = lambda r: (r.args, r.ch_in[0], r.repeats)         # Upsample, RFBblock: channels unchanged


_PLACEABLE = (Conv, C2f, SPPF, Upsample)          # top-level modules whose last kernel can write into a caller-provided NHWC view


def _out_hw(m, x):
    """Spatial size of m(x) for the placeable module types."""
    H, W = x.shape[2], x.shape[3]
    if isinstance(m, Conv):
        c = m.conv
        k, st, p, d = c.kernel_size[0], c.stride[0], c.padding[0], c.dilation[0]
        return (H + 2 * p - d * (k - 1) - 1) // st + 1, (W + 2 * p - d * (k - 1) - 1) // st + 1
    if isinstance(m, Upsample):
        return H * m.scale_factor, W * m.scale_factor
    return H, W


class GraphPlan:
    """Flat execution plan of a layer graph: nodes[i] = (module, sources, takes_list); a source is an absolute node index or -1
    for the network input.  `dead_after[i]` lists the nodes whose outputs have their last consumer at node i.

    yaml-level Concat without copies: when every source of a Concat node is a Conv / C2f / SPPF / Upsample node that is not already
    placed elsewhere (and all widths are multiples of 8), `place[s] = (concat node, channel offset)`: node s then writes its output
    straight into its channel slice of the concat buffer (allocated when the first source runs; other consumers of s read the
    slice as a strided view), and Concat finds its inputs already in place (reference conv.py:462-473 = torch.cat copies)."""

    def __init__(self, layers):
        self.nodes, n = [], len(layers)
        last_use = [-1] * n
        for i, m in enumerate(layers):
            f = m.f
            rel = [f] if isinstance(f, int) else list(f)
            src = tuple((i - 1 if j == -1 else (j if j >= 0 else i + j)) for j in rel)
            self.nodes.append((m, src, not isinstance(f, int)))
            for sidx in src:
                if sidx >= 0:
                    last_use[sidx] = i
        self.dead_after = [[] for _ in range(n)]
        for sidx, i in enumerate(last_use):
            if 0 <= i < n - 1:
                self.dead_after[i].append(sidx)
        self.save = sorted({sidx for m, src, _ in self.nodes for sidx, j in zip(src, ([m.f] if isinstance(m.f, int) else m.f)) if j != -1})
        self.place, self.concat_width = {}, {}
        for i, (m, src, _) in enumerate(self.nodes):
            if not isinstance(m, Concat) or len(set(src)) != len(src):
                continue
            widths = [getattr(layers[sidx], "c_out", None) if sidx >= 0 else None for sidx in src]
            ok = all(sidx >= 0 and isinstance(layers[sidx], _PLACEABLE) and sidx not in self.place and w and w % 8 == 0
                     for sidx, w in zip(src, widths))
            if not ok:
                continue
            off = 0
            for sidx, w in zip(src, widths):
                self.place[sidx] = (i, off)
                off += w
            self.concat_width[i] = off

    # ---- training: the whole graph behind ONE autograd.Function (_GraphFn below)
    def trainable(self):
        """Every node is a DyModule (or an nn.Sequential of them): the explicit backward walk covers the graph."""
        ok = self.__dict__.get("_trainable")
        if ok is None:
            ok = all(all(isinstance(mod, DyModule) for mod in (m if isinstance(m, nn.Sequential) else [m])) for m, _, _ in self.nodes)
            self.__dict__["_trainable"] = ok
        return ok

    def forward_train(self, x, front_args):
        n = len(self.nodes)
        outs, bufs = [None] * n, {}
        st = dict(tapes=[None] * n, mods=[None] * n, metas=[None] * n)
        for i, (m, src, as_list) in enumerate(self.nodes):
            mods = list(m) if isinstance(m, nn.Sequential) else [m]
            cur = [x if sidx < 0 else outs[sidx] for sidx in src]
            slot = self.place.get(i)
            view = None
            if slot is not None:
                c, off = slot
                buf = bufs.get(c)
                if buf is None:
                    t = cur[0]
                    Ho, Wo = _out_hw(m, t)
                    buf = bufs[c] = ops.empty_nhwc(t.shape[0], self.concat_width[c], Ho, Wo, ops.get_compute_dtype(), t.device)
                view = buf[:, off:off + m.c_out]
            tapes, metas = [], []
            for j, mod in enumerate(mods):
                tape = Tape()
                if isinstance(mod, lowlight_recovery):
                    mod._A, mod._I = front_args           # (None, None) in training: the extractor predicts the filter parameters
                xs = [mod._adapt(t) for t in cur]
                o = mod._fwd(tape, *xs, out=view) if (view is not None and j == len(mods) - 1) else mod._fwd(tape, *xs)
                ol = list(o) if isinstance(o, (list, tuple)) else [o]
                tapes.append(tape)
                metas.append([(t.shape, t.dtype, t.device) for t in ol])
                cur = ol
            st["tapes"][i], st["mods"][i], st["metas"][i] = tapes, mods, metas
            outs[i] = cur if len(cur) > 1 or isinstance(o, (list, tuple)) else cur[0]
            if i in bufs:
                del bufs[i]
            for sidx in self.dead_after[i]:
                outs[sidx] = None
        st["multi"] = isinstance(outs[-1], (list, tuple))
        return outs[-1], st

    def backward_train(self, st, gouts, x_needs):
        from ..ops import as_nhwc, copy2d
        n = len(self.nodes)
        grads = [None] * n
        grads[-1] = list(gouts) if st["multi"] else gouts[0]
        pgrads, gx = {}, None
        for i in reversed(range(n)):
            m, src, _ = self.nodes[i]
            g, grads[i] = grads[i], None
            tapes, mods, metas = st["tapes"][i], st["mods"][i], st["metas"][i]
            st["tapes"][i] = None
            if g is None:
                continue                                   # nothing downstream of this node reached the loss
            gl = list(g) if isinstance(g, (list, tuple)) else [g]
            for j in reversed(range(len(mods))):
                mod, tape, meta = mods[j], tapes[j], metas[j]
                gl = [gg if gg is not None else torch.zeros(mm[0], dtype=mm[1], device=mm[2]) for gg, mm in zip(gl, meta)]
                if not (getattr(mod, "_planar_grad_ok", False) and all(gg.is_contiguous() and gg.dtype == mm[1] for gg, mm in zip(gl, meta))):
                    gl = [as_nhwc(gg, mm[1]) for gg, mm in zip(gl, meta)]
                needs = [(sidx >= 0 or x_needs) for sidx in src] if j == 0 else [True]
                into = None
                if j == 0 and getattr(mod, "_takes_into", False):
                    # fan-out: a later consumer already left a gradient for an input of this node -- the module adds into it (the
                    # data gradient of a Conv, the blend / pooling / resize adjoints of an ASFF level) and returns None for that input
                    # (only a gradient that is a dense NHWC tensor of exactly the producer's output shape qualifies: the modules add
                    # through raw pointers; anything else takes the out-of-place sum below)
                    into = [grads[sidx] if (sidx >= 0 and _accumulable(grads[sidx], st["metas"][sidx], meta[0][1])) else None for sidx in src]
                    if all(t is None for t in into):
                        into = None
                gins = mod._bwd(tape, *gl, needs=needs, into=into) if into is not None else mod._bwd(tape, *gl, needs=needs)
                gl = list(gins) if isinstance(gins, (list, tuple)) else [gins]
                assert not tape.stack, f"{type(mod).__name__}: unbalanced tape"
                for p, gp in tape.pgrads.items():
                    pgrads[id(p)] = gp if id(p) not in pgrads else pgrads[id(p)] + gp
            cb = m.__dict__.get("_dy_after_backward")
            if cb is not None:
                cb()                                       # data-parallel trainer: this layer's gradients are enqueued
            for sidx, gin in zip(src, gl):
                if gin is None:
                    continue
                if sidx < 0:
                    gx = gin if gx is None else gx + gin
                elif grads[sidx] is None:
                    grads[sidx] = gin
                elif isinstance(gin, torch.Tensor) and gin.dim() == 4 and gin.dtype == grads[sidx].dtype and gin.shape == grads[sidx].shape:
                    copy2d(gin, grads[sidx], accumulate=True)      # fan-out: add into the gradient the later consumer left (ours alone)
                else:
                    grads[sidx] = grads[sidx] + gin
        return gx, pgrads

    def run(self, x, call_layer):
        outs = [None] * len(self.nodes)
        bufs = {}                                       # concat node -> its buffer of this pass
        for i, (m, src, as_list) in enumerate(self.nodes):
            ins = [x if sidx < 0 else outs[sidx] for sidx in src]
            slot = self.place.get(i)
            if slot is None:
                outs[i] = call_layer(m, ins if as_list else ins[0])
            else:
                c, off = slot
                buf = bufs.get(c)
                if buf is None:
                    t = ins[0]
                    Ho, Wo = _out_hw(m, t)
                    buf = bufs[c] = ops.empty_nhwc(t.shape[0], self.concat_width[c], Ho, Wo, ops.get_compute_dtype(), t.device)
                outs[i] = call_layer(m, ins if as_list else ins[0], out=buf[:, off:off + m.c_out])
            if i in bufs:
                del bufs[i]                             # the Concat node has run: its output tensor owns the buffer now
            for sidx in self.dead_after[i]:
                outs[sidx] = None                     # last consumer done: the buffer goes back to the allocator now
        return outs[-1]


def _accumulable(g, producer_metas, dtype):
    """may a module add into gradient tensor `g` (left by another consumer of the same producer) through raw pointers?"""
    if not (torch.is_tensor(g) and g.dim() == 4 and g.dtype == dtype):
        return False
    want = producer_metas[-1][0][0] if producer_metas else None          # output shape of the producer node's last module
    if want is not None and tuple(g.shape) != tuple(want):
        return False
    st = g.stride()
    return st[1] == 1 and st[3] >= g.shape[1] and st[2] == g.shape[3] * st[3] and st[0] == g.shape[2] * st[2]      # pixel-major view


class _GraphFn(torch.autograd.Function):
    """forward(plan, front_args, n_params, x, *params) -> the last node's output(s); backward = GraphPlan.backward_train."""

    @staticmethod
    def forward(ctx, plan, front_args, params, x, *ptensors):
        out, st = plan.forward_train(x, front_args)
        ctx.plan, ctx.st, ctx.params = plan, st, params
        return tuple(out) if st["multi"] else out

    @staticmethod
    def backward(ctx, *gouts):
        gx, pgrads = ctx.plan.backward_train(ctx.st, gouts, bool(ctx.needs_input_grad[3]))
        ctx.st = None
        return (None, None, None, gx, *[pgrads.get(id(p)) for p in ctx.params])


def parse_model(d, ch, verbose=False):
    """yaml dict -> (nn.Sequential of layers carrying .i / .f / .type / .np, save list) -- the reference's return contract."""
    import ast
    nc, scales = d.get("nc"), d.get("scales")
    depth, width, max_ch = d.get("depth_multiple", 1.0), d.get("width_multiple", 1.0), float("inf")
    if scales:
        depth, width, max_ch = scales[d.get("scale") or next(iter(scales))]
    widths, layers = [], []                  # output channels per node
    for index, (src, repeats, kind, args) in enumerate(d["backbone"] + d["head"]):
        cls = _REGISTRY.get(kind)
        if cls is None:
            raise NotImplementedError(f"module '{kind}' is outside the Dedark-YOLO hot path (SURVEY.md 8)")
        row = _Row()
        row.index, row.src, row.kind, row.nc, row.width, row.max_ch = index, src, kind, nc, width, max_ch
        row.args = [nc if a == "nc" else _literal(a, ast) for a in args]
        row.repeats = max(round(repeats * depth), 1) if repeats > 1 else repeats
        rel = [src] if isinstance(src, int) else list(src)
        row.ch_in = [(widths[j] if widths else ch) if j == -1 else widths[j] for j in rel]
        ctor_args, c_out, n_mod = _RULES.get(cls, _PASS_THROUGH)(row)
        layer = nn.Sequential(*(cls(*ctor_args) for _ in range(n_mod))) if n_mod > 1 else cls(*ctor_args)
        layer.np = sum(p.numel() for p in layer.parameters())
        layer.i, layer.f, layer.type = index, src, kind
        layer.c_out = c_out                       # output channels (GraphPlan places producers inside yaml-level Concat buffers)
        if verbose:
            print(f"{index:>3}{str(src):>20}{row.repeats:>3}{layer.np:10.0f}  {kind:<45}{str(ctor_args):<30}")
        layers.append(layer)
        widths.append(c_out)
    model = nn.Sequential(*layers)
    return model, GraphPlan(layers).save


def _literal(a, ast):
    if isinstance(a, str):
        with contextlib.suppress(ValueError, SyntaxError):
            return ast.literal_eval(a)
    return a


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
        plan = self.__dict__.get("_plan")
        if plan is None or len(plan.nodes) != len(self.model):
            plan = self.__dict__["_plan"] = GraphPlan(list(self.model))
        eval_front = not self.training
        if _GRAPH_BACKWARD and self.training and torch.is_grad_enabled() and torch.is_tensor(x) and plan.trainable():
            # (un)freezing after train() must not go unnoticed: the requires_grad flags are read every step -- from a cached tuple of
            # the Parameter objects (walking the module tree cost ~1 ms of a 9 ms C2 step; the set of parameters only changes with the
            # modules: train() / _apply() / fuse() drop the cache)
            allp = self.__dict__.get("_all_params")
            if allp is None:
                allp = self.__dict__["_all_params"] = tuple(self.model.parameters())
            sig = tuple([p.requires_grad for p in allp])
            cached = self.__dict__.get("_graph_params")
            if cached is None or cached[0] != sig:
                cached = self.__dict__["_graph_params"] = (sig, tuple(p for p in allp if p.requires_grad))
            params = cached[1]
            if params or x.requires_grad:
                out = _GraphFn.apply(plan, (None, None), params, x, *params)
                return list(out) if isinstance(out, tuple) else out

        def call_layer(m, inp, out=None):
            if eval_front and isinstance(m, lowlight_recovery):
                return m(inp, self.current_dedark_A, self.current_IcA)
            return m(inp) if out is None else m(inp, out=out)
        return plan.run(x, call_layer)

    def train(self, mode=True):
        self.__dict__.pop("_graph_params", None)          # (requires_grad flags are read when the mode is set)
        self.__dict__.pop("_all_params", None)
        return super().train(mode)

    def loss(self, batch, preds=None):
        if not hasattr(self, "criterion"):
            self.criterion = self.init_criterion()
        preds = self._predict_once(batch) if preds is None else preds
        return self.criterion(preds, batch)

    def init_criterion(self):
        raise NotImplementedError

    def _bn_pairs(self):
        import torch.nn as nn_
        for m in self.modules():
            bn = getattr(m, "bn", None) or getattr(m, "batch_norm", None)
            conv = getattr(m, "conv", None)
            if isinstance(bn, nn_.BatchNorm2d) and isinstance(conv, nn_.Conv2d):
                yield conv, bn

    def fuse(self, verbose=True):
        """reference tasks.py:153-178 / fuse_conv_and_bn (torch_utils.py:123-144).  The HIP conv applies BatchNorm as a per-channel
        affine in its epilogue, so fusing = computing that affine once: every Conv / add_conv gets its folded (scale, shift) cached
        and the eval forward launches no fold kernel afterwards.  The parameters stay untouched (training can continue; the caches
        are invalidated by any weight or buffer update)."""
        n = 0
        for conv, bn in self._bn_pairs():
            if bn.weight.is_cuda:
                ops.bn_fold(bn, ops.round_up(conv.out_channels, ops.vec_elems(ops.get_compute_dtype())))
                n += 1
        if verbose and n == 0:
            print("fuse(): move the model to the GPU first (the folded affines live next to the weights)")
        return self

    def is_fused(self, thresh=10):
        pairs = list(self._bn_pairs())
        return bool(pairs) and all(ops.bn_fold_is_current(bn) for _, bn in pairs)

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
        self.__dict__.pop("_graph_params", None)
        self.__dict__.pop("_all_params", None)
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

    @torch.no_grad()
    def reference_initial_buffers(self, s=256):
        """Give the BatchNorm buffers the values a freshly constructed REFERENCE model has.  Its constructor probes the strides with
        two train-mode forward passes of zeros(1, ch, 256, 256) before initialize_weights() sets eps / momentum (tasks.py:284-292,
        torch_utils.py:257-267), so its running statistics start from two eps-1e-5 / momentum-0.1 updates on the zero image and
        num_batches_tracked = 2 -- not from (0, 1).  Optional: only a from-scratch run that must follow the reference's own
        trajectory needs it (checkpoints carry their buffers).  Call after .cuda(); pinned by tests/golden/g10_initbuf.npz."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("reference_initial_buffers() runs two forward passes: move the model to the GPU first")
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        keep = [(m.eps, m.momentum) for m in bns]
        was_training = self.training
        dt = ops.get_compute_dtype()
        ops.set_compute_dtype(torch.float32)          # the reference builds its models in fp32
        try:
            for m in bns:
                m.eps, m.momentum = 1e-5, 0.1
            self.train()
            zero = torch.zeros(1, self.yaml.get("ch", 3), s, s, device=dev)
            for _ in range(2):
                self._predict_once(zero)
            ops.flush_bn_counters()
        finally:
            for m, (e, mo) in zip(bns, keep):
                m.eps, m.momentum = e, mo
            self.train(was_training)
            ops.set_compute_dtype(dt)
        return self

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
            elif isinstance(L, (AsffTribeLevel, AsffDoubLevel)):
                s.append(s[f[L.level]])
            elif isinstance(L, MFRU):
                s.append(s[f[2]])                      # fuses at the finest of its three inputs
            elif isinstance(L, Detect):
                return [float(s[j]) for j in f]
            else:
                s.append(prev if f == -1 else s[f])
        raise RuntimeError("no Detect layer")

    def init_criterion(self):
        from ..utils.loss import RcoveryDetectionLoss
        return RcoveryDetectionLoss(self)
