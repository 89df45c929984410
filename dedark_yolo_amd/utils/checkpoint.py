"""Reading reference checkpoints (`last.pt` / `best.pt`) without the reference package.

The reference saves `{'epoch', 'best_fitness', 'model': deepcopy(de_parallel(model)).half(), 'ema': deepcopy(ema.ema).half(),
'updates', 'optimizer', 'train_args', 'date', 'version'}` with torch.save (ultralytics/engine/trainer.py:408-433): `model` and
`ema` are PICKLED MODULE OBJECTS (`ultralytics.nn.tasks.DetectionModel` holding `ultralytics.nn.modules.*`), which only the
reference's own classes can rebuild.  `attempt_load_one_weight` / `torch_safe_load` (ultralytics/nn/tasks.py:592-630,674-707) then
take `ckpt.get('ema') or ckpt['model']`, cast to fp32 and read `.yaml`, `.names`, `.args`.

Here a restricted unpickler maps every class outside torch / the standard containers to an inert record that only keeps the
pickled attribute dict; walking `_modules` / `_parameters` / `_buffers` of those records reproduces `nn.Module.state_dict()`
(same key order, same names `model.<i>.<sub>...`), which `DetectionModel.load` consumes.  The only callables a file can reach are
the ones on an EXACT (module, name) allow-list (tensor / storage rebuilders, plain containers, numpy array reconstruction); every
other global -- any other torch.* or numpy.* name included -- becomes an inert record.  State-dict checkpoints written by this
package are tried with `torch.load(weights_only=True)` first and never reach the unpickler.
"""
import collections
import pickle
from types import SimpleNamespace

import torch


class _Record:
    """Stand-in for a pickled object of a class we do not have (reference modules, namespaces, paths)."""

    def __init__(self, *args, **kwargs):
        self._dy_args = args

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):      # (dict, slots) protocol
            if isinstance(state[0], dict):
                self.__dict__.update(state[0])
            self.__dict__.update(state[1])
        else:
            self._dy_state = state

    def __setitem__(self, k, v):                          # dict / list subclasses we do not have (easydict.EasyDict): SETITEM(S) / APPEND(S)
        self.__dict__.setdefault("_dy_items", {})[k] = v

    def append(self, v):
        self.__dict__.setdefault("_dy_list", []).append(v)

    def extend(self, vs):
        self.__dict__.setdefault("_dy_list", []).extend(vs)


_RECORD_TYPES = {}


def _record_type(module, name):
    key = (module, name)
    t = _RECORD_TYPES.get(key)
    if t is None:
        t = _RECORD_TYPES[key] = type(name, (_Record,), {"_dy_module": module})
    return t


# EXACT (module, name) allow-list: a pickle REDUCE may call whatever find_class returns with arguments of the file's choosing, so
# a prefix rule such as "everything under torch.* / numpy.*" hands the file every callable of those packages (exec / shell helpers
# included).  Only what torch.save needs to rebuild tensors, containers and numpy scalars is real; every other name -- other
# torch.* / numpy.* names too -- becomes an inert record.
_STORAGES = ("Float", "Half", "BFloat16", "Double", "Long", "Int", "Short", "Char", "Byte", "Bool", "ComplexFloat", "ComplexDouble")
_DTYPES = ("float32", "float", "float16", "half", "bfloat16", "float64", "double", "int64", "long", "int32", "int", "int16", "short",
           "int8", "uint8", "bool", "complex64", "complex128")
_ALLOWED_EXACT = (
    {("builtins", n) for n in ("set", "frozenset", "dict", "list", "tuple", "int", "float", "bool", "str", "bytes", "complex", "slice",
                               "range", "bytearray", "object")}
    | {("collections", "OrderedDict"), ("collections", "defaultdict"), ("copyreg", "_reconstructor"), ("_codecs", "encode")}
    | {("torch._utils", n) for n in ("_rebuild_tensor_v2", "_rebuild_tensor", "_rebuild_parameter", "_rebuild_parameter_with_state")}
    | {("torch", n + "Storage") for n in _STORAGES} | {("torch", "UntypedStorage"), ("torch.storage", "UntypedStorage"),
                                                        ("torch.storage", "TypedStorage"), ("torch", "Size"), ("torch", "device"),
                                                        ("torch", "dtype"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter")}
    | {("torch", n) for n in _DTYPES}
    | {(m, n) for m in ("numpy.core.multiarray", "numpy._core.multiarray") for n in ("_reconstruct", "scalar")}
    | {("numpy", "ndarray"), ("numpy", "dtype")})


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "__builtin__":                       # protocol-2 spelling (torch.save's default protocol)
            module = "builtins"
        if (module, name) in _ALLOWED_EXACT:
            return super().find_class(module, name)
        return _record_type(module, name)                  # ultralytics.*, torch.nn.modules.*, pathlib.*, any other torch.* / numpy.* ...


class _PickleModule:
    """The `pickle_module` argument of torch.load: same surface as `pickle`, with the restricted Unpickler."""
    __name__ = "dedark_yolo_amd.utils.checkpoint"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    PickleError = pickle.PickleError
    UnpicklingError = pickle.UnpicklingError


def _is_module_record(o):
    return isinstance(o, _Record) and isinstance(getattr(o, "_modules", None), dict) and isinstance(getattr(o, "_parameters", None), dict)


def module_state_dict(rec, prefix="", out=None):
    """nn.Module.state_dict() of a module record: own parameters, persistent buffers, then the children, in registration order."""
    out = collections.OrderedDict() if out is None else out
    for k, v in rec._parameters.items():
        if v is not None:
            out[prefix + k] = v.detach() if isinstance(v, torch.Tensor) else v
    skip = getattr(rec, "_non_persistent_buffers_set", set()) or set()
    for k, v in getattr(rec, "_buffers", {}).items():
        if v is not None and k not in skip:
            out[prefix + k] = v
    for k, m in rec._modules.items():
        if m is not None and _is_module_record(m):
            module_state_dict(m, prefix + k + ".", out)
    return out


def _plain(o, depth=0):
    """Records of namespaces / paths -> plain python values (for yaml / args / names)."""
    if isinstance(o, _Record):
        if depth > 6:
            return None
        d = {k: _plain(v, depth + 1) for k, v in o.__dict__.items() if not k.startswith("_dy_")}
        return d if d else [_plain(a, depth + 1) for a in getattr(o, "_dy_args", ())]
    if isinstance(o, dict):
        return {k: _plain(v, depth + 1) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_plain(v, depth + 1) for v in o)
    return o


def load_raw(path, device="cpu"):
    """torch.load that never runs code from the file: weights_only=True first (plain tensors / containers), the restricted
    unpickler for pickled module objects."""
    try:                                               # this package's own checkpoints (plain state_dicts) and any tensor-only file
        return torch.load(path, map_location=device, weights_only=True)
    except Exception:                                  # pickled module objects (reference last.pt / best.pt): restricted unpickler
        return torch.load(path, map_location=device, pickle_module=_PickleModule, weights_only=False)


def load_checkpoint(path, device="cpu"):
    """Reads a checkpoint written by the reference trainer OR by this package's DetectionTrainer.save_model / YOLO.save.

    Returns a SimpleNamespace with
      state_dict   fp32 weights of `ckpt.get('ema') or ckpt['model']` (reference precedence, tasks.py:640,682)
      model_sd     fp32 weights of ckpt['model'] (the raw training weights; None if absent)
      yaml, nc, names, train_args, epoch, best_fitness, updates, optimizer, source ('reference-pickle' | 'state-dict')
    """
    ck = load_raw(path, device)
    if not isinstance(ck, dict):                       # a bare pickled model
        ck = dict(model=ck)

    def to_sd(obj):
        if obj is None:
            return None, None
        if _is_module_record(obj):
            return module_state_dict(obj), obj
        if isinstance(obj, dict):
            return collections.OrderedDict(obj), None
        raise RuntimeError(f"checkpoint entry of type {type(obj).__name__} is neither a module nor a state_dict")

    ema_sd, ema_rec = to_sd(ck.get("ema"))
    model_sd, model_rec = to_sd(ck.get("model") if ck.get("model") is not None else ck.get("state_dict"))
    sd = ema_sd if ema_sd else model_sd
    if sd is None:
        raise RuntimeError(f"{path}: no 'ema' / 'model' / 'state_dict' entry")
    f32 = lambda d: None if d is None else collections.OrderedDict(
        (k, v.float() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in d.items())
    rec = ema_rec if ema_sd else model_rec
    yaml_ = ck.get("yaml") or ck.get("cfg")
    names, nc = None, ck.get("nc")
    if rec is not None:
        yaml_ = _plain(getattr(rec, "yaml", None)) or yaml_
        names = _plain(getattr(rec, "names", None))
        if isinstance(yaml_, dict) and nc is None:
            nc = yaml_.get("nc")
    ta = ck.get("train_args")
    return SimpleNamespace(state_dict=f32(sd), model_sd=f32(model_sd), yaml=yaml_, nc=nc, names=names,
                           train_args=_plain(ta) if ta is not None else None, epoch=ck.get("epoch"),
                           best_fitness=ck.get("best_fitness"), updates=ck.get("updates"), optimizer=_plain(ck.get("optimizer")),
                           dy_state=_plain(ck.get("dy_state")),
                           source="reference-pickle" if rec is not None else "state-dict")


def intersect_dicts(da, db, exclude=()):
    """reference torch_utils.py:303-305: keys of `da` that exist in `db` with the same shape (minus excluded substrings)."""
    return {k: v for k, v in da.items() if k in db and all(x not in k for x in exclude) and v.shape == db[k].shape}


# ----------------------------------------------------------------------------------------------------------------------------
# Writing checkpoints the REFERENCE loads (write side).
#
# The reference trainer pickles module OBJECTS: {'model': deepcopy(model).half(), 'ema': deepcopy(ema.ema).half(), ...}
# (ultralytics/engine/trainer.py:408-433), and its loader does `(ckpt.get('ema') or ckpt['model']).to(device).float()`, `.fuse()`,
# `.eval()` on what torch.load returns (ultralytics/nn/tasks.py:592-630,674-707).  A file it can load therefore has to name ITS
# classes (`ultralytics.nn.tasks.DetectionModel`, `ultralytics.nn.modules.*`) in the pickle stream and give every instance the
# attribute dict its forward() reads.  Nothing of the reference ships here: each module of THIS package's model (whose constructor
# arguments, attribute and child names mirror the reference's, that is what makes the state_dict keys equal) is turned into an
# inert stand-in object whose class only carries the reference's module path and name; a pickler subclass writes that path as the
# GLOBAL opcode.  torch.nn leaves (Conv2d, BatchNorm2d, SiLU, Sequential, ...) are real torch modules.  The layout is checked on the
# CPU against tests/golden/g12_ref_skeleton.json (what the reference itself pickles) and, in the build container, by loading the
# file with the reference (tests/golden/make_ckpt_interop.py).
_REF_HOME = {"DetectionModel": "ultralytics.nn.tasks"}
_REF_HOME.update({n: "ultralytics.nn.modules.conv" for n in ("Conv", "Concat", "SCConv", "SRU", "CRU", "GroupBatchnorm2d")})
_REF_HOME.update({n: "ultralytics.nn.modules.block" for n in ("C2f", "Bottleneck", "SPPF", "DFL", "AsffTribeLevel", "AsffDoubLevel", "MFRU",
                                                               "RFBblock")})
_REF_HOME.update({n: "ultralytics.nn.modules.head" for n in ("Detect", "AsffDetect")})
_REF_HOME.update({"lowlight_recovery": "ultralytics.nn.modules.llie", "ExtractParameters2": "ultralytics.nn.modules.common",
                  "ConvBlock": "ultralytics.nn.modules.common"})
# the front-end's filter objects (ultralytics/nn/modules/filtersB.py, filter_cfg.py:17-75): parameter-free modules holding these
# constants; the product's fused kernels have no such objects, so the writer emits them from this table
_FILTER_CFG = dict(num_filter_parameters=15, dedark_begin_param=0, wb_begin_param=1, gamma_begin_param=4, tone_begin_param=5,
                   contrast_begin_param=13, usm_begin_param=14, curve_steps=8, gamma_range=3, exposure_range=3.5, wb_range=1.1,
                   color_curve_range=(0.90, 1.10), lab_curve_range=(0.90, 1.10), tone_curve_range=(0.5, 2), defog_range=(0.1, 1.0),
                   usm_range=(0.0, 5), masking=False, minimum_strength=0.3, maximum_sharpness=1, clamp=False, source_img_size=64,
                   base_channels=32, dropout_keep_prob=0.5, share_feed_dict=True, shared_feature_extractor=True, fc1_size=128, bnw=False,
                   feature_extractor_dims=4096)
_FILTERS = (("DeDarkFilter", dict(num_filter_parameters=1, short_name="DF", filter_parameters=None, begin_filter_parameter=0)),
            ("ImprovedWhiteBalanceFilter", dict(num_filter_parameters=3, short_name="W", filter_parameters=None, channels=3,
                                                begin_filter_parameter=1)),
            ("GammaFilter", dict(num_filter_parameters=1, short_name="G", filter_parameters=None, begin_filter_parameter=4)),
            ("ContrastFilter", dict(num_filter_parameters=1, short_name="Ct", filter_parameters=None, begin_filter_parameter=13)),
            ("UsmFilter", dict(num_filter_parameters=1, short_name="UF", filter_parameters=None, begin_filter_parameter=14)))

# plain (non-module, non-parameter) attributes an instance of each reference class carries (g12_ref_skeleton.json); the product's
# modules hold the same names with the same values plus a few of their own, which must not travel
_REF_ATTRS = {"Conv": (), "Concat": ("d",), "C2f": ("c",), "Bottleneck": ("add",), "SPPF": (), "DFL": ("c1",),
              "AsffTribeLevel": ("level", "dim", "inter_dim"), "AsffDoubLevel": ("level", "dim", "inter_dim"), "MFRU": (), "RFBblock": (),
              "SCConv": (), "SRU": ("gate_treshold",), "CRU": ("up_channel", "low_channel"), "GroupBatchnorm2d": ("group_num", "eps"),
              "Detect": ("nc", "nl", "reg_max", "no", "stride"), "AsffDetect": ("nc", "nl", "reg_max", "no", "stride"),
              "lowlight_recovery": (), "ExtractParameters2": ("output_dim", "channels"), "ConvBlock": ()}

_STANDINS = {}


class _StandIn:
    """Instance of a class this package does not have; pickled as `<module> <name>` + attribute dict."""


def _standin_type(module, name, base=_StandIn):
    t = _STANDINS.get((module, name))
    if t is None:
        t = _STANDINS[(module, name)] = type(name, (base,), {"__module__": module, "__qualname__": name, "_dy_standin": True})
    return t


class _RefPickler(pickle._Pickler):
    """Pure-python pickler (torch.save subclasses `pickle_module.Pickler`) that writes stand-in classes as bare GLOBAL references:
    the stock save_global insists on importing the named module, which by design does not exist here."""

    def save_global(self, obj, name=None):
        if isinstance(obj, type) and obj.__dict__.get("_dy_standin"):
            self.write(pickle.GLOBAL + obj.__module__.encode() + b"\n" + obj.__qualname__.encode() + b"\n")
            self.memoize(obj)
            return
        super().save_global(obj, name)

    dispatch = dict(pickle._Pickler.dispatch)
    dispatch[type] = save_global


class _RefPickleModule:
    __name__ = "dedark_yolo_amd.utils.checkpoint"
    Pickler = _RefPickler
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    dump = staticmethod(lambda o, f, protocol=2: _RefPickler(f, protocol).dump(o))
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    PickleError = pickle.PickleError
    PicklingError = pickle.PicklingError
    UnpicklingError = pickle.UnpicklingError


_NN_BOOKKEEPING = None


def _nn_base_state(training):
    """the attribute dict nn.Module.__init__ creates (hook tables etc.), fresh per object"""
    global _NN_BOOKKEEPING
    if _NN_BOOKKEEPING is None:
        _NN_BOOKKEEPING = list(torch.nn.Module().__dict__.keys())
    d = torch.nn.Module().__dict__
    d["training"] = bool(training)
    return d


def _half(v):
    return v.detach().to("cpu").half() if v.is_floating_point() else v.detach().to("cpu").clone()


def _plain_attr(v):
    if isinstance(v, (bool, int, float, str, type(None))):
        return True
    if isinstance(v, (list, tuple)):
        return all(_plain_attr(x) for x in v)
    if isinstance(v, dict):
        return all(isinstance(k, (str, int)) and _plain_attr(x) for k, x in v.items())
    return False


class _RefWriter:
    def __init__(self, state, training):
        self.sd, self.training = state, training
        ed = _standin_type("easydict", "EasyDict", dict)
        self.cfg = ed(_FILTER_CFG)                       # ONE object shared by the extractor and the five filters, like filter_cfg.cfg
        self.cfg.__dict__.update(_FILTER_CFG)            # (easydict keeps items and attributes in step)
        self.filters = None

    def tensor(self, key, like):
        v = self.sd.get(key)
        if v is None:
            raise KeyError(f"save_reference_checkpoint: state has no entry '{key}'")
        if tuple(v.shape) != tuple(like.shape):
            raise ValueError(f"save_reference_checkpoint: '{key}' has shape {tuple(v.shape)}, the model expects {tuple(like.shape)}")
        return _half(v)

    def fill(self, state, mod, prefix):
        """_parameters / _buffers / _modules of `mod` into `state` (half precision, children converted)"""
        state["_parameters"] = collections.OrderedDict(
            (k, None if p is None else torch.nn.Parameter(self.tensor(prefix + k, p), requires_grad=p.requires_grad))
            for k, p in mod._parameters.items())
        nonp = getattr(mod, "_non_persistent_buffers_set", set())
        state["_buffers"] = collections.OrderedDict(
            (k, None if b is None else (_half(b) if k in nonp else self.tensor(prefix + k, b))) for k, b in mod._buffers.items())
        state["_non_persistent_buffers_set"] = set(nonp)
        state["_modules"] = collections.OrderedDict(
            (k, None if c is None else self.convert(c, prefix + k + ".")) for k, c in mod._modules.items())

    def torch_leaf(self, mod, prefix, cls=None, extra=None, children=None):
        """a REAL torch.nn module object (Conv2d, BatchNorm2d, activations, containers): own attribute dict, fresh hook tables"""
        cls = cls or type(mod)
        obj = cls.__new__(cls)
        state = _nn_base_state(self.training)
        for k, v in mod.__dict__.items():
            if k not in state and not k.startswith("_dy") and k != "_plist":
                state[k] = v
        self.fill(state, mod, prefix)
        if children is not None:
            state["_modules"] = children
        if extra:
            state.update(extra)
        obj.__dict__.update(state)
        return obj

    def graph_attrs(self, mod, ref_type):
        out = {}
        if hasattr(mod, "i") and hasattr(mod, "f"):
            out.update(i=mod.i, f=mod.f, type=ref_type)
        return out

    def convert(self, mod, prefix):
        nn = torch.nn
        name = type(mod).__name__
        if type(mod).__module__.startswith("torch.nn."):
            extra = {}
            if isinstance(mod, (nn.SiLU, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.Hardswish)):
                extra["inplace"] = True                   # initialize_weights (ultralytics/utils/torch_utils.py:266-267)
            extra.update(self.graph_attrs(mod, type(mod).__module__ + "." + name))
            return self.torch_leaf(mod, prefix, extra=extra)
        if name == "Upsample":                            # yaml-level nn.Upsample (parse_model resolves 'nn.Upsample' to torch's class)
            real = nn.Upsample(None, float(mod.scale_factor), mod.mode)
            return self.torch_leaf(real, prefix, extra=self.graph_attrs(mod, "torch.nn.modules.upsampling.Upsample"))
        if name == "AddConv":                             # add_conv(): nn.Sequential(conv, batch_norm, leaky) (block.py:24-45)
            kids = collections.OrderedDict((k, self.convert(c, prefix + k + ".")) for k, c in mod._modules.items())
            return self.torch_leaf(nn.Sequential(), prefix, children=kids)
        home = _REF_HOME.get(name)
        if home is None:
            raise NotImplementedError(f"save_reference_checkpoint: no reference class known for module {type(mod).__module__}.{name}")
        obj = _standin_type(home, name)()
        state = _nn_base_state(self.training)
        for k in _REF_ATTRS[name]:                        # the constructor's plain attributes (c, add, level, dim, nc, nl, reg_max, no, ...)
            if k not in mod.__dict__:
                raise RuntimeError(f"save_reference_checkpoint: {name} lacks the attribute '{k}' the reference's forward reads")
            v = mod.__dict__[k]
            state[k] = _half(v) if torch.is_tensor(v) else v
        self.fill(state, mod, prefix)
        kids = state["_modules"]
        if name == "lowlight_recovery":
            kids["filters"] = self.filter_list()
        elif name == "ExtractParameters2":
            state["cfg"] = self.cfg
        elif name == "SPPF":                              # the pooling layer is an object there (block.py:331), a kernel argument here
            kids["m"] = self.torch_leaf(nn.MaxPool2d(kernel_size=mod.k, stride=1, padding=mod.k // 2), "")
        elif name == "SRU":
            kids["sigomid"] = self.torch_leaf(nn.Sigmoid(), "")
        elif name == "CRU":
            kids["advavg"] = self.torch_leaf(nn.AdaptiveAvgPool2d(1), "")
        elif name in ("Detect", "AsffDetect"):
            state["inplace"] = True
            state["anchors"] = torch.empty(0, dtype=torch.float16)       # BaseModel._apply moves stride / anchors / strides
            state["strides"] = torch.empty(0, dtype=torch.float16)       # (tasks.py:203-220); rebuilt at the first eval forward
            state.pop("shape", None)
        state.update(self.graph_attrs(mod, home + "." + name))
        obj.__dict__.update(state)
        return obj

    def filter_list(self):
        if self.filters is None:
            nn = torch.nn
            objs = collections.OrderedDict()
            for k, (fname, attrs) in enumerate(_FILTERS):
                o = _standin_type("ultralytics.nn.modules.filtersB", fname)()
                st = _nn_base_state(self.training)
                st.update(cfg=self.cfg, **attrs)
                o.__dict__.update(st)
                objs[str(k)] = o
            self.cfg["filters"] = list(objs.values())          # filter_cfg.py:75
            self.cfg.__dict__["filters"] = self.cfg["filters"]
            self.filters = self.torch_leaf(nn.ModuleList(), "", children=objs)
        return self.filters

    def model(self, model, args):
        obj = _standin_type("ultralytics.nn.tasks", "DetectionModel")()
        state = _nn_base_state(self.training)
        layers = list(model.model)
        save = sorted(x % m.i for m in layers for x in ([m.f] if isinstance(m.f, int) else m.f) if x != -1)      # tasks.py:913
        yaml_ = dict(model.yaml) if isinstance(getattr(model, "yaml", None), dict) else {}
        nc = int(getattr(layers[-1], "nc", yaml_.get("nc", 0)))
        names = getattr(model, "names", None) or {i: f"{i}" for i in range(nc)}
        state.update(current_dedark_A=None, current_IcA=None, yaml=yaml_, save=save, names=dict(names), inplace=True)
        state["_modules"] = collections.OrderedDict(model=self.torch_leaf(
            model.model, "model.", children=collections.OrderedDict((k, self.convert(c, f"model.{k}.")) for k, c in model.model._modules.items())))
        stride = getattr(model, "stride", None)
        state["stride"] = stride.detach().float().cpu().clone() if torch.is_tensor(stride) else torch.tensor([32.0])
        if args is not None:
            state["args"] = dict(args)
        for k in ("nc", "task", "pt_path"):
            if k in model.__dict__ and _plain_attr(model.__dict__[k]):
                state[k] = model.__dict__[k]
        obj.__dict__.update(state)
        return obj


def reference_module_object(model, state=None, training=True, args=None):
    """The stand-in object tree for `model` (a dedark_yolo_amd DetectionModel) with half-precision tensors taken from `state`
    (a state_dict-like mapping with the reference's keys; default: the model's own)."""
    sd = model.state_dict() if state is None else state
    return _RefWriter(sd, training).model(model, args)


def save_reference_checkpoint(path, model, ema_state=None, epoch=-1, best_fitness=None, updates=0, optimizer=None, train_args=None,
                              model_args=None, version="8.0.142", extra=None):
    """Writes `path` in the reference trainer's own format (ultralytics/engine/trainer.py:408-433): pickled half-precision
    DetectionModel objects under 'model' (the training weights, train mode) and 'ema' (`ema_state`, eval mode; None when no EMA
    exists), counters, `optimizer` (a torch.optim-style state_dict or None), `train_args`, date, version.  The reference's
    attempt_load_one_weight / torch_safe_load / check_resume read it; so does load_checkpoint above."""
    import datetime
    ta = dict(train_args or {})
    margs = model_args if model_args is not None else {k: ta[k] for k in ("box", "cls", "dfl", "lrl") if k in ta} or None
    ckpt = {"epoch": int(epoch), "best_fitness": best_fitness,
            "model": reference_module_object(model, None, True, margs),
            "ema": None if ema_state is None else reference_module_object(model, ema_state, False, margs),
            "updates": int(updates), "optimizer": optimizer, "train_args": ta, "date": datetime.datetime.now().isoformat(),
            "version": version}
    ckpt.update(extra or {})                              # keys the reference ignores (it reads the ones above by name)
    torch.save(ckpt, path, pickle_module=_RefPickleModule, pickle_protocol=2)
    return path
