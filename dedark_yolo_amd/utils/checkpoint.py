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
                           best_fitness=ck.get("best_fitness"), updates=ck.get("updates"), optimizer=ck.get("optimizer"),
                           source="reference-pickle" if rec is not None else "state-dict")


def intersect_dicts(da, db, exclude=()):
    """reference torch_utils.py:303-305: keys of `da` that exist in `db` with the same shape (minus excluded substrings)."""
    return {k: v for k, v in da.items() if k in db and all(x not in k for x in exclude) and v.shape == db[k].shape}
