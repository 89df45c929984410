"""Host-side operator layer: tensors -> C-ABI calls (libdedark_yolo.so).  PyTorch is plumbing only (device memory, streams).

Tensor convention between modules: logical shape [B, C, H, W] (the reference's module interface), physical layout NHWC
(`channels_last`), dtype = compute dtype (fp32 parity path or bf16 throughput path).  A channel slice of a wider buffer is a
legal input/output ("view" = pointer + pixel stride), which is how C2f / SPPF / Detect concats cost nothing.
"""
import ctypes as C
import os

import torch

from . import _C
from ._C import ACT_LEAKY, ACT_NONE, ACT_SILU, ConvDesc, DetMaps, call

_compute_dtype = torch.float32
_pack_generation = 0        # bumped whenever _pack() allocates a new packed copy
_weights_epoch = 0          # bumped by the fused optimizer (it writes parameters through raw pointers)


def set_compute_dtype(dt):
    """fp32 = the parity path (exact-f32 MFMA), bf16 = the throughput path, fp16 = the reference's AMP dtype (BASELINE configs[4];
    the trainer adds dynamic loss scaling)."""
    global _compute_dtype
    if dt not in (torch.float32, torch.bfloat16, torch.float16):
        raise ValueError(f"dedark_yolo_amd: unsupported compute dtype {dt}")
    _compute_dtype = dt


def get_compute_dtype():
    return _compute_dtype


_emulate_storage = None     # TEST HOOK: fp32 path with every stored activation / gradient tensor rounded to this 16-bit dtype


def set_storage_emulation(dt):
    """Test hook (tests/test_gpu_lowprec.py): with the fp32 compute dtype, round what the 16-bit paths STORE -- pre-BatchNorm conv
    outputs, activations, data gradients -- to `dt` right after the kernel that produced it, while all arithmetic stays on the
    golden-pinned fp32 kernels.  That is an ideal 16-bit-storage implementation running on the same GPU: the yardstick that
    separates kernel error from the rounding error any implementation of that storage format has.  None switches it off."""
    global _emulate_storage
    if dt not in (None, torch.bfloat16, torch.float16):
        raise ValueError("storage emulation: bf16 / fp16 / None")
    _emulate_storage = dt


def emu_round(*tensors):
    if _emulate_storage is not None:
        for t in tensors:
            if t is not None and t.dtype == torch.float32:
                t.copy_(t.to(_emulate_storage))


def bump_weights_epoch():
    global _weights_epoch
    _weights_epoch += 1


def dt_id(dtype):
    if dtype == torch.float32:
        return _C.DY_F32
    if dtype == torch.bfloat16:
        return _C.DY_BF16
    if dtype == torch.float16:
        return _C.DY_F16
    raise RuntimeError(f"dedark_yolo_amd: unsupported dtype {dtype}")


def vec_elems(dtype):
    return 4 if dtype == torch.float32 else 8


def round_up(c, m):
    return (c + m - 1) // m * m


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """hipStream_t of torch's current stream (the raw accessor: torch.cuda.current_stream() costs ~8 us per call)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("dedark_yolo_amd: the HIP path needs tensors on the GPU (there is no CPU fallback)")


def empty_nhwc(B, C_, H, W, dtype, device):
    return torch.empty((B, C_, H, W), dtype=dtype, device=device, memory_format=torch.channels_last)


def zeros_nhwc(B, C_, H, W, dtype, device):
    return torch.zeros((B, H, W, C_), dtype=dtype, device=device).permute(0, 3, 1, 2)


def ld_of(t):
    """Pixel stride (elements) of an NHWC view [B,C,H,W]; raises if `t` is not such a view."""
    B, Cc, H, W = t.shape
    sb, sc, sh, sw = t.stride()              # one call: this runs ~400 times per training step
    if Cc > 1 and sc != 1:
        raise RuntimeError(f"dedark_yolo_amd: expected an NHWC (channels_last) view, got strides {t.stride()} for {tuple(t.shape)}")
    if W > 1:
        ld = sw
    elif H > 1:
        ld = sh
    elif B > 1:
        ld = sb
    else:
        ld = Cc
    if not (ld >= Cc and (W == 1 or sw == ld) and (H == 1 or sh == W * ld) and (B == 1 or sb == H * W * ld)):
        raise RuntimeError(f"dedark_yolo_amd: not a dense NHWC view: shape {tuple(t.shape)} strides {t.stride()}")
    return ld


def is_nhwc_view(t):
    try:
        ld_of(t)
        return True
    except RuntimeError:
        return False


def as_nhwc(t, dtype=None):
    """Boundary adapter: any [B,C,H,W] tensor -> NHWC view in `dtype` (torch relayout only when the caller hands NCHW)."""
    dtype = dtype or _compute_dtype
    require_gpu(t)
    if t.dtype != dtype:
        t = t.to(dtype)
    ve = vec_elems(dtype)
    if is_nhwc_view(t) and (ld_of(t) >= round_up(t.shape[1], ve)) and (ld_of(t) * t.element_size()) % 16 == 0 \
            and t.data_ptr() % 16 == 0:
        return t          # (a narrower-than-ld view is one of our own zero-padded buffers)
    Cc = t.shape[1]
    Cp = round_up(Cc, ve)
    if Cp == Cc:
        return t.contiguous(memory_format=torch.channels_last)
    buf = torch.zeros((t.shape[0], t.shape[2], t.shape[3], Cp), dtype=dtype, device=t.device)
    buf[..., :Cc] = t.permute(0, 2, 3, 1)
    return buf.permute(0, 3, 1, 2)[:, :Cc]


def padded_channels(t):
    """Channels physically readable behind view `t` when C is not a vector multiple (pad lanes must be zero)."""
    ve = vec_elems(t.dtype)
    Cp = round_up(t.shape[1], ve)
    if Cp != t.shape[1] and ld_of(t) < Cp:
        raise RuntimeError("dedark_yolo_amd: channel count needs zero padding to a 16-byte multiple")
    return Cp


# ------------------------------------------------------------------------------------------------ scratch arena
class _Arena:
    """Zero-initialised double scratch for per-channel statistics; one fill per step instead of one per conv.

    Streams: reset() runs on the compute stream at the start of a forward pass and remembers it.  A chunk that has to be added in
    the middle of a pass (the first pass, before the right size is known) may be requested from a branch stream (Detect levels,
    weight gradients): it is then allocated AND zeroed on the compute stream, every side stream is made to wait for that fill
    (dy_stream_fork), and the chunk it replaces stays alive until the next reset(), so that no kernel of another stream can see
    memory that is being filled or has gone back to the allocator."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.used = 0          # doubles handed out since the last reset
        self.chunks = 0        # buffers created since the last reset
        self.retired = []      # chunks replaced since the last reset (kept alive: other streams may still use their slices)
        self.main_raw = None   # hipStream_t of the compute stream (set by reset)

    def _new_chunk(self, cap, device):
        cur = stream()
        main_raw = self.main_raw if self.main_raw is not None else cur
        if cur == main_raw:
            buf = torch.zeros(cap, dtype=torch.float64, device=device)
        else:
            with torch.cuda.stream(torch.cuda.ExternalStream(main_raw, device=device)):
                buf = torch.zeros(cap, dtype=torch.float64, device=device)
        # every side stream that may take a slice of this chunk waits for the fill
        others = [s.cuda_stream for s in _branch["streams"]]
        if _wg_side.raw is not None:
            others.append(_wg_side.raw)
        if cur != main_raw and cur not in others:
            others.append(cur)
        for raw in others:
            call("dy_stream_fork", main_raw, raw)
        return buf

    def alloc(self, n, device):
        n = round_up(n, 2)
        self.used += n
        if self.buf is None or self.buf.device != device or self.off + n > self.buf.numel():
            cap = max(1 << 18, 4 * n)                # overflow chunk; reset() replaces the chunks by one buffer of the right size
            if self.buf is not None:
                self.retired.append(self.buf)
            self.buf = self._new_chunk(cap, device)
            self.off = 0
            self.chunks += 1
        out = self.buf[self.off:self.off + n]
        self.off += n
        return out

    def reset(self):
        self.main_raw = stream() if torch.cuda.is_available() else None
        self.retired.clear()
        if self.chunks > 1 and self.buf is not None:
            self.buf = torch.zeros(int(self.used * 1.25) + 1024, dtype=torch.float64, device=self.buf.device)
        elif self.buf is not None and self.off:
            self.buf[:self.off].zero_()
        self.off = 0
        self.used = 0
        self.chunks = 1 if self.buf is not None else 0


arena = _Arena()


# ------------------------------------------------------------------------------------------------ weights
def _pack(weight, cout_pad, cin_pad, transposed, dtype):
    key = (cout_pad, cin_pad, transposed, dtype)
    cache = weight.__dict__.setdefault("_dy_pack", {})
    hit = cache.get(key)
    tag = (_weights_epoch, weight._version)
    if hit is not None and hit[0] == tag:
        return hit[1]
    Co, Ci, KH, KW = weight.shape
    if hit is None:
        global _pack_generation
        _pack_generation += 1                  # a new packed copy exists: PackPlan must re-collect
    out = hit[1] if hit is not None else torch.empty(cout_pad * KH * KW * cin_pad, dtype=dtype, device=weight.device)
    w32 = weight.detach()
    if w32.dtype != torch.float32 or not w32.is_contiguous():
        w32 = w32.float().contiguous()
    call("dy_pack_weight", ptr(w32), ptr(out), Co, cout_pad, Ci, cin_pad, KH, KW, 1 if transposed else 0, dt_id(dtype), stream())
    cache[key] = (tag, out)
    return out


_extra_pack_views = []      # weakrefs of persistent weight VIEWS (fully connected layers run as convs) that PackPlan re-packs too


def register_pack_view(t):
    import weakref
    _extra_pack_views.append(weakref.ref(t))


class PackPlan:
    """Re-packs every cached packed weight of a model with one launch (dy_pack_weights_multi) right after the optimizer
    step wrote the f32 masters; the per-conv lazy path of _pack() then only ever hits its cache."""

    def __init__(self):
        self.sig = None
        self.table = None
        self.n_blocks = 0
        self.entries = []

    def _collect(self, model):
        ent = []
        _extra_pack_views[:] = [r for r in _extra_pack_views if r() is not None]
        for w in list(model.parameters()) + [r() for r in _extra_pack_views]:
            cache = w.__dict__.get("_dy_pack")
            if not cache or w.dtype != torch.float32 or not w.is_contiguous() or w.dim() != 4:
                continue
            for key, (tag, out) in cache.items():
                ent.append((w, key, out))
        return ent

    def repack(self, model):
        # the entry list only changes when _pack() creates a new packed copy (first use of a layout) or the parameters move:
        # walking model.parameters() and rebuilding the signature cost ~1 ms of host time per step
        key = (id(model), _pack_generation)
        if key != getattr(self, "_ent_key", None):
            self._ent = self._collect(model)
            self._ent_sig = tuple((w.data_ptr(), out.data_ptr(), k) for w, k, out in self._ent)
            self._ent_key = key
        ent = self._ent
        if not ent:
            return
        sig = self._ent_sig
        if any(w.data_ptr() != sg[0] for (w, _, _), sg in zip(ent[:4], sig[:4])):      # parameters re-bound (model.to(), new flat state)
            self._ent_key = None
            return self.repack(model)
        if sig != self.sig:
            items = (_C.PackItem * len(ent))()
            blk = 0
            lib = _C.lib()
            for it, (w, (cout_pad, cin_pad, transposed, dtype), out) in zip(items, ent):
                Co, Ci, KH, KW = w.shape
                it.w, it.packed = w.data_ptr(), out.data_ptr()
                it.Cout, it.Cout_pad, it.Cin, it.Cin_pad, it.KH, it.KW = Co, cout_pad, Ci, cin_pad, KH, KW
                it.transposed, it.dtype, it.first_block = int(transposed), dt_id(dtype), blk
                blk += lib.dy_pack_item_blocks(cout_pad, cin_pad, KH, KW)
            raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
            self.table = raw.to(ent[0][0].device)
            self.n_blocks, self.sig = blk, sig
        call("dy_pack_weights_multi", ptr(self.table), len(ent), self.n_blocks, stream())
        for w, key, out in ent:
            w.__dict__["_dy_pack"][key] = ((_weights_epoch, w._version), out)


def _padded_vec(v, n):
    """f32 per-channel vector zero-padded to n (bias for padded output channels)."""
    if v is None:
        return None
    v = v.detach()
    if v.numel() == n and v.dtype == torch.float32:
        return v
    out = torch.zeros(n, dtype=torch.float32, device=v.device)
    out[:v.numel()] = v
    return out


# ------------------------------------------------------------------------------------------------ conv + bn + act
def bn_fold(bn, cout_pad):
    """scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale as a cached [2, cout_pad] f32 tensor on the
    BatchNorm module.  The fold is recomputed only when a weight or buffer changed (optimizer step, load_state_dict, a training
    forward): `BaseModel.fuse()` fills every cache up front, afterwards an eval forward launches no fold kernel at all."""
    tag = (_weights_epoch, bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.running_mean.data_ptr(), cout_pad)
    hit = bn.__dict__.get("_dy_fold")
    if hit is not None and hit[0] == tag:
        return hit[1]
    aff = hit[1] if (hit is not None and hit[1].shape[1] == cout_pad and hit[1].device == bn.weight.device) else \
        torch.empty((2, cout_pad), dtype=torch.float32, device=bn.weight.device)
    call("dy_bn_fold_eval", ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var), float(bn.eps),
         ptr(aff[0]), ptr(aff[1]), cout_pad, stream())
    bn.__dict__["_dy_fold"] = (tag, aff)
    return aff


def bn_fold_is_current(bn):
    hit = bn.__dict__.get("_dy_fold")
    return hit is not None and hit[0][:7] == (_weights_epoch, bn.weight._version, bn.bias._version, bn.running_mean._version,
                                               bn.running_var._version, bn.weight.data_ptr(), bn.running_mean.data_ptr())


class ConvCtx:
    __slots__ = ("x", "z", "aff", "weight", "bias", "bn", "act", "k", "stride", "pad", "dil", "cout", "cout_pad", "cin_pad",
                 "has_bn", "y", "owner", "shared")


def _conv_desc(src, w, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, KH, KW, stride, pad, dil, scale, shift, act, stats, accumulate, dtype):
    # positional construction (field order of _C.ConvDesc): ~4x cheaper than 25 attribute stores, and this runs twice per conv
    if dst is not None:
        dptr, dld = dst.data_ptr(), ld_of(dst)
    else:                                   # dgrad into a planar tensor (dst_planar is set by the caller)
        dptr, dld = None, Cd
    return ConvDesc(src.data_ptr(), ld_of(src), N, Hs, Ws, Cs, w.data_ptr(), dptr, dld, Hd, Wd, Cd, KH, KW, stride, pad, dil,
                    None if scale is None else scale.data_ptr(), None if shift is None else shift.data_ptr(), act,
                    None if stats is None else stats.data_ptr(), 1 if accumulate else 0, dt_id(dtype))


_bn_pending = {}


def flush_bn_counters():
    """num_batches_tracked is bumped lazily (one tiny kernel per BN per step would be pure launch overhead)."""
    for bn, n in _bn_pending.items():
        bn.num_batches_tracked += n
    _bn_pending.clear()


def conv_forward(tape, x, weight, bias=None, bn=None, act=ACT_NONE, stride=1, pad=0, dil=1, training=False, out=None,
                 residual=None, owner=None, shared=False):
    """y = act(bn(conv(x) [+ bias])) [+ residual].  x: NHWC view; returns an NHWC view (into `out` when given).

    Training + bn: conv kernel (raw z + batch statistics) -> finalize -> fused affine/activation/residual pass.
    Otherwise one kernel with the affine (+bias / folded BN) and activation in the epilogue.
    A context is pushed on `tape` when it is not None.
    """
    dtype = x.dtype
    B, Cin, H, W = x.shape
    Cout, Cw, KH, KW = weight.shape
    if Cw != Cin:
        raise RuntimeError(f"conv: weight expects {Cw} input channels, got {Cin}")
    ve = vec_elems(dtype)
    cin_pad = padded_channels(x)
    cout_pad = round_up(Cout, ve)
    Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    wp = _pack(weight, cout_pad, cin_pad, False, dtype)
    dev = x.device
    has_bn = bn is not None
    batch_stats = has_bn and training
    ctx = None
    if tape is not None:
        ctx = ConvCtx()
        ctx.x, ctx.weight, ctx.bias, ctx.bn, ctx.act = x, weight, bias, bn, act
        ctx.owner = owner if owner is not None else weight
        ctx.shared = shared        # parameters used more than once per step (MFRU): gradients go through tape.pgrads, which adds them up
        ctx.k, ctx.stride, ctx.pad, ctx.dil = (KH, KW), stride, pad, dil
        ctx.cout, ctx.cout_pad, ctx.cin_pad, ctx.has_bn = Cout, cout_pad, cin_pad, has_bn
    if batch_stats:
        z = empty_nhwc(B, cout_pad, Ho, Wo, dtype, dev)
        stats = arena.alloc(2 * cout_pad * _C.STATS_REPLICAS, dev)
        d = _conv_desc(x, wp, z, B, H, W, cin_pad, Ho, Wo, cout_pad, KH, KW, stride, pad, dil, None, None, ACT_NONE, stats,
                       False, dtype)
        aff = torch.empty((4, cout_pad), dtype=torch.float32, device=dev)     # scale, shift, mean, invstd
        pa, sa = aff.data_ptr(), 4 * cout_pad
        pixels = B * Ho * Wo
        _bn_pending[bn] = _bn_pending.get(bn, 0) + 1
        bn.__dict__.pop("_dy_fold", None)          # the kernels below rewrite the running statistics through raw pointers
        y = out if out is not None else empty_nhwc(B, cout_pad, Ho, Wo, dtype, dev)
        st = stream()
        rp, rld = (residual.data_ptr(), ld_of(residual)) if residual is not None else (None, 0)
        bp = bn._parameters
        if _C._prof is None and _emulate_storage is None:
            # conv (raw z + statistics) -> finalize -> affine/activation/residual: three launches, ONE foreign call
            call("dy_conv2d_bn_act_fwd", C.byref(d), pixels, ptr(bp["weight"]), ptr(bp["bias"]), ptr(bn.running_mean), ptr(bn.running_var),
                 float(bn.momentum), float(bn.eps), pa, act, rp, rld, y.data_ptr(), ld_of(y), st)
        else:                               # per-entry timing (bench.py roofline leg, tools/layer_profile.py)
            _C.set_meta(kind="conv_fwd", shape=f"{Cin}->{Cout} k{KH} s{stride} in {B}x{H}x{W}", dtype=str(dtype), flops=2.0 * pixels * Cout * KH * KW * Cin,
                        bytes=float((B * H * W * Cin + pixels * Cout + Cout * KH * KW * Cin) * x.element_size()))
            call("dy_conv2d_fwd", C.byref(d), st)
            emu_round(z)                    # (the statistics come from the f32 accumulators in every dtype)
            call("dy_bn_finalize", ptr(stats), pixels, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
                 ptr(bn.running_var), float(bn.momentum), float(bn.eps), pa, pa + sa, pa + 2 * sa, pa + 3 * sa, cout_pad, st)
            _C.set_meta(kind="bn_act_fwd", shape=f"{cout_pad}ch {B}x{Ho}x{Wo}", dtype=str(dtype), flops=0.0,
                        bytes=float(pixels * cout_pad * x.element_size() * (3 if residual is not None else 2)))
            call("dy_bn_act_fwd", ptr(z), ld_of(z), pa, pa + sa, act, rp, rld, ptr(y), ld_of(y), pixels, cout_pad, dt_id(dtype), st)
            emu_round(y)
        if ctx is not None:
            ctx.z, ctx.aff, ctx.y = z, aff, None
    else:
        if has_bn:                      # eval: running statistics folded into the conv epilogue (fuse_conv_and_bn, torch_utils.py:123-144)
            aff = bn_fold(bn, cout_pad)
            scale, shift = aff[0], aff[1]
        else:
            scale, shift = None, _padded_vec(bias, cout_pad)
        direct = residual is None
        y = out if (out is not None and direct) else empty_nhwc(B, cout_pad, Ho, Wo, dtype, dev)
        d = _conv_desc(x, wp, y, B, H, W, cin_pad, Ho, Wo, cout_pad, KH, KW, stride, pad, dil, scale, shift, act, None, False,
                       dtype)
        _C._prof is not None and _C.set_meta(kind="conv_fwd", shape=f"{Cin}->{Cout} k{KH} s{stride} in {B}x{H}x{W}", dtype=str(dtype), flops=2.0 * B * Ho * Wo * Cout * KH * KW * Cin,
                    bytes=float((B * H * W * Cin + B * Ho * Wo * Cout + Cout * KH * KW * Cin) * x.element_size()))
        call("dy_conv2d_fwd", C.byref(d), stream())
        if not direct:                  # eval-time residual: y_out = y + residual
            tgt = out if out is not None else y
            if tgt is not y:
                call("dy_copy2d", ptr(y), ld_of(y), ptr(tgt), ld_of(tgt), B * Ho * Wo, cout_pad, 0, dt_id(dtype), stream())
            call("dy_copy2d", ptr(residual), ld_of(residual), ptr(tgt), ld_of(tgt), B * Ho * Wo, cout_pad, 1, dt_id(dtype),
                 stream())
            y = tgt
        emu_round(y)
        if ctx is not None:
            if has_bn:
                raise RuntimeError("conv: gradients through an eval-mode BatchNorm are not supported")
            ctx.z, ctx.aff, ctx.y = None, None, y
    if tape is not None:
        tape.push(ctx)
    return y if cout_pad == Cout else y[:, :Cout]


def _add_pgrad(tape, p, g):
    if p is None or not p.requires_grad:
        return
    if p in tape.pgrads:
        tape.pgrads[p] = tape.pgrads[p] + g
    else:
        tape.pgrads[p] = g


_wg_scratch = {}


def wgrad_scratch(device, elems=32 << 20, tag=0):
    """Reusable f32 workspace for the split-pixel partial tiles of dy_conv2d_wgrad (128 MiB; stream-ordered reuse, one per
    launch stream: `tag` = the raw stream handle.  A shared workspace was a race once the Detect levels ran on branch streams
    with gradients handed back to autograd -- their weight gradients then run on three streams at once)."""
    t = _wg_scratch.get((device, tag))
    if t is None or t.numel() < elems:
        t = torch.empty(elems, dtype=torch.float32, device=device)
        _wg_scratch[(device, tag)] = t
    return t


# ---- weight gradients on a second HIP stream.  dgrad and wgrad of a conv both depend only on dz; the next layer's backward
# depends only on the dgrad.  At the batch sizes of BASELINE configs[1] most kernels are launch-/latency-bound (20-50 us, a few
# hundred blocks), so the wgrad + its split reduction are issued on a side stream where they fill the gaps of the
# dz -> dgrad -> BN-backward chain.  Only used with direct gradient placement (the trainer's flat gradient buffer): gradients
# handed back to autograd are accumulated on the compute stream and stay there.  The operands are kept alive until the join.
class _WgradSide:
    GROUP = 8                      # operands are released in groups of this many weight gradients (one event per group)

    def __init__(self):
        self.on = False
        self.stream = None
        self.raw = None            # hipStream_t of the side stream
        self.cur = []              # operands of the group being filled
        self.groups = []           # [(event recorded on the side stream after the group's last wgrad, operands)], issue order
        self.free_events = []
        self.cb_queued = False


_wg_side = _WgradSide()


def enable_wgrad_stream(on=True):
    """Trainer switch (DY_WGRAD_STREAM=0 keeps everything on the compute stream)."""
    _wg_side.on = bool(on) and os.environ.get("DY_WGRAD_STREAM", "1") != "0"


def wgrad_stream_enabled():
    return bool(_wg_side.on)


def wgrad_side_stream(device=None):
    s = _wg_side
    if not s.on:
        return None
    if s.stream is None:
        prio = int(os.environ.get("DY_WGRAD_PRIO", "0"))       # experiments: 1 = below the compute stream where HIP offers it
        s.stream = torch.cuda.Stream(device=device, priority=prio)
        s.raw = s.stream.cuda_stream
    return s.stream


def _side_wait_main(side=None):
    """side stream waits for everything issued so far on the current (compute) stream."""
    call("dy_stream_fork", stream(), _wg_side.raw)


def _wg_track(x, dz):
    """Keeps the operands of a side-stream wgrad alive; finished groups (oldest first) are released so that their memory
    returns to the allocator during the backward pass instead of at the join."""
    s = _wg_side
    s.cur.append((x, dz))
    if len(s.cur) < s.GROUP or torch.cuda.is_current_stream_capturing():
        return
    ev = s.free_events.pop() if s.free_events else torch.cuda.Event()
    ev.record(s.stream)
    s.groups.append((ev, s.cur))
    s.cur = []
    while s.groups and s.groups[0][0].query():
        s.free_events.append(s.groups.pop(0)[0])


def wgrad_pending():
    return bool(_wg_side.cur or _wg_side.groups)


def wgrad_join():
    """The compute stream waits for the weight gradients issued so far; their operands may be freed afterwards."""
    s = _wg_side
    if s.cur or s.groups:
        call("dy_stream_fork", s.raw, stream())
        s.free_events.extend(g[0] for g in s.groups)
        s.groups.clear()
        s.cur = []
    s.cb_queued = False


# ---- independent branches on their own streams (the three pyramid levels of the Detect head, forward and backward)
_branch = {"on": False, "streams": [], "by_device": {}}      # "streams": every side stream ever made (arena fills fork to all)


def enable_branch_streams(on=True):
    """Trainer switch (DY_BRANCH_STREAMS=0 keeps the branches on the compute stream)."""
    _branch["on"] = bool(on) and os.environ.get("DY_BRANCH_STREAMS", "1") != "0"


def branch_streams(n, device):
    """n side streams for independent branches, or None when the switch is off."""
    if not _branch["on"] or n <= 0:
        return None
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    mine = _branch["by_device"].setdefault(key, [])            # a stream belongs to ONE device: one list per ordinal
    while len(mine) < n:
        s = torch.cuda.Stream(device=device)
        mine.append(s)
        _branch["streams"].append(s)
    return mine[:n]


def _grad_dst(p):
    """p.grad storage when the trainer enabled direct gradient placement (kernels write there; autograd gets None)."""
    if p is not None and p.requires_grad and getattr(p, "_dy_direct", False) and p.grad is not None:
        return p.grad
    return None


def conv_backward(tape, dy, need_dx=True, dx_out=None, accumulate=False, add_src=None):
    """Backward of the matching conv_forward (pops its context). dy: NHWC view of the gradient wrt the conv's output
    (for a residual conv the caller routes dy to the residual branch itself). Returns dx (NHWC view) or None.
    add_src: optional NHWC view of dx's shape added to the data gradient in the same call (dx = [dx_out +] dgrad + add_src): the
    shortcut gradient of a Bottleneck."""
    ctx = tape.pop()
    x = ctx.x
    dtype = x.dtype
    dev = x.device
    B, Cin, H, W = x.shape
    KH, KW = ctx.k
    Cout, cout_pad, cin_pad = ctx.cout, ctx.cout_pad, ctx.cin_pad
    Ho, Wo = dy.shape[2], dy.shape[3]
    if cout_pad != Cout:
        if padded_channels(dy) != cout_pad:
            raise RuntimeError("conv_backward: gradient view lacks zero padding")
    pixels = B * Ho * Wo
    did = dt_id(dtype)
    st = stream()
    _grad_dst = (lambda p: None) if getattr(ctx, "shared", False) else globals()["_grad_dst"]
    if ctx.has_bn:
        aff, z, bn = ctx.aff, ctx.z, ctx.bn
        sums = arena.alloc(2 * cout_pad * _C.BN_BWD_REPLICAS, dev)
        pa, sa = aff.data_ptr(), 4 * cout_pad                               # rows of aff: scale, shift, mean, invstd
        dz = empty_nhwc(B, cout_pad, Ho, Wo, dtype, dev)
        gw_, gb_ = _grad_dst(bn.weight), _grad_dst(bn.bias)
        direct = gw_ is not None and gb_ is not None
        if not direct:
            dgb = torch.empty((2, cout_pad), dtype=torch.float32, device=dev)
            gw_, gb_ = dgb[0], dgb[1]
        if _C._prof is None or _emulate_storage is not None:
            call("dy_bn_act_bwd", dy.data_ptr(), ld_of(dy), z.data_ptr(), ld_of(z), pa, ptr(bn._parameters["weight"]), ctx.act,
                 sums.data_ptr(), dz.data_ptr(), ld_of(dz), gw_.data_ptr(), gb_.data_ptr(), pixels, cout_pad, did, st)
        else:
            _C.set_meta(kind="bn_act_bwd_reduce", shape=f"{cout_pad}ch {pixels}px", dtype=str(dtype), flops=0.0, bytes=float(pixels * cout_pad * x.element_size() * 2))
            call("dy_bn_act_bwd_reduce", ptr(dy), ld_of(dy), ptr(z), ld_of(z), pa, pa + sa, pa + 2 * sa, pa + 3 * sa,
                 ctx.act, 1, ptr(sums), pixels, cout_pad, did, st)
            _C.set_meta(kind="bn_act_bwd_apply", shape=f"{cout_pad}ch {pixels}px", dtype=str(dtype), flops=0.0, bytes=float(pixels * cout_pad * x.element_size() * 3))
            call("dy_bn_act_bwd_apply", ptr(dy), ld_of(dy), ptr(z), ld_of(z), pa, pa + sa, pa + 2 * sa, pa + 3 * sa,
                 ptr(bn.weight), ctx.act, 1, ptr(sums), ptr(dz), ld_of(dz), ptr(gw_), ptr(gb_), pixels, cout_pad, did, st)
        emu_round(dz)
        if not direct:
            _add_pgrad(tape, bn.weight, gw_)
            _add_pgrad(tape, bn.bias, gb_)
    else:
        y = ctx.y
        need_bias = ctx.bias is not None and ctx.bias.requires_grad
        if ctx.act == ACT_NONE and not need_bias:
            dz = dy
        else:
            sums = arena.alloc(2 * cout_pad * _C.BN_BWD_REPLICAS, dev)
            call("dy_bn_act_bwd_reduce", ptr(dy), ld_of(dy), ptr(y), ld_of(y), None, None, None, None, ctx.act, 0, ptr(sums),
                 pixels, cout_pad, did, st)
            gb_ = _grad_dst(ctx.bias) if (need_bias and cout_pad == Cout) else None
            db = gb_ if gb_ is not None else torch.empty(cout_pad, dtype=torch.float32, device=dev)
            if ctx.act == ACT_NONE:
                dz = dy
                call("dy_bn_act_bwd_apply", ptr(dy), ld_of(dy), ptr(y), ld_of(y), None, None, None, None, None, ctx.act, 0,
                     ptr(sums), ptr(dy), ld_of(dy), None, ptr(db), 0, cout_pad, did, st)      # 0 pixels: only dbias
            else:
                dz = empty_nhwc(B, cout_pad, Ho, Wo, dtype, dev)
                call("dy_bn_act_bwd_apply", ptr(dy), ld_of(dy), ptr(y), ld_of(y), None, None, None, None, None, ctx.act, 0,
                     ptr(sums), ptr(dz), ld_of(dz), None, ptr(db), pixels, cout_pad, did, st)
                emu_round(dz)
            if need_bias and gb_ is None:
                gd = _grad_dst(ctx.bias)
                if gd is not None:
                    gd.copy_(db[:Cout])
                else:
                    _add_pgrad(tape, ctx.bias, db[:Cout])
    # weight gradient
    if ctx.owner.requires_grad:
        gd = _grad_dst(ctx.owner)
        gw = gd if gd is not None else torch.empty(ctx.weight.shape, dtype=torch.float32, device=dev)
        side = wgrad_side_stream(dev) if gd is not None else None
        st_w = st
        if side is not None:
            if _C._prof is not None:
                _side_wait_main()                      # dz (and, for a first step, x) are ready
            if not _wg_side.cb_queued:                 # join at the end of this backward pass even without a trainer
                try:
                    torch.autograd.Variable._execution_engine.queue_callback(wgrad_join)
                    _wg_side.cb_queued = True
                except RuntimeError:
                    pass
            st_w = side.cuda_stream
        scratch = wgrad_scratch(dev, tag=st_w)         # one workspace per launch stream: Detect's levels may run their wgrads side by side
        _C._prof is not None and _C.set_meta(kind="conv_wgrad", shape=f"{Cin}->{Cout} k{KH} s{ctx.stride} in {B}x{H}x{W}", dtype=str(dtype), flops=2.0 * pixels * Cout * KH * KW * Cin,
                    bytes=float((B * H * W * Cin + pixels * Cout) * x.element_size() + Cout * KH * KW * Cin * 4))
        wargs = (ptr(x), ld_of(x), B, H, W, cin_pad, ptr(dz), ld_of(dz), Ho, Wo, cout_pad, KH, KW, ctx.stride, ctx.pad, ctx.dil, Cout, Cin,
                 ptr(scratch), scratch.numel(), ptr(gw), did)
        if side is not None and _C._prof is not None:
            with torch.cuda.stream(side):              # per-call timing: the events must sit on the launch stream
                call("dy_conv2d_wgrad", *wargs, st_w)
        elif side is not None:
            call("dy_conv2d_wgrad_forked", st, *wargs, st_w)        # fork from the compute stream + wgrad: one foreign call
        else:
            call("dy_conv2d_wgrad", *wargs, st_w)
        if side is not None:
            _wg_track(x, dz)
        if gd is None:
            _add_pgrad(tape, ctx.owner, gw.view(ctx.owner.shape))
    if not need_dx:
        return None
    wt = _pack(ctx.weight, cout_pad, cin_pad, True, dtype)
    # network stem (3 -> c, 3x3 stride 2): the direct kernel writes dx planar [B,Cin,H,W], the layout the front-end's
    # backward consumes (6 B/pixel instead of a 16 B NHWC8 vector that is 5/8 padding)
    planar = (dx_out is None and add_src is None and dtype in (torch.bfloat16, torch.float16) and cin_pad == 8 and Cin <= 4 and (KH, KW) == (3, 3) and ctx.stride == 2
              and ctx.pad == 1 and ctx.dil == 1 and cout_pad in (16, 32, 64) and os.environ.get("DY_NO_CONV_SMALL") is None)
    if planar:
        dxp = torch.empty((B, Cin, H, W), dtype=dtype, device=dev)
        d = _conv_desc(dz, wt, None, B, Ho, Wo, cout_pad, H, W, cin_pad, KH, KW, ctx.stride, ctx.pad, ctx.dil, None, None, ACT_NONE,
                       None, False, dtype)
        d.dst_valid_channels = Cin
        d.dst_planar = dxp.data_ptr()
        _C._prof is not None and _C.set_meta(kind="conv_dgrad", shape=f"{Cin}->{Cout} k{KH} s{ctx.stride} in {B}x{H}x{W} (planar)", dtype=str(dtype),
                    flops=2.0 * pixels * Cout * KH * KW * Cin, bytes=float((B * H * W * Cin + pixels * Cout) * x.element_size()))
        call("dy_conv2d_dgrad", C.byref(d), st)
        return dxp
    if dx_out is None:
        dxb = empty_nhwc(B, cin_pad, H, W, dtype, dev)
        accumulate = False
    else:
        dxb = dx_out
        if padded_channels(dxb) != cin_pad:
            raise RuntimeError("conv_backward: dx_out view too narrow")
    d = _conv_desc(dz, wt, dxb, B, Ho, Wo, cout_pad, H, W, cin_pad, KH, KW, ctx.stride, ctx.pad, ctx.dil, None, None, ACT_NONE,
                   None, accumulate, dtype)
    d.dst_valid_channels = Cin
    if add_src is not None:
        if tuple(add_src.shape) != (B, Cin, H, W) or add_src.dtype != dtype or padded_channels(add_src) != cin_pad:
            raise RuntimeError("conv_backward: add_src must be an NHWC view of dx's shape and dtype")
        d.add_src, d.add_src_ld = add_src.data_ptr(), ld_of(add_src)
    _C._prof is not None and _C.set_meta(kind="conv_dgrad", shape=f"{Cin}->{Cout} k{KH} s{ctx.stride} in {B}x{H}x{W}", dtype=str(dtype), flops=2.0 * pixels * Cout * KH * KW * Cin,
                bytes=float((B * H * W * Cin * (2 if accumulate else 1) + pixels * Cout + Cout * KH * KW * Cin) * x.element_size()))
    call("dy_conv2d_dgrad", C.byref(d), st)
    emu_round(dxb)
    if dx_out is not None:
        return dx_out
    return dxb if cin_pad == Cin else dxb[:, :Cin]


# ------------------------------------------------------------------------------------------------ small ops
def copy2d(src, dst, accumulate=False):
    B, Cc, H, W = src.shape
    Cp = padded_channels(src)
    call("dy_copy2d", ptr(src), ld_of(src), ptr(dst), ld_of(dst), B * H * W, Cp, 1 if accumulate else 0, dt_id(src.dtype), stream())


def maxpool_fwd(x, k, stride, pad, out=None, want_arg=True):
    B, Cc, H, W = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = out if out is not None else empty_nhwc(B, Cc, Ho, Wo, x.dtype, x.device)
    arg = torch.empty((B, Ho, Wo, Cc), dtype=torch.uint8, device=x.device) if want_arg else None
    call("dy_maxpool_fwd", ptr(x), ld_of(x), ptr(y), ld_of(y), ptr(arg), B, H, W, Cc, k, stride, pad, Ho, Wo, dt_id(x.dtype),
         stream())
    return y, arg


def maxpool_bwd(dy, arg, in_shape, k, stride, pad, dx_out=None, accumulate=False):
    B, Cc, H, W = in_shape
    Ho, Wo = dy.shape[2], dy.shape[3]
    dx = dx_out if dx_out is not None else empty_nhwc(B, Cc, H, W, dy.dtype, dy.device)
    call("dy_maxpool_bwd", ptr(dy), ld_of(dy), ptr(arg), ptr(dx), ld_of(dx), B, H, W, Cc, k, stride, pad, Ho, Wo,
         1 if (accumulate and dx_out is not None) else 0, dt_id(dy.dtype), stream())
    emu_round(dx)
    return dx


def upsample_fwd(x, scale, out=None):
    B, Cc, H, W = x.shape
    y = out if out is not None else empty_nhwc(B, Cc, H * scale, W * scale, x.dtype, x.device)
    call("dy_upsample_nearest_fwd", ptr(x), ld_of(x), ptr(y), ld_of(y), B, H, W, Cc, scale, dt_id(x.dtype), stream())
    return y


def upsample_bwd(dy, scale, dx_out=None, accumulate=False):
    B, Cc, Ho, Wo = dy.shape
    H, W = Ho // scale, Wo // scale
    dx = dx_out if dx_out is not None else empty_nhwc(B, Cc, H, W, dy.dtype, dy.device)
    call("dy_upsample_nearest_bwd", ptr(dy), ld_of(dy), ptr(dx), ld_of(dx), B, H, W, Cc, scale,
         1 if (accumulate and dx_out is not None) else 0, dt_id(dy.dtype), stream())
    emu_round(dx)
    return dx


def det_maps(maps, strides, nc):
    """Build the dy_det_maps descriptor for 1-3 NHWC Detect maps [B, 64+nc, h, w]."""
    m = DetMaps()
    m.B, m.nc, m.n_levels = maps[0].shape[0], nc, len(maps)
    m.dtype = dt_id(maps[0].dtype)
    for i, t in enumerate(maps):
        m.map[i] = t.data_ptr()
        m.map_ld[i] = ld_of(t)
        m.h[i], m.w[i] = t.shape[2], t.shape[3]
        m.stride[i] = float(strides[i])
    return m
