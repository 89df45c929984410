"""Module registry of the MI355X-native Dedark-YOLO path.

Same class names, constructor arguments, call signatures and state_dict key names as the reference modules (cited per
class), so `parse_model` / checkpoints / user scripts are drop-in.  Every module runs hand-written HIP kernels through
`ops` (C-ABI); nn.Conv2d / nn.BatchNorm2d / nn.Linear instances are used ONLY as parameter containers (their forward is
never called).  Each module implements `_fwd(tape, ...)` / `_bwd(tape, ...)`; a top-level call wraps the pair in one
autograd.Function so torch.autograd links modules while everything inside a module is explicit.
"""
import ctypes as C
import math
import os

import torch
import torch.nn as nn

from .. import ops
from .._C import ACT_LEAKY, ACT_NONE, ACT_SILU, call
from ..ops import as_nhwc, conv_backward, conv_forward, copy2d, empty_nhwc, ld_of, ptr, stream

__all__ = ("Conv", "Concat", "Bottleneck", "C2f", "SPPF", "Upsample", "AsffTribeLevel", "AsffDoubLevel", "MFRU", "SCConv", "RFBblock", "DFL", "Detect",
           "AsffDetect",
           "lowlight_recovery", "ExtractParameters2", "autopad")


def autopad(k, p=None, d=1):
    """Pad to 'same' (reference ultralytics/nn/modules/conv.py:15-21)."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


class Tape:
    """LIFO of per-op contexts recorded by a module's _fwd and consumed in reverse by its _bwd."""

    def __init__(self):
        self.stack = []
        self.pgrads = {}

    def push(self, c):
        self.stack.append(c)

    def pop(self):
        return self.stack.pop()


class _ModuleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, n_in, out, *args):
        tape = Tape()
        outs = module._fwd(tape, *args[:n_in]) if out is None else module._fwd(tape, *args[:n_in], out=out)
        ctx.module, ctx.tape, ctx.n_in = module, tape, n_in
        ctx.multi = isinstance(outs, (list, tuple))
        if ctx.multi:
            ctx.out_meta = [(o.shape, o.dtype, o.device) for o in outs]
            return tuple(outs)
        ctx.out_meta = [(outs.shape, outs.dtype, outs.device)]
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        gouts = [g if g is not None else torch.zeros(m[0], dtype=m[1], device=m[2]) for g, m in zip(gouts, ctx.out_meta)]
        module, tape = ctx.module, ctx.tape
        if getattr(module, "_planar_grad_ok", False) and all(g.is_contiguous() and g.dtype == m[1] for g, m in zip(gouts, ctx.out_meta)):
            pass             # the module's backward consumes planar [B,C,H,W] gradients as they are (front-end <- direct stem dgrad)
        else:
            gouts = [as_nhwc(g, m[1]) for g, m in zip(gouts, ctx.out_meta)]
        gins = module._bwd(tape, *gouts, needs=list(ctx.needs_input_grad[3:3 + ctx.n_in]))
        if not isinstance(gins, (list, tuple)):
            gins = (gins,)
        assert not tape.stack, f"{type(module).__name__}: unbalanced tape"
        cb = module.__dict__.get("_dy_after_backward")
        if cb is not None:
            cb()                 # data-parallel trainer: this layer's gradients are enqueued -> maybe launch its bucket
        pg = [tape.pgrads.get(p) for p in module._plist]
        return (None, None, None, *gins, *pg)


class DyModule(nn.Module):
    """Base: dispatches a call either through autograd (_ModuleFn) or straight to _fwd (no_grad / eval)."""
    in_dtype = None      # None -> compute dtype

    def _params(self):
        pl = self.__dict__.get("_plist")
        if pl is None:
            pl = [p for p in self.parameters() if p.requires_grad]
            self.__dict__["_plist"] = pl
        return pl

    def forward(self, x, *extra, out=None):
        """`out`: optional NHWC view the module's result is written into (GraphPlan: a producer's slice of a yaml-level Concat
        buffer); only modules whose _fwd takes `out` are called with it."""
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        xs = [self._adapt(t) for t in xs]
        pl = self._params()
        if torch.is_grad_enabled() and (any(t.requires_grad for t in xs) or any(p.requires_grad for p in pl)):
            return self._wrap(_ModuleFn.apply(self, len(xs), out, *xs, *pl))
        with torch.no_grad():
            return self._wrap(self._fwd(None, *xs) if out is None else self._fwd(None, *xs, out=out))

    def _adapt(self, t):
        return as_nhwc(t, self.in_dtype)

    def _wrap(self, out):
        return list(out) if isinstance(out, tuple) else out

    def train(self, mode=True):
        self.__dict__.pop("_plist", None)
        return super().train(mode)


def _act_code(act):
    if act is True or isinstance(act, nn.SiLU):
        return ACT_SILU
    if isinstance(act, nn.LeakyReLU):
        if abs(act.negative_slope - 0.1) > 1e-12:
            raise NotImplementedError("only LeakyReLU(0.1) is implemented in the HIP epilogue")
        return ACT_LEAKY
    if act is False or act is None or isinstance(act, nn.Identity):
        return ACT_NONE
    raise NotImplementedError(f"activation {act!r} has no HIP epilogue")


class Conv(DyModule):
    """Conv2d(bias=False) + BatchNorm2d + SiLU (reference ultralytics/nn/modules/conv.py:38-55)."""
    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if g != 1:
            raise NotImplementedError("grouped convolution is outside the Dedark-YOLO hot path")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()
        self._act = _act_code(self.act)

    def _fwd(self, tape, x, out=None, residual=None):
        c = self.conv
        return conv_forward(tape, x, c.weight, None, self.bn, self._act, c.stride[0], c.padding[0], c.dilation[0],
                            self.training, out=out, residual=residual)

    _takes_into = True       # GraphPlan.backward_train: `into[k]` = gradient another consumer of input k already left; add into it

    def _bwd(self, tape, dy, needs=(True,), dx_out=None, accumulate=False, add_src=None, into=None):
        if into is not None and into[0] is not None:
            conv_backward(tape, dy, need_dx=True, dx_out=into[0], accumulate=True)
            return None
        return conv_backward(tape, dy, need_dx=needs[0], dx_out=dx_out, accumulate=accumulate, add_src=add_src)


class AddConv(nn.Module):
    """Parameter container for add_conv (reference ultralytics/nn/modules/block.py:24-45): conv + batch_norm + LeakyReLU(0.1)."""

    def __init__(self, in_ch, out_ch, ksize, stride):
        super().__init__()
        self.conv = nn.Conv2d(in_ch, out_ch, ksize, stride, (ksize - 1) // 2, bias=False)
        self.batch_norm = nn.BatchNorm2d(out_ch)
        self.leaky = nn.LeakyReLU(0.1)

    def _fwd(self, tape, x, training, out=None):
        c = self.conv
        return conv_forward(tape, x, c.weight, None, self.batch_norm, ACT_LEAKY, c.stride[0], c.padding[0], 1, training, out=out)


def plain_conv_fwd(tape, m, x, act=ACT_NONE, out=None):
    """nn.Conv2d-with-bias container `m` run as a HIP conv (Detect 1x1 heads, ASFF weight_levels, RFB, extractor)."""
    return conv_forward(tape, x, m.weight, m.bias, None, act, m.stride[0], m.padding[0], m.dilation[0], False, out=out)


class Concat(DyModule):
    """torch.cat along channels (reference conv.py:462-473) as strided slab copies into one NHWC buffer."""

    def __init__(self, dimension=1):
        super().__init__()
        if dimension != 1:
            raise NotImplementedError("Concat: only channel concat is on the hot path")
        self.d = dimension

    @staticmethod
    def _in_place(xs, tot):
        """The inputs already ARE the channel slices, in order, of one [B, tot, H, W] NHWC buffer (GraphPlan placed their
        producers there): return that buffer as a view, else None."""
        t0 = xs[0]
        if ld_of(t0) != tot or t0.dim() != 4:
            return None
        es, base, o = t0.element_size(), t0.data_ptr(), 0
        for t in xs:
            if (t.dtype != t0.dtype or tuple(t.shape[2:]) != tuple(t0.shape[2:]) or t.shape[0] != t0.shape[0] or ld_of(t) != tot
                    or t.stride() != t0.stride() or t.data_ptr() != base + o * es
                    or t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr()):
                return None
            o += t.shape[1]
        return torch.as_strided(t0, (t0.shape[0], tot, t0.shape[2], t0.shape[3]), t0.stride(), t0.storage_offset())

    def _fwd(self, tape, *xs):
        B, _, H, W = xs[0].shape
        tot = sum(t.shape[1] for t in xs)
        whole = self._in_place(xs, tot)
        if whole is not None:
            if tape is not None:
                tape.push([t.shape[1] for t in xs])
            return whole
        out = empty_nhwc(B, tot, H, W, xs[0].dtype, xs[0].device)
        o = 0
        for t in xs:
            copy2d(t, out[:, o:o + t.shape[1]])
            o += t.shape[1]
        if tape is not None:
            tape.push([t.shape[1] for t in xs])
        return out

    def _bwd(self, tape, dy, needs=None):
        sizes = tape.pop()
        outs, o = [], 0
        for c in sizes:
            outs.append(dy[:, o:o + c])
            o += c
        return outs


class Upsample(DyModule):
    """nn.Upsample(size=None, scale_factor, 'nearest') of the yaml head (reference yolov8.yaml:32,36)."""

    def __init__(self, size=None, scale_factor=None, mode="nearest"):
        super().__init__()
        if size is not None or mode != "nearest" or int(scale_factor) != scale_factor:
            raise NotImplementedError("Upsample: integer nearest-neighbour scale only")
        self.scale_factor = int(scale_factor)
        self.mode = mode

    def _fwd(self, tape, x, out=None):
        return ops.upsample_fwd(x, self.scale_factor, out=out)

    _takes_into = True

    def _bwd(self, tape, dy, needs=None, into=None):
        if into is not None and into[0] is not None:
            ops.upsample_bwd(dy, self.scale_factor, dx_out=into[0], accumulate=True)
            return None
        return ops.upsample_bwd(dy, self.scale_factor)


class Bottleneck(DyModule):
    """x + cv2(cv1(x)) (reference ultralytics/nn/modules/block.py:553-565)."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def _fwd(self, tape, x, out=None):
        t = self.cv1._fwd(tape, x)
        return self.cv2._fwd(tape, t, out=out, residual=x if self.add else None)

    def _bwd(self, tape, dy, needs=(True,), dx_out=None, accumulate=False):
        dt = self.cv2._bwd(tape, dy)
        # shortcut: dx = d cv1 + dy, added inside cv1's data gradient (dy_conv_desc.add_src) instead of by a copy pass
        return self.cv1._bwd(tape, dt, dx_out=dx_out, accumulate=accumulate, add_src=dy if self.add else None)


class C2f(DyModule):
    """CSP block (reference block.py:373-393).  chunk / cat are free: cv1 and every Bottleneck write straight into
    channel slices of ONE NHWC buffer that cv2 then reads; backward mirrors it with one gradient buffer."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    def _fwd(self, tape, x, out=None):
        c, n = self.c, len(self.m)
        B, _, H, W = x.shape
        Y = empty_nhwc(B, (2 + n) * c, H, W, x.dtype, x.device)
        self.cv1._fwd(tape, x, out=Y[:, :2 * c])
        for i, m in enumerate(self.m):
            m._fwd(tape, Y[:, (1 + i) * c:(2 + i) * c], out=Y[:, (2 + i) * c:(3 + i) * c])
        return self.cv2._fwd(tape, Y, out=out)

    def _bwd(self, tape, dy, needs=(True,)):
        c, n = self.c, len(self.m)
        dY = self.cv2._bwd(tape, dy)
        for i in reversed(range(n)):
            self.m[i]._bwd(tape, dY[:, (2 + i) * c:(3 + i) * c], dx_out=dY[:, (1 + i) * c:(2 + i) * c], accumulate=True)
        return self.cv1._bwd(tape, dY[:, :2 * c], needs=needs)


class SPPF(DyModule):
    """cv1 -> 3 chained MaxPool2d(5,1,2) -> cat -> cv2 (reference block.py:323-338); pools write into slices."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.k = k
        self.c_ = c_

    def _fwd(self, tape, x, out=None):
        c_, k = self.c_, self.k
        B, _, H, W = x.shape
        Y = empty_nhwc(B, 4 * c_, H, W, x.dtype, x.device)
        self.cv1._fwd(tape, x, out=Y[:, :c_])
        args = []
        for i in range(3):
            _, a = ops.maxpool_fwd(Y[:, i * c_:(i + 1) * c_], k, 1, k // 2, out=Y[:, (i + 1) * c_:(i + 2) * c_],
                                   want_arg=tape is not None)
            args.append(a)
        if tape is not None:
            tape.push(args)
        return self.cv2._fwd(tape, Y, out=out)

    def _bwd(self, tape, dy, needs=(True,)):
        c_, k = self.c_, self.k
        dY = self.cv2._bwd(tape, dy)
        args = tape.pop()
        B, _, H, W = dY.shape
        for i in reversed(range(3)):
            ops.maxpool_bwd(dY[:, (i + 1) * c_:(i + 2) * c_], args[i], (B, c_, H, W), k, 1, k // 2,
                            dx_out=dY[:, i * c_:(i + 1) * c_], accumulate=True)
        return self.cv1._bwd(tape, dY[:, :c_], needs=needs)


class AsffTribeLevel(DyModule):
    """3-level adaptive spatial feature fusion (reference block.py:48-115). Inputs (P5, P4, P3)."""

    def __init__(self, level):
        super().__init__()
        self.level = level
        self.dim = [512, 512, 256]
        self.inter_dim = self.dim[level]
        if level == 0:
            self.stride_level_1 = nn.MaxPool2d(kernel_size=2, stride=2)
            self.stride_level_2 = AddConv(256, self.inter_dim, 3, 2)
            self.expand = AddConv(self.inter_dim, 512, 3, 1)
        elif level == 1:
            self.stride_level_2 = AddConv(256, self.inter_dim, 3, 2)
            self.expand = AddConv(self.inter_dim, 512, 3, 1)
        elif level == 2:
            self.compress_level_0 = AddConv(512, self.inter_dim, 1, 1)
            self.compress_level_1 = AddConv(512, self.inter_dim, 1, 1)
            self.expand = AddConv(self.inter_dim, 256, 3, 1)
        compress_c = 8
        self.weight_level_0 = AddConv(self.inter_dim, compress_c, 1, 1)
        self.weight_level_1 = AddConv(self.inter_dim, compress_c, 1, 1)
        self.weight_level_2 = AddConv(self.inter_dim, compress_c, 1, 1)
        self.weight_levels = nn.Conv2d(compress_c * 3, 3, kernel_size=1, stride=1, padding=0)

    def _fwd(self, tape, x0, x1, x2):
        tr = self.training
        saved = {}
        if self.level == 0:
            r0 = x0
            r1, saved["a1"] = ops.maxpool_fwd(x1, 2, 2, 0, want_arg=tape is not None)
            p2, saved["a2"] = ops.maxpool_fwd(x2, 3, 2, 1, want_arg=tape is not None)
            r2 = self.stride_level_2._fwd(tape, p2, tr)
        elif self.level == 1:
            r0 = ops.upsample_fwd(x0, 2)
            r1 = x1
            r2 = self.stride_level_2._fwd(tape, x2, tr)
        else:
            r0 = ops.upsample_fwd(self.compress_level_0._fwd(tape, x0, tr), 4)
            r1 = ops.upsample_fwd(self.compress_level_1._fwd(tape, x1, tr), 2)
            r2 = x2
        B, Cc, H, W = r0.shape
        wv = empty_nhwc(B, 24, H, W, r0.dtype, r0.device)
        self.weight_level_0._fwd(tape, r0, tr, out=wv[:, 0:8])
        self.weight_level_1._fwd(tape, r1, tr, out=wv[:, 8:16])
        self.weight_level_2._fwd(tape, r2, tr, out=wv[:, 16:24])
        logits = plain_conv_fwd(tape, self.weight_levels, wv)                 # [B,3,H,W] view of an 8/4-channel padded buffer
        fused = empty_nhwc(B, Cc, H, W, r0.dtype, r0.device)
        call("dy_asff_fuse_fwd", ptr(r0), ld_of(r0), ptr(r1), ld_of(r1), ptr(r2), ld_of(r2), ptr(logits), ld_of(logits),
             ptr(fused), ld_of(fused), B * H * W, Cc, ops.dt_id(r0.dtype), stream())
        ops.emu_round(fused)
        if tape is not None:
            saved.update(r0=r0, r1=r1, r2=r2, logits=logits, shapes=(x0.shape, x1.shape, x2.shape))
            tape.push(saved)
        return self.expand._fwd(tape, fused, tr)

    _takes_into = True

    def _bwd(self, tape, dy, needs=(True, True, True), into=None):
        dfused = conv_backward(tape, dy)
        s = tape.pop()
        r0, r1, r2, logits = s["r0"], s["r1"], s["r2"], s["logits"]
        B, Cc, H, W = r0.shape
        dt, dev = r0.dtype, r0.device
        into = list(into) if into is not None else [None, None, None]
        # the input this level takes as it is (r_k = x_k): the blend's gradient adds straight into what another consumer left
        same = self.level
        acc = [1 if (k == same and into[k] is not None) else 0 for k in range(3)]
        dr = [into[k] if acc[k] else empty_nhwc(B, Cc, H, W, dt, dev) for k in range(3)]
        lw = ld_of(logits)
        dlog = empty_nhwc(B, lw, H, W, dt, dev)
        call("dy_asff_fuse_bwd", ptr(dfused), ld_of(dfused), ptr(r0), ld_of(r0), ptr(r1), ld_of(r1), ptr(r2), ld_of(r2),
             ptr(logits), lw, ptr(dr[0]), ld_of(dr[0]), ptr(dr[1]), ld_of(dr[1]), ptr(dr[2]), ld_of(dr[2]), ptr(dlog), lw,
             B * H * W, Cc, acc[0], acc[1], acc[2], ops.dt_id(dt), stream())
        ops.emu_round(dr[0], dr[1], dr[2], dlog)
        dwv = conv_backward(tape, dlog[:, :3])                                # weight_levels -> [B,24,H,W]
        conv_backward(tape, dwv[:, 16:24], dx_out=dr[2], accumulate=True)     # weight_level_2
        conv_backward(tape, dwv[:, 8:16], dx_out=dr[1], accumulate=True)
        conv_backward(tape, dwv[:, 0:8], dx_out=dr[0], accumulate=True)
        (s0, s1, s2) = s["shapes"]
        A = [dict(dx_out=t, accumulate=True) if t is not None else {} for t in into]      # add into an existing gradient
        if self.level == 0:
            dp2 = conv_backward(tape, dr[2])                                  # stride_level_2
            dx2 = ops.maxpool_bwd(dp2, s["a2"], tuple(s2), 3, 2, 1, **A[2])
            dx1 = ops.maxpool_bwd(dr[1], s["a1"], tuple(s1), 2, 2, 0, **A[1])
            dx0 = dr[0]
        elif self.level == 1:
            dx2 = conv_backward(tape, dr[2], **A[2])
            dx1 = dr[1]
            dx0 = ops.upsample_bwd(dr[0], 2, **A[0])
        else:
            dx2 = dr[2]
            dx1 = conv_backward(tape, ops.upsample_bwd(dr[1], 2), **A[1])     # compress_level_1
            dx0 = conv_backward(tape, ops.upsample_bwd(dr[0], 4), **A[0])     # compress_level_0
        return tuple(None if into[k] is not None else d for k, d in enumerate((dx0, dx1, dx2)))


class GroupBatchnorm2d(nn.Module):
    """Parameter container (reference ultralytics/nn/modules/conv.py:323-343): weight ~ randn(c, 1, 1), bias = 0, eps 1e-10."""

    def __init__(self, c_num, group_num=16, eps=1e-10):
        super().__init__()
        assert c_num >= group_num
        self.group_num = group_num
        self.weight = nn.Parameter(torch.randn(c_num, 1, 1))
        self.bias = nn.Parameter(torch.zeros(c_num, 1, 1))
        self.eps = eps


class SRU(nn.Module):
    """Spatial reconstruction unit (conv.py:346-376): container; the arithmetic is dy_chan_moments + dy_sru_fwd / dy_sru_bwd."""

    def __init__(self, oup_channels, group_num=16, gate_treshold=0.5, torch_gn=False):
        super().__init__()
        if torch_gn or gate_treshold != 0.5:
            raise NotImplementedError("SRU: only the GroupBatchnorm2d variant with the 0.5 gate (what SCConv builds) is implemented")
        self.gn = GroupBatchnorm2d(oup_channels, group_num=group_num)
        self.gate_treshold = gate_treshold


class CRU(nn.Module):
    """Channel reconstruction unit (conv.py:379-417): container of its six convolutions (alpha 1/2, squeeze 2, 2 groups, 3x3)."""

    def __init__(self, op_channel, alpha=1 / 2, squeeze_radio=2, group_size=2, group_kernel_size=3):
        super().__init__()
        if (alpha, squeeze_radio, group_size, group_kernel_size) != (1 / 2, 2, 2, 3):
            raise NotImplementedError("CRU: only the default split (what SCConv / MFRU build) is implemented")
        self.up_channel = up = int(alpha * op_channel)
        self.low_channel = low = op_channel - up
        self.squeeze1 = nn.Conv2d(up, up // squeeze_radio, kernel_size=1, bias=False)
        self.squeeze2 = nn.Conv2d(low, low // squeeze_radio, kernel_size=1, bias=False)
        self.GWC = nn.Conv2d(up // squeeze_radio, op_channel, kernel_size=group_kernel_size, stride=1, padding=group_kernel_size // 2,
                             groups=group_size)
        self.PWC1 = nn.Conv2d(up // squeeze_radio, op_channel, kernel_size=1, bias=False)
        self.PWC2 = nn.Conv2d(low // squeeze_radio, op_channel - low // squeeze_radio, kernel_size=1, bias=False)


class SCConv(DyModule):
    """Spatial and channel reconstruction convolution (reference conv.py:420-440): SRU then CRU, channels preserved.
    Every parameter may be used more than once per step (MFRU applies one SCConv to two inputs), so all gradients are collected
    on the tape (which adds them up) instead of being written in place."""

    def __init__(self, op_channel, group_num=4, gate_treshold=0.5, alpha=1 / 2, squeeze_radio=2, group_size=2, group_kernel_size=3):
        super().__init__()
        if op_channel % 64 != 0:
            raise NotImplementedError("SCConv: channel count must be a multiple of 64 (16-byte channel vectors of the C/8 group slices)")
        self.SRU = SRU(op_channel, group_num=group_num, gate_treshold=gate_treshold)
        self.CRU = CRU(op_channel, alpha=alpha, squeeze_radio=squeeze_radio, group_size=group_size, group_kernel_size=group_kernel_size)
        self.c = op_channel

    def _gwc_views(self):
        g = self.CRU.GWC
        key = (g.weight.data_ptr(), g.bias.data_ptr())
        c = self.__dict__.get("_gwc_cache")
        if c is None or c[0] != key:
            h = self.c // 2
            # slices of the parameters as gradient-taking leaves of their own: conv_backward files their gradients under these
            # objects, _bwd copies them into the halves of the full GWC gradient
            c = (key, [g.weight.detach()[j * h:(j + 1) * h].requires_grad_(True) for j in range(2)],
                 [g.bias.detach()[j * h:(j + 1) * h].requires_grad_(True) for j in range(2)])
            self.__dict__["_gwc_cache"] = c
            for w in c[1]:
                ops.register_pack_view(w)
        return c[1], c[2]

    def _fwd(self, tape, x):
        B, C_, H, W = x.shape
        if C_ != self.c:
            raise RuntimeError(f"SCConv: expected {self.c} channels, got {C_}")
        dt, dev, did, st = x.dtype, x.device, ops.dt_id(x.dtype), stream()
        gn, cru = self.SRU.gn, self.CRU
        mom = torch.zeros((B, C_, 2), dtype=torch.float64, device=dev)
        call("dy_chan_moments", ptr(x), ld_of(x), B, H * W, C_, ptr(mom), did, st)
        y = empty_nhwc(B, C_, H, W, dt, dev)
        call("dy_sru_fwd", ptr(x), ld_of(x), ptr(y), ld_of(y), B, H * W, C_, gn.group_num, ptr(mom), ptr(gn.weight), ptr(gn.bias),
             float(gn.eps), did, st)
        h, q, e = C_ // 2, C_ // 4, C_ // 8
        buf = empty_nhwc(B, 2 * C_, H, W, dt, dev)                       # cat(Y1 [C], PWC2(low) [3C/4], low [C/4])
        kw = dict(shared=True)
        sq1 = conv_forward(tape, y[:, :h], cru.squeeze1.weight, None, None, ACT_NONE, 1, 0, 1, False, **kw)
        sq2 = conv_forward(tape, y[:, h:], cru.squeeze2.weight, None, None, ACT_NONE, 1, 0, 1, False, out=buf[:, 2 * C_ - q:], **kw)
        wv, bv = self._gwc_views()
        for j in range(2):                                               # grouped 3x3: group j maps channels [j e, (j+1) e) to [j h, (j+1) h)
            conv_forward(tape, sq1[:, j * e:(j + 1) * e], wv[j], bv[j], None, ACT_NONE, 1, 1, 1, False, out=buf[:, j * h:(j + 1) * h], **kw)
        pw1 = conv_forward(tape, sq1, cru.PWC1.weight, None, None, ACT_NONE, 1, 0, 1, False, **kw)
        copy2d(pw1, buf[:, :C_], accumulate=True)                        # Y1 = GWC(up) + PWC1(up)
        conv_forward(tape, sq2, cru.PWC2.weight, None, None, ACT_NONE, 1, 0, 1, False, out=buf[:, C_:2 * C_ - q], **kw)
        mom2 = torch.zeros((B, 2 * C_, 2), dtype=torch.float64, device=dev)
        call("dy_chan_moments", ptr(buf), ld_of(buf), B, H * W, 2 * C_, ptr(mom2), did, st)
        res = empty_nhwc(B, C_, H, W, dt, dev)
        call("dy_cru_fuse_fwd", ptr(buf), ld_of(buf), ptr(res), ld_of(res), B, H * W, C_, ptr(mom2), did, st)
        if tape is not None:
            tape.push(dict(x=x, mom=mom, buf=buf, mom2=mom2, views=(wv, bv)))
        return res

    def _bwd(self, tape, dres, needs=(True,)):
        s = tape.pop()
        x, mom, buf, mom2, (wv, bv) = s["x"], s["mom"], s["buf"], s["mom2"], s["views"]
        B, C_, H, W = x.shape
        dt, dev, did, st = x.dtype, x.device, ops.dt_id(x.dtype), stream()
        gn, cru = self.SRU.gn, self.CRU
        h, q, e = C_ // 2, C_ // 4, C_ // 8
        dbuf = empty_nhwc(B, 2 * C_, H, W, dt, dev)
        ds = torch.zeros((B, 2 * C_, 2), dtype=torch.float64, device=dev)
        call("dy_cru_fuse_bwd", ptr(buf), ld_of(buf), ptr(dres), ld_of(dres), ptr(dbuf), ld_of(dbuf), B, H * W, C_, ptr(mom2), ptr(ds), did, st)
        conv_backward(tape, dbuf[:, C_:2 * C_ - q], dx_out=dbuf[:, 2 * C_ - q:], accumulate=True)       # PWC2: d low' += ...
        dsq1 = conv_backward(tape, dbuf[:, :C_])                                                       # PWC1
        gw = torch.empty_like(cru.GWC.weight)
        gb = torch.empty_like(cru.GWC.bias)
        for j in (1, 0):                                                                               # GWC groups, reverse order
            conv_backward(tape, dbuf[:, j * h:(j + 1) * h], dx_out=dsq1[:, j * e:(j + 1) * e], accumulate=True)
            gw[j * h:(j + 1) * h].copy_(tape.pgrads.pop(wv[j]).view(h, e, 3, 3))
            gb[j * h:(j + 1) * h].copy_(tape.pgrads.pop(bv[j]).view(h))
        ops._add_pgrad(tape, cru.GWC.weight, gw)
        ops._add_pgrad(tape, cru.GWC.bias, gb)
        dy = empty_nhwc(B, C_, H, W, dt, dev)
        conv_backward(tape, dbuf[:, 2 * C_ - q:], dx_out=dy[:, h:])                                    # squeeze2
        conv_backward(tape, dsq1, dx_out=dy[:, :h])                                                    # squeeze1
        dx = empty_nhwc(B, C_, H, W, dt, dev)
        red = torch.zeros((B, C_, 2), dtype=torch.float64, device=dev)
        call("dy_sru_bwd", ptr(x), ld_of(x), ptr(dy), ld_of(dy), ptr(dx), ld_of(dx), B, H * W, C_, gn.group_num, ptr(mom), ptr(gn.weight),
             ptr(gn.bias), float(gn.eps), ptr(red), did, st)
        tot = red.sum(0).float()
        ops._add_pgrad(tape, gn.weight, tot[:, 1].reshape(C_, 1, 1))
        ops._add_pgrad(tape, gn.bias, tot[:, 0].reshape(C_, 1, 1))
        return dx


def _flush_shared_grads(tape):
    """Gradients collected on the tape for parameters that a module uses more than once: under the trainer's direct placement they
    are copied into the parameter's slot of the flat gradient buffer (and leave the tape), otherwise autograd gets them."""
    for p in list(tape.pgrads):
        gd = ops._grad_dst(p)
        if gd is not None:
            gd.copy_(tape.pgrads.pop(p).view_as(gd))


class MFRU(DyModule):
    """Multi-scale feature reconstruction unit (reference block.py:164-217, yolov8-3.yaml).  Inputs (P5 512 ch, P4 512 ch, P3 256 ch);
    ONE SCConv(512) + 1x1 conv serve both coarse levels and ONE SCConv(256) serves the fine level and the fused map; output 256
    channels at the P3 size."""

    def __init__(self, level=None):
        super().__init__()
        compress_c = 16
        self.scconv512 = SCConv(512)
        self.scconv256 = SCConv(256)
        self.pwconv = nn.Conv2d(512, 256, 1, 1, 0)
        self.weight_level_0 = nn.Conv2d(256, compress_c, 1, 1, 0)
        self.weight_level_1 = nn.Conv2d(256, compress_c, 1, 1, 0)
        self.weight_level_2 = nn.Conv2d(256, compress_c, 1, 1, 0)
        self.weight_levels = nn.Conv2d(compress_c * 3, 3, 1, 1, 0)

    def _pw(self, tape, m, x, out=None):
        return conv_forward(tape, x, m.weight, m.bias, None, ACT_NONE, 1, 0, 1, False, out=out, shared=True)

    def _fwd(self, tape, x0, x1, x2):
        r0 = ops.upsample_fwd(self._pw(tape, self.pwconv, self.scconv512._fwd(tape, x0)), 4)
        r1 = ops.upsample_fwd(self._pw(tape, self.pwconv, self.scconv512._fwd(tape, x1)), 2)
        r2 = self.scconv256._fwd(tape, x2)
        B, Cc, H, W = r2.shape
        if tuple(r0.shape) != tuple(r2.shape) or tuple(r1.shape) != tuple(r2.shape):
            raise RuntimeError("MFRU: inputs must be at strides 4 : 2 : 1")
        wv = empty_nhwc(B, 48, H, W, r2.dtype, r2.device)
        self._pw(tape, self.weight_level_0, r0, out=wv[:, 0:16])
        self._pw(tape, self.weight_level_1, r1, out=wv[:, 16:32])
        self._pw(tape, self.weight_level_2, r2, out=wv[:, 32:48])
        logits = self._pw(tape, self.weight_levels, wv)
        fused = empty_nhwc(B, Cc, H, W, r2.dtype, r2.device)
        call("dy_asff_fuse_fwd", ptr(r0), ld_of(r0), ptr(r1), ld_of(r1), ptr(r2), ld_of(r2), ptr(logits), ld_of(logits),
             ptr(fused), ld_of(fused), B * H * W, Cc, ops.dt_id(r2.dtype), stream())
        if tape is not None:
            tape.push(dict(r0=r0, r1=r1, r2=r2, logits=logits))
        return self.scconv256._fwd(tape, fused)

    def _bwd(self, tape, dy, needs=(True, True, True)):
        dfused = self.scconv256._bwd(tape, dy)
        s = tape.pop()
        r0, r1, r2, logits = s["r0"], s["r1"], s["r2"], s["logits"]
        B, Cc, H, W = r2.shape
        dt, dev = r2.dtype, r2.device
        dr = [empty_nhwc(B, Cc, H, W, dt, dev) for _ in range(3)]
        lw = ld_of(logits)
        dlog = empty_nhwc(B, lw, H, W, dt, dev)
        call("dy_asff_fuse_bwd", ptr(dfused), ld_of(dfused), ptr(r0), ld_of(r0), ptr(r1), ld_of(r1), ptr(r2), ld_of(r2),
             ptr(logits), lw, ptr(dr[0]), ld_of(dr[0]), ptr(dr[1]), ld_of(dr[1]), ptr(dr[2]), ld_of(dr[2]), ptr(dlog), lw,
             B * H * W, Cc, 0, 0, 0, ops.dt_id(dt), stream())
        dwv = conv_backward(tape, dlog[:, :3])                                # weight_levels -> [B,48,H,W]
        conv_backward(tape, dwv[:, 32:48], dx_out=dr[2], accumulate=True)
        conv_backward(tape, dwv[:, 16:32], dx_out=dr[1], accumulate=True)
        conv_backward(tape, dwv[:, 0:16], dx_out=dr[0], accumulate=True)
        dx2 = self.scconv256._bwd(tape, dr[2])
        dx1 = self.scconv512._bwd(tape, conv_backward(tape, ops.upsample_bwd(dr[1], 2)))
        dx0 = self.scconv512._bwd(tape, conv_backward(tape, ops.upsample_bwd(dr[0], 4)))
        _flush_shared_grads(tape)
        return dx0, dx1, dx2


class AsffDoubLevel(DyModule):
    """2-level adaptive spatial feature fusion (reference block.py:118-162).  Inputs (P4-like 512 ch, P3-like 256 ch at twice the
    resolution); level 0 fuses at the coarse size (512 ch out), level 1 at the fine size (256 ch out)."""

    def __init__(self, level):
        super().__init__()
        self.level = level
        self.dim = [512, 256]
        self.inter_dim = self.dim[level]
        if level == 0:
            self.stride_level_1 = AddConv(256, self.inter_dim, 3, 2)
            self.expand = AddConv(self.inter_dim, 512, 3, 1)
        elif level == 1:
            self.compress_level_0 = AddConv(512, self.inter_dim, 1, 1)
            self.expand = AddConv(self.inter_dim, 256, 3, 1)
        compress_c = 16
        self.weight_level_0 = AddConv(self.inter_dim, compress_c, 1, 1)
        self.weight_level_1 = AddConv(self.inter_dim, compress_c, 1, 1)
        self.weight_levels = nn.Conv2d(compress_c * 2, 2, kernel_size=1, stride=1, padding=0)

    def _fwd(self, tape, x0, x1):
        tr = self.training
        if self.level == 0:
            r0 = x0
            r1 = self.stride_level_1._fwd(tape, x1, tr)
        else:
            r0 = ops.upsample_fwd(self.compress_level_0._fwd(tape, x0, tr), 2)
            r1 = x1
        B, Cc, H, W = r0.shape
        wv = empty_nhwc(B, 32, H, W, r0.dtype, r0.device)
        self.weight_level_0._fwd(tape, r0, tr, out=wv[:, 0:16])
        self.weight_level_1._fwd(tape, r1, tr, out=wv[:, 16:32])
        logits = plain_conv_fwd(tape, self.weight_levels, wv)                 # [B,2,H,W] view of a channel-padded buffer
        fused = empty_nhwc(B, Cc, H, W, r0.dtype, r0.device)
        call("dy_asff_fuse_fwd", ptr(r0), ld_of(r0), ptr(r1), ld_of(r1), None, 0, ptr(logits), ld_of(logits),
             ptr(fused), ld_of(fused), B * H * W, Cc, ops.dt_id(r0.dtype), stream())
        ops.emu_round(fused)
        if tape is not None:
            tape.push(dict(r0=r0, r1=r1, logits=logits))
        return self.expand._fwd(tape, fused, tr)

    def _bwd(self, tape, dy, needs=(True, True)):
        dfused = conv_backward(tape, dy)
        s = tape.pop()
        r0, r1, logits = s["r0"], s["r1"], s["logits"]
        B, Cc, H, W = r0.shape
        dt, dev = r0.dtype, r0.device
        dr = [empty_nhwc(B, Cc, H, W, dt, dev) for _ in range(2)]
        lw = ld_of(logits)
        dlog = empty_nhwc(B, lw, H, W, dt, dev)
        call("dy_asff_fuse_bwd", ptr(dfused), ld_of(dfused), ptr(r0), ld_of(r0), ptr(r1), ld_of(r1), None, 0,
             ptr(logits), lw, ptr(dr[0]), ld_of(dr[0]), ptr(dr[1]), ld_of(dr[1]), None, 0, ptr(dlog), lw,
             B * H * W, Cc, 0, 0, 0, ops.dt_id(dt), stream())
        dwv = conv_backward(tape, dlog[:, :2])                                # weight_levels -> [B,32,H,W]
        conv_backward(tape, dwv[:, 16:32], dx_out=dr[1], accumulate=True)     # weight_level_1
        conv_backward(tape, dwv[:, 0:16], dx_out=dr[0], accumulate=True)
        if self.level == 0:
            dx1 = conv_backward(tape, dr[1])                                  # stride_level_1
            dx0 = dr[0]
        else:
            dx1 = dr[1]
            dx0 = conv_backward(tape, ops.upsample_bwd(dr[0], 2))             # compress_level_0
        return dx0, dx1


class RFBblock(DyModule):
    """Receptive-field block: four branches of biased convs with dilations 1/1/2/3, concatenated
    (reference block.py:703-734).  Branch outputs are written into slices of the output buffer."""

    def __init__(self, in_ch):
        super().__init__()
        q = in_ch // 4
        self.branch_0 = nn.Sequential(nn.Conv2d(in_ch, q, 1, 1, 0))
        self.branch_1 = nn.Sequential(nn.Conv2d(in_ch, q, 1, 1, 0), nn.Conv2d(q, q, 3, 1, 1))
        self.branch_2 = nn.Sequential(nn.Conv2d(in_ch, q, 1, 1, 0), nn.Conv2d(q, q, 3, 1, 1), nn.Conv2d(q, q, 3, 1, dilation=2, padding=2))
        self.branch_3 = nn.Sequential(nn.Conv2d(in_ch, q, 1, 1, 0), nn.Conv2d(q, q, 5, 1, 2), nn.Conv2d(q, q, 3, 1, dilation=3, padding=3))
        self.q = q

    def _branches(self):
        return (self.branch_0, self.branch_1, self.branch_2, self.branch_3)

    def _fwd(self, tape, x):
        B, _, H, W = x.shape
        q = self.q
        out = empty_nhwc(B, 4 * q, H, W, x.dtype, x.device)
        for i, br in enumerate(self._branches()):
            t = x
            for j, m in enumerate(br):
                t = plain_conv_fwd(tape, m, t, out=out[:, i * q:(i + 1) * q] if j == len(br) - 1 else None)
        return out

    def _bwd(self, tape, dy, needs=(True,)):
        q = self.q
        dx = None
        for i in reversed(range(4)):
            g = dy[:, i * q:(i + 1) * q]
            br = self._branches()[i]
            for j in reversed(range(len(br))):
                if j == 0:
                    dx = conv_backward(tape, g, dx_out=dx, accumulate=dx is not None)
                else:
                    g = conv_backward(tape, g)
        return dx


class DFL(nn.Module):
    """Integral of the distribution focal loss bins (reference block.py:220-238). The frozen 1x1 conv (weights = arange)
    is kept for state_dict compatibility; the softmax-expectation itself runs inside the decode kernels."""

    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1


class Detect(DyModule):
    """YOLOv8 detect head (reference ultralytics/nn/modules/head.py:19-102). Train: list of raw maps [B, 64+nc, h, w];
    eval: (y [B, 4+nc, A], maps)."""
    dynamic = False
    export = False
    shape = None
    anchors = torch.empty(0)
    strides = torch.empty(0)

    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], min(self.nc, 100))
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, self.nc, 1)) for x in ch)
        self.dfl = DFL(self.reg_max)

    def _level_fwd(self, tape, i, x):
        B, _, H, W = x.shape
        ve = ops.vec_elems(x.dtype)
        nc_pad = ops.round_up(self.nc, ve)
        buf = empty_nhwc(B, 4 * self.reg_max + nc_pad, H, W, x.dtype, x.device)
        a, b, c = self.cv2[i]
        t = b._fwd(tape, a._fwd(tape, x))
        plain_conv_fwd(tape, c, t, out=buf[:, :4 * self.reg_max])
        a, b, c = self.cv3[i]
        t = b._fwd(tape, a._fwd(tape, x))
        plain_conv_fwd(tape, c, t, out=buf[:, 4 * self.reg_max:4 * self.reg_max + nc_pad])
        return buf[:, :self.no]

    def _level_bwd(self, tape, i, g):
        r = 4 * self.reg_max
        nc_pad = ops.round_up(self.nc, ops.vec_elems(g.dtype))
        if ld_of(g) < r + nc_pad:
            raise RuntimeError("Detect: gradient map lacks channel padding")
        gt = conv_backward(tape, g[:, r:r + self.nc])                      # cv3[i][2]
        gt = self.cv3[i][1]._bwd(tape, gt)
        dx = self.cv3[i][0]._bwd(tape, gt)
        gt = conv_backward(tape, g[:, :r])                                 # cv2[i][2]
        gt = self.cv2[i][1]._bwd(tape, gt)
        return self.cv2[i][0]._bwd(tape, gt, dx_out=dx, accumulate=True)

    def _run_levels(self, fn, tapes, args):
        """fn(tapes[i], i, args[i]) for every pyramid level.  The levels are independent chains of small kernels (three convs
        + BatchNorm each way), so when the trainer enabled branch streams the coarser levels run on side streams next to
        level 0 on the compute stream: fork before, join after (everything later on the compute stream is ordered behind
        them).  The operand of a side level was allocated on the compute stream: it is marked as used by the side stream
        (record_stream), because the host drops its last reference (tape pop in the backward pass) while the side stream's
        kernels that read it are still queued -- without the mark the caching allocator hands the block to the next
        compute-stream allocation (level 0's weight gradient) at once and that kernel overwrites it first."""
        n = len(args)
        side = ops.branch_streams(n - 1, args[0].device) if (self.training and tapes[0] is not None) else None
        out = [None] * n
        if side is None:
            for i in range(n):
                out[i] = fn(tapes[i], i, args[i])
            return out
        main_raw = stream()
        for i in range(1, n):                       # issue the side levels first: they run while level 0 is being issued
            s = side[i - 1]
            call("dy_stream_fork", main_raw, s.cuda_stream)
            args[i].record_stream(s)
            with torch.cuda.stream(s):
                out[i] = fn(tapes[i], i, args[i])
        out[0] = fn(tapes[0], 0, args[0])
        for s in side:
            call("dy_stream_fork", s.cuda_stream, main_raw)
        return out

    def _fwd(self, tape, *xs):
        subs = [Tape() if tape is not None else None for _ in xs]          # one tape per level: the levels are independent
        maps = self._run_levels(self._level_fwd, subs, list(xs))
        if tape is not None:
            tape.push(subs)
        if self.training:
            return maps
        m = ops.det_maps(maps, self.strides_as_floats(), self.nc)
        A = sum(t.shape[2] * t.shape[3] for t in maps)
        y = torch.empty((maps[0].shape[0], 4 + self.nc, A), dtype=torch.float32, device=maps[0].device)
        call("dy_detect_decode", C.byref(m), ptr(y), stream())
        return (y, *maps)

    def strides_as_floats(self):
        """self.stride as host floats, read back once (float(tensor) per call is a device synchronisation)."""
        key = (id(self.stride), self.stride._version)
        if self.__dict__.get("_stride_key") != key:
            self.__dict__["_stride_host"] = [float(v) for v in self.stride.detach().cpu()]
            self.__dict__["_stride_key"] = key
        return self.__dict__["_stride_host"]

    def _wrap(self, out):
        if self.training:
            return list(out)
        return out[0], list(out[1:])

    def _bwd(self, tape, *dmaps, needs=None):
        if not self.training:
            raise RuntimeError("Detect: backward through the eval decode is not supported")
        subs = tape.pop()
        dxs = self._run_levels(self._level_bwd, subs, list(dmaps))
        for t in subs:
            assert not t.stack, "Detect: unbalanced level tape"
            for p, g in t.pgrads.items():
                ops._add_pgrad(tape, p, g)
        return dxs

    def bias_init(self):
        """reference head.py:95-102."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[:self.nc] = math.log(5 / self.nc / (640 / s) ** 2)


class AsffDetect(Detect):
    """Detect head with one 1x1 conv per branch and level (reference head.py:105-174): cv2[i] = Conv2d(ch_i, 64, 1),
    cv3[i] = Conv2d(ch_i, nc, 1); decode, anchors and bias_init as Detect."""

    def __init__(self, nc=80, ch=()):
        DyModule.__init__(self)
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        self.cv2 = nn.ModuleList(nn.Sequential(nn.Conv2d(x, 4 * self.reg_max, 1)) for x in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(nn.Conv2d(x, self.nc, 1)) for x in ch)
        self.dfl = DFL(self.reg_max)

    def _level_fwd(self, tape, i, x):
        B, _, H, W = x.shape
        nc_pad = ops.round_up(self.nc, ops.vec_elems(x.dtype))
        r = 4 * self.reg_max
        buf = empty_nhwc(B, r + nc_pad, H, W, x.dtype, x.device)
        plain_conv_fwd(tape, self.cv2[i][0], x, out=buf[:, :r])
        plain_conv_fwd(tape, self.cv3[i][0], x, out=buf[:, r:r + nc_pad])
        return buf[:, :self.no]

    def _level_bwd(self, tape, i, g):
        r = 4 * self.reg_max
        nc_pad = ops.round_up(self.nc, ops.vec_elems(g.dtype))
        if ld_of(g) < r + nc_pad:
            raise RuntimeError("AsffDetect: gradient map lacks channel padding")
        dx = conv_backward(tape, g[:, r:r + self.nc])                      # cv3[i][0]
        return conv_backward(tape, g[:, :r], dx_out=dx, accumulate=True)   # cv2[i][0]


# ------------------------------------------------------------------------------------------------ low-light front-end
class ConvBlock(nn.Module):
    """Parameter container of the extractor's conv + LeakyReLU(0.1) block (reference ultralytics/nn/modules/common.py:9-23)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, downsample=False, bn=False, activate=True):
        super().__init__()
        if bn:
            raise NotImplementedError("the front-end extractor uses bn=False")
        layers = [nn.Conv2d(in_channels, out_channels, kernel_size, 2 if downsample else 1, (kernel_size - 1) // 2)]
        if activate:
            layers.append(nn.LeakyReLU(0.1, inplace=False))
        self.conv_block = nn.Sequential(*layers)


_EXTRACTOR_F32 = os.environ.get("DY_EXTRACTOR_F32") is not None


class ExtractParameters2(nn.Module):
    """CNN parameter regressor (reference common.py:52-78): 5 x (conv3x3 s2 + LeakyReLU) 256^2 -> 8^2, fc 2048-64-15."""

    def __init__(self, cfg=None):
        super().__init__()
        self.output_dim = 15
        self.channels = 16
        c = self.channels
        self.conv_layers = nn.Sequential(ConvBlock(3, c, downsample=True), ConvBlock(c, 2 * c, downsample=True),
                                         ConvBlock(2 * c, 2 * c, downsample=True), ConvBlock(2 * c, 2 * c, downsample=True),
                                         ConvBlock(2 * c, 2 * c, downsample=True))
        self.fc1 = nn.Linear(2048, 64)
        self.fc2 = nn.Linear(64, self.output_dim)


class lowlight_recovery(DyModule):
    """Image-adaptive enhancement front-end (reference ultralytics/nn/modules/llie.py:11-54).

    forward(x [B,3,H,W] fp32, dedark_A [B,3] | None, IcA [B,1,H,W] | None) -> enhanced image [B,3,H,W] (a 3-channel view of an
    NHWC8 buffer in the compute dtype, consumed in place by the stem conv).  All filter math is fp32.
    """
    in_dtype = torch.float32

    _planar_grad_ok = True          # _bwd takes the planar gradient written by the direct stem dgrad kernel

    def __init__(self, in_channels=3, out_channels=3):
        super().__init__()
        self.extractor = ExtractParameters2()

    def _adapt(self, t):
        ops.require_gpu(t)
        return t.float().contiguous()          # NCHW fp32 image

    def forward(self, x, dedark_A=None, IcA=None):
        self._A = None if dedark_A is None else dedark_A.detach().float().contiguous()
        self._I = None if IcA is None else IcA.detach().float().contiguous()
        return super().forward(x)

    def _fwd(self, tape, x):
        B, _, H, W = x.shape
        dev = x.device
        f32 = torch.float32
        A, I = getattr(self, "_A", None), getattr(self, "_I", None)
        st = stream()
        # the five stride-2 convs of the regressor run in the compute dtype (the reference's autocast covers them as well); the two
        # fully connected layers and all filter math stay fp32.  DY_EXTRACTOR_F32=1 keeps the whole regressor in fp32.
        xd = f32 if _EXTRACTOR_F32 else ops.get_compute_dtype()
        r8 = torch.empty((B, 256, 256, 8), dtype=xd, device=dev).permute(0, 3, 1, 2)
        call("dy_image_to_nhwc8", ptr(x), B, H, W, ptr(r8), 256, 256, ops.dt_id(xd), st)
        t = r8[:, :3]
        ex = self.extractor
        for blk in ex.conv_layers:
            t = plain_conv_fwd(tape, blk.conv_block[0], t, act=ACT_LEAKY)
        if xd != f32:
            t = as_nhwc(t, f32)                          # [B,32,8,8]: 64 K values
        w1, w2 = self._fc_views()
        t = conv_forward(tape, t, w1, ex.fc1.bias, None, ACT_LEAKY, 1, 0, 1, False, owner=ex.fc1.weight)
        feat = conv_forward(tape, t, w2, ex.fc2.bias, None, ACT_NONE, 1, 0, 1, False, owner=ex.fc2.weight)    # [B,15,1,1], ld 16
        params = torch.empty((B, 8), dtype=f32, device=dev)
        call("dy_filter_params_fwd", ptr(feat), ld_of(feat), ptr(params), B, st)
        s4 = torch.empty((B, 3, H, W), dtype=f32, device=dev)
        call("dy_filters_pointwise_fwd", ptr(x), ptr(params), ptr(A), ptr(I), ptr(s4), B, H, W, int(ops.get_compute_dtype() != torch.float32), st)
        cd = ops.get_compute_dtype()
        out8 = torch.empty((B, H, W, 8), dtype=cd, device=dev).permute(0, 3, 1, 2)
        hp = torch.empty((B, 3, H, W), dtype=f32, device=dev) if tape is not None else None
        call("dy_usm_fwd", ptr(s4), ptr(params), None, ptr(out8), ptr(hp), B, H, W, ops.dt_id(cd), st)
        ops.emu_round(out8)
        if tape is not None:
            tape.push(dict(x=x, feat=feat, params=params, hp=hp, A=A, I=I))
        return out8[:, :3]

    def _fc_views(self):
        """The fully connected weights as conv weights ([64,32,8,8] window = whole input, [15,64,1,1]).  The view OBJECTS are kept:
        the packed-weight cache lives on the tensor object, and a fresh view per step meant four re-pack launches per step."""
        ex = self.extractor
        key = (ex.fc1.weight.data_ptr(), ex.fc2.weight.data_ptr())
        c = self.__dict__.get("_fc_view_cache")
        if c is None or c[0] != key:
            c = (key, ex.fc1.weight.detach().view(64, 32, 8, 8), ex.fc2.weight.detach().view(15, 64, 1, 1))
            self.__dict__["_fc_view_cache"] = c
            ops.register_pack_view(c[1])
            ops.register_pack_view(c[2])
        return c[1], c[2]

    def _bwd(self, tape, dout, needs=(False,)):
        s = tape.pop()
        x, feat, params, hp = s["x"], s["feat"], s["params"], s["hp"]
        B, _, H, W = x.shape
        dev, f32, st = x.device, torch.float32, stream()
        need_dx = bool(needs[0])
        dparams = torch.zeros((B, 8), dtype=torch.float64, device=dev)      # f64 atomics: order-free sums (frontend.hip)
        ds4 = torch.empty((B, 3, H, W), dtype=f32, device=dev)
        if dout.is_contiguous() and dout.shape[1] == 3:       # planar gradient from the direct stem dgrad kernel
            dld = 0
        else:
            ops.padded_channels(dout)           # raises unless dout is a zero-padded NHWC view
            dld = ld_of(dout)
        call("dy_usm_bwd", None, ptr(dout), dld, ptr(hp), ptr(params), ptr(ds4), ptr(dparams), B, H, W,
             ops.dt_id(dout.dtype), st)
        dx = torch.empty((B, 3, H, W), dtype=f32, device=dev) if need_dx else None
        call("dy_filters_pointwise_bwd", ptr(x), ptr(params), ptr(s["A"]), ptr(s["I"]), ptr(ds4), ptr(dx), ptr(dparams), B, H, W, 0,
             int(ops.get_compute_dtype() != torch.float32), st)
        fl = ld_of(feat)
        dfeat = torch.empty((B, 1, 1, fl), dtype=f32, device=dev).permute(0, 3, 1, 2)
        call("dy_filter_params_bwd", ptr(feat), fl, ptr(dparams), ptr(dfeat), B, st)
        g = conv_backward(tape, dfeat[:, :15])            # fc2
        g = conv_backward(tape, g)                        # fc1
        if not _EXTRACTOR_F32 and ops.get_compute_dtype() != f32:
            g = as_nhwc(g, ops.get_compute_dtype())
        for k in reversed(range(5)):
            g = conv_backward(tape, g, need_dx=(k > 0 or need_dx))
        if need_dx:
            if g.dtype != f32:
                g = as_nhwc(g, f32)
            call("dy_resize_bwd", ptr(g), ld_of(g), B, H, W, 256, 256, ptr(dx), st)
        return dx
