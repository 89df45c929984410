"""Oracle: YOLOv8 graph.  TEST INFRASTRUCTURE.

A model is (plan, sd): `plan` is a list of layer specs derived from the yaml dict, `sd` a flat dict of
fp32 tensors keyed exactly like the reference `state_dict()` (`model.<i>.<sub>.weight` ...).  Forward is
plain functional PyTorch on NCHW fp32; BatchNorm running buffers in `sd` are updated in place in train mode.
"""
import math

import torch
import torch.nn.functional as F

from . import frontend

BN_EPS = 1e-3        # U/utils/torch_utils.py:263
BN_MOM = 0.03        # U/utils/torch_utils.py:264
REG_MAX = 16         # U/nn/modules/head.py:33
ASFF_DIMS = (512, 512, 256)   # U/nn/modules/block.py:52


def _round_up(x, d):
    """make_divisible (U/utils/ops.py:128-142)."""
    return int(math.ceil(x / d) * d)


# ----------------------------------------------------------------------------- plan (registry / channel math)
def build_plan(cfg, scale=None, nc=None, ch=3):
    """yaml dict -> list of dict(i, f, kind, ...) following parse_model (U/nn/tasks.py:803-921).

    Channel rule: c2 = make_divisible(min(c2, max_ch) * width, 8) for Conv/C2f/SPPF (:861-868);
    repeats n = max(round(n*depth), 1) if n > 1 (:853); Concat sums (:884-885); lowlight_recovery c2 = args[0]
    (:886-887); AsffTribeLevel c2 = 512 for level 0/1 else 256 (:892-896); RFBblock keeps ch[f] (:903-904);
    Detect gets the list of input channels (:897-898).
    """
    nc = cfg["nc"] if nc is None else nc
    depth, width, max_ch = 1.0, 1.0, float("inf")
    if cfg.get("scales"):
        scale = scale or cfg.get("scale") or next(iter(cfg["scales"]))
        depth, width, max_ch = cfg["scales"][scale]
    chans, plan, save = [ch], [], []
    for i, (f, n, kind, args) in enumerate(cfg["backbone"] + cfg["head"]):
        args = [nc if a == "nc" else a for a in args]
        n = max(round(n * depth), 1) if n > 1 else n
        spec = dict(i=i, f=f, kind=kind)
        cin = chans[f] if isinstance(f, int) else [chans[j] for j in f]
        if kind in ("Conv", "C2f", "SPPF"):
            c2 = args[0]
            if c2 != nc:
                c2 = _round_up(min(c2, max_ch) * width, 8)
            spec.update(c1=cin, c2=c2)
            if kind == "Conv":
                spec.update(k=args[1] if len(args) > 1 else 1, s=args[2] if len(args) > 2 else 1)
            elif kind == "C2f":
                spec.update(n=n, shortcut=bool(args[1]) if len(args) > 1 else False)
            else:
                spec.update(k=args[1] if len(args) > 1 else 5)
        elif kind == "nn.Upsample":
            c2 = cin
            spec.update(scale=args[1])
        elif kind == "Concat":
            c2 = sum(cin)
        elif kind == "lowlight_recovery":
            c2 = args[0]
        elif kind == "AsffTribeLevel":
            c2 = 512 if args[0] in (0, 1) else 256
            spec.update(level=args[0])
        elif kind == "RFBblock":
            c2 = cin
            spec.update(c1=args[0])
        elif kind == "MFRU":
            c2 = 256                                  # tasks.py:890-891
        elif kind == "Detect":
            c2 = None
            spec.update(nc=args[0], ch=cin)
        else:
            raise NotImplementedError(kind)
        plan.append(spec)
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        if i == 0:
            chans = []
        chans.append(c2)
    return plan, sorted(set(save))


def scconv_shapes(out, p, c):
    """state_dict entries of SCConv(c) (conv.py:420-440; GroupBatchnorm2d :323-335, CRU :379-404)."""
    out[p + "SRU.gn.weight"] = (c, 1, 1)
    out[p + "SRU.gn.bias"] = (c, 1, 1)
    out[p + "CRU.squeeze1.weight"] = (c // 4, c // 2, 1, 1)
    out[p + "CRU.squeeze2.weight"] = (c // 4, c // 2, 1, 1)
    out[p + "CRU.GWC.weight"] = (c, c // 8, 3, 3)
    out[p + "CRU.GWC.bias"] = (c,)
    out[p + "CRU.PWC1.weight"] = (c, c // 4, 1, 1)
    out[p + "CRU.PWC2.weight"] = (c - c // 4, c // 4, 1, 1)


def detect_widths(nc, ch):
    """c2, c3 of Detect (U/nn/modules/head.py:38)."""
    return max(16, ch[0] // 4, REG_MAX * 4), max(ch[0], min(nc, 100))


def param_shapes(plan):
    """All state_dict entries (name -> shape) the reference module tree would hold, for `plan`."""
    out = {}

    def conv(p, c1, c2, k):      # Conv (U/nn/modules/conv.py:38-55): conv(bias=False)+bn
        out[p + "conv.weight"] = (c2, c1, k, k)
        bn(p + "bn.", c2)

    def bn(p, c):
        out[p + "weight"] = (c,)
        out[p + "bias"] = (c,)
        out[p + "running_mean"] = (c,)
        out[p + "running_var"] = (c,)
        out[p + "num_batches_tracked"] = ()

    def addconv(p, c1, c2, k):   # add_conv (block.py:24-45)
        out[p + "conv.weight"] = (c2, c1, k, k)
        bn(p + "batch_norm.", c2)

    def plain(p, c1, c2, k):     # nn.Conv2d with bias
        out[p + "weight"] = (c2, c1, k, k)
        out[p + "bias"] = (c2,)

    for L in plan:
        p = f"model.{L['i']}."
        kind = L["kind"]
        if kind == "Conv":
            conv(p, L["c1"], L["c2"], L["k"])
        elif kind == "C2f":
            c = int(L["c2"] * 0.5)
            conv(p + "cv1.", L["c1"], 2 * c, 1)
            conv(p + "cv2.", (2 + L["n"]) * c, L["c2"], 1)
            for j in range(L["n"]):
                conv(p + f"m.{j}.cv1.", c, c, 3)
                conv(p + f"m.{j}.cv2.", c, c, 3)
        elif kind == "SPPF":
            c_ = L["c1"] // 2
            conv(p + "cv1.", L["c1"], c_, 1)
            conv(p + "cv2.", 4 * c_, L["c2"], 1)
        elif kind == "lowlight_recovery":
            chs = [3, 16, 32, 32, 32, 32]
            for k in range(5):
                plain(p + f"extractor.conv_layers.{k}.conv_block.0.", chs[k], chs[k + 1], 3)
            out[p + "extractor.fc1.weight"] = (64, 2048)
            out[p + "extractor.fc1.bias"] = (64,)
            out[p + "extractor.fc2.weight"] = (frontend.N_FEAT, 64)
            out[p + "extractor.fc2.bias"] = (frontend.N_FEAT,)
        elif kind == "AsffTribeLevel":
            lv = L["level"]
            d = ASFF_DIMS[lv]
            if lv in (0, 1):
                addconv(p + "stride_level_2.", 256, d, 3)
                addconv(p + "expand.", d, 512, 3)
            else:
                addconv(p + "compress_level_0.", 512, d, 1)
                addconv(p + "compress_level_1.", 512, d, 1)
                addconv(p + "expand.", d, 256, 3)
            for j in range(3):
                addconv(p + f"weight_level_{j}.", d, 8, 1)
            plain(p + "weight_levels.", 24, 3, 1)
        elif kind == "MFRU":
            scconv_shapes(out, p + "scconv512.", 512)
            scconv_shapes(out, p + "scconv256.", 256)
            plain(p + "pwconv.", 512, 256, 1)
            for j in range(3):
                plain(p + f"weight_level_{j}.", 256, 16, 1)
            plain(p + "weight_levels.", 48, 3, 1)
        elif kind == "SCConv":
            scconv_shapes(out, p, L["c"])
        elif kind == "AsffDoubLevel":
            lv = L["level"]
            d = (512, 256)[lv]
            if lv == 0:
                addconv(p + "stride_level_1.", 256, d, 3)
                addconv(p + "expand.", d, 512, 3)
            else:
                addconv(p + "compress_level_0.", 512, d, 1)
                addconv(p + "expand.", d, 256, 3)
            for j in range(2):
                addconv(p + f"weight_level_{j}.", d, 16, 1)
            plain(p + "weight_levels.", 32, 2, 1)
        elif kind == "AsffDetect":
            for j, cj in enumerate(L["ch"]):
                plain(p + f"cv2.{j}.0.", cj, 4 * REG_MAX, 1)
                plain(p + f"cv3.{j}.0.", cj, L["nc"], 1)
            out[p + "dfl.conv.weight"] = (1, REG_MAX, 1, 1)
        elif kind == "RFBblock":
            c1 = L["c1"]
            q = c1 // 4
            plain(p + "branch_0.0.", c1, q, 1)
            plain(p + "branch_1.0.", c1, q, 1)
            plain(p + "branch_1.1.", q, q, 3)
            plain(p + "branch_2.0.", c1, q, 1)
            plain(p + "branch_2.1.", q, q, 3)
            plain(p + "branch_2.2.", q, q, 3)
            plain(p + "branch_3.0.", c1, q, 1)
            plain(p + "branch_3.1.", q, q, 5)
            plain(p + "branch_3.2.", q, q, 3)
        elif kind == "Detect":
            c2, c3 = detect_widths(L["nc"], L["ch"])
            for j, cj in enumerate(L["ch"]):
                conv(p + f"cv2.{j}.0.", cj, c2, 3)
                conv(p + f"cv2.{j}.1.", c2, c2, 3)
                plain(p + f"cv2.{j}.2.", c2, 4 * REG_MAX, 1)
                conv(p + f"cv3.{j}.0.", cj, c3, 3)
                conv(p + f"cv3.{j}.1.", c3, c3, 3)
                plain(p + f"cv3.{j}.2.", c3, L["nc"], 1)
            out[p + "dfl.conv.weight"] = (1, REG_MAX, 1, 1)
    return out


def rng_fill(shapes, seed):
    """Deterministic, platform-independent parameter fill shared by the golden generator and the tests.

    Keys are visited in sorted order with one numpy PCG64 stream; the value law depends on the key suffix only.
    """
    import numpy as np
    g = np.random.default_rng(seed)
    sd = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.int64)
            continue
        if ".dfl." in name or name.startswith("dfl."):
            sd[name] = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
            continue
        u = torch.from_numpy(g.random(shp, dtype=np.float32)) if shp else torch.tensor(float(g.random()))
        if name.endswith("running_var"):
            v = 0.5 + u
        elif name.endswith("running_mean"):
            v = 0.2 * u - 0.1
        elif name.endswith("bias"):
            v = 0.2 * u - 0.1
        elif len(shp) == 1:                      # bn / batch_norm weight
            v = 0.5 + u
        else:                                    # conv / linear weight: U(-a, a), a = sqrt(3 / fan_in)
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            a = math.sqrt(3.0 / fan_in)
            v = (2.0 * u - 1.0) * a
        sd[name] = v.contiguous()
    return sd


# ----------------------------------------------------------------------------- layers
def _bn(sd, p, x, train):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        train, BN_MOM, BN_EPS)


def _bn_tick(sd, p, train):
    if train:
        sd[p + "num_batches_tracked"] += 1


def conv_bn_silu(sd, p, x, k, s, train):
    """Conv.forward (conv.py:49-51): SiLU(BN(conv2d(x, pad=k//2, bias=False)))."""
    y = F.conv2d(x, sd[p + "conv.weight"], None, s, k // 2)
    _bn_tick(sd, p + "bn.", train)
    return F.silu(_bn(sd, p + "bn.", y, train))


def conv_bn_leaky(sd, p, x, k, s, train):
    """add_conv (block.py:24-45): conv(pad=(k-1)//2, bias=False) + BN + LeakyReLU(0.1)."""
    y = F.conv2d(x, sd[p + "conv.weight"], None, s, (k - 1) // 2)
    _bn_tick(sd, p + "batch_norm.", train)
    return F.leaky_relu(_bn(sd, p + "batch_norm.", y, train), 0.1)


def c2f(sd, p, x, n, shortcut, train):
    """C2f.forward (block.py:383-387) with Bottleneck (block.py:553-565; add only if shortcut, c1==c2 always here)."""
    y = conv_bn_silu(sd, p + "cv1.", x, 1, 1, train)
    c = y.shape[1] // 2
    parts = [y[:, :c], y[:, c:]]
    for j in range(n):
        t = conv_bn_silu(sd, p + f"m.{j}.cv1.", parts[-1], 3, 1, train)
        t = conv_bn_silu(sd, p + f"m.{j}.cv2.", t, 3, 1, train)
        parts.append(parts[-1] + t if shortcut else t)
    return conv_bn_silu(sd, p + "cv2.", torch.cat(parts, 1), 1, 1, train)


def sppf(sd, p, x, k, train):
    """SPPF.forward (block.py:333-338)."""
    a = conv_bn_silu(sd, p + "cv1.", x, 1, 1, train)
    b = F.max_pool2d(a, k, 1, k // 2)
    c = F.max_pool2d(b, k, 1, k // 2)
    d = F.max_pool2d(c, k, 1, k // 2)
    return conv_bn_silu(sd, p + "cv2.", torch.cat((a, b, c, d), 1), 1, 1, train)


def asff(sd, p, xs, level, train):
    """AsffTribeLevel.forward (block.py:82-115). xs = (deepest P5, P4, P3)."""
    x0, x1, x2 = xs
    if level == 0:
        r0 = x0
        r1 = F.max_pool2d(x1, 2, 2)
        r2 = conv_bn_leaky(sd, p + "stride_level_2.", F.max_pool2d(x2, 3, 2, 1), 3, 2, train)
    elif level == 1:
        r0 = F.interpolate(x0, scale_factor=2, mode="nearest")
        r1 = x1
        r2 = conv_bn_leaky(sd, p + "stride_level_2.", x2, 3, 2, train)
    else:
        r0 = F.interpolate(conv_bn_leaky(sd, p + "compress_level_0.", x0, 1, 1, train), scale_factor=4, mode="nearest")
        r1 = F.interpolate(conv_bn_leaky(sd, p + "compress_level_1.", x1, 1, 1, train), scale_factor=2, mode="nearest")
        r2 = x2
    w = torch.cat([conv_bn_leaky(sd, p + f"weight_level_{j}.", r, 1, 1, train) for j, r in enumerate((r0, r1, r2))], 1)
    w = F.conv2d(w, sd[p + "weight_levels.weight"], sd[p + "weight_levels.bias"])
    w = F.softmax(w, dim=1)
    fused = r0 * w[:, 0:1] + r1 * w[:, 1:2] + r2 * w[:, 2:]
    return conv_bn_leaky(sd, p + "expand.", fused, 3, 1, train)


def asff2(sd, p, xs, level, train):
    """AsffDoubLevel.forward (block.py:140-162). xs = (coarse 512-channel map, fine 256-channel map at twice the resolution)."""
    x0, x1 = xs
    if level == 0:
        r0 = x0
        r1 = conv_bn_leaky(sd, p + "stride_level_1.", x1, 3, 2, train)
    else:
        r0 = F.interpolate(conv_bn_leaky(sd, p + "compress_level_0.", x0, 1, 1, train), scale_factor=2, mode="nearest")
        r1 = x1
    w = torch.cat([conv_bn_leaky(sd, p + f"weight_level_{j}.", r, 1, 1, train) for j, r in enumerate((r0, r1))], 1)
    w = F.softmax(F.conv2d(w, sd[p + "weight_levels.weight"], sd[p + "weight_levels.bias"]), dim=1)
    fused = r0 * w[:, 0:1] + r1 * w[:, 1:2]
    return conv_bn_leaky(sd, p + "expand.", fused, 3, 1, train)


def scconv(sd, p, x, groups=4):
    """SCConv.forward (conv.py:420-440): SRU (:346-376 with GroupBatchnorm2d :323-343) then CRU (:379-417).
    SRU: normalise each of `groups` channel groups of every image by its mean and UNBIASED std (+1e-10 outside the root), affine;
    channels whose sigmoid(gn * gamma / sum(gamma)) reaches 0.5 are "informative"; the output adds every informative value to the
    non-informative value of the channel half a tensor away.  CRU: halves squeezed 2x by 1x1 convs; upper = grouped 3x3 (2 groups,
    bias) + 1x1 to C channels; lower = cat(1x1 to 3C/4, itself); the 2C channels are scaled by a softmax over their global means
    and the two halves added."""
    N, C, H, W = x.shape
    gw, gb = sd[p + "SRU.gn.weight"], sd[p + "SRU.gn.bias"]
    t = x.reshape(N, groups, -1)
    t = (t - t.mean(2, keepdim=True)) / (t.std(2, keepdim=True) + 1e-10)
    gn = t.reshape(N, C, H, W) * gw + gb
    informative = torch.sigmoid(gn * (gw / gw.sum()).view(1, C, 1, 1)) >= 0.5
    keep, rest = gn * informative, gn * ~informative
    h = C // 2
    y = torch.cat((keep[:, :h] + rest[:, h:], keep[:, h:] + rest[:, :h]), 1)
    up = F.conv2d(y[:, :h], sd[p + "CRU.squeeze1.weight"])
    low = F.conv2d(y[:, h:], sd[p + "CRU.squeeze2.weight"])
    y1 = F.conv2d(up, sd[p + "CRU.GWC.weight"], sd[p + "CRU.GWC.bias"], 1, 1, 1, 2) + F.conv2d(up, sd[p + "CRU.PWC1.weight"])
    o = torch.cat((y1, F.conv2d(low, sd[p + "CRU.PWC2.weight"]), low), 1)
    o = F.softmax(o.mean((2, 3), keepdim=True), dim=1) * o
    return o[:, :C] + o[:, C:]


def mfru(sd, p, xs):
    """MFRU.forward (block.py:188-217). xs = (P5 512 ch, P4 512 ch, P3 256 ch); the same scconv512 + pwconv serve P5 and P4, the same
    scconv256 serves P3 and the fused map."""
    x0, x1, x2 = xs
    pw = lambda t: F.conv2d(t, sd[p + "pwconv.weight"], sd[p + "pwconv.bias"])
    r0 = F.interpolate(pw(scconv(sd, p + "scconv512.", x0)), scale_factor=4, mode="nearest")
    r1 = F.interpolate(pw(scconv(sd, p + "scconv512.", x1)), scale_factor=2, mode="nearest")
    r2 = scconv(sd, p + "scconv256.", x2)
    w = torch.cat([F.conv2d(r, sd[p + f"weight_level_{j}.weight"], sd[p + f"weight_level_{j}.bias"]) for j, r in enumerate((r0, r1, r2))], 1)
    w = F.softmax(F.conv2d(w, sd[p + "weight_levels.weight"], sd[p + "weight_levels.bias"]), dim=1)
    return scconv(sd, p + "scconv256.", r0 * w[:, 0:1] + r1 * w[:, 1:2] + r2 * w[:, 2:])


def rfb(sd, p, x):
    """RFBblock.forward (block.py:703-734): 4 branches of biased convs (no BN/act), dilations 1/1/2/3, cat."""
    def cv(name, t, k, pad, dil=1):
        return F.conv2d(t, sd[p + name + "weight"], sd[p + name + "bias"], 1, pad, dil)
    b0 = cv("branch_0.0.", x, 1, 0)
    b1 = cv("branch_1.1.", cv("branch_1.0.", x, 1, 0), 3, 1)
    b2 = cv("branch_2.2.", cv("branch_2.1.", cv("branch_2.0.", x, 1, 0), 3, 1), 3, 2, 2)
    b3 = cv("branch_3.2.", cv("branch_3.1.", cv("branch_3.0.", x, 1, 0), 5, 2), 3, 3, 3)
    return torch.cat((b0, b1, b2, b3), 1)


def make_anchors(shapes, strides, offset=0.5, dtype=torch.float32):
    """tal.py:246-259: per level, x fastest then y; returns points [A,2] (grid units) and stride column [A,1]."""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sy, sx = torch.meshgrid(torch.arange(h, dtype=dtype) + offset, torch.arange(w, dtype=dtype) + offset,
                                indexing="ij")
        pts.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=dtype))
    return torch.cat(pts), torch.cat(st)


def dfl_expect(box):
    """DFL.forward (block.py:234-237): box [B, 4*16, A] -> expected distance [B, 4, A]."""
    b, _, a = box.shape
    prob = box.view(b, 4, REG_MAX, a).softmax(2)
    return (prob * torch.arange(REG_MAX, dtype=box.dtype).view(1, 1, REG_MAX, 1)).sum(2)


def detect(sd, p, xs, nc, strides, train):
    """Detect.forward (head.py:50-93). Train: list of 3 raw maps [B, 64+nc, h, w].
    Eval: (y [B, 4+nc, A] = cat(xywh*stride, sigmoid(cls)), maps)."""
    maps = []
    for j, x in enumerate(xs):
        t = conv_bn_silu(sd, p + f"cv2.{j}.0.", x, 3, 1, train)
        t = conv_bn_silu(sd, p + f"cv2.{j}.1.", t, 3, 1, train)
        box = F.conv2d(t, sd[p + f"cv2.{j}.2.weight"], sd[p + f"cv2.{j}.2.bias"])
        t = conv_bn_silu(sd, p + f"cv3.{j}.0.", x, 3, 1, train)
        t = conv_bn_silu(sd, p + f"cv3.{j}.1.", t, 3, 1, train)
        cls = F.conv2d(t, sd[p + f"cv3.{j}.2.weight"], sd[p + f"cv3.{j}.2.bias"])
        maps.append(torch.cat((box, cls), 1))
    if train:
        return maps
    b = maps[0].shape[0]
    no = 4 * REG_MAX + nc
    cat = torch.cat([m.view(b, no, -1) for m in maps], 2)
    box, cls = cat[:, :4 * REG_MAX], cat[:, 4 * REG_MAX:]
    pts, st = make_anchors([m.shape[2:] for m in maps], strides)
    d = dfl_expect(box)                                  # [B,4,A] l,t,r,b
    lt, rb = d[:, :2], d[:, 2:]
    a = pts.t()[None]                                    # [1,2,A]
    x1y1, x2y2 = a - lt, a + rb
    xywh = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * st.t()[None]      # dist2bbox xywh (tal.py:262-271)
    return torch.cat((xywh, cls.sigmoid()), 1), maps


def asff_detect(sd, p, xs, nc, strides, train):
    """AsffDetect.forward (head.py:128-165): one biased 1x1 conv per branch and level, then Detect's decode."""
    maps = []
    for j, x in enumerate(xs):
        box = F.conv2d(x, sd[p + f"cv2.{j}.0.weight"], sd[p + f"cv2.{j}.0.bias"])
        cls = F.conv2d(x, sd[p + f"cv3.{j}.0.weight"], sd[p + f"cv3.{j}.0.bias"])
        maps.append(torch.cat((box, cls), 1))
    if train:
        return maps
    b = maps[0].shape[0]
    no = 4 * REG_MAX + nc
    cat = torch.cat([m.view(b, no, -1) for m in maps], 2)
    box, cls = cat[:, :4 * REG_MAX], cat[:, 4 * REG_MAX:]
    pts, st = make_anchors([m.shape[2:] for m in maps], strides)
    d = dfl_expect(box)
    a = pts.t()[None]
    x1y1, x2y2 = a - d[:, :2], a + d[:, 2:]
    xywh = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * st.t()[None]
    return torch.cat((xywh, cls.sigmoid()), 1), maps


# ----------------------------------------------------------------------------- whole graph
def forward(plan, save, sd, x, train, A=None, IcA=None, want_front=False):
    """BaseModel._predict_once (tasks.py:74-118). In train mode the front-end always gets the defaults (:107-110)."""
    ys, front = [], None
    in_h = x.shape[2]
    for L in plan:
        f, p, kind = L["f"], f"model.{L['i']}.", L["kind"]
        if f != -1:
            x = ys[f] if isinstance(f, int) else [x if j == -1 else ys[j] for j in f]
        if kind == "lowlight_recovery":
            x = frontend.lowlight_recovery(sd, p, x, None if train else A, None if train else IcA)
            front = x
        elif kind == "Conv":
            x = conv_bn_silu(sd, p, x, L["k"], L["s"], train)
        elif kind == "C2f":
            x = c2f(sd, p, x, L["n"], L["shortcut"], train)
        elif kind == "SPPF":
            x = sppf(sd, p, x, L["k"], train)
        elif kind == "nn.Upsample":
            x = F.interpolate(x, scale_factor=L["scale"], mode="nearest")
        elif kind == "Concat":
            x = torch.cat(x, 1)
        elif kind == "AsffTribeLevel":
            x = asff(sd, p, x, L["level"], train)
        elif kind == "RFBblock":
            x = rfb(sd, p, x)
        elif kind == "MFRU":
            x = mfru(sd, p, x)
        elif kind == "Detect":
            # strides as the reference's 256x256 probe would find them (tasks.py:284-292): input H / map H
            strides = [float(in_h // t.shape[2]) for t in x]
            x = detect(sd, p, x, L["nc"], strides, train)
        ys.append(x if L["i"] in save else None)
    return (x, front) if want_front else x


def reference_initial_buffers(plan, save, sd, ch=3, s=256):
    """State of the BatchNorm buffers of a freshly constructed reference model: DetectionModel.__init__ (tasks.py:284-292) probes
    the strides with TWO train-mode forward passes of zeros(1, ch, 256, 256) BEFORE initialize_weights() (torch_utils.py:257-267)
    sets eps = 1e-3 / momentum = 0.03, i.e. with torch's BatchNorm defaults eps = 1e-5 / momentum = 0.1.  Updates `sd` in place
    (running_mean, running_var, num_batches_tracked); pinned by tests/golden/g10_initbuf.npz."""
    global BN_EPS, BN_MOM
    keep = (BN_EPS, BN_MOM)
    BN_EPS, BN_MOM = 1e-5, 0.1
    try:
        with torch.no_grad():
            for _ in range(2):
                forward(plan, save, sd, torch.zeros(1, ch, s, s), True)
    finally:
        BN_EPS, BN_MOM = keep
    return sd
