"""Oracle: low-light enhancement front-end.  TEST INFRASTRUCTURE.

Functional fp32 restatement; parameters come from a flat dict keyed like the
reference state_dict (`<prefix>extractor.conv_layers.<k>.conv_block.0.weight`, ...).
"""
import math

import torch
import torch.nn.functional as F

N_FEAT = 15            # U/nn/modules/filter_cfg.py:17
SLOT_DEDARK = 0        # filter_cfg.py:19
SLOT_WB = 1            # filter_cfg.py:21 (3 slots)
SLOT_GAMMA = 4         # filter_cfg.py:22
SLOT_CONTRAST = 13     # filter_cfg.py:24
SLOT_USM = 14          # filter_cfg.py:25
DEFOG_RANGE = (0.1, 1.0)   # filter_cfg.py:35
USM_RANGE = (0.0, 5.0)     # filter_cfg.py:36
GAMMA_RANGE = 3.0          # filter_cfg.py:29
USM_RADIUS = 12            # filtersB.py:152
USM_SIGMA = 5.0            # filtersB.py:159


def squash(x, lo, hi):
    """tanh_range(l, r): tanh(x)*(r-l)/2 + (r+l)/2 ; `initial` is ignored (U/nn/modules/util_filters.py:295-304)."""
    return torch.tanh(x) * ((hi - lo) / 2.0) + ((hi + lo) / 2.0)


def regress(feat):
    """Per-image scalar parameters of the five filters from the 15 regressed features.

    Returns dict: omega[B,1], wb[B,3], gamma[B,1], alpha[B,1], lam[B,1].
    Follows the `filter_param_regressor`s: filtersB.py:151 (usm), :187 (dedark), :227-229 (gamma),
    :246-256 (white balance), :296-297 (contrast).
    """
    omega = squash(feat[:, SLOT_DEDARK:SLOT_DEDARK + 1], *DEFOG_RANGE)
    wb_in = feat[:, SLOT_WB:SLOT_WB + 3] * feat.new_tensor([[0.0, 1.0, 1.0]])     # R slot is masked out
    s = torch.exp(squash(wb_in, -0.5, 0.5))
    s = s / (1e-5 + 0.27 * s[:, 0] + 0.67 * s[:, 1] + 0.06 * s[:, 2])[:, None]
    lg = math.log(GAMMA_RANGE)
    gamma = torch.exp(squash(feat[:, SLOT_GAMMA:SLOT_GAMMA + 1], -lg, lg))
    alpha = torch.tanh(feat[:, SLOT_CONTRAST:SLOT_CONTRAST + 1])
    lam = squash(feat[:, SLOT_USM:SLOT_USM + 1], *USM_RANGE)
    return dict(omega=omega, wb=s, gamma=gamma, alpha=alpha, lam=lam)


def f_dedark(img, omega, A, IcA):
    """filtersB.py:190-216: (img - A) / max(1 - omega*IcA, 0.01) + A."""
    tx = 1.0 - omega[:, :, None, None] * IcA                 # [B,1,H,W]
    return (img - A[:, :, None, None]) / tx.clamp(min=0.01) + A[:, :, None, None]


def f_wb(img, s):
    """filtersB.py:258-259."""
    return img * s[:, :, None, None]


def f_gamma(img, gamma):
    """filtersB.py:231-233: pow(max(img, 1e-4), gamma)."""
    return torch.pow(img.clamp(min=1e-4), gamma[:, :, None, None])


def f_contrast(img, alpha):
    """filtersB.py:299-303 with rgb2lum of util_filters.py:270-273.

    NOTE (must-reproduce quirk): rgb2lum indexes the LAST dim of the NCHW tensor, so `lum` is a
    per-(b, c, row) scalar built from pixel COLUMNS 0, 1, 2 -> shape [B,3,H,1].
    """
    lum = (0.27 * img[..., 0] + 0.67 * img[..., 1] + 0.06 * img[..., 2])[..., None].clamp(0.0, 1.0)
    cl = -torch.cos(math.pi * lum) * 0.5 + 0.5
    ci = img / (lum + 1e-6) * cl
    a = alpha[:, :, None, None]
    return (1.0 - a) * img + a * ci


def gaussian_taps(dtype=torch.float32):
    """1-D taps of filtersB.py:152-161 (sigma 5, radius 12, normalised)."""
    x = torch.arange(-USM_RADIUS, USM_RADIUS + 1, dtype=dtype)
    k = torch.exp(-0.5 * (x / USM_SIGMA) ** 2)
    return k / k.sum()


def f_usm(img, lam):
    """filtersB.py:153-175: dense 25x25 gaussian (outer product), reflect pad 12, per channel."""
    k1 = gaussian_taps(img.dtype).to(img.device)
    k2 = (k1[:, None] * k1[None, :])[None, None]
    pad = F.pad(img, (USM_RADIUS,) * 4, mode="reflect")
    b, c, h, w = img.shape
    blur = F.conv2d(pad.reshape(b * c, 1, h + 2 * USM_RADIUS, w + 2 * USM_RADIUS), k2).reshape(b, c, h, w)
    return (img - blur) * lam[:, :, None, None] + img


def filter_chain(x, feat, A=None, IcA=None, stages=False):
    """DeDark -> WhiteBalance -> Gamma -> Contrast -> Usm (filter_cfg.py:75; llie.py:49-52)."""
    b, _, h, w = x.shape
    if A is None:
        A = torch.full((b, 3), 0.8, dtype=x.dtype, device=x.device)          # llie.py:34-36
    if IcA is None:
        IcA = torch.full((b, 1, h, w), 0.5, dtype=x.dtype, device=x.device)  # llie.py:38-40
    p = regress(feat)
    s1 = f_dedark(x, p["omega"], A, IcA)
    s2 = f_wb(s1, p["wb"])
    s3 = f_gamma(s2, p["gamma"])
    s4 = f_contrast(s3, p["alpha"])
    s5 = f_usm(s4, p["lam"])
    if stages:
        return s5, (s1, s2, s3, s4, s5), p
    return s5


def extractor(sd, prefix, r):
    """ExtractParameters2 (U/nn/modules/common.py:52-78): 5x(conv3x3 s2 p1 + bias, LeakyReLU 0.1),
    flatten NCHW -> 2048, fc 2048->64 (LeakyReLU 0.1), fc 64->15."""
    t = r
    for k in range(5):
        w = sd[f"{prefix}conv_layers.{k}.conv_block.0.weight"]
        bias = sd[f"{prefix}conv_layers.{k}.conv_block.0.bias"]
        t = F.leaky_relu(F.conv2d(t, w, bias, stride=2, padding=1), 0.1)
    t = t.reshape(-1, 2048)
    t = F.leaky_relu(F.linear(t, sd[f"{prefix}fc1.weight"], sd[f"{prefix}fc1.bias"]), 0.1)
    return F.linear(t, sd[f"{prefix}fc2.weight"], sd[f"{prefix}fc2.bias"])


def lowlight_recovery(sd, prefix, x, A=None, IcA=None, stages=False):
    """llie.py:17-54.  Bilinear (align_corners=False) 256x256 copy -> extractor -> filter chain at full res."""
    r = F.interpolate(x, size=(256, 256), mode="bilinear", align_corners=False)
    feat = extractor(sd, prefix + "extractor.", r)
    if stages:
        out, st, p = filter_chain(x, feat, A, IcA, stages=True)
        return out, feat, st, p
    return filter_chain(x, feat, A, IcA)
