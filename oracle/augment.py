"""Oracle: the input pipeline in front of the hot path.  TEST INFRASTRUCTURE -- numpy on the CPU.

Two kinds of code live here:

* restatements of the REFERENCE's own numpy / Python code (ultralytics/data/augment.py: Mosaic._mosaic4 :158-195, _cat_labels :268-288,
  RandomPerspective.affine_transform / apply_bboxes / box_candidates :310-478, RandomHSV's lookup tables :493-497, RandomFlip :517-537,
  LetterBox :559-603, Format :715-751; ultralytics/data/base.py:142-169 load_image; ultralytics/utils/instance.py box bookkeeping;
  ultralytics/models/yolo/detect/train.py:42-68 DarkChannel / AtmLight / DarkIcA).  These are PINNED: tests/golden/make_augment_golden.py
  runs the reference's transform objects on synthetic images with fixed RNG seeds and records the mosaic canvas, the affine matrix, the
  lookup tables and the final labels (g13_augment.npz).

* restatements of the OpenCV routines the reference calls on pixels -- cv2.resize(INTER_LINEAR), cv2.warpAffine(INTER_LINEAR,
  borderValue=114), cv2.cvtColor(BGR2HSV / HSV2BGR), cv2.LUT, cv2.copyMakeBorder, cv2.getRotationMatrix2D -- for 8-bit images.
  `opencv-python` is a dependency the reference does not vendor, does not pin (README: `pip install opencv-python`) and that is
  absent from this image: **parity unpinned** for these.  They follow OpenCV 4.x's published fixed-point algorithms
  (modules/imgproc/src/resize.cpp: INTER_RESIZE_COEF_BITS = 11, the `>> 4 ... >> 16 ... + 2 >> 2` vertical pass; imgwarp.cpp: AB_BITS = 10,
  INTER_BITS = 5, 15-bit bilinear weights; color_hsv.cpp: the 12-bit division tables of RGB2HSV_b and the float path of HSV2RGB_b).
"""
import math

import numpy as np

# ------------------------------------------------------------------------------------------------ OpenCV restatements (unpinned)
def cv_round(x):
    """cvRound / saturate_cast<int>(double): round half to even"""
    return np.rint(x)


def get_rotation_matrix_2d(angle, center, scale):
    """cv2.getRotationMatrix2D (documented closed form): [[a, b, (1-a) cx - b cy], [-b, a, b cx + (1-a) cy]], a = s cos, b = s sin."""
    a = scale * math.cos(angle * math.pi / 180)
    b = scale * math.sin(angle * math.pi / 180)
    cx, cy = center
    return np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]], dtype=np.float64)


def _resize_axis(src_n, dst_n):
    """per destination index: first source tap and the two 11-bit coefficients (resize.cpp, INTER_LINEAR)"""
    scale = src_n / dst_n                                   # double, like `scale_x = 1. / inv_scale_x`
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def cv_resize_linear_u8(img, dsize):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 HWC."""
    sh, sw = img.shape[:2]
    dw, dh = int(dsize[0]), int(dsize[1])
    sx, fx = _resize_axis(sw, dw)
    lo, hi = sx < 0, sx >= sw - 1                           # columns: taps clamped AND the fraction zeroed
    fx = np.where(lo | hi, np.float32(0), fx)
    sx = np.clip(sx, 0, sw - 1)
    sx1 = np.minimum(sx + 1, sw - 1)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    sy, fy = _resize_axis(sh, dh)                           # rows: taps clamped, fraction kept
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    y0, y1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    s = img.astype(np.int64)
    rows = s[:, sx] * a0[None, :, None] + s[:, sx1] * a1[None, :, None]          # horizontal pass, ints scaled by 2048
    r0, r1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def invert_affine(M):
    """the inversion at the top of cv::warpAffine (double)"""
    m = np.array(M, dtype=np.float64).reshape(2, 3).copy()
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[1, 1] * D, m[0, 0] * D
    m[0, 0] = A11
    m[0, 1] *= -D
    m[1, 0] *= -D
    m[1, 1] = A22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    return m


def warp_coords(minv, dw, dh):
    """fixed-point source coordinates of cv::warpAffine (AB_BITS 10, INTER_BITS 5): integer part and 5-bit fractions per pixel"""
    x = np.arange(dw, dtype=np.float64)
    y = np.arange(dh, dtype=np.float64)
    adelta = cv_round(minv[0, 0] * x * 1024).astype(np.int64)
    bdelta = cv_round(minv[1, 0] * x * 1024).astype(np.int64)
    X0 = cv_round((minv[0, 1] * y + minv[0, 2]) * 1024).astype(np.int64) + 16
    Y0 = cv_round((minv[1, 1] * y + minv[1, 2]) * 1024).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sat = lambda v: np.clip(v, -32768, 32767)
    return sat(X >> 5), sat(Y >> 5), X & 31, Y & 31


def cv_warp_affine_linear_u8(img, M, dsize, border=114, fetch=None):
    """cv2.warpAffine(img, M, dsize=(w, h), borderValue=(114,)*3) for uint8 HWC, INTER_LINEAR (flags default).
    `fetch(ys, xs)` (optional) returns the source pixel [..., C] and a validity mask instead of indexing `img` (mosaic: the
    canvas is never materialised)."""
    dw, dh = int(dsize[0]), int(dsize[1])
    sx, sy, fx, fy = warp_coords(invert_affine(M), dw, dh)
    if fetch is None:
        sh, sw = img.shape[:2]

        def fetch(ys, xs):
            ok = (ys >= 0) & (ys < sh) & (xs >= 0) & (xs < sw)
            return img[np.clip(ys, 0, sh - 1), np.clip(xs, 0, sw - 1)].astype(np.int64), ok
    acc = 0
    for dy_, dx_, w in ((0, 0, (32 - fx) * (32 - fy)), (0, 1, fx * (32 - fy)), (1, 0, (32 - fx) * fy), (1, 1, fx * fy)):
        p, ok = fetch(sy + dy_, sx + dx_)
        p = np.where(ok[..., None], p, border)
        acc = acc + p * (32 * w)[..., None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


_SDIV = np.array([0] + [int(np.rint((255 << 12) / (1.0 * i))) for i in range(1, 256)], dtype=np.int64)
_HDIV180 = np.array([0] + [int(np.rint((180 << 12) / (6.0 * i))) for i in range(1, 256)], dtype=np.int64)


def cv_bgr2hsv_u8(img):
    """cv2.cvtColor(img, cv2.COLOR_BGR2HSV) for uint8 (H in [0, 180)): RGB2HSV_b's integer arithmetic."""
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    s = (diff * _SDIV[v] + (1 << 11)) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * _HDIV180[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack((h, s, v), -1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])


def cv_hsv2bgr_u8(img):
    """cv2.cvtColor(img, cv2.COLOR_HSV2BGR) for uint8: HSV2RGB_b = float conversion of (h, s / 255, v / 255) and back."""
    f = np.float32
    h = img[..., 0].astype(f) * f(6.0 / 180.0)
    s = img[..., 1].astype(f) * f(1.0 / 255.0)
    v = img[..., 2].astype(f) * f(1.0 / 255.0)
    h = np.where(h >= 6, h - f(6), h).astype(f)             # h < 6 * 255 / 180: at most one wrap; h >= 0 always
    sector = np.floor(h).astype(np.int64)
    h = (h - sector.astype(f)).astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    h = np.where(bad, f(0), h).astype(f)
    tab = np.stack((v, (v * (f(1) - s)).astype(f), (v * (f(1) - (s * h).astype(f)).astype(f)).astype(f),
                    (v * (f(1) - (s * (f(1) - h).astype(f)).astype(f)).astype(f)).astype(f)), -1)
    idx = _SECTOR[sector]                                   # [..., 3] -> b, g, r
    bgr = np.take_along_axis(tab, idx, -1)
    bgr = np.where((img[..., 1] == 0)[..., None], v[..., None], bgr).astype(f)
    return np.clip(np.rint((bgr * f(255)).astype(f)), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ reference numpy / Python code (pinned)
def hsv_luts(r):
    """RandomHSV's three lookup tables for gains r = uniform(-1, 1, 3) * [h, s, v] + 1 (augment.py:490-497)."""
    x = np.arange(0, 256, dtype=r.dtype)
    return (((x * r[0]) % 180).astype(np.uint8), np.clip(x * r[1], 0, 255).astype(np.uint8), np.clip(x * r[2], 0, 255).astype(np.uint8))


def random_hsv(img, r):
    """RandomHSV.__call__ on a BGR image (augment.py:486-499)."""
    hsv = cv_bgr2hsv_u8(img)
    lh, ls, lv = hsv_luts(r)
    return cv_hsv2bgr_u8(np.stack((lh[hsv[..., 0]], ls[hsv[..., 1]], lv[hsv[..., 2]]), -1))


def load_resize_shape(h0, w0, imgsz):
    """BaseDataset.load_image's target size (base.py:152-157): long side -> imgsz, (w, h) or None when r == 1."""
    r = imgsz / max(h0, w0)
    if r == 1:
        return None
    return min(math.ceil(w0 * r), imgsz), min(math.ceil(h0 * r), imgsz)


def mosaic4_rects(s, yc, xc, shapes):
    """Mosaic._mosaic4's placement (augment.py:166-186): for each of the 4 images (h, w) the canvas rectangle (x1a, y1a, x2a, y2a) and
    the source rectangle (x1b, y1b, x2b, y2b); padw = x1a - x1b, padh = y1a - y1b."""
    out = []
    for i, (h, w) in enumerate(shapes):
        if i == 0:
            x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
            x1b, y1b, x2b, y2b = w - (x2a - x1a), h - (y2a - y1a), w, h
        elif i == 1:
            x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s * 2), yc
            x1b, y1b, x2b, y2b = 0, h - (y2a - y1a), min(w, x2a - x1a), h
        elif i == 2:
            x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s * 2, yc + h)
            x1b, y1b, x2b, y2b = w - (x2a - x1a), 0, w, min(y2a - y1a, h)
        else:
            x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s * 2), min(s * 2, yc + h)
            x1b, y1b, x2b, y2b = 0, 0, min(w, x2a - x1a), min(y2a - y1a, h)
        out.append((x1a, y1a, x2a, y2a, x1b, y1b, x2b, y2b))
    return out


def mosaic4_canvas(s, rects, imgs):
    """the 2s x 2s canvas (augment.py:165,188): later images overwrite earlier ones where rectangles are empty / overlap"""
    img4 = np.full((s * 2, s * 2, 3), 114, dtype=np.uint8)
    for (x1a, y1a, x2a, y2a, x1b, y1b, x2b, y2b), im in zip(rects, imgs):
        img4[y1a:y2a, x1a:x2a] = im[y1b:y2b, x1b:x2b]
    return img4


def affine_matrix(draws, img_hw, border, degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0):
    """RandomPerspective.affine_transform's matrix (augment.py:310-345) from its 8 draws, in call order:
    draws = (p_x, p_y, angle, scale, shear_x, shear_y, t_x, t_y) already in their own ranges.  Returns (M float32 3x3, s, (w, h))."""
    f = np.float32
    size = img_hw[1] + border[1] * 2, img_hw[0] + border[0] * 2          # w, h (augment.py:441)
    C = np.eye(3, dtype=f)
    C[0, 2] = -img_hw[1] / 2
    C[1, 2] = -img_hw[0] / 2
    P = np.eye(3, dtype=f)
    P[2, 0], P[2, 1] = draws[0], draws[1]
    R = np.eye(3, dtype=f)
    a, s = draws[2], draws[3]
    R[:2] = get_rotation_matrix_2d(angle=a, center=(0, 0), scale=s)
    S = np.eye(3, dtype=f)
    S[0, 1] = math.tan(draws[4] * math.pi / 180)
    S[1, 0] = math.tan(draws[5] * math.pi / 180)
    T = np.eye(3, dtype=f)
    T[0, 2] = draws[6] * size[0]
    T[1, 2] = draws[7] * size[1]
    return T @ S @ R @ P @ C, s, size


def apply_bboxes(bboxes, M):
    """RandomPerspective.apply_bboxes (augment.py:352-371), affine case."""
    n = len(bboxes)
    if n == 0:
        return bboxes
    xy = np.ones((n * 4, 3), dtype=bboxes.dtype)
    xy[:, :2] = bboxes[:, [0, 1, 2, 3, 0, 3, 2, 1]].reshape(n * 4, 2)
    xy = xy @ M.T
    xy = xy[:, :2].reshape(n, 8)
    x, y = xy[:, [0, 2, 4, 6]], xy[:, [1, 3, 5, 7]]
    return np.concatenate((x.min(1), y.min(1), x.max(1), y.max(1)), dtype=bboxes.dtype).reshape(4, n).T


def box_candidates(box1, box2, wh_thr=2, ar_thr=100, area_thr=0.1, eps=1e-16):
    """augment.py:471-477; box1 / box2 [4, n]"""
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + eps) > area_thr) & (ar < ar_thr)


def xywhn_to_xyxy_px(b, w, h):
    """Instances.convert_bbox('xyxy') + denormalize(w, h) of normalised xywh rows (utils/instance.py, ops.xywh2xyxy), float32, in place
    semantics restated on a copy"""
    b = b.astype(np.float32).copy()
    y = np.empty_like(b)
    dw, dh = b[:, 2] / 2, b[:, 3] / 2
    y[:, 0], y[:, 1], y[:, 2], y[:, 3] = b[:, 0] - dw, b[:, 1] - dh, b[:, 0] + dw, b[:, 1] + dh
    y[:, 0] *= w
    y[:, 2] *= w
    y[:, 1] *= h
    y[:, 3] *= h
    return y


def letterbox_geometry(shape, new_shape=(640, 640), scaleup=True, center=True):
    """LetterBox's numbers (augment.py:566-590): ratio r, resized (w, h), padding (top, bottom, left, right), (dw, dh)."""
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if center:
        dw /= 2
        dh /= 2
    top, bottom = (int(round(dh - 0.1)) if center else 0), int(round(dh + 0.1))
    left, right = (int(round(dw - 0.1)) if center else 0), int(round(dw + 0.1))
    return r, new_unpad, (top, bottom, left, right), (dw, dh)


def letterbox(img, new_shape=(640, 640), scaleup=True):
    """LetterBox()(image=img): resize if needed + constant border 114 (augment.py:586-591)."""
    r, new_unpad, (top, bottom, left, right), _ = letterbox_geometry(img.shape[:2], new_shape, scaleup)
    if img.shape[:2][::-1] != new_unpad:
        img = cv_resize_linear_u8(img, new_unpad)
    out = np.full((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), 114, dtype=np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out


def format_img(img):
    """Format._format_img (augment.py:745-751): HWC BGR -> CHW RGB, contiguous"""
    return np.ascontiguousarray(img.transpose(2, 0, 1)[::-1])


# ------------------------------------------------------------------------------------------------ dark-channel prior (detect/train.py:42-68)
def dark_channel(im):
    """DarkChannel (:42-45): per-pixel minimum over the 3 channels of an HWC uint8 image"""
    return im.min(2)


def atm_light_reference(im, dark):
    """AtmLight exactly as written (:47-63), numpy's default (unstable) argsort included: only deterministic across numpy builds
    when the dark values around the cut are distinct."""
    h, w = im.shape[:2]
    imsz = h * w
    numpx = int(max(math.floor(imsz / 1000), 1))
    darkvec, imvec = dark.reshape(imsz, 1), im.reshape(imsz, 3)
    indices = darkvec.argsort(0)[(imsz - numpx):imsz]
    atmsum = np.zeros([1, 3])
    for ind in range(1, numpx):
        atmsum = atmsum + imvec[indices[ind]]
    return atmsum / numpx


def atm_light(im, dark):
    """The DEFINED version the product implements: the same sum with ties broken by pixel index (a stable ascending sort: among equal
    dark values the later pixel ranks higher), so the numpx brightest-dark pixels are well defined; like the reference it leaves out
    the FIRST of them (its loop starts at 1) and still divides by numpx."""
    h, w = im.shape[:2]
    imsz = h * w
    numpx = int(max(math.floor(imsz / 1000), 1))
    order = np.argsort(dark.reshape(imsz), kind="stable")[(imsz - numpx):]
    return im.reshape(imsz, 3)[order[1:]].astype(np.float64).sum(0, keepdims=True) / numpx


def dark_ica(im, A):
    """DarkIcA (:65-68).  As written it fills rows 0..2 of an UNINITIALISED uint8 array with `row / A[0, row]` (the loop indexes rows
    of the HWC image where channels were meant) and takes the channel minimum: rows 0..2 of the result are defined -- returned
    here exactly -- and rows >= 3 are whatever np.empty held.  The product DEFINES those rows by the formula the fork's own test
    script uses (utils/test_code/test_dedark_preprocess.py:50-53: channel c divided by A[0, c]), which is what the method computes
    once the index is put on the axis it was meant for.  Returns (IcA [H, W] uint8, defined_rows = 3)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.empty(im.shape, im.dtype)
        q = im.astype(np.float64) / A[0][None, None, :]                       # rows >= 3: per-channel (the defined extension)
        out[...] = _to_u8(q)
        for ind in range(3):
            out[ind, :, :] = _to_u8(im[ind, :, :].astype(np.float64) / A[0, ind])    # rows 0..2: as the reference writes them
    return out.min(2), 3


def _to_u8(q):
    """float64 -> uint8 assignment as numpy does it on x86-64 (C cast through a wider integer: truncation toward zero, modulo 256);
    inf / nan (division by a zero A) land on 0"""
    q = np.where(np.isfinite(q), q, 0.0)
    return (np.trunc(q).astype(np.int64) & 255).astype(np.uint8)
