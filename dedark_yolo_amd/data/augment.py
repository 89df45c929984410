"""Training / validation transforms of the reference's data pipeline with the PIXEL work on the device.

Reference (host, per sample, cv2 + numpy in dataloader workers): ultralytics/data/base.py:142-169 load_image (resize so that the long
side is imgsz), ultralytics/data/augment.py `v8_transforms` :753-783 = Mosaic :118-216 -> CopyPaste(p=0) -> RandomPerspective :292-478 ->
MixUp(p=0) -> Albumentations (package absent: bookkeeping only) -> RandomHSV :480-499 -> RandomFlip x2 :502-537, then Format :697-751 and
YOLODataset.collate_fn (dataset.py:172-188).  Validation: LetterBox(scaleup=False) :540-603 + Format.

Here the split is:
  * `plan_train_sample` draws the random numbers in the reference's CALL ORDER from the same generators (`random`, `numpy.random`), so a
    run seeded like the reference picks the same mosaic partners, centre, affine matrix, HSV gains and flips (pinned by
    tests/golden/g13_augment.npz);
  * `train_labels` / `val_labels` move the boxes through the same float32 steps as the reference's Instances bookkeeping (a few dozen
    numbers per image: host numpy, like the reference);
  * the pixels never exist on the host in augmented form: `DeviceAugmenter` keeps the decoded uint8 images in HBM (or uploads them) and
    ONE kernel per batch (dy_aug_mosaic_warp) samples mosaic canvas -> affine warp (cv2.warpAffine's fixed-point bilinear) -> HSV
    gains (cv2's 8-bit BGR<->HSV + the three lookup tables) -> flips -> CHW RGB uint8, i.e. batch['img'] of the reference's batch dict.
    The 2s x 2s mosaic canvas is never materialised: every bilinear tap is resolved through the four placement rectangles.
There is no CPU pixel path: without the library the augmenter raises.
"""
import math
import random as _random
from types import SimpleNamespace

import numpy as np
import torch

F32 = np.float32


def AugmentHyp(**kw):
    """the reference's augmentation hyper-parameters (cfg/default.yaml:101-113)"""
    d = dict(mosaic=1.0, copy_paste=0.0, degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0, mixup=0.0, hsv_h=0.015,
             hsv_s=0.7, hsv_v=0.4, flipud=0.0, fliplr=0.5)
    d.update(kw)
    return SimpleNamespace(**d)


def _check_hyp(hyp):
    if hyp.mixup or hyp.copy_paste or hyp.perspective:
        raise NotImplementedError("mixup / copy_paste / perspective are 0 in the reference's configuration and are not implemented")


def plan_train_sample(index, shapes, buffer, imgsz, hyp, rnd=_random, nprnd=np.random):
    """Random draws of ONE training sample, in the call order of the reference's transform chain:
      Mosaic.__call__ (augment.py:86-104): uniform(0, 1) against p; 3 partners with random.choices(buffer, k=3) (:145-150);
        centre yc, xc = int(uniform(-x, 2 s + x)) for x in border = (-s // 2, -s // 2) (:161);
      RandomPerspective.affine_transform (:317-339): 2 perspective, rotation, scale, 2 shear, 2 translation draws;
      MixUp.__call__: uniform(0, 1) against p = 0 (still one draw);
      RandomHSV (:490): numpy.random.uniform(-1, 1, 3) when any gain is non-zero;
      RandomFlip vertical (:527): random.random(); RandomFlip horizontal (:530): random.random().
    `shapes[i]` = (h, w) of dataset image i at its load_image size.  Returns a SimpleNamespace plan."""
    _check_hyp(hyp)
    s = int(imgsz)
    p = SimpleNamespace(index=int(index), imgsz=s)
    p.mosaic = not (rnd.uniform(0, 1) > hyp.mosaic)
    if p.mosaic:
        p.sources = [int(index)] + [int(i) for i in rnd.choices(list(buffer), k=3)]
        border = (-s // 2, -s // 2)
        p.yc, p.xc = (int(rnd.uniform(-x, 2 * s + x)) for x in border)
        p.border = border
        p.canvas_hw = (2 * s, 2 * s)
        p.rects = mosaic4_rects(s, p.yc, p.xc, [shapes[i] for i in p.sources])
    else:                                                     # RandomPerspective's pre_transform: LetterBox((s, s)) (:767)
        p.sources = [int(index)]
        p.border = (0, 0)
        h, w = shapes[index]
        geo = letterbox_geometry((h, w), (s, s), scaleup=True)
        if (w, h) != geo.new_unpad:
            raise NotImplementedError("letterbox with resize inside the training chain: images must be at their load_image size")
        p.canvas_hw = (s, s)
        p.letterbox = geo
        p.rects = [(geo.left, geo.top, geo.left + w, geo.top + h, 0, 0, w, h)]
    draws = [rnd.uniform(-hyp.perspective, hyp.perspective), rnd.uniform(-hyp.perspective, hyp.perspective),
             rnd.uniform(-hyp.degrees, hyp.degrees), rnd.uniform(1 - hyp.scale, 1 + hyp.scale),
             rnd.uniform(-hyp.shear, hyp.shear), rnd.uniform(-hyp.shear, hyp.shear),
             rnd.uniform(0.5 - hyp.translate, 0.5 + hyp.translate), rnd.uniform(0.5 - hyp.translate, 0.5 + hyp.translate)]
    p.M, p.scale, p.size = affine_matrix(draws, p.canvas_hw, p.border)
    rnd.uniform(0, 1)                                          # MixUp's own coin (p = 0: never taken)
    p.hsv_gains = None
    if hyp.hsv_h or hyp.hsv_s or hyp.hsv_v:
        p.hsv_gains = nprnd.uniform(-1, 1, 3) * [hyp.hsv_h, hyp.hsv_s, hyp.hsv_v] + 1
        p.luts = hsv_luts(p.hsv_gains)
    p.flipud = rnd.random() < hyp.flipud
    p.fliplr = rnd.random() < hyp.fliplr
    return p


def mosaic4_rects(s, yc, xc, shapes):
    """placement of the four images around the centre (Mosaic._mosaic4, augment.py:166-186): canvas rectangle (x1a, y1a, x2a, y2a) and
    source rectangle (x1b, y1b, x2b, y2b) per image"""
    rects = []
    for i, (h, w) in enumerate(shapes):
        if i == 0:                                            # top left
            a = (max(xc - w, 0), max(yc - h, 0), xc, yc)
            b = (w - (a[2] - a[0]), h - (a[3] - a[1]), w, h)
        elif i == 1:                                          # top right
            a = (xc, max(yc - h, 0), min(xc + w, s * 2), yc)
            b = (0, h - (a[3] - a[1]), min(w, a[2] - a[0]), h)
        elif i == 2:                                          # bottom left
            a = (max(xc - w, 0), yc, xc, min(s * 2, yc + h))
            b = (w - (a[2] - a[0]), 0, w, min(a[3] - a[1], h))
        else:                                                 # bottom right
            a = (xc, yc, min(xc + w, s * 2), min(s * 2, yc + h))
            b = (0, 0, min(w, a[2] - a[0]), min(a[3] - a[1], h))
        rects.append(a + b)
    return rects


def rotation_matrix_2d(angle, scale):
    """cv2.getRotationMatrix2D(angle, (0, 0), scale) (OpenCV's documented closed form)"""
    a = scale * math.cos(angle * math.pi / 180)
    b = scale * math.sin(angle * math.pi / 180)
    return np.array([[a, b, 0.0], [-b, a, 0.0]], dtype=np.float64)


def affine_matrix(draws, canvas_hw, border):
    """T S R P C of RandomPerspective.affine_transform (augment.py:310-345), float32 like the reference.  Returns (M 3x3, scale, (w, h))."""
    size = canvas_hw[1] + border[1] * 2, canvas_hw[0] + border[0] * 2
    C = np.eye(3, dtype=F32)
    C[0, 2] = -canvas_hw[1] / 2
    C[1, 2] = -canvas_hw[0] / 2
    P = np.eye(3, dtype=F32)
    P[2, 0], P[2, 1] = draws[0], draws[1]
    R = np.eye(3, dtype=F32)
    R[:2] = rotation_matrix_2d(draws[2], draws[3])
    S = np.eye(3, dtype=F32)
    S[0, 1] = math.tan(draws[4] * math.pi / 180)
    S[1, 0] = math.tan(draws[5] * math.pi / 180)
    T = np.eye(3, dtype=F32)
    T[0, 2] = draws[6] * size[0]
    T[1, 2] = draws[7] * size[1]
    return T @ S @ R @ P @ C, draws[3], size


def hsv_luts(r):
    """RandomHSV's tables (augment.py:493-497)"""
    x = np.arange(0, 256, dtype=r.dtype)
    return (((x * r[0]) % 180).astype(np.uint8), np.clip(x * r[1], 0, 255).astype(np.uint8), np.clip(x * r[2], 0, 255).astype(np.uint8))


def letterbox_geometry(shape, new_shape, scaleup=True):
    """LetterBox's arithmetic (augment.py:566-590, center=True)"""
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = (new_shape[1] - new_unpad[0]) / 2, (new_shape[0] - new_unpad[1]) / 2
    return SimpleNamespace(r=r, new_unpad=new_unpad, dw=dw, dh=dh, top=int(round(dh - 0.1)), bottom=int(round(dh + 0.1)),
                           left=int(round(dw - 0.1)), right=int(round(dw + 0.1)))


# ---------------------------------------------------------------------------------------------------------------- labels (host, float32)
def _xywh2xyxy(x):
    y = np.empty_like(x)
    dw, dh = x[..., 2] / 2, x[..., 3] / 2
    y[..., 0], y[..., 1], y[..., 2], y[..., 3] = x[..., 0] - dw, x[..., 1] - dh, x[..., 0] + dw, x[..., 1] + dh
    return y


def _xyxy2xywh(x):
    y = np.copy(x)
    y[..., 0] = (x[..., 0] + x[..., 2]) / 2
    y[..., 1] = (x[..., 1] + x[..., 3]) / 2
    y[..., 2] = x[..., 2] - x[..., 0]
    y[..., 3] = x[..., 3] - x[..., 1]
    return y


def _mul(b, sx, sy):
    """Bboxes.mul (utils/instance.py:103-115): column by column, in place, python scalars"""
    b[:, 0] *= sx
    b[:, 1] *= sy
    b[:, 2] *= sx
    b[:, 3] *= sy


def _apply_affine(bboxes, M):
    n = len(bboxes)
    if n == 0:
        return bboxes
    xy = np.ones((n * 4, 3), dtype=bboxes.dtype)
    xy[:, :2] = bboxes[:, [0, 1, 2, 3, 0, 3, 2, 1]].reshape(n * 4, 2)
    xy = (xy @ M.T)[:, :2].reshape(n, 8)
    x, y = xy[:, [0, 2, 4, 6]], xy[:, [1, 3, 5, 7]]
    return np.concatenate((x.min(1), y.min(1), x.max(1), y.max(1)), dtype=bboxes.dtype).reshape(4, n).T


def _candidates(box1, box2, wh_thr=2, ar_thr=100, area_thr=0.1, eps=1e-16):
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + eps) > area_thr) & (ar < ar_thr)


def train_labels(plan, labels, shapes):
    """The boxes of one planned sample through the reference's bookkeeping: per source xywhn -> xyxy pixels + mosaic offset
    (Mosaic._update_labels :262-268), concatenation, clip to the canvas and zero-area removal (_cat_labels :270-288), affine + clip
    + box_candidates against the scaled originals (RandomPerspective.__call__ :432-468), xywh-normalised (Albumentations' bookkeeping
    :681-692), flips on normalised centres (RandomFlip :521-534), Format's denormalise / normalise round trip (:719-733).
    labels[i] = dict(cls [n,1] float32, bboxes [n,4] normalised xywh float32).  Returns (cls [m,1], bboxes [m,4]) float32."""
    cls, boxes = [], []
    for src, rect in zip(plan.sources, plan.rects):
        h, w = shapes[src]
        b = _xywh2xyxy(np.array(labels[src]["bboxes"], dtype=F32, copy=True).reshape(-1, 4))
        _mul(b, w, h)
        if plan.mosaic:
            padw, padh = rect[0] - rect[4], rect[1] - rect[5]           # Mosaic._update_labels: integer paste offset (augment.py:189-192)
        else:
            padw, padh = plan.letterbox.dw, plan.letterbox.dh          # LetterBox._update_labels: the UNROUNDED half padding (:593-603)
        b[:, 0] += padw
        b[:, 1] += padh
        b[:, 2] += padw
        b[:, 3] += padh
        boxes.append(b)
        cls.append(np.array(labels[src]["cls"], dtype=F32).reshape(-1, 1))
    b, c = np.concatenate(boxes, 0), np.concatenate(cls, 0)
    if plan.mosaic:
        ch, cw = plan.canvas_hw
        b[:, [0, 2]] = b[:, [0, 2]].clip(0, cw)
        b[:, [1, 3]] = b[:, [1, 3]].clip(0, ch)
        good = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) > 0
        b, c = b[good], c[good]
    nb = _apply_affine(b, plan.M)
    w, h = plan.size
    nb[:, [0, 2]] = nb[:, [0, 2]].clip(0, w)
    nb[:, [1, 3]] = nb[:, [1, 3]].clip(0, h)
    _mul(b, plan.scale, plan.scale)
    keep = _candidates(b.T, nb.T, area_thr=0.10)
    nb, c = nb[keep], c[keep]
    if len(c):                                                # Albumentations.__call__ touches the boxes only when there are any
        nb = _xyxy2xywh(nb)
        _mul(nb, 1 / w, 1 / h)
        normalized = True
    else:
        normalized = False
    # RandomFlip: convert_bbox('xywh') first (a real conversion when Albumentations skipped the empty set)
    if not normalized:
        nb = _xyxy2xywh(nb)
    fh, fw = (1, 1) if normalized else (h, w)
    if plan.flipud:
        nb[:, 1] = fh - nb[:, 1]
    if plan.fliplr:
        nb[:, 0] = fw - nb[:, 0]
    if normalized:                                            # Format: denormalize(w, h) ...
        _mul(nb, w, h)
    _mul(nb, 1 / w, 1 / h)                                    # ... then normalize(w, h)
    return c, nb


def val_labels(bboxes, shape, imgsz):
    """LetterBox(scaleup=False)._update_labels + Format for the validation set (augment.py:593-603, 719-733).  Returns (bboxes
    normalised xywh float32, ratio_pad ((r, r), (dw, dh)), geometry)."""
    h, w = shape
    geo = letterbox_geometry((h, w), (imgsz, imgsz), scaleup=False)
    b = _xywh2xyxy(np.array(bboxes, dtype=F32, copy=True).reshape(-1, 4))
    _mul(b, w, h)
    _mul(b, geo.r, geo.r)
    b[:, 0] += geo.dw
    b[:, 1] += geo.dh
    b[:, 2] += geo.dw
    b[:, 3] += geo.dh
    b = _xyxy2xywh(b)
    _mul(b, 1 / imgsz, 1 / imgsz)
    return b, ((geo.r, geo.r), (geo.dw, geo.dh)), geo


def collate(samples):
    """YOLODataset.collate_fn (dataset.py:172-188) for (cls, bboxes) pairs: concatenation + batch_idx"""
    cls = np.concatenate([c for c, _ in samples], 0) if samples else np.zeros((0, 1), F32)
    bb = np.concatenate([b for _, b in samples], 0) if samples else np.zeros((0, 4), F32)
    bi = np.concatenate([np.full(len(c), i, dtype=F32) for i, (c, _) in enumerate(samples)]) if samples else np.zeros(0, F32)
    return torch.from_numpy(bi), torch.from_numpy(cls), torch.from_numpy(bb)


# ---------------------------------------------------------------------------------------------------------------- device side
class DeviceAugmenter:
    """Owns the decoded dataset images (uint8 HWC BGR at their load_image size, device-resident) and produces the reference's batch dict
    {img uint8 [B,3,s,s] RGB, cls, bboxes, batch_idx, n_max} for lists of sample plans.  `images`: list of uint8 HWC numpy arrays or
    device tensors; `labels`: list of dict(cls, bboxes normalised xywh)."""

    def __init__(self, images, labels, imgsz, hyp=None, device="cuda"):
        from .. import ops
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceAugmenter: the pixel pipeline only exists on the device")
        self.imgsz = int(imgsz)
        self.hyp = hyp or AugmentHyp()
        self.images = [ops.require_gpu(im) if torch.is_tensor(im) else torch.from_numpy(np.ascontiguousarray(im)).to(self.device) for im in images]
        for im in self.images:
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_contiguous():
                raise ValueError("DeviceAugmenter: images must be contiguous uint8 HWC with 3 channels")
        self.shapes = [(int(im.shape[0]), int(im.shape[1])) for im in self.images]
        self.labels = labels
        self.buffer = list(range(len(self.images)))

    def plan(self, index, rnd=_random, nprnd=np.random):
        return plan_train_sample(index, self.shapes, self.buffer, self.imgsz, self.hyp, rnd, nprnd)

    def render(self, plans, staging=None):
        """pixels of a list of plans: uint8 [B, 3, s, s] RGB on the device (one launch on the current stream).  `staging`: optional
        pinned uint8 host tensor (>= B * sizeof(dy_aug_sample)) the descriptors are written into and copied from asynchronously;
        without it the copy comes from pageable memory and blocks the host."""
        from .._C import call
        from ..ops import ptr, stream
        B, s = len(plans), self.imgsz
        nbytes = B * descriptor_bytes()
        host = staging[:nbytes] if staging is not None else torch.empty(nbytes, dtype=torch.uint8)
        fill_descriptors(plans, self.images, host.data_ptr())
        dev = host.to(self.device, non_blocking=staging is not None)
        out = torch.empty((B, 3, s, s), dtype=torch.uint8, device=self.device)
        call("dy_aug_mosaic_warp", ptr(dev), B, s, s, ptr(out), stream())
        self._keep = dev                                      # descriptor array stays alive until the next call
        return out

    def batch(self, indices, rnd=_random, nprnd=np.random):
        plans = [self.plan(i, rnd, nprnd) for i in indices]
        img = self.render(plans)
        lab = [train_labels(p, self.labels, self.shapes) for p in plans]
        bi, cls, bb = collate(lab)
        n_max = max([len(c) for c, _ in lab] + [0])
        return dict(img=img, batch_idx=bi, cls=cls, bboxes=bb, n_max=n_max, im_file=[f"{i}" for i in indices],
                    ori_shape=[self.shapes[i] for i in indices], resized_shape=[(self.imgsz, self.imgsz)] * len(indices))


def descriptor_bytes():
    import ctypes as C
    from .._C import AugSample
    return C.sizeof(AugSample)


def fill_descriptors(plans, images, address):
    """dy_aug_sample[len(plans)] at `address` (host memory owned by the caller) for plans over `images` (device tensors by dataset index)"""
    import ctypes as C
    from .._C import AugSample
    arr = (AugSample * len(plans)).from_address(address)
    for k, p in enumerate(plans):
        a = arr[k]
        a.n_src = len(p.sources)
        for j, (src, r) in enumerate(zip(p.sources, p.rects)):
            im = images[src]
            a.src[j], a.sh[j], a.sw[j], a.pitch[j] = im.data_ptr(), im.shape[0], im.shape[1], im.stride(0)
            rj = a.rect[j]
            rj[0], rj[1], rj[2], rj[3], rj[4], rj[5] = int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5])
        a.canvas_h, a.canvas_w = p.canvas_hw
        minv = invert_affine(p.M[:2]).reshape(-1)
        m = a.minv
        m[0], m[1], m[2], m[3], m[4], m[5] = (float(v) for v in minv)
        a.hsv = int(p.hsv_gains is not None)
        if a.hsv:
            for c_ in range(3):
                C.memmove(a.lut[c_], p.luts[c_].ctypes.data, 256)
        a.flipud, a.fliplr = int(p.flipud), int(p.fliplr)


def invert_affine(M):
    """the inversion cv::warpAffine applies to its 2x3 matrix (double)"""
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


def load_resize(image, imgsz):
    """BaseDataset.load_image's resize (base.py:152-157, training: INTER_LINEAR) of a decoded uint8 HWC device image: long side -> imgsz.
    Returns the image itself when the ratio is 1."""
    from .._C import call
    from ..ops import ptr, stream
    h0, w0 = int(image.shape[0]), int(image.shape[1])
    r = imgsz / max(h0, w0)
    if r == 1:
        return image
    w, h = min(math.ceil(w0 * r), imgsz), min(math.ceil(h0 * r), imgsz)
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=image.device)
    call("dy_aug_resize_u8", ptr(image), h0, w0, image.stride(0), ptr(out), h, w, out.stride(0), stream())
    return out


def letterbox_batch(images, imgsz, scaleup=False):
    """Validation transform: LetterBox(new_shape=(imgsz, imgsz), scaleup) + Format's CHW RGB for a list of decoded uint8 HWC device images
    -> uint8 [B, 3, imgsz, imgsz] (resize + constant border 114 + channel flip in one launch per image)."""
    from .._C import call
    from ..ops import ptr, stream
    dev = images[0].device
    out = torch.empty((len(images), 3, imgsz, imgsz), dtype=torch.uint8, device=dev)
    geos = []
    for k, im in enumerate(images):
        h, w = int(im.shape[0]), int(im.shape[1])
        g = letterbox_geometry((h, w), (imgsz, imgsz), scaleup)
        geos.append(g)
        call("dy_aug_letterbox", ptr(im), h, w, im.stride(0), g.new_unpad[1], g.new_unpad[0], g.top, g.left, imgsz, imgsz, ptr(out[k]), stream())
    return out, geos


def dark_channel_prior(img):
    """Deterministic device version of the trainer's DarkChannel / AtmLight / DarkIcA (models/yolo/detect/train.py:42-68) on the darkened
    float image batch [B, 3, H, W] in [0, 1]: returns (dedark_A [B, 3], IcA [B, 1, H, W]) as preprocess_batch stores them (:95-96).
    Semantics where the reference leaves them open (ties of its unstable argsort, the uninitialised rows of DarkIcA's buffer) are the
    ones oracle/augment.py documents: ties by pixel index, rows >= 3 by the per-channel formula."""
    from .._C import call
    from ..ops import ptr, stream
    if img.dtype != torch.float32 or not img.is_cuda or img.dim() != 4 or img.shape[1] != 3:
        raise RuntimeError("dark_channel_prior expects the f32 device image batch [B, 3, H, W]")
    img = img.contiguous()
    B, _, H, W = img.shape
    A = torch.empty((B, 3), dtype=torch.float32, device=img.device)
    ica = torch.empty((B, 1, H, W), dtype=torch.float32, device=img.device)
    call("dy_dark_channel_prior", ptr(img), B, H, W, ptr(A), ptr(ica), stream())
    return A, ica
