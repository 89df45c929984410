"""Host side of the device input pipeline against the reference's own transform objects (g13_augment.npz, written by
tests/golden/make_augment_golden.py from ultralytics/data/augment.py with fixed RNG seeds): RNG call order, mosaic geometry and canvas
pixels, the affine matrix, RandomHSV's tables, flips, label bookkeeping, collate; LetterBox geometry / labels for validation; the
defined part of the trainer's DarkChannel / AtmLight / DarkIcA.  No GPU: planner + label math are host numpy, the pixel oracle too."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


@pytest.fixture(scope="module")
def g13():
    return np.load(os.path.join(ROOT, "tests", "golden", "g13_augment.npz"))


def synth_dataset(seed, n, imgsz):
    """same law as make_augment_golden.synth_dataset (kept in step by the canvas comparison below)"""
    g = np.random.default_rng(seed)
    ims, labels = [], []
    for i in range(n):
        if i % 2:
            h, w = imgsz, int(g.integers(imgsz // 2, imgsz + 1))
        else:
            h, w = int(g.integers(imgsz // 2, imgsz + 1)), imgsz
        ims.append(g.integers(0, 256, (h, w, 3), dtype=np.uint8))
        k = int(g.integers(1, 5))
        xy = g.uniform(0.2, 0.8, (k, 2))
        wh = g.uniform(0.1, 0.5, (k, 2))
        labels.append(dict(cls=g.integers(0, 20, (k, 1)).astype(np.float32), bboxes=np.concatenate((xy, wh), 1).astype(np.float32)))
    return ims, labels


def _hyp(v):
    from dedark_yolo_amd.data.augment import AugmentHyp
    return AugmentHyp(degrees=float(v[0]), translate=float(v[1]), scale=float(v[2]), shear=float(v[3]), perspective=float(v[4]), hsv_h=float(v[5]),
                      hsv_s=float(v[6]), hsv_v=float(v[7]), flipud=float(v[8]), fliplr=float(v[9]), mosaic=float(v[10]))


@pytest.mark.parametrize("tag", ["t0", "t1", "t2"])
def test_train_plan_and_labels_follow_the_reference(g13, tag):
    from dedark_yolo_amd.data import augment as A
    from oracle import augment as oa
    z = g13
    imgsz, seed, picks = int(z[f"{tag}_imgsz"]), int(z[f"{tag}_data_seed"]), [int(i) for i in z[f"{tag}_picks"]]
    ims, labels = synth_dataset(seed, int(z[f"{tag}_n_img"]), imgsz)
    shapes = [im.shape[:2] for im in ims]
    hyp = _hyp(z[f"{tag}_hyp"])
    random.seed(seed + 1)
    np.random.seed(seed + 2)
    lab, flips = [], 0
    for k, idx in enumerate(picks):
        p = A.plan_train_sample(idx, shapes, list(range(len(ims))), imgsz, hyp)
        assert p.mosaic == (tag != "t2") and tuple(z[f"{tag}_s{k}_dsize"]) == tuple(p.size)
        # the canvas the reference hands to cv2.warpAffine = its own numpy paste of the four images (t2: Mosaic's coin fails and
        # RandomPerspective's LetterBox pre_transform pads the single image instead, augment.py:432-433,767)
        if p.mosaic:
            canvas = oa.mosaic4_canvas(imgsz, p.rects, [ims[i] for i in p.sources])
        else:
            canvas = oa.letterbox(ims[p.sources[0]], (imgsz, imgsz))
            r = p.rects[0]
            assert np.array_equal(canvas[r[1]:r[3], r[0]:r[2]], ims[p.sources[0]]) and (r[2] - r[0], r[3] - r[1]) == shapes[p.sources[0]][::-1]
        assert np.array_equal(canvas, z[f"{tag}_s{k}_canvas"]), f"sample {k}: mosaic canvas"
        assert np.array_equal(p.M[:2], z[f"{tag}_s{k}_M"]) and p.M.dtype == np.float32, f"sample {k}: affine matrix"
        assert np.array_equal(np.stack(p.luts), z[f"{tag}_s{k}_lut"]), f"sample {k}: HSV tables"
        # flips + Format on the stand-in's placeholder pattern
        yy, xx, cc = np.meshgrid(np.arange(p.size[1]), np.arange(p.size[0]), np.arange(3), indexing="ij")
        pat = ((xx + 3 * yy + 5 * cc) % 251).astype(np.uint8)
        if p.flipud:
            pat = np.flipud(pat)
        if p.fliplr:
            pat = np.fliplr(pat)
        flips += int(p.flipud) + int(p.fliplr)
        assert np.array_equal(oa.format_img(pat), z[f"{tag}_s{k}_img"]), f"sample {k}: flips / channel order"
        c, b = A.train_labels(p, labels, shapes)
        assert np.array_equal(c.reshape(-1), z[f"{tag}_s{k}_cls"].reshape(-1)), f"sample {k}: classes kept"
        assert b.dtype == np.float32 and np.array_equal(b, z[f"{tag}_s{k}_bboxes"].reshape(-1, 4)), f"sample {k}: boxes"
        lab.append((c, b))
    assert flips > 0
    bi, cls, bb = A.collate(lab)
    assert np.array_equal(bi.numpy(), z[f"{tag}_batch_idx"]) and np.array_equal(bb.numpy(), z[f"{tag}_batch_bboxes"])
    assert np.array_equal(cls.numpy().reshape(-1), z[f"{tag}_batch_cls"].reshape(-1))
    # both generators were consumed exactly as far as the reference consumed them
    assert np.array_equal(np.array([random.random(), np.random.uniform()]), z[f"{tag}_rng_after"])


def test_val_letterbox_geometry_and_labels(g13):
    from dedark_yolo_amd.data import augment as A
    z = g13
    imgsz = int(z["val_imgsz"])
    for k, (h, w) in enumerate(z["val_shapes"]):
        b, rp, geo = A.val_labels(z[f"val_v{k}_in_bboxes"], (int(h), int(w)), imgsz)
        assert np.array_equal(b, z[f"val_v{k}_bboxes"]), k
        rs = tuple(z[f"val_v{k}_resize"])
        assert rs == ((-1, -1) if (int(w), int(h)) == geo.new_unpad else geo.new_unpad)
        assert tuple(z[f"val_v{k}_border"]) == (geo.top, geo.bottom, geo.left, geo.right)


def test_dark_channel_prior_defined_part(g13):
    """oracle/augment.py against the trainer's own methods where those are defined: dark channel everywhere, AtmLight when the cut has
    no ties (the fixture's images), rows 0..2 of DarkIcA.  The product kernel is held to this oracle on the GPU (test_gpu_augment.py)."""
    from oracle import augment as oa
    z = g13
    for k in range(3):
        im = z[f"dark_d{k}_im"]
        dark = oa.dark_channel(im)
        assert np.array_equal(dark, z[f"dark_d{k}_dark"])
        A = oa.atm_light(im, dark)
        assert np.array_equal(A, z[f"dark_d{k}_A"]) and np.array_equal(oa.atm_light_reference(im, dark), A)
        ica, rows = oa.dark_ica(im, A)
        assert rows == 3 and np.array_equal(ica[:3], z[f"dark_d{k}_ica_rows012"])


def test_pixel_oracle_sanity():
    """properties of the OpenCV restatements (cv2 is absent, so these are the only independent checks): identity resize / warp are exact,
    a constant image stays constant, an integer translation is a shift with a 114 border, BGR -> HSV -> BGR returns within 2 levels,
    H stays below 180."""
    from oracle import augment as oa
    g = np.random.default_rng(5)
    im = g.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(oa.cv_resize_linear_u8(im, (53, 37)), im)
    assert np.array_equal(oa.cv_warp_affine_linear_u8(im, np.array([[1, 0, 0], [0, 1, 0]], np.float32), (53, 37)), im)
    flat = np.full((20, 30, 3), 77, np.uint8)
    assert (oa.cv_resize_linear_u8(flat, (45, 31)) == 77).all()
    sh = oa.cv_warp_affine_linear_u8(im, np.array([[1, 0, 5], [0, 1, -3]], np.float32), (53, 37))
    assert np.array_equal(sh[:34, 5:], im[3:, :48]) and (sh[34:] == 114).all() and (sh[:, :5] == 114).all()
    up = oa.cv_resize_linear_u8(im, (106, 74))
    assert up.shape == (74, 106, 3) and abs(float(up.mean()) - float(im.mean())) < 1.5
    hsv = oa.cv_bgr2hsv_u8(im)
    assert hsv[..., 0].max() < 180
    back = oa.cv_hsv2bgr_u8(hsv)
    assert np.abs(back.astype(int) - im.astype(int)).max() <= 6          # hue has 180 levels
    gray = np.repeat(g.integers(0, 256, (5, 5, 1), dtype=np.uint8), 3, 2)
    assert (oa.cv_bgr2hsv_u8(gray)[..., :2] == 0).all() and np.array_equal(oa.cv_hsv2bgr_u8(oa.cv_bgr2hsv_u8(gray)), gray)
