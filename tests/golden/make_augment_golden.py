#!/usr/bin/env python3
"""tests/golden/g13_augment.npz: the reference's training transforms (ultralytics/data/augment.py:118-795 `v8_transforms` + `Format`,
dataset hyper-parameters of cfg/default.yaml:101-113) run on synthetic images with FIXED RNG seeds, in the build container.

cv2 is absent from the image and the reference calls it for every pixel operation, so the stand-in below RECORDS instead of computing:
what reaches cv2.warpAffine (the mosaic canvas -- assembled by the reference's own numpy code -- the 2x3 matrix, dsize, border value),
the three lookup tables RandomHSV hands to cv2.LUT, and cv2.copyMakeBorder / cv2.resize arguments of LetterBox.  Everything the reference
computes ITSELF is therefore pinned: RNG call order, mosaic geometry and canvas pixels, the affine matrix T S R P C, the HSV tables, flips,
label clipping / filtering / normalisation and the collated batch layout.  The pixel arithmetic of cv2 (bilinear warp, resize, colour
conversion) stays unpinned (oracle/augment.py restates it from OpenCV's published source).  The one cv2 function whose RESULT the
reference's numpy code consumes, getRotationMatrix2D, is given its documented closed form (oracle.augment.get_rotation_matrix_2d).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_augment_golden.py
"""
import os
import random
import sys
import types
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.argv = [sys.argv[0]]
sys.dont_write_bytecode = True
os.environ.setdefault("YOLO_CONFIG_DIR", "/tmp/yolo_cfg_golden")
os.makedirs(os.environ["YOLO_CONFIG_DIR"], exist_ok=True)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import augment as oa  # noqa: E402

REC = dict(warp=[], lut=[], border=[], resize=[], flipped=[])


def _install_import_stubs():
    class _Names(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return 0
    cv2 = _Names("cv2")
    cv2.setNumThreads = lambda *a, **k: None
    cv2.imshow = lambda *a, **k: None
    cv2.getRotationMatrix2D = lambda angle, center, scale: oa.get_rotation_matrix_2d(angle, center, scale)

    def warp_affine(img, M, dsize=None, borderValue=None, **kw):
        REC["warp"].append(dict(img=img.copy(), M=np.array(M).copy(), dsize=tuple(dsize), border=tuple(borderValue)))
        yy, xx, cc = np.meshgrid(np.arange(dsize[1]), np.arange(dsize[0]), np.arange(img.shape[2]), indexing="ij")
        return ((xx + 3 * yy + 5 * cc) % 251).astype(np.uint8)       # placeholder pixels (cv2 absent): a pattern that shows flips / channel order
    cv2.warpAffine = warp_affine
    cv2.cvtColor = lambda img, code, dst=None: img
    cv2.split = lambda img: tuple(img[..., k] for k in range(img.shape[2]))
    cv2.merge = lambda chans: np.stack(chans, -1)

    def lut(src, table):
        REC["lut"].append(np.array(table).copy())
        return src
    cv2.LUT = lut

    def resize(img, dsize, interpolation=None, **kw):
        REC["resize"].append(dict(shape=img.shape[:2], dsize=tuple(dsize)))
        return np.full((dsize[1], dsize[0], img.shape[2]), 9, dtype=np.uint8)
    cv2.resize = resize

    def copy_make_border(img, top, bottom, left, right, btype, value=None):
        REC["border"].append((top, bottom, left, right, tuple(value)))
        out = np.full((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]), value[0], dtype=np.uint8)
        out[top:top + img.shape[0], left:left + img.shape[1]] = img
        return out
    cv2.copyMakeBorder = copy_make_border
    sys.modules["cv2"] = cv2
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)
        __setattr__ = dict.__setitem__
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.0.0"
    tv.ops = types.ModuleType("torchvision.ops")
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules.update({"torchvision": tv, "torchvision.ops": tv.ops, "torchvision.transforms": tv.transforms})


_install_import_stubs()
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
from ultralytics.data.augment import Compose, Format, LetterBox, v8_transforms  # noqa: E402
from ultralytics.data.dataset import YOLODataset  # noqa: E402
from ultralytics.utils.instance import Instances  # noqa: E402

HYP = dict(mosaic=1.0, copy_paste=0.0, degrees=0.0, translate=0.1, scale=0.5, shear=0.0, perspective=0.0, mixup=0.0, hsv_h=0.015, hsv_s=0.7,
           hsv_v=0.4, flipud=0.0, fliplr=0.5, mask_ratio=4, overlap_mask=True)


def synth_dataset(seed, n, imgsz):
    """n decoded BGR images already at their load_image size (long side == imgsz) with 1..4 normalised xywh boxes each"""
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


class FakeDataset:
    """what Mosaic / MixUp need from BaseDataset (base.py:236-257): buffer, __len__, get_image_and_label"""

    def __init__(self, ims, labels):
        self.ims, self.labels = ims, labels
        self.buffer = list(range(len(ims)))
        self.data = {}
        self.use_keypoints = False

    def __len__(self):
        return len(self.ims)

    def get_image_and_label(self, i):
        im = self.ims[i].copy()
        lab = dict(im_file=f"img{i}.jpg", cls=self.labels[i]["cls"].copy(), img=im, ori_shape=im.shape[:2], resized_shape=im.shape[:2],
                   ratio_pad=(1.0, 1.0))
        lab["instances"] = Instances(self.labels[i]["bboxes"].copy(), np.zeros((0, 1000, 2), dtype=np.float32), None, bbox_format="xywh",
                                     normalized=True)
        return lab


def run_train(seed, imgsz, n_img, picks, **over):
    hyp = SimpleNamespace(**dict(HYP, **over))
    ims, labels = synth_dataset(seed, n_img, imgsz)
    ds = FakeDataset(ims, labels)
    tf = v8_transforms(ds, imgsz, hyp)
    tf.append(Format(bbox_format="xywh", normalize=True, return_mask=False, return_keypoint=False, batch_idx=True, mask_ratio=4,
                     mask_overlap=True))
    out = dict(imgsz=imgsz, n_img=n_img, data_seed=seed, picks=np.array(picks), hyp=np.array([hyp.degrees, hyp.translate, hyp.scale,
                                                                                              hyp.shear, hyp.perspective, hyp.hsv_h, hyp.hsv_s, hyp.hsv_v, hyp.flipud, hyp.fliplr, hyp.mosaic]))
    random.seed(seed + 1)
    np.random.seed(seed + 2)
    samples = []
    for k, idx in enumerate(picks):
        for v in REC.values():
            v.clear()
        s = tf(ds.get_image_and_label(idx))
        w = REC["warp"][0]
        out[f"s{k}_canvas"], out[f"s{k}_M"], out[f"s{k}_dsize"] = w["img"], w["M"], np.array(w["dsize"])
        assert w["border"] == (114, 114, 114) and len(REC["warp"]) == 1 and len(REC["lut"]) == 3
        out[f"s{k}_lut"] = np.stack(REC["lut"])
        out[f"s{k}_cls"], out[f"s{k}_bboxes"] = s["cls"].numpy(), s["bboxes"].numpy()
        out[f"s{k}_img"] = s["img"].numpy()                  # flips + Format (CHW, RGB) of the placeholder pattern
        samples.append(s)
    batch = YOLODataset.collate_fn(samples)
    out["batch_idx"], out["batch_cls"], out["batch_bboxes"] = batch["batch_idx"].numpy(), batch["cls"].numpy(), batch["bboxes"].numpy()
    out["rng_after"] = np.array([random.random(), np.random.uniform()])          # both generators consumed exactly as far as the reference
    return out


def run_val(seed, imgsz, shapes):
    """the validation transform (dataset.py:141 LetterBox(scaleup=False) + Format) on images of several shapes"""
    g = np.random.default_rng(seed)
    out = dict(imgsz=imgsz, shapes=np.array(shapes))
    tf = Compose([LetterBox(new_shape=(imgsz, imgsz), scaleup=False)])
    tf.append(Format(bbox_format="xywh", normalize=True, return_mask=False, return_keypoint=False, batch_idx=True, mask_ratio=4, mask_overlap=True))
    for k, (h, w) in enumerate(shapes):
        for v in REC.values():
            v.clear()
        im = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        bb = np.concatenate((g.uniform(0.3, 0.7, (3, 2)), g.uniform(0.1, 0.4, (3, 2))), 1).astype(np.float32)
        lab = dict(im_file="v.jpg", cls=g.integers(0, 20, (3, 1)).astype(np.float32), img=im, ori_shape=(h, w), resized_shape=(h, w),
                   ratio_pad=(1.0, 1.0), instances=Instances(bb.copy(), np.zeros((0, 1000, 2), dtype=np.float32), None, bbox_format="xywh",
                                                             normalized=True))
        s = tf(lab)
        out[f"v{k}_in_bboxes"], out[f"v{k}_bboxes"] = bb, s["bboxes"].numpy()
        out[f"v{k}_resize"] = np.array(REC["resize"][0]["dsize"] if REC["resize"] else (-1, -1))
        out[f"v{k}_border"] = np.array(REC["border"][0][:4])
        out[f"v{k}_ratio_pad"] = np.array([s["ratio_pad"][0][0], s["ratio_pad"][0][1], s["ratio_pad"][1][0], s["ratio_pad"][1][1]], dtype=np.float64)
    return out


def run_dark(seed):
    """DarkChannel / AtmLight / DarkIcA of the trainer (models/yolo/detect/train.py:42-68; cv2.split / cv2.min are the stand-ins'
    channel views / np.minimum) on small images whose brightest dark-channel values are DISTINCT around the cut, so that numpy's
    unstable argsort has one answer.  Rows 0..2 of DarkIcA are the defined part (the rest of its np.empty buffer is not recorded)."""
    import cv2
    cv2.min = np.minimum
    from ultralytics.models.yolo.detect.train import DetectionTrainer
    tr = DetectionTrainer.__new__(DetectionTrainer)
    g = np.random.default_rng(seed)
    out = {}
    for k, (h, w) in enumerate(((48, 64), (40, 100), (64, 64))):
        im = g.integers(0, 200, (h, w, 3), dtype=np.uint8)
        numpx = max(h * w // 1000, 1)
        pos = g.choice(h * w, numpx + 2, replace=False)                 # distinct bright dark-channel values at the top
        for j, p in enumerate(pos):                                    # pixel p: channel minimum exactly 255 - j, the other two >= it
            px = np.minimum(255, 255 - j + g.integers(0, 3, 3))
            px[int(g.integers(0, 3))] = 255 - j
            im.reshape(-1, 3)[p] = px
        dark = tr.DarkChannel(im)
        A = tr.AtmLight(im, dark)
        ica = tr.DarkIcA(im, A)
        out[f"d{k}_im"], out[f"d{k}_dark"], out[f"d{k}_A"], out[f"d{k}_ica_rows012"] = im, dark, A, np.array(ica[:3])
    return out


def main():
    out = {}
    for tag, res in (("t0", run_train(1301, 64, 6, [0, 3, 5, 1])),
                     ("t1", run_train(1302, 96, 5, [4, 2, 2], degrees=10.0, shear=2.0, flipud=0.5, translate=0.2, scale=0.3)),
                     ("t2", run_train(1305, 64, 6, [0, 1, 2, 3, 4], mosaic=0.0, degrees=5.0)),     # Mosaic's coin fails: LetterBox pre_transform path
                     ("val", run_val(1303, 64, [(48, 64), (64, 40), (64, 64), (30, 50)])),
                     ("dark", run_dark(1304))):
        for k, v in res.items():
            out[f"{tag}_{k}"] = v
    path = os.path.join(HERE, "g13_augment.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
