"""GPU tests of the device input pipeline (csrc/augment.hip) through the C-ABI: every kernel bit-exact against
oracle/augment.py (numpy restatement of the reference's numpy code -- pinned by g13_augment.npz in tests/test_augment_cpu.py -- and of
the OpenCV routines it calls, which are absent from the image: parity unpinned for those), the reference's own mosaic canvases from the
fixture pushed through the kernel, the batch dict against the reference's collated labels, and a training step on a device-augmented batch."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from test_augment_cpu import _hyp, synth_dataset  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g13():
    return np.load(os.path.join(ROOT, "tests", "golden", "g13_augment.npz"))


def _oracle_render(plan, ims):
    """the reference's chain on the host: canvas -> warpAffine -> RandomHSV -> flips -> Format (oracle/augment.py)"""
    from oracle import augment as oa
    if plan.mosaic:
        canvas = oa.mosaic4_canvas(plan.imgsz, plan.rects, [ims[i] for i in plan.sources])
    else:
        canvas = oa.letterbox(ims[plan.sources[0]], (plan.imgsz, plan.imgsz))
    img = oa.cv_warp_affine_linear_u8(canvas, plan.M[:2], plan.size)
    if plan.hsv_gains is not None:
        img = oa.random_hsv(img, plan.hsv_gains)
    if plan.flipud:
        img = np.flipud(img)
    if plan.fliplr:
        img = np.fliplr(img)
    return oa.format_img(img)


@pytest.mark.parametrize("tag", ["t0", "t1"])
def test_mosaic_warp_batch_vs_oracle_and_reference_labels(g13, tag):
    from dedark_yolo_amd.data import DeviceAugmenter
    z = g13
    imgsz, seed, picks = int(z[f"{tag}_imgsz"]), int(z[f"{tag}_data_seed"]), [int(i) for i in z[f"{tag}_picks"]]
    ims, labels = synth_dataset(seed, int(z[f"{tag}_n_img"]), imgsz)
    aug = DeviceAugmenter(ims, labels, imgsz, _hyp(z[f"{tag}_hyp"]))
    random.seed(seed + 1)
    np.random.seed(seed + 2)
    plans = [aug.plan(i) for i in picks]
    got = aug.render(plans).cpu().numpy()
    assert got.shape == (len(picks), 3, imgsz, imgsz) and got.dtype == np.uint8
    for k, p in enumerate(plans):
        want = _oracle_render(p, ims)
        assert np.array_equal(got[k], want), f"{tag} sample {k}: {int((got[k] != want).sum())} of {want.size} bytes differ"
    # the whole batch dict in one call, same seeds: labels are the reference's collated ones
    random.seed(seed + 1)
    np.random.seed(seed + 2)
    b = aug.batch(picks)
    assert np.array_equal(b["img"].cpu().numpy(), got)
    assert np.array_equal(b["batch_idx"].numpy(), z[f"{tag}_batch_idx"]) and np.array_equal(b["bboxes"].numpy(), z[f"{tag}_batch_bboxes"])
    assert b["n_max"] == max(np.bincount(z[f"{tag}_batch_idx"].astype(int)))


def test_letterbox_path_and_hsv_off():
    """mosaic probability 0: RandomPerspective's LetterBox pre_transform (augment.py:767) -> one placement rectangle; no HSV gains"""
    from dedark_yolo_amd.data import AugmentHyp, DeviceAugmenter
    ims, labels = synth_dataset(77, 4, 64)
    aug = DeviceAugmenter(ims, labels, 64, AugmentHyp(mosaic=0.0, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0, degrees=5.0, fliplr=1.0))
    random.seed(3)
    np.random.seed(4)
    plans = [aug.plan(i) for i in range(4)]
    assert not any(p.mosaic for p in plans) and all(p.fliplr for p in plans)
    got = aug.render(plans).cpu().numpy()
    for k, p in enumerate(plans):
        assert np.array_equal(got[k], _oracle_render(p, ims)), k


@pytest.mark.parametrize("shape,imgsz", [((480, 640), 640), ((375, 500), 640), ((1080, 810), 640), ((64, 64), 96), ((700, 333), 320)])
def test_load_resize_and_letterbox_vs_oracle(shape, imgsz):
    """load_image's resize (base.py:152-157) and the validation LetterBox + Format on decoded images of dataset-like shapes"""
    from dedark_yolo_amd.data import augment as A
    from oracle import augment as oa
    g = np.random.default_rng(shape[0] + shape[1])
    im = g.integers(0, 256, shape + (3,), dtype=np.uint8)
    d = torch.from_numpy(im).cuda()
    r = A.load_resize(d, imgsz)
    tgt = oa.load_resize_shape(shape[0], shape[1], imgsz)
    want = im if tgt is None else oa.cv_resize_linear_u8(im, tgt)
    assert np.array_equal(r.cpu().numpy(), want)
    out, geos = A.letterbox_batch([r], imgsz, scaleup=False)
    assert np.array_equal(out[0].cpu().numpy(), oa.format_img(oa.letterbox(want, (imgsz, imgsz), scaleup=False)))
    small = torch.from_numpy(np.ascontiguousarray(im[:shape[0] // 3, :shape[1] // 3])).cuda()       # scaleup=True really enlarges
    out2, _ = A.letterbox_batch([small], imgsz, scaleup=True)
    assert np.array_equal(out2[0].cpu().numpy(), oa.format_img(oa.letterbox(small.cpu().numpy(), (imgsz, imgsz), scaleup=True)))


def test_dark_channel_prior_vs_oracle_and_reference(g13):
    """dy_dark_channel_prior against oracle/augment.py on the fixture's images (where the oracle equals the trainer's own methods:
    test_augment_cpu.py) and on random images full of ties (the oracle's documented tie rule)."""
    from dedark_yolo_amd.data.augment import dark_channel_prior
    from oracle import augment as oa
    z = g13
    ims = [z[f"dark_d{k}_im"] for k in range(3)]
    g = np.random.default_rng(9)
    ims += [g.integers(0, 256, (70, 90, 3), dtype=np.uint8), g.integers(100, 104, (64, 64, 3), dtype=np.uint8), np.full((40, 50, 3), 200, np.uint8)]
    for k, im in enumerate(ims):
        x = torch.from_numpy(np.ascontiguousarray(im.transpose(2, 0, 1))).float().div(255.0)
        # the trainer's own quantisation of the float batch (train.py:81): exactly the uint8 image again
        assert np.array_equal((x.permute(1, 2, 0).numpy() * 255).astype(np.uint8), im)
        A, ica = dark_channel_prior(x[None].cuda())
        dark = oa.dark_channel(im)
        wantA = oa.atm_light(im, dark)
        wantI, _ = oa.dark_ica(im, wantA)
        assert np.array_equal(A.cpu().numpy().astype(np.float64), wantA.astype(np.float32).astype(np.float64)), k
        assert np.array_equal(ica[0, 0].cpu().numpy(), wantI.astype(np.float32)), k
        if k < 3:
            assert np.array_equal(wantA, z[f"dark_d{k}_A"]) and np.array_equal(wantI[:3], z[f"dark_d{k}_ica_rows012"])


def test_full_size_batch_properties_and_train_step():
    """B = 16 mosaics at 640x640 from 640-long images: every output byte either comes from a source image / the grey border (checked
    through the oracle on two samples, shape and range on all), a second render of the same plans is identical, and the batch feeds a
    training step (preprocess_batch -> model -> loss) with the opt-in device dark-channel prior."""
    import dedark_yolo_amd as dy
    from dedark_yolo_amd.data import DeviceAugmenter
    from dedark_yolo_amd.engine.trainer import DetectionTrainer, get_cfg
    from dedark_yolo_amd.nn.tasks import DetectionModel
    g = np.random.default_rng(21)
    ims, labels = [], []
    for i in range(24):
        h, w = (640, int(g.integers(400, 641))) if i % 2 else (int(g.integers(400, 641)), 640)
        ims.append(g.integers(0, 256, (h, w, 3), dtype=np.uint8))
        k = int(g.integers(1, 6))
        labels.append(dict(cls=g.integers(0, 20, (k, 1)).astype(np.float32),
                           bboxes=np.concatenate((g.uniform(0.2, 0.8, (k, 2)), g.uniform(0.1, 0.5, (k, 2))), 1).astype(np.float32)))
    aug = DeviceAugmenter(ims, labels, 640)
    random.seed(11)
    np.random.seed(12)
    plans = [aug.plan(i) for i in range(16)]
    a = aug.render(plans)
    b = aug.render(plans)
    assert a.shape == (16, 3, 640, 640) and torch.equal(a, b)
    for k in (0, 9):
        assert np.array_equal(a[k].cpu().numpy(), _oracle_render(plans[k], ims)), k
    random.seed(11)
    np.random.seed(12)
    batch = aug.batch(list(range(16)))
    assert float(batch["bboxes"].min()) >= 0 and float(batch["bboxes"].max()) <= 1 and batch["img"].dtype == torch.uint8
    try:
        cfg = get_cfg(dict(model="yolov8n-lowlight.yaml", dtype="bf16", optimizer="SGD", batch=16, imgsz=640, lowlight_FLAG=True, dedark_FLAG=True,
                           dark_channel_prior=True, dark_param=2.0))
        tr = DetectionTrainer(cfg)
        tr.setup(DetectionModel("yolov8n-lowlight.yaml", nc=20))
        pb = tr.preprocess_batch(dict(batch))
        assert pb["dedark_A"].shape == (16, 3) and pb["IcA"].shape == (16, 1, 640, 640) and torch.isfinite(pb["IcA"]).all()
        loss, items = tr.train_step(dict(batch))
        assert np.isfinite(float(loss))
    finally:
        dy.set_compute_dtype(torch.float32)


def test_loader_resident_and_host_modes_agree():
    """DeviceAugmentLoader: the decoded dataset in HBM (resident) or in pinned host memory with the next batch's source images uploaded on
    a copy stream -- same seeds, same batches, bit for bit; the host mode really moves bytes; the caller's global RNG state is untouched."""
    from dedark_yolo_amd.data import DeviceAugmentLoader
    ims, labels = synth_dataset(5, 12, 96)
    random.seed(123)
    before = random.getstate()
    a = list(DeviceAugmentLoader(ims, labels, 96, 4, resident=True, seed=7))
    lh = DeviceAugmentLoader(ims, labels, 96, 4, resident=False, seed=7)
    b = list(lh)
    torch.cuda.synchronize()
    assert len(a) == len(b) == 3 and lh.uploaded_bytes > 0 and random.getstate() == before
    for x, y in zip(a, b):
        assert torch.equal(x["img"], y["img"]) and torch.equal(x["bboxes"], y["bboxes"]) and torch.equal(x["batch_idx"], y["batch_idx"])
        assert x["img"].shape == (4, 3, 96, 96) and x["n_max"] == y["n_max"]
