import sys, os, random
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_augment_cpu import _hyp, synth_dataset
from test_gpu_augment import _oracle_render
from dedark_yolo_amd.data import DeviceAugmenter
from oracle import augment as oa
z = np.load(os.path.join(ROOT, "tests/golden/g13_augment.npz"))
tag = "t1"
imgsz, seed, picks = int(z[f"{tag}_imgsz"]), int(z[f"{tag}_data_seed"]), [int(i) for i in z[f"{tag}_picks"]]
ims, labels = synth_dataset(seed, int(z[f"{tag}_n_img"]), imgsz)
aug = DeviceAugmenter(ims, labels, imgsz, _hyp(z[f"{tag}_hyp"]))
random.seed(seed + 1); np.random.seed(seed + 2)
plans = [aug.plan(i) for i in picks]
for k, p in enumerate(plans):
    for stage in ("warp", "hsv", "full"):
        import copy
        q = copy.copy(p)
        if stage == "warp":
            q.hsv_gains = None; q.flipud = q.fliplr = False
        elif stage == "hsv":
            q.flipud = q.fliplr = False
        got = aug.render([q]).cpu().numpy()[0]
        want = _oracle_render(q, ims)
        d = np.argwhere(got != want)
        print(k, stage, "diffs", len(d), "flipud", p.flipud, "fliplr", p.fliplr)
        for c, y, x in d[:4]:
            print("   at", c, y, x, "got", got[c, y, x], "want", want[c, y, x], "all ch got", got[:, y, x], "want", want[:, y, x])
            if stage != "warp":
                canvas = oa.mosaic4_canvas(p.imgsz, p.rects, [ims[i] for i in p.sources])
                w = oa.cv_warp_affine_linear_u8(canvas, p.M[:2], p.size)
                px = w[y, x]
                hsv = oa.cv_bgr2hsv_u8(px[None, None])[0, 0]
                lh, ls, lv = oa.hsv_luts(p.hsv_gains)
                print("   bgr", px, "hsv", hsv, "after lut", lh[hsv[0]], ls[hsv[1]], lv[hsv[2]])
