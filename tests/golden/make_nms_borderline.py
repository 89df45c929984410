"""Writes tests/golden/g11_nms_borderline.npz: box pairs that separate f32 NMS arithmetic (torchvision's CPU kernel) from f64.

Needs neither the reference nor a GPU (pure numpy search, ~2 min): `python tests/golden/make_nms_borderline.py`.  The fixture is
DATA for tests/test_oracle_golden.py (the oracle must follow the f32 rule and differ from its own f64 evaluation on it) and for
tests/test_gpu_val.py (the HIP kernel must follow the f32 rule)."""
import os

import numpy as np
import torch


def nms_borderline_pred(seed, nc=20, pairs=24, thr=0.7, S=640.0):
    """One image [1, 4+nc, 2*pairs] of (keeper, victim) box pairs whose IoU -- on the f32 class-offset xyxy boxes the reference
    hands torchvision.ops.nms (U/utils/ops.py:236,259-261) -- falls on OPPOSITE sides of `thr` when evaluated in f32 (torchvision's
    CPU kernel, scalar_t = float) and in f64.  Found by search: for a random keeper and class, slide the victim's centre x over
    consecutive f32 values around the f64 crossing and keep the first value where the two evaluations disagree.  Every pair sits in
    its own class, so the keep set of the image is the union of the pairs' outcomes.  Returns (pred, n_f32_suppresses, n_f64_suppresses)."""
    g = np.random.default_rng(seed)
    f = np.float32

    def boxes(x, y, w, h, c):
        x, y, w, h = f(x), f(y), f(w), f(h)
        hw, hh = f(w / f(2)), f(h / f(2))
        off = f(f(c) * f(7680))
        return np.array([f(f(x - hw) + off), f(f(y - hh) + off), f(f(x + hw) + off), f(f(y + hh) + off)], dtype=f)

    def iou(bi, bj, ft):
        bi, bj = bi.astype(ft), bj.astype(ft)
        ai = ft((bi[2] - bi[0]) * (bi[3] - bi[1]))
        aj = ft((bj[2] - bj[0]) * (bj[3] - bj[1]))
        w = max(ft(0), ft(min(bi[2], bj[2]) - max(bi[0], bj[0])))
        h = max(ft(0), ft(min(bi[3], bj[3]) - max(bi[1], bj[1])))
        inter = ft(w * h)
        return ft(inter / ft(ft(ai + aj) - inter))

    cols, used = [], set()
    want = {True: pairs - pairs // 2, False: pairs // 2}       # key: "f32 suppresses" (and f64 does not) / the converse
    tries = 0
    while (want[True] or want[False]) and tries < 20000:
        tries += 1
        c = int(g.integers(1, nc))                           # class >= 1: the offset makes the f32 coordinates coarse (ulp 2^-11..2^-7)
        x, y = g.uniform(0.2 * S, 0.8 * S, 2)
        w, h = g.uniform(0.1 * S, 0.3 * S, 2)
        bi = boxes(x, y, w, h, c)
        if tries % 2:                                        # same size, shifted along x
            w2, h2, y2 = w, h, y
        else:                                                # slightly different size and row
            w2, h2, y2 = w * g.uniform(0.9, 1.1), h * g.uniform(0.9, 1.1), y + g.uniform(-0.03, 0.03) * h
        lo, hi = 0.0, float(w)                               # f64 crossing of thr by bisection on the shift
        if not iou(bi, boxes(x, y2, w2, h2, c), np.float64) > thr:
            continue
        for _ in range(60):
            mid = (lo + hi) / 2
            if iou(bi, boxes(x + mid, y2, w2, h2, c), np.float64) > thr:
                lo = mid
            else:
                hi = mid
        x0 = f(x + lo)
        xs, a, b = [x0], x0, x0
        for _ in range(200):
            a = np.nextafter(a, f(-np.inf)); b = np.nextafter(b, f(np.inf))
            xs += [a, b]
        for xv in xs:
            bj = boxes(xv, y2, w2, h2, c)
            s32 = bool(np.float64(iou(bi, bj, np.float32)) > thr)
            s64 = bool(iou(bi, bj, np.float64) > thr)
            if s32 != s64 and want[s32]:
                want[s32] -= 1
                hi_s, lo_s = g.uniform(0.8, 0.95), g.uniform(0.3, 0.6)
                # pairs of one class must not interact: park each pair in its own corner of the class plane via the score order only
                # (keeper first, its victim second; other pairs of the class are far away or harmless: checked by the caller's oracle)
                for (bx, by, bw, bh, sc) in ((f(x), f(y), f(w), f(h), hi_s), (xv, f(y2), f(w2), f(h2), lo_s)):
                    col = np.zeros(4 + nc, dtype=f)
                    col[:4] = (bx, by, bw, bh)
                    col[4 + c] = sc
                    cols.append(col)
                break
    n32, n64 = pairs - pairs // 2, pairs // 2
    assert len(cols) == 2 * pairs, f"borderline search found only {len(cols) // 2} pairs"
    pred = np.stack(cols, 1)[None]
    return torch.from_numpy(np.ascontiguousarray(pred)), n32, n64


if __name__ == "__main__":
    out = {}
    for k, seed in enumerate((3, 11)):
        pred, n32, n64 = nms_borderline_pred(seed)
        out[f"pred{k}"] = pred.numpy()
        out[f"n32_{k}"], out[f"n64_{k}"] = n32, n64
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g11_nms_borderline.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)
