"""Shared helpers for the tests (golden loading, yaml plans, synthetic batches)."""
import os

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CFG = os.path.join(ROOT, "dedark_yolo_amd", "cfg", "models", "v8")


def gold(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: (torch.from_numpy(np.ascontiguousarray(z[k])) if z[k].dtype.kind in "fiub" else z[k]) for k in z.files}


def load_yaml(name):
    with open(os.path.join(CFG, name)) as f:
        return yaml.safe_load(f)


def rnd(seed, *shape, lo=0.0, hi=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((lo + (hi - lo) * g.random(shape, dtype=np.float32)).astype(np.float32))


def make_batch(seed, B, S, nbox, nc=20):
    """Same law as tests/golden/make_golden.py::make_batch (kept in sync by the golden tests)."""
    g = np.random.default_rng(seed)
    img = torch.from_numpy(g.random((B, 3, S, S), dtype=np.float32))
    bi, cls, bb = [], [], []
    for b in range(B):
        for _ in range(nbox[b]):
            bi.append(b)
            cls.append(int(g.integers(0, nc)))
            cx, cy = g.uniform(0.25, 0.75, 2)
            w, h = g.uniform(0.15, 0.5, 2)
            bb.append([cx, cy, w, h])
    return dict(img=img, batch_idx=torch.tensor(bi, dtype=torch.float32),
                cls=torch.tensor(cls, dtype=torch.float32).view(-1, 1),
                bboxes=torch.tensor(bb, dtype=torch.float32).view(-1, 4))


def close(a, b, rtol=1e-4, atol=1e-5, what=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    if a.numel() == 1 and b.numel() == 1:
        a, b = a.reshape(()), b.reshape(())
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max abs err {float(err.max()):.3e} " \
                          f"(ref max {float(b.abs().max()):.3e})"

