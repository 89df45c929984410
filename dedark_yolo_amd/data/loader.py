"""Batches of the device input pipeline for the training loop: what `build_dataloader` + the dataloader workers are in the
reference (ultralytics/data/build.py:72-109, dataset.py:171-188), with the decoded images kept in device memory.

`DeviceAugmentLoader` owns the decoded uint8 images.  `resident=True` (the MI355X-sized default: 288 GB of HBM holds a VOC-scale dataset
several times over -- 16.5 k images x ~0.9 MB) keeps them in HBM, so a step reads nothing over PCIe; `resident=False` keeps them in pinned
host memory and uploads the source images of batch i+1 on a copy stream while step i runs (the PCIe-inclusive mode bench.py reports).
Either way the augmented pixels are produced on the device by ONE launch per batch (DeviceAugmenter.render)."""
import random as _random

import numpy as np
import torch

from .augment import AugmentHyp, DeviceAugmenter, collate, plan_train_sample, train_labels


class DeviceAugmentLoader:
    def __init__(self, images, labels, imgsz, batch_size, hyp=None, device="cuda", resident=True, seed=0, shuffle=True, drop_last=True):
        self.device = torch.device(device)
        self.imgsz, self.bs, self.hyp = int(imgsz), int(batch_size), hyp or AugmentHyp()
        self.labels = labels
        self.resident = bool(resident)
        self.shapes = [(int(im.shape[0]), int(im.shape[1])) for im in images]
        self.shuffle, self.drop_last = shuffle, drop_last
        self.rnd = _random.Random(seed)                      # own generators: the loader must not disturb the caller's global RNG state
        self.nprnd = np.random.RandomState(seed + 1)
        if self.resident:
            self.aug = DeviceAugmenter(images, labels, imgsz, self.hyp, device)
            self.host = None
        else:
            self.host = [(im if torch.is_tensor(im) else torch.from_numpy(np.ascontiguousarray(im))).pin_memory() for im in images]
            self.aug = None
            self.copy_stream = torch.cuda.Stream(device=self.device)
        self.uploaded_bytes = 0

    def __len__(self):
        n = len(self.shapes)
        return n // self.bs if self.drop_last else -(-n // self.bs)

    def _plans(self, indices):
        buf = list(range(len(self.shapes)))
        return [plan_train_sample(i, self.shapes, buf, self.imgsz, self.hyp, self.rnd, self.nprnd) for i in indices]

    def _stage(self, indices):
        """plans + (non-resident) the upload of every source image the batch touches, on the copy stream"""
        plans = self._plans(indices)
        if self.resident:
            return plans, None
        need = sorted({s for p in plans for s in p.sources})
        with torch.cuda.stream(self.copy_stream):
            dev = {s: self.host[s].to(self.device, non_blocking=True) for s in need}
        self.uploaded_bytes += sum(self.host[s].numel() for s in need)
        return plans, dev

    def _finish(self, indices, plans, dev):
        if self.resident:
            img = self.aug.render(plans)
        else:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self.copy_stream)
            for t in dev.values():
                t.record_stream(cur)
            tmp = DeviceAugmenter.__new__(DeviceAugmenter)           # a view of the uploaded subset with the full index space
            tmp.device, tmp.imgsz, tmp.hyp = self.device, self.imgsz, self.hyp
            tmp.images = _Sparse(dev)
            img = DeviceAugmenter.render(tmp, plans)
        lab = [train_labels(p, self.labels, self.shapes) for p in plans]
        bi, cls, bb = collate(lab)
        return dict(img=img, batch_idx=bi, cls=cls, bboxes=bb, n_max=max([len(c) for c, _ in lab] + [0]))

    def __iter__(self):
        order = list(range(len(self.shapes)))
        if self.shuffle:
            self.rnd.shuffle(order)
        chunks = [order[i:i + self.bs] for i in range(0, len(order), self.bs)]
        if self.drop_last:
            chunks = [c for c in chunks if len(c) == self.bs]
        staged = self._stage(chunks[0]) if chunks else None
        for k, idx in enumerate(chunks):
            plans, dev = staged
            staged = self._stage(chunks[k + 1]) if k + 1 < len(chunks) else None      # upload of the next batch overlaps this step
            yield self._finish(idx, plans, dev)


class _Sparse:
    """list-like over the uploaded subset of the dataset"""

    def __init__(self, d):
        self.d = d

    def __getitem__(self, i):
        return self.d[i]
