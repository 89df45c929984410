"""Batches of the device input pipeline for the training loop: what `build_dataloader` + the dataloader workers are in the
reference (ultralytics/data/build.py:72-109, dataset.py:171-188), with the decoded images kept in device memory.

`DeviceAugmentLoader` owns the decoded uint8 images.  `resident=True` (the MI355X-sized default: 288 GB of HBM holds a VOC-scale dataset
several times over -- 16.5 k images x ~0.9 MB) keeps them in HBM, so a step reads nothing over PCIe; `resident=False` keeps them in pinned
host memory and uploads the source images of batch i+1 on a copy stream while step i runs (the PCIe-inclusive mode bench.py reports).
Either way the augmented pixels are produced on the device by ONE launch per batch (DeviceAugmenter.render)."""
import random as _random

import numpy as np
import torch

from .augment import AugmentHyp, DeviceAugmenter, plan_train_sample, train_labels


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
        self.uploaded_bytes = 0

    def __len__(self):
        n = len(self.shapes)
        return n // self.bs if self.drop_last else -(-n // self.bs)

    def _plans(self, indices):
        buf = list(range(len(self.shapes)))
        return [plan_train_sample(i, self.shapes, buf, self.imgsz, self.hyp, self.rnd, self.nprnd) for i in indices]

    RING = 3                                                  # staging buffers in flight (pinned host memory is expensive to allocate)

    def _slots(self):
        from .augment import descriptor_bytes
        if getattr(self, "_ring", None) is None:
            cap = 64 * self.bs                                   # label rows per batch the pinned buffer holds (grown on demand)
            self._ring = [dict(desc=torch.empty(self.bs * descriptor_bytes(), dtype=torch.uint8).pin_memory(),
                               lab=torch.empty((cap, 6), dtype=torch.float32).pin_memory(), ev=torch.cuda.Event()) for _ in range(self.RING)]
            self._turn = 0
            self.stream = torch.cuda.Stream(device=self.device)
        slot = self._ring[self._turn % self.RING]
        self._turn += 1
        slot["ev"].synchronize()                                 # its last copies (three batches ago) have long finished
        return slot

    def _prepare(self, indices):
        """One whole batch -- plans, label bookkeeping, descriptor / label / (host mode) image uploads from pinned memory, the render
        launch -- on the loader's own stream, without a host synchronisation: it is issued while the previous training step is still
        running on the compute stream and shares the GPU with it.  Returns (batch dict of device tensors, event)."""
        plans = self._plans(indices)
        lab = [train_labels(p, self.labels, self.shapes) for p in plans]
        n = sum(len(c) for c, _ in lab)
        slot = self._slots()
        if n > slot["lab"].shape[0]:
            slot["lab"] = torch.empty((2 * n, 6), dtype=torch.float32).pin_memory()
        rows = slot["lab"][:n].numpy()
        o = 0
        for i, (c, b) in enumerate(lab):
            m = len(c)
            rows[o:o + m, 0], rows[o:o + m, 1], rows[o:o + m, 2:6] = i, c.reshape(-1), b
            o += m
        with torch.cuda.stream(self.stream):
            if self.resident:
                aug = self.aug
            else:
                need = sorted({s for p in plans for s in p.sources})
                dev = {s: self.host[s].to(self.device, non_blocking=True) for s in need}
                self.uploaded_bytes += sum(self.host[s].numel() for s in need)
                aug = DeviceAugmenter.__new__(DeviceAugmenter)           # a view of the uploaded subset with the full index space
                aug.device, aug.imgsz, aug.hyp, aug.images = self.device, self.imgsz, self.hyp, _Sparse(dev)
            img = aug.render(plans, staging=slot["desc"])
            labd = slot["lab"][:n].to(self.device, non_blocking=True)
            slot["ev"].record(self.stream)
        batch = dict(img=img, batch_idx=labd[:, 0], cls=labd[:, 1:2], bboxes=labd[:, 2:6], n_max=max([len(c) for c, _ in lab] + [0]))
        return batch, slot["ev"], (img, labd)

    def __iter__(self):
        order = list(range(len(self.shapes)))
        if self.shuffle:
            self.rnd.shuffle(order)
        chunks = [order[i:i + self.bs] for i in range(0, len(order), self.bs)]
        if self.drop_last:
            chunks = [c for c in chunks if len(c) == self.bs]
        nxt = self._prepare(chunks[0]) if chunks else None
        for k in range(len(chunks)):
            batch, ev, tensors = nxt
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for t in tensors:
                t.record_stream(cur)                             # allocated on the loader stream, consumed (and freed) on the compute stream
            # batch k+1 is prepared when the consumer comes back for it, i.e. right after step k has been ISSUED: the host work and the
            # uploads / render on the loader stream overlap step k on the GPU
            yield batch
            nxt = self._prepare(chunks[k + 1]) if k + 1 < len(chunks) else None


class _Sparse:
    """list-like over the uploaded subset of the dataset"""

    def __init__(self, d):
        self.d = d

    def __getitem__(self, i):
        return self.d[i]
