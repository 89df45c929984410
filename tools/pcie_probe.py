#!/usr/bin/env python3
"""PCIe-inclusive rate of the C2 step: the batch dict lives in (pinned) host memory, as a dataloader hands it over, and every
step uploads its uint8 images (39 MB at B = 32) and targets before preprocess_batch.  bench.py's `value` has them resident."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dedark_yolo_amd.engine.trainer import DetectionTrainer, get_cfg  # noqa: E402
from dedark_yolo_amd.nn.tasks import DetectionModel  # noqa: E402

torch.cuda.set_device(0)
torch.manual_seed(0)
cfg = get_cfg(dict(model="yolov8n-lowlight.yaml", dtype="bf16", optimizer="SGD", batch=32, imgsz=640, lowlight_FLAG=True, dedark_FLAG=True))
tr = DetectionTrainer(cfg)
tr.setup(DetectionModel("yolov8n-lowlight.yaml", nc=20))
dev_batches = [bench.synth_batch(1234 + i, 32, 640, 20, "cuda") for i in range(2)]


def to_host(b, pin):
    out = {}
    for k, v in b.items():
        if torch.is_tensor(v):
            v = v.cpu()
            out[k] = v.pin_memory() if pin else v
        else:
            out[k] = v
    return out


for pin in (True, False):
    host = [to_host(b, pin) for b in dev_batches]
    for mode, src in (("resident", dev_batches), ("pinned host" if pin else "pageable host", host)):
        for i in range(10):
            b = dict(src[i % 2])
            tr.args.dark_param = b["gamma"]
            tr.train_step(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 60
        for i in range(n):
            b = dict(src[i % 2])
            tr.args.dark_param = b["gamma"]
            tr.train_step(b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{mode:14s}: {1e3 * dt:7.3f} ms/step  {32 / dt:8.1f} img/s", flush=True)
    if pin:                                           # the trainer's one-ahead upload on a copy stream (engine/trainer.py)
        from dedark_yolo_amd.engine.trainer import DevicePrefetcher
        n = 60
        src = [host[i % 2] for i in range(n + 10)]
        it = DevicePrefetcher(src, tr.device)
        for i, b in enumerate(it):
            if i == 10:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            b = dict(b)
            tr.args.dark_param = b["gamma"]
            tr.train_step(b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{'pinned + prefetch':14s}: {1e3 * dt:7.3f} ms/step  {32 / dt:8.1f} img/s", flush=True)
    if not pin:
        break
