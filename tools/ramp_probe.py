#!/usr/bin/env python3
"""Step time against time since process start (does the device need a sustained load before it reaches its clocks?)."""
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
batches = [bench.synth_batch(1234 + i, 32, 640, 20, "cuda") for i in range(2)]
t_start = time.perf_counter()
for blk in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        b = dict(batches[i % 2])
        tr.args.dark_param = b["gamma"]
        tr.train_step(b)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"t={t1 - t_start:6.2f}s  {1e3 * (t1 - t0) / 20:.3f} ms/step", flush=True)
