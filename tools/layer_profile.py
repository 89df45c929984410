#!/usr/bin/env python3
"""Per-call breakdown of one training step: every C-ABI entry timed with events on the launch stream, grouped by
(entry point, layer shape).  python tools/layer_profile.py [--model yolov8n-lowlight.yaml] [--batch 32] [--top 60]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="yolov8n-lowlight.yaml")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    from dedark_yolo_amd import _C
    from dedark_yolo_amd.engine.trainer import DetectionTrainer, get_cfg
    from dedark_yolo_amd.nn.tasks import DetectionModel
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    cfg = get_cfg(dict(model=a.model, dtype=a.dtype, optimizer="SGD", batch=a.batch, imgsz=a.imgsz, lowlight_FLAG=True, dedark_FLAG=True))
    tr = DetectionTrainer(cfg)
    tr.setup(DetectionModel(a.model, nc=20))
    batches = [bench.synth_batch(1234 + i, a.batch, a.imgsz, 20, "cuda") for i in range(2)]
    for i in range(3):
        b = dict(batches[i % 2])
        tr.args.dark_param = b["gamma"]
        tr.train_step(b)
    torch.cuda.synchronize()
    from dedark_yolo_amd import ops
    ops.enable_wgrad_stream(False)          # one stream: per-call durations are only meaningful without co-running kernels
    ops.enable_branch_streams(False)
    _C._prof = []
    for i in range(a.steps):
        b = dict(batches[i % 2])
        tr.args.dark_param = b["gamma"]
        tr.train_step(b)
    torch.cuda.synchronize()
    rec, _C._prof = _C._prof, None
    agg = {}
    for name, e0, e1, meta, kern in rec:
        key = (name, (meta or {}).get("shape", ""))
        v = agg.setdefault(key, [0.0, 0, 0.0, 0.0, set()])
        if kern:
            v[4].add(kern.split("+")[0])
        v[0] += e0.elapsed_time(e1)
        v[1] += 1
        if meta:
            v[2] += meta.get("flops", 0.0)
            v[3] += meta.get("bytes", 0.0)
    tot = sum(v[0] for v in agg.values())
    print(f"total event time {tot / a.steps:.3f} ms/step over {sum(v[1] for v in agg.values()) // a.steps} calls/step")
    for (name, shape), v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
        ms = v[0] / a.steps
        us = 1e3 * v[0] / v[1]
        tf = v[2] / v[1] / (us * 1e-6) / 1e12 if v[2] else 0.0
        gbs = v[3] / v[1] / (us * 1e-6) / 1e9 if v[3] else 0.0
        print(f"{ms:8.3f} ms/step {v[1] // a.steps:4d}x {us:9.1f} us  {tf:7.1f} TF {gbs:7.0f} GB/s  {name:26s} {shape}  [{', '.join(sorted(v[4]))}]")


if __name__ == "__main__":
    main()
