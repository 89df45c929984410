#!/usr/bin/env python3
"""Experiment: capture one whole training step (preprocess + forward + loss + backward + optimizer) in a HIP graph through
torch.cuda.graph and replay it.  By-value scalars (lr, EMA decay, gamma) are frozen at capture time, so this is a timing probe,
not a trainer.  python tools/graph_probe.py [--model ...] [--batch 32]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="yolov8n-lowlight.yaml")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    from dedark_yolo_amd.engine.trainer import DetectionTrainer, get_cfg
    from dedark_yolo_amd.nn.tasks import DetectionModel
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    cfg = get_cfg(dict(model=a.model, dtype="bf16", optimizer="SGD", batch=a.batch, imgsz=640, lowlight_FLAG=True, dedark_FLAG=True))
    tr = DetectionTrainer(cfg)
    tr.setup(DetectionModel(a.model, nc=20))
    static = bench.synth_batch(1234, a.batch, 640, 20, "cuda")
    tr.args.dark_param = static["gamma"]

    def step():
        return tr.train_step(dict(static))

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(5):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / a.steps
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"eager {1e3 * eager:.3f} ms/step   graph replay: issue {1e3 * (t1 - t0) / a.steps:.3f}  complete {1e3 * (t2 - t0) / a.steps:.3f} ms/step   "
          f"loss {float(out[0]):.4f}")


if __name__ == "__main__":
    main()
