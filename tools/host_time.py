#!/usr/bin/env python3
"""Host enqueue time against GPU time of the training step: python tools/host_time.py [--model ...] [--batch 32].
Prints the wall time to ISSUE n steps (no synchronisation) and the wall time until the GPU has finished them."""
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
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    from dedark_yolo_amd.engine.trainer import DetectionTrainer, get_cfg
    from dedark_yolo_amd.nn.tasks import DetectionModel
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    cfg = get_cfg(dict(model=a.model, dtype="bf16", optimizer="SGD", batch=a.batch, imgsz=640, lowlight_FLAG=True, dedark_FLAG=True))
    tr = DetectionTrainer(cfg)
    tr.setup(DetectionModel(a.model, nc=20))
    batches = [bench.synth_batch(1234 + i, a.batch, 640, 20, "cuda") for i in range(2)]

    def step(i):
        b = dict(batches[i % 2])
        tr.args.dark_param = b["gamma"]
        tr.train_step(b)

    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"issue {1e3 * (t1 - t0) / a.steps:.3f} ms/step   complete {1e3 * (t2 - t0) / a.steps:.3f} ms/step")
    iss, comp = [], []
    for i in range(10):                      # one step at a time from an idle GPU: the queue depth cannot hide the host time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        iss.append(t1 - t0)
        comp.append(t2 - t0)
    print(f"single step from idle: issue {1e3 * sorted(iss)[5]:.3f} ms   complete {1e3 * sorted(comp)[5]:.3f} ms")
    # phases of the host time (from an idle GPU, nothing blocks)
    ph = {"preprocess": [], "forward+loss": [], "backward": [], "optimizer": []}
    for i in range(10):
        torch.cuda.synchronize()
        b = dict(batches[i % 2])
        tr.args.dark_param = b["gamma"]
        t0 = time.perf_counter()
        b = tr.preprocess_batch(b)
        t1 = time.perf_counter()
        loss, items = tr.model(b)
        t2 = time.perf_counter()
        loss.backward()
        from dedark_yolo_amd import ops
        ops.wgrad_join()
        t3 = time.perf_counter()
        tr.optimizer_step([tr.lr0] * 3, tr.momentum)
        t4 = time.perf_counter()
        for k, v in zip(ph, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            ph[k].append(v)
    print("host phases (median): " + "  ".join(f"{k} {1e3 * sorted(v)[5]:.3f} ms" for k, v in ph.items()))
    # forward / backward split of the host time
    import cProfile
    import pstats
    # the autograd engine runs a CUDA graph's backward on its own thread, which cProfile does not see: keep it on this one
    torch.autograd.set_multithreading_enabled(False)
    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(5):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
    pstats.Stats(pr).sort_stats("tottime").print_stats(45)


if __name__ == "__main__":
    main()
