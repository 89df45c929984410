#!/usr/bin/env python3
"""Times dy_conv2d_wgrad on the YOLOv8-n (C2) layer shapes; used to tune the split heuristics (env DY_WGRAD_BLOCKS / _STEPS)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dedark_yolo_amd import ops
from dedark_yolo_amd._C import call
from dedark_yolo_amd.ops import ptr, stream, ld_of
shapes = [(32, 64, 64, 40, 3, 1), (32, 32, 32, 80, 3, 1), (32, 64, 64, 80, 3, 1), (32, 16, 16, 160, 3, 1), (32, 3, 16, 640, 3, 2),
          (32, 16, 32, 320, 3, 2), (32, 128, 128, 20, 3, 1), (32, 128, 64, 40, 3, 1), (32, 192, 128, 40, 1, 1), (32, 384, 256, 20, 1, 1),
          (32, 64, 128, 80, 3, 2), (32, 128, 256, 40, 3, 2), (32, 256, 256, 20, 1, 1), (32, 128, 128, 40, 1, 1), (32, 64, 64, 80, 1, 1),
          (32, 96, 64, 80, 1, 1), (32, 64, 20, 80, 1, 1), (32, 128, 64, 20, 3, 1), (32, 256, 64, 20, 3, 1)]
dt = torch.bfloat16
scratch = ops.wgrad_scratch(torch.device("cuda"))
tot = 0.0
for B, Ci, Co, H, k, s in shapes:
    p = k // 2
    Ho = (H + 2 * p - k) // s + 1
    cip, cop = ops.round_up(Ci, 8), ops.round_up(Co, 8)
    x = ops.as_nhwc(torch.randn(B, Ci, H, H, device="cuda"), dt)
    dz = ops.as_nhwc(torch.randn(B, Co, Ho, Ho, device="cuda"), dt)
    g = torch.empty(Co, Ci, k, k, device="cuda")
    def run():
        call("dy_conv2d_wgrad", ptr(x), ld_of(x), B, H, H, cip, ptr(dz), ld_of(dz), Ho, Ho, cop, k, k, s, p, 1, Co, Ci, ptr(scratch),
             scratch.numel(), ptr(g), 1, stream())
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tot += us
    print(f"{Ci:4d}->{Co:4d} k{k} s{s} @{H:3d}: {us:8.1f} us")
print(f"sum {tot:.1f} us")
