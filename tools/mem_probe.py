#!/usr/bin/env python3
"""Debug aid: allocator history around the saved input of model.26.cv2.2.0 (who else got that memory, and what sits right before it)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dedark_yolo_amd import ops
from parity_helpers import build_models, make_batch
import dedark_yolo_amd as dy
dy.set_compute_dtype(torch.float32)
ops.enable_branch_streams(True)
model, (plan, save, sd) = build_models("yolov8.yaml", "l", None, 404)
batch = make_batch(405, 4, 128, [3, 2, 5, 1])
batch["img"] = batch["img"].pow(3.0)
batch["recovery_loss_batch"] = torch.tensor(0.0123)
gb = dict(batch); gb["img"] = batch["img"].cuda(); gb["recovery_loss_batch"] = batch["recovery_loss_batch"].cuda()
model.train()
target = dict(model.named_parameters())["model.26.cv2.2.0.conv.weight"]
cap = []
orig_call = ops.call
def spy(name, *a):
    r = orig_call(name, *a)
    if name in ("dy_conv2d_wgrad", "dy_conv2d_wgrad_forked"):
        loc = sys._getframe(1).f_locals
        ctx = loc.get("ctx")
        if ctx is not None and ctx.weight is target:
            cap.append((loc["x"].data_ptr(), loc["x"].numel() * loc["x"].element_size(), loc["x"].untyped_storage().data_ptr(), loc["x"].untyped_storage().nbytes()))
    return r
ops.call = spy
torch.cuda.memory._record_memory_history(max_entries=200000, context="all", stacks="python")
loss, items = model(gb)
torch.cuda.synchronize()
mark = len(torch.cuda.memory._snapshot()["device_traces"][0])
loss.backward()
torch.cuda.synchronize()
snap = torch.cuda.memory._snapshot()
tr = snap["device_traces"][0]
xptr, xbytes, sptr, sbytes = cap[0]
print("x ptr %#x bytes %d storage %#x bytes %d; forward events %d, total %d" % (xptr, xbytes, sptr, sbytes, mark, len(tr)))
def site(e):
    fr = [f for f in e.get("frames", []) if "dedark_yolo_amd" in f["filename"] or "tests" in f["filename"] or "tools" in f["filename"]]
    return " < ".join("%s:%d:%s" % (os.path.basename(f["filename"]), f["line"], f["name"]) for f in fr[:4])
lo, hi = sptr - 65536, sptr + sbytes
for i, e in enumerate(tr):
    a, s = e.get("addr", 0), e.get("size", 0)
    if a < hi and a + s > lo and e["action"] in ("alloc", "free_requested", "free_completed", "free"):
        rel = "INSIDE" if (a < sptr + sbytes and a + s > sptr) else "before"
        print("%5d %s %-14s addr %#x (+%d) size %d stream %#x  %s  %s" % (i, "F" if i < mark else "B", e["action"], a, a - sptr, s, e.get("stream", 0), rel, site(e)))
