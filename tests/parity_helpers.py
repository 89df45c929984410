"""Helpers shared by the GPU parity tests, __graft_entry__.smoke() and bench.py's checker: run the HIP product path and the
CPU oracle on the same seeded inputs.  (Test infrastructure: the only place product and oracle meet.)"""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from util import load_yaml, make_batch  # noqa: E402

HYP = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5, lrl=2.0)


def load_sd(module, sd, prefix=""):
    """Copy an oracle-style flat dict into a product module (strict on names and shapes)."""
    own = module.state_dict()
    missing = [k for k in own if prefix + k not in sd]
    extra = [k for k in sd if k.startswith(prefix) and k[len(prefix):] not in own]
    assert not missing and not extra, f"state_dict mismatch: missing {missing[:5]} extra {extra[:5]}"
    module.load_state_dict({k: sd[prefix + k].detach().clone() for k in own}, strict=True)
    return module


def set_bn(module):
    import torch.nn as nn
    for m in module.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    return module


def build_models(yaml_name, scale, scale_def, seed, nc=20, device="cuda"):
    """(product DetectionModel on GPU, oracle plan/save/sd) with identical rng_fill weights."""
    from dedark_yolo_amd.nn.tasks import DetectionModel
    from oracle import model as om
    cfg = load_yaml(yaml_name)
    if scale_def is not None:
        cfg["scales"][scale] = list(scale_def)
    cfg["scale"] = scale
    plan, save = om.build_plan(cfg, scale=scale, nc=nc)
    sd = om.rng_fill(om.param_shapes(plan), seed)
    model = DetectionModel(dict(cfg), ch=3, nc=nc)
    model.args = HYP
    load_sd(model, sd)
    model = model.to(device)
    return model, (plan, save, sd)


def model_parity_case(yaml_name, scale, scale_def, seed, S, B, nbox, dtype=torch.float32, with_oracle=True):
    """One training step (forward + loss + backward) on the HIP path and on the oracle; returns scalars + grad errors."""
    import dedark_yolo_amd as dy
    from oracle import loss as oloss
    from oracle import model as om
    dy.set_compute_dtype(dtype)
    model, (plan, save, sd) = build_models(yaml_name, scale, scale_def, seed)
    batch = make_batch(seed + 1, B, S, nbox)
    batch["img"] = batch["img"].pow(3.0)
    batch["recovery_loss_batch"] = torch.tensor(0.0123)
    gb = dict(batch)
    gb["img"] = batch["img"].cuda()
    gb["recovery_loss_batch"] = batch["recovery_loss_batch"].cuda()
    model.train()
    loss, items = model(gb)
    loss.backward()
    torch.cuda.synchronize()
    out = dict(loss=float(loss), items=[float(v) for v in items])
    named = dict(model.named_parameters())
    out["grad_finite"] = all(bool(torch.isfinite(p.grad).all()) for p in named.values() if p.grad is not None)
    out["n_nograd"] = sum(1 for p in named.values() if p.requires_grad and p.grad is None)
    if with_oracle:
        for k, v in sd.items():
            v.requires_grad_(v.is_floating_point() and v.ndim > 0 and ".dfl." not in k and "running_" not in k)
        maps = om.forward(plan, save, sd, batch["img"], True)
        strides = [float(S // m.shape[2]) for m in maps]
        ol, oi = oloss.recovery_detection_loss(maps, batch, strides, 20, oloss.default_hyp())
        ol.backward()
        out["oracle_loss"] = float(ol)
        out["oracle_items"] = [float(v) for v in oi]
        worst, worst_k = 0.0, None
        for k, p in named.items():
            if p.grad is None or sd[k].grad is None:
                continue
            g, r = p.grad.detach().float().cpu(), sd[k].grad
            e = float((g - r).norm() / (r.norm() + 1e-12)) if float(r.norm()) > 1e-10 else float((g - r).norm())
            if e > worst:
                worst, worst_k = e, k
        out["worst_grad_rel"] = worst
        out["worst_grad_key"] = worst_k
        # running statistics after the step
        msd = model.state_dict()
        rs = 0.0
        for k in msd:
            if k.endswith("running_var") or k.endswith("running_mean"):
                rs = max(rs, float((msd[k].cpu() - sd[k]).abs().max()))
        out["worst_running_stat_abs"] = rs
    return out
