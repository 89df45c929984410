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


def model_parity_case(yaml_name, scale, scale_def, seed, S, B, nbox, dtype=torch.float32, with_oracle=True, fp64=False, keep_grads=False):
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
    if keep_grads:
        out["grads"] = {k: p.grad.detach().float().cpu().clone() for k, p in named.items() if p.grad is not None}
    if with_oracle:
        for k, v in sd.items():
            v.requires_grad_(v.is_floating_point() and v.ndim > 0 and ".dfl." not in k and "running_" not in k)
        maps = om.forward(plan, save, sd, batch["img"], True)
        strides = [float(S // m.shape[2]) for m in maps]
        ol, oi = oloss.recovery_detection_loss(maps, batch, strides, 20, oloss.default_hyp())
        ol.backward()
        out["oracle_loss"] = float(ol)
        out["oracle_items"] = [float(v) for v in oi]
        ref64 = None
        if fp64:      # conditioning probe: the same oracle in float64 tells how much fp32 rounding alone moves each gradient
            sd64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v.clone())
                    for k, v in om.rng_fill(om.param_shapes(plan), seed).items()}
            for k, v in sd64.items():
                if v.is_floating_point():
                    v.requires_grad_(v.ndim > 0 and ".dfl." not in k and "running_" not in k)
            b64 = dict(batch)
            b64["img"] = batch["img"].double()
            m64 = om.forward(plan, save, sd64, b64["img"], True)
            l64, _ = oloss.recovery_detection_loss(m64, b64, strides, 20, oloss.default_hyp())
            l64.backward()
            ref64 = {k: v.grad for k, v in sd64.items() if v.is_floating_point() and v.grad is not None}
        errs = []
        for k, p in named.items():
            if p.grad is None or sd[k].grad is None:
                continue
            g, r = p.grad.detach().float().cpu(), sd[k].grad
            tgt = ref64[k].float() if ref64 is not None and k in ref64 else r
            den = float(tgt.norm())
            e = float((g - tgt).norm()) / den if den > 1e-10 else float((g - tgt).norm())
            eo = (float((r - tgt).norm()) / den if den > 1e-10 else float((r - tgt).norm())) if ref64 is not None else 0.0
            errs.append((e, eo, k))
        # direction of the step: cosine between the concatenated gradients (all tensors / without the front-end, whose gradient has
        # passed through the whole network AND the image filters) and per tensor
        trip = {k: (float((p.grad.detach().double().cpu() * sd[k].grad.double()).sum()), float(p.grad.detach().double().norm()) ** 2,
                    float(sd[k].grad.double().norm()) ** 2) for k, p in named.items() if p.grad is not None and sd[k].grad is not None}

        def cos(keys):
            d = [trip[k] for k in keys]
            return sum(t[0] for t in d) / max((sum(t[1] for t in d) * sum(t[2] for t in d)) ** 0.5, 1e-300)
        out["grad_cosine"] = cos(trip)
        out["grad_cosine_net"] = cos([k for k in trip if not k.startswith("model.0.")] or list(trip))
        per = sorted(t[0] / max((t[1] * t[2]) ** 0.5, 1e-300) for t in trip.values())
        out["grad_cosine_median"] = per[len(per) // 2]
        out["grad_cosine_p10"] = per[len(per) // 10]
        out["top_grad_norms"] = sorted(((round(t[2] ** 0.5, 3), k) for k, t in trip.items()), reverse=True)[:4]
        errs.sort(reverse=True)
        out["worst_grad_rel"] = errs[0][0]
        out["worst_grad_key"] = errs[0][2]
        out["worst5"] = [(round(e, 5), round(eo, 5), k) for e, eo, k in errs[:5]]
        out["median_grad_rel"] = errs[len(errs) // 2][0]
        if ref64 is not None:
            out["worst_excess_over_oracle32"] = max(e / max(eo, 1e-4) for e, eo, _ in errs)
        # running statistics after the step
        msd = model.state_dict()
        rs = 0.0
        for k in msd:
            if k.endswith("running_var") or k.endswith("running_mean"):
                rs = max(rs, float((msd[k].cpu() - sd[k]).abs().max()))
        out["worst_running_stat_abs"] = rs
    return out
