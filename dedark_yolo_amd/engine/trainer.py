"""Training harness of the hot path: the caller contract of the reference BaseTrainer / DetectionTrainer
(ultralytics/engine/trainer.py:191-266,292-359,459-467,611-665; ultralytics/models/yolo/detect/train.py:70-111,123-129;
ModelEMA ultralytics/utils/torch_utils.py:344-377) re-designed around flat HBM-resident state:

  * every trainable parameter, its gradient, momentum and EMA live in ONE flat f32 buffer each (module order); the conv /
    BN kernels write gradients straight into it, the optimizer + clip + EMA is one fused kernel (3 launches per step);
  * data parallel: one process per GPU, gradients are SUM-all-reduced over RCCL in layer-aligned buckets that are issued
    as soon as backward has passed the bucket (overlap with the remaining wgrad kernels); loss.py's per-rank normaliser
    and `loss *= world_size` + DDP mean (trainer.py:334-335) == plain sum of per-rank gradients;
  * preprocess_batch's tensor part (uint8 -> /255 -> x^dark_param, recovery mse) is one HIP kernel; the numpy dark-channel
    branch of the reference (train.py:81-97) is dead downstream and reads uninitialised memory: not reproduced.
"""
import contextlib
import math
import os
from pathlib import Path
from types import SimpleNamespace

import torch
import torch.distributed as dist
import yaml

from .. import ops
from .._C import call
from ..nn.modules import DyModule
from ..nn.tasks import DetectionModel
from ..ops import ptr, stream

DEFAULT_CFG_PATH = Path(__file__).resolve().parent.parent / "cfg" / "default.yaml"


def get_cfg(overrides=None):
    with open(DEFAULT_CFG_PATH) as f:
        d = yaml.safe_load(f)
    for k, v in (overrides or {}).items():
        d[k] = v
    return SimpleNamespace(**d)


class FlatState:
    """Flat parameter / gradient / optimizer-state storage for a model (module order, 16-byte aligned slots)."""

    def __init__(self, model, with_ema=True):
        import torch.nn as nn
        params, gids = [], []
        bn_types = tuple(v for k, v in nn.__dict__.items() if "Norm" in k and isinstance(v, type))
        seen = set()
        self.layer_ranges = []                       # (top-level layer index, start, end) in elements
        off = 0
        slots = []
        for li, layer in enumerate(model.model):
            start = off
            for mod in layer.modules():
                for name, p in mod.named_parameters(recurse=False):
                    if not p.requires_grad or id(p) in seen:
                        continue
                    seen.add(id(p))
                    g = 2 if name == "bias" else (1 if isinstance(mod, bn_types) else 0)     # trainer.py:626-634
                    n = p.numel()
                    slots.append((p, off, n, g))
                    off += ops.round_up(n, 4)
            self.layer_ranges.append((li, start, off))
        self.n = off
        dev = next(model.parameters()).device
        f32 = torch.float32
        self.p = torch.zeros(off, dtype=f32, device=dev)
        self.g = torch.zeros(off, dtype=f32, device=dev)
        self.m = torch.zeros(off, dtype=f32, device=dev)
        self.m2 = None
        self.gid = torch.zeros(off, dtype=torch.uint8, device=dev)
        for p, o, n, g in slots:
            self.p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.p[o:o + n].view(p.shape)
            p.grad = self.g[o:o + n].view(p.shape)
            p._dy_direct = True
            self.gid[o:o + n] = g
        self.slots = slots
        self.ema = self.p.clone() if with_ema else None
        # float buffers (BN running statistics) take part in the EMA like in the reference (state_dict based)
        bufs = [b for b in model.buffers() if b.dtype.is_floating_point and b.numel() > 0 and b.is_cuda]
        self.buf_list = bufs
        nb = sum(ops.round_up(b.numel(), 4) for b in bufs)
        self.buf_flat = torch.zeros(nb, dtype=f32, device=dev)
        o = 0
        for b in bufs:
            n = b.numel()
            self.buf_flat[o:o + n].copy_(b.reshape(-1))
            b.data = self.buf_flat[o:o + n].view(b.shape)
            o += ops.round_up(n, 4)
        self.buf_ema = self.buf_flat.clone() if with_ema else None
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        ops.bump_weights_epoch()

    def ema_state_dict(self, model):
        """state_dict of the EMA model (for checkpoints / validation)."""
        sd = {}
        names = {id(p): k for k, p in model.named_parameters()}
        for p, o, n, _ in self.slots:
            sd[names[id(p)]] = self.ema[o:o + n].view(p.shape).clone()
        bnames = {id(b): k for k, b in model.named_buffers()}
        o = 0
        for b in self.buf_list:
            n = b.numel()
            sd[bnames[id(b)]] = self.buf_ema[o:o + n].view(b.shape).clone()
            o += ops.round_up(n, 4)
        return sd


class EarlyStopping:
    """Stops training after `patience` epochs without a fitness improvement (reference ultralytics/utils/torch_utils.py:478-516:
    `>=` keeps the early zero-fitness epochs from counting, `possible_stop` announces the stop one epoch ahead so that the last
    epoch is validated even with val=False)."""

    def __init__(self, patience=50):
        self.best_fitness, self.best_epoch = 0.0, 0
        self.patience = patience or float("inf")
        self.possible_stop = False

    def __call__(self, epoch, fitness):
        if fitness is None:
            return False
        if fitness >= self.best_fitness:
            self.best_epoch, self.best_fitness = epoch, fitness
        delta = epoch - self.best_epoch
        self.possible_stop = delta >= (self.patience - 1)
        return delta >= self.patience


class GradBuckets:
    """Layer-aligned gradient buckets.  The all-reduce(sum) of a bucket is issued from inside the backward pass as soon as
    every parameterised layer of the bucket has produced its gradients (RCCL waits on the compute stream at issue time and
    runs on its own stream, so it overlaps the remaining dgrad / wgrad kernels)."""

    def __init__(self, flat, model, bucket_bytes=None):
        # about six buckets per backward pass (so that all but the last overlap with it), capped at 32 MB: a ring all-reduce over
        # xGMI is per-link bound, so few large messages beat many small ones, but one single bucket cannot overlap at all
        if bucket_bytes is None:
            bucket_bytes = min(32 << 20, max(1 << 20, flat.n * 4 // 6))
        self.flat = flat
        self.works = []
        self.buckets = []          # dict(start, end, layers)
        cur_end, layers = flat.n, []
        for li, s, e in reversed(flat.layer_ranges):
            if e > s:
                layers.append(li)
            if (cur_end - s) * 4 >= bucket_bytes or li == 0:
                if cur_end > s:
                    self.buckets.append(dict(start=s, end=cur_end, layers=set(layers)))
                cur_end, layers = s, []
        self.owner = {li: b for b in self.buckets for li in b["layers"]}
        self._reset()
        for li, layer in enumerate(model.model):
            if li in self.owner:
                layer._dy_after_backward = (lambda idx: (lambda: self._layer_done(idx)))(li)

    def _reset(self):
        for b in self.buckets:
            b["left"] = len(b["layers"])

    def _layer_done(self, li):
        b = self.owner[li]
        b["left"] -= 1
        if b["left"] == 0:
            g = self.flat.g[b["start"]:b["end"]]
            side = ops.wgrad_side_stream() if ops.wgrad_pending() else None
            if side is None:
                self.works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))
            else:
                # the bucket's weight gradients are on the side stream: issue the collective from there (after everything the
                # compute stream has produced so far) instead of stalling the compute stream on a join
                ops._side_wait_main()
                with torch.cuda.stream(side):
                    self.works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        for b in self.buckets:              # a layer that took no part in this backward (should not happen) is reduced here
            if b["left"] > 0:
                self.works.append(dist.all_reduce(self.flat.g[b["start"]:b["end"]], op=dist.ReduceOp.SUM, async_op=True))
        ev = self.exposed_events
        if ev is not None:                  # bench.py --gpus N: how long the compute stream sits in these waits = the part of the
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)     # all-reduce backward did not hide
            e0.record()
        for w in self.works:
            w.wait()                        # (stream-side wait for the collective's stream; the host does not block)
        if ev is not None:
            e1.record()
            ev.append((e0, e1))
        self.works.clear()
        self._reset()

    exposed_events = None                   # set to a list to collect (start, end) events around the waits of every finish()


class DevicePrefetcher:
    """Uploads batch i+1 (host tensors of the dataloader's dict: uint8 images, cls, bboxes, batch_idx) on a copy stream while
    step i runs on the compute stream (first piece).  Tensors that already live on the device pass through."""

    def __init__(self, loader, device):
        self.it = iter(loader)
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self.nxt = None
        self._load()

    def _load(self):
        try:
            b = next(self.it)
        except StopIteration:
            self.nxt = None
            return
        bi = b.get("batch_idx")
        if "n_max" not in b and torch.is_tensor(bi) and not bi.is_cuda and torch.is_tensor(b.get("img")):
            from ..utils.loss import n_max_of
            b = dict(b, n_max=n_max_of(bi, int(b["img"].shape[0])))       # host-side bincount: no device sync in the criterion
        with torch.cuda.stream(self.stream):
            self.nxt = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) and not v.is_cuda else v) for k, v in b.items()}

    def __iter__(self):
        return self

    def __next__(self):
        if self.nxt is None:
            raise StopIteration
        cur = torch.cuda.current_stream()
        cur.wait_stream(self.stream)
        b = self.nxt
        for v in b.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)                 # allocated on the copy stream, consumed (and freed) on the compute stream
        self._load()                                 # the next upload overlaps the step that is about to be issued
        return b


class DetectionTrainer:
    """Minimal, faithful training loop for the detect task.  `train(loader)` consumes batch dicts with the reference schema
    (img uint8 [B,3,H,W] RGB, cls [N,1], bboxes [N,4] normalised xywh, batch_idx [N])."""

    def __init__(self, cfg=None, overrides=None):
        self.args = cfg if cfg is not None else get_cfg(overrides)
        self.rank = int(os.environ.get("RANK", -1))
        self.local_rank = int(os.environ.get("LOCAL_RANK", 0))
        self.world_size = int(os.environ.get("WORLD_SIZE", 1))
        self.device = torch.device("cuda", self.local_rank)
        self.model = None
        self.flat = None
        self.buckets = None
        self.updates = 0
        self.step_count = 0
        self.mse_acc = None
        self.pack_plan = ops.PackPlan()

    # ---------------------------------------------------------------- setup
    def get_model(self, cfg=None, weights=None, nc=None):
        model = DetectionModel(cfg or self.args.model, nc=nc)
        if weights is not None:
            model.load(weights)
        return model

    def setup(self, model=None, nc=None, total_iterations=None):
        torch.cuda.set_device(self.device)
        dt = str(getattr(self.args, "dtype", "fp32"))
        self.compute_dtype = (torch.bfloat16 if dt in ("bf16", "bfloat16") else
                              torch.float16 if dt in ("fp16", "float16", "half") else torch.float32)
        ops.set_compute_dtype(self.compute_dtype)
        # fp16 = the reference's AMP (trainer.py:221 GradScaler): dynamic loss scale kept ON THE DEVICE, {scale, finite steps, skipped steps}
        self.loss_scale = (torch.tensor([65536.0, 0.0, 0.0], dtype=torch.float32, device=self.device)
                           if self.compute_dtype == torch.float16 else None)
        self.model = (model if model is not None else self.get_model(nc=nc)).to(self.device)
        self.model.args = self.args
        self.model.train()
        if self.world_size > 1:
            if not dist.is_initialized():
                dist.init_process_group(os.environ.get("DY_DIST_BACKEND", "nccl" if dist.is_nccl_available() else "gloo"))
            for t in list(self.model.parameters()) + list(self.model.buffers()):       # K3: one broadcast of the start state
                dist.broadcast(t.data, src=0)
        self.flat = FlatState(self.model, with_ema=True)
        # `deterministic` (reference cfg/default.yaml:23, True by default there and here): one stream, kernels in program order.  What
        # stays free is the arrival order of the f64 atomics behind the BatchNorm / loss sums (last-bit effects on the f32 statistics):
        # repeated runs agree to f32 round-off, not bit for bit (tests/test_gpu_trainer.py measures it).  False = the throughput
        # schedule: weight gradients and the coarser Detect levels on side streams.
        det = bool(getattr(self.args, "deterministic", False))
        ops.enable_wgrad_stream(not det)
        ops.enable_branch_streams(not det)
        if self.world_size > 1:
            self.buckets = GradBuckets(self.flat, self.model)
        a = self.args
        name = a.optimizer
        if name == "auto":                                       # trainer.py:617-623
            nc_ = getattr(self.model.model[-1], "nc", 10)
            lr_fit = round(0.002 * 5 / (4 + nc_), 6)
            iters = total_iterations if total_iterations is not None else 1e9
            name, self.lr0, self.momentum = ("SGD", 0.01, 0.9) if iters > 10000 else ("AdamW", lr_fit, 0.9)
            a.warmup_bias_lr = 0.0
        else:
            self.lr0, self.momentum = a.lr0, a.momentum
        self.opt_name = name
        if name == "AdamW":
            self.flat.m2 = torch.zeros_like(self.flat.m)
        self.mse_acc = torch.zeros(1, dtype=torch.float64, device=self.device)
        # trainer.py:248-249: accumulate gradients up to the nominal batch size nbs, scale weight_decay accordingly
        nbs, bs = float(getattr(a, "nbs", 64)), max(int(getattr(a, "batch", 64)), 1)
        self.accumulate = max(round(nbs / bs), 1)
        self.weight_decay = a.weight_decay * bs * self.accumulate / nbs
        self.last_opt_step = -1
        self.acc_count = 0            # backward passes accumulated in flat.g_acc since the last optimizer step
        return self

    # ---------------------------------------------------------------- batch
    def preprocess_batch(self, batch):
        """reference detect/train.py:70-111 (tensor part): clean_img, img (darkened), recovery_loss_batch."""
        a = self.args
        img = batch["img"]
        if img.dtype != torch.uint8:
            raise RuntimeError("preprocess_batch expects the dataloader's uint8 image tensor")
        bi = batch.get("batch_idx")
        if "n_max" not in batch and torch.is_tensor(bi) and not bi.is_cuda:
            from ..utils.loss import n_max_of
            batch["n_max"] = n_max_of(bi, int(img.shape[0]))
        img = img.to(self.device, non_blocking=True).contiguous()
        n = img.numel()
        out = torch.empty(img.shape, dtype=torch.float32, device=self.device)
        low, ded = bool(getattr(a, "lowlight_FLAG", False)), bool(getattr(a, "dedark_FLAG", False))
        both = low and ded
        clean = out if both else torch.empty_like(out)
        self.mse_acc.zero_()
        call("dy_preprocess_batch", ptr(img), ptr(out), None if both else ptr(clean), float(a.dark_param), int(low), int(ded),
             ptr(self.mse_acc), n, stream())
        batch["img"] = out
        batch["clean_img"] = clean
        batch["recovery_loss_batch"] = (self.mse_acc / n).float().reshape(())
        if both and getattr(a, "dark_channel_prior", False):
            # detect/train.py:81-97: dedark_A / IcA of the darkened batch.  The reference computes them per image on the host (a
            # device->host copy + a Python loop) and nothing downstream reads them in train mode, so the device version
            # is opt-in (args.dark_channel_prior); it is deterministic where the reference's is not (data/augment.py).
            from ..data.augment import dark_channel_prior
            batch["dedark_A"], batch["IcA"] = dark_channel_prior(out)
        return batch

    # ---------------------------------------------------------------- step
    def lr_factors(self, ni, nw, epoch, epochs):
        """warm-up interpolation of trainer.py:318-327 + linear / cosine schedule (:255-259)."""
        a = self.args
        lf = ((1 - math.cos(epoch * math.pi / epochs)) / 2) * (a.lrf - 1) + 1 if a.cos_lr else (1 - epoch / epochs) * (1.0 - a.lrf) + a.lrf
        lr = [self.lr0 * lf] * 3
        mom = self.momentum
        if ni <= nw:
            xi = ni / max(nw, 1)
            lr = [self.lr0 * lf * xi, self.lr0 * lf * xi, a.warmup_bias_lr + (self.lr0 * lf - a.warmup_bias_lr) * xi]
            mom = a.warmup_momentum + (self.momentum - a.warmup_momentum) * xi
        return lr, mom

    def optimizer_step(self, lr, mom):
        """clip_grad_norm_(10.0) + optimizer.step() + zero_grad() + ema.update() (trainer.py:459-467) in 2 kernels."""
        f = self.flat
        st = stream()
        g = f.g
        if self.acc_count > 0:                  # gradients of the earlier backward passes wait in g_acc
            call("dy_grad_accumulate", ptr(f.g_acc), ptr(f.g), f.n, st)
            g = f.g_acc
        f.sumsq.zero_()
        call("dy_sumsq", ptr(g), f.n, ptr(f.sumsq), st)
        self.updates += 1
        d = 0.9999 * (1 - math.exp(-self.updates / 2000))                       # torch_utils.py:357
        wd = self.weight_decay
        ls = getattr(self, "loss_scale", None)      # fp16: unscale inside the step, skip it on overflow, then GradScaler.update
        if self.opt_name == "AdamW":
            call("dy_adamw_step_scaled", ptr(f.p), ptr(g), ptr(f.m), ptr(f.m2), ptr(f.ema), ptr(f.gid), lr[0], lr[1], lr[2], wd, 0.0, 0.0,
                 mom, 0.999, 1e-8, self.updates, d, ptr(f.sumsq), 10.0, 1.0, ptr(ls), f.n, st)
        else:
            call("dy_sgd_step_scaled", ptr(f.p), ptr(g), ptr(f.m), ptr(f.ema), ptr(f.gid), lr[0], lr[1], lr[2], wd, 0.0, 0.0, mom, 1, d,
                 ptr(f.sumsq), 10.0, 1.0, ptr(ls), f.n, st)
        if ls is not None:
            call("dy_loss_scale_update", ptr(ls), ptr(f.sumsq), 2.0, 0.5, 2000, st)
        if f.buf_ema is not None and f.buf_flat.numel():
            call("dy_ema_lerp", ptr(f.buf_ema), ptr(f.buf_flat), d, f.buf_flat.numel(), st)
        if self.acc_count > 0:
            f.g_acc.zero_()
            self.acc_count = 0
        ops.bump_weights_epoch()
        self.pack_plan.repack(self.model)

    def accumulate_gradients(self):
        """Keep this backward pass's gradients for a later optimizer step (the conv kernels overwrite flat.g)."""
        f = self.flat
        if getattr(f, "g_acc", None) is None:
            f.g_acc = torch.zeros_like(f.g)
        call("dy_grad_accumulate", ptr(f.g_acc), ptr(f.g), f.n, stream())
        self.acc_count += 1

    def train_step(self, batch, lr=None, mom=None, step_optimizer=True):
        """preprocess + forward + loss + backward (+ bucketed all-reduce) + optimizer/EMA. Returns (loss, loss_items).
        step_optimizer=False only accumulates the gradients (trainer.py:340-342)."""
        batch = self.preprocess_batch(batch)
        loss, items = self.model(batch)
        if getattr(self, "loss_scale", None) is not None:
            (loss * self.loss_scale[0]).backward()      # scaler.scale(loss).backward() (trainer.py:340)
        else:
            loss.backward()
        ops.wgrad_join()
        if self.buckets is not None:
            self.buckets.finish()
        if lr is None:
            lr, mom = [self.lr0] * 3, self.momentum
        if step_optimizer:
            self.optimizer_step(lr, mom)
        else:
            self.accumulate_gradients()
        self.step_count += 1
        return loss.detach(), items

    # ---------------------------------------------------------------- checkpoints (the reference's own format, both ways)
    def _param_order(self):
        """The reference optimizer's parameter numbering (build_optimizer, U/engine/trainer.py:626-660): torch numbers the
        parameters group by group -- group 0 = biases, group 1 = decayed weights, group 2 = BatchNorm weights -- each in
        model.named_modules() order.  Returns [(flat offset, numel, shape)] in that numbering and the three group sizes."""
        f = self.flat
        order = [[], [], []]
        for p, o, n, g in f.slots:                            # FlatState walks the modules in the same order with the same rule
            order[{2: 0, 0: 1, 1: 2}[g]].append((o, n, tuple(p.shape)))
        return order[0] + order[1] + order[2], [len(x) for x in order]

    def optimizer_state_dict(self):
        """torch.optim-style state_dict of the fused optimizer -- what `optimizer.state_dict()` is in the reference's checkpoint
        (trainer.py:423) and what its resume hands to `optimizer.load_state_dict` (:591): per-parameter momentum_buffer (SGD) or
        step / exp_avg / exp_avg_sq (AdamW) + the three param_groups."""
        f = self.flat
        flat, sizes = self._param_order()
        m, m2 = f.m.detach().cpu(), None if f.m2 is None else f.m2.detach().cpu()
        state = {}
        for i, (o, n, shape) in enumerate(flat):
            if self.opt_name == "AdamW":
                state[i] = dict(step=torch.tensor(float(self.updates)), exp_avg=m[o:o + n].view(shape).clone(),
                                exp_avg_sq=m2[o:o + n].view(shape).clone())
            else:
                state[i] = dict(momentum_buffer=m[o:o + n].view(shape).clone())
        groups, k = [], 0
        for gi, cnt in enumerate(sizes):
            common = dict(lr=float(self.lr0), initial_lr=float(self.lr0), weight_decay=float(self.weight_decay) if gi == 1 else 0.0,
                          maximize=False, foreach=None, differentiable=False, params=list(range(k, k + cnt)))
            if self.opt_name == "AdamW":
                common.update(betas=(float(self.momentum), 0.999), eps=1e-8, amsgrad=False, capturable=False, fused=None)
            else:
                common.update(momentum=float(self.momentum), dampening=0, nesterov=True, fused=None)
            groups.append(common)
            k += cnt
        return dict(state=state, param_groups=groups)

    def load_optimizer_state_dict(self, opt):
        """Inverse of optimizer_state_dict; also reads the optimizer entry of a checkpoint the REFERENCE wrote."""
        f = self.flat
        flat, sizes = self._param_order()
        groups = opt.get("param_groups", [])
        if [len(g["params"]) for g in groups] != sizes:
            raise RuntimeError(f"resume: optimizer groups {[len(g['params']) for g in groups]} do not match this model's {sizes}")
        adam = any("exp_avg" in st for st in opt["state"].values())
        if adam != (self.opt_name == "AdamW"):
            raise RuntimeError(f"resume: checkpoint optimizer state does not belong to {self.opt_name}")
        f.m.zero_()
        if f.m2 is not None:
            f.m2.zero_()
        for i, (o, n, shape) in enumerate(flat):
            st = opt["state"].get(i)
            if not st:
                continue                                      # torch creates state lazily: a parameter that never stepped has none
            if adam:
                f.m[o:o + n].copy_(st["exp_avg"].reshape(-1).float())
                f.m2[o:o + n].copy_(st["exp_avg_sq"].reshape(-1).float())
            elif st.get("momentum_buffer") is not None:
                f.m[o:o + n].copy_(st["momentum_buffer"].reshape(-1).float())

    def save_model(self, wdir, epoch=0, fitness=None):
        """trainer.save_model (U/engine/trainer.py:408-433): last.pt always, best.pt when `fitness` is the best so far -- in the
        reference's own on-disk format: `model` / `ema` are pickled half-precision DetectionModel OBJECTS under the reference's
        class paths (utils/checkpoint.py:save_reference_checkpoint; the reference's attempt_load_one_weight loads them, verified by
        tests/golden/make_ckpt_interop.py), `optimizer` is a torch.optim-style state_dict in the reference's parameter numbering,
        `train_args` the argument dict.  `dy_state` carries what the reference has no slot for (fp16 loss scale, step counters)."""
        from ..utils.checkpoint import save_reference_checkpoint
        os.makedirs(wdir, exist_ok=True)
        f = self.flat
        best = getattr(self, "best_fitness", None)
        if fitness is not None and (best is None or fitness >= best):
            self.best_fitness = best = fitness
        ops.flush_bn_counters()
        ema_sd = dict(self.model.state_dict())
        ema_sd.update(f.ema_state_dict(self.model))           # EMA parameters + float buffers; integer buffers / frozen DFL as they are
        ta = {k: v for k, v in vars(self.args).items() if isinstance(v, (int, float, str, bool, type(None), list, tuple))}
        ls = getattr(self, "loss_scale", None)
        extra = dict(dy_state=dict(opt_name=self.opt_name, step_count=int(self.step_count), last_opt_step=int(self.last_opt_step),
                                   loss_scale=None if ls is None else [float(v) for v in ls.cpu()], version="dedark_yolo_amd-2"))
        last = os.path.join(wdir, "last.pt")
        save_reference_checkpoint(last, self.model, ema_state=ema_sd, epoch=int(epoch), best_fitness=best, updates=int(self.updates),
                                  optimizer=self.optimizer_state_dict(), train_args=ta, extra=extra)
        if fitness is not None and best == fitness:
            import shutil
            shutil.copyfile(last, os.path.join(wdir, "best.pt"))
        return last

    def resume_training(self, ckpt):
        """Restores parameters, EMA, optimizer state and counters from last.pt -- written by save_model above OR by the reference's
        trainer (pickled module objects; read without the reference package by utils/checkpoint.py) -- and returns the epoch to
        continue with (`U/engine/trainer.py:580-609`: start_epoch = ckpt['epoch'] + 1; the half-precision weights are converted
        back with .float() exactly like the reference does)."""
        from ..utils.checkpoint import load_checkpoint
        ck = load_checkpoint(ckpt) if isinstance(ckpt, (str, os.PathLike)) else ckpt
        f = self.flat
        msd = ck.model_sd if ck.model_sd is not None else ck.state_dict
        esd = ck.state_dict                                   # `ema` when present, else `model` (reference precedence)
        names = {id(p): k for k, p in self.model.named_parameters()}
        for p, o, n, _ in f.slots:
            k = names[id(p)]
            f.p[o:o + n].copy_(msd[k].float().reshape(-1).to(f.p.device))
            f.ema[o:o + n].copy_(esd[k].float().reshape(-1).to(f.p.device))
        bnames = {id(b): k for k, b in self.model.named_buffers()}
        o = 0
        for b in f.buf_list:
            n, k = b.numel(), bnames[id(b)]
            f.buf_flat[o:o + n].copy_(msd[k].float().reshape(-1).to(f.p.device))
            f.buf_ema[o:o + n].copy_(esd[k].float().reshape(-1).to(f.p.device))
            o += ops.round_up(n, 4)
        for k, b in self.model.named_buffers():                     # integer buffers (num_batches_tracked)
            if not b.dtype.is_floating_point and k in msd:
                b.copy_(msd[k].to(b.device))
        if ck.optimizer is not None:
            self.load_optimizer_state_dict(ck.optimizer)
        self.best_fitness = ck.best_fitness
        self.updates = int(ck.updates or 0)
        dy_state = getattr(ck, "dy_state", None) or {}
        self.step_count = int(dy_state.get("step_count", 0))
        self.last_opt_step = int(dy_state.get("last_opt_step", -1))
        ls = dy_state.get("loss_scale")
        if ls is not None and getattr(self, "loss_scale", None) is not None:      # GradScaler state: {scale, finite steps so far, skipped}
            self.loss_scale.copy_(torch.tensor((list(ls) + [0.0])[:3], dtype=torch.float32))
        ops.bump_weights_epoch()
        self.pack_plan.repack(self.model)
        return int(ck.epoch if ck.epoch is not None else -1) + 1

    @contextlib.contextmanager
    def ema_weights(self):
        """The model on its EMA parameters and buffers, in place (the parameters are views of the flat buffers): what the reference
        validates and checkpoints (`trainer.ema.ema`, engine/validator.py:104).  Swapped back on exit."""
        f = self.flat
        if f.ema is None:
            yield self.model
            return
        has_buf = f.buf_ema is not None and f.buf_flat.numel() > 0
        keep_p, keep_b = f.p.clone(), (f.buf_flat.clone() if has_buf else None)
        try:
            f.p.copy_(f.ema)
            if has_buf:
                f.buf_flat.copy_(f.buf_ema)
            ops.bump_weights_epoch()
            yield self.model
        finally:
            f.p.copy_(keep_p)
            if has_buf:
                f.buf_flat.copy_(keep_b)
            ops.bump_weights_epoch()
            self.pack_plan.repack(self.model)                 # one launch; a lazy re-pack would cost one per layer in the next step

    def validate(self, val_loader):
        """trainer.validate() (engine/trainer.py:471-480): the validator on the EMA weights, forced to fp32 as the fork does
        (engine/validator.py:105-107 `self.args.half = False`).  Returns (metrics dict, fitness)."""
        from .validator import DetectionValidator
        dt = ops.get_compute_dtype()
        try:
            with self.ema_weights() as model:
                metrics = DetectionValidator(self.args)(model, val_loader, dtype=torch.float32)
        finally:
            ops.set_compute_dtype(dt)
            self.model.train()
        fitness = metrics.get("fitness")
        if fitness is None:
            fitness = 0.1 * metrics.get("metrics/mAP50(B)", 0.0) + 0.9 * metrics.get("metrics/mAP50-95(B)", 0.0)
        return metrics, float(fitness)

    def _batches(self, loader):
        return DevicePrefetcher(loader, self.device)

    def train(self, loader, epochs=None, val_loader=None, save_dir=None, start_epoch=0):
        """The epoch loop of BaseTrainer._do_train (engine/trainer.py:300-400) around train_step: warm-up / accumulation schedule,
        and at the end of every epoch on rank 0 -- validation on the EMA weights when `val_loader` is given and (args.val or last
        epoch), fitness / best tracking, last.pt / best.pt when `save_dir` is given and (args.save or last epoch)
        (:366-380, 408-433); then the stop flag of EarlyStopping(args.patience) is broadcast from rank 0 (:389-395) -- the
        collective doubles as the epoch-boundary barrier: the other ranks wait HERE, on the host, for rank 0's validation / save,
        not inside the first gradient all-reduce of the next epoch where a long validation would run into the RCCL watchdog.
        `start_epoch`: resume_training's return value.  Returns the per-epoch loss items; self.metrics / self.fitness /
        self.best_fitness / self.stop hold the rest."""
        epochs = epochs or self.args.epochs
        nb = len(loader)
        nw = max(round(self.args.warmup_epochs * nb), 100) if self.args.warmup_epochs > 0 else -1
        history = []
        self.metrics, self.fitness = {}, None
        nbs, bs = float(getattr(self.args, "nbs", 64)), max(int(getattr(self.args, "batch", 64)), 1)
        self.stopper = getattr(self, "stopper", None) or EarlyStopping(getattr(self.args, "patience", 50))
        self.stop = False
        for epoch in range(start_epoch, epochs):
            for i, batch in enumerate(self._batches(loader)):
                ni = i + nb * epoch
                lr, mom = self.lr_factors(ni, nw, epoch, epochs)
                if ni <= nw:                                   # trainer.py:320-322: accumulate ramps from 1 to nbs / batch
                    xi = ni / max(nw, 1)
                    self.accumulate = max(1, round(1 + xi * (nbs / bs - 1)))
                step = ni - self.last_opt_step >= self.accumulate        # trainer.py:340-342
                loss, items = self.train_step(batch, lr, mom, step_optimizer=step)
                if step:
                    self.last_opt_step = ni
            history.append([float(v) for v in items])
            if self.rank in (-1, 0):
                final_epoch = epoch + 1 == epochs or self.stopper.possible_stop
                if val_loader is not None and (getattr(self.args, "val", True) or final_epoch):
                    self.metrics, self.fitness = self.validate(val_loader)
                self.stop = self.stopper(epoch + 1, self.fitness)
                if save_dir is not None and (getattr(self.args, "save", True) or epoch + 1 == epochs):
                    self.save_model(os.path.join(str(save_dir), "weights"), epoch=epoch, fitness=self.fitness)
            if self.world_size > 1 and dist.is_initialized():         # K5: broadcast_object_list([stop], 0) (trainer.py:389-393)
                flag = [self.stop if self.rank == 0 else None]
                dist.broadcast_object_list(flag, 0)
                self.stop = bool(flag[0])
            if self.stop:
                break                                                 # every rank leaves the loop in the same epoch
        return history
