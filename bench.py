#!/usr/bin/env python3
"""Headline benchmark of the Dedark-YOLO hot path on MI355X (BASELINE.json metric: training img/s at 640x640).

  python bench.py --gpus N --steps K --warmup W
      N > 1 without WORLD_SIZE in the environment: bench.py starts `python -m torch.distributed.run` itself (one rank per GPU, a
      child process created BEFORE anything touches the GPU) and exits with the child's code; the driver's own
      `python -m torch.distributed.run ... bench.py --gpus N` works the same way.

A step = one training pass of the hot path over one synthetic batch that is ALREADY RESIDENT IN HBM:
preprocess_batch (uint8 -> /255 -> x^gamma, recovery mse) -> lowlight_recovery -> YOLOv8 backbone / ASFF neck / Detect ->
RcoveryDetectionLoss + TaskAlignedAssigner -> backward -> (N>1: bucketed RCCL all-reduce overlapped with backward) ->
fused clip + SGD-nesterov + EMA.

Default workload = BASELINE configs[2] (C3, the configuration the metric's target is quoted on): repo yolov8.yaml at scale L
(lowlight_recovery + ASFF), 640x640, bf16, batch 64 per GPU, gamma ~ U(5,10), nc=20; weak scaling (per-GPU batch fixed).
`--model yolov8n-lowlight.yaml --batch 32` = configs[1] (C2); `--imgsz 1280 --batch 16 --dtype fp16` = configs[4] (C5).

Rank 0 prints ONE JSON line with the contract fields plus
  * `roofline`: ONE GPU kernel symbol -- the one with the largest share of the step -- timed live with events on the launch
    stream over extra instrumented steps after the timed region (its launches, its own algorithmic flops / bytes), next to the
    per-kernel breakdown; `traffic` = HBM-side bytes per launch of that symbol from the committed rocprofv3 PMC passes of this
    command (profiles/, file named in the line);
  * `cpu_baseline` (N=1 only): the CPU oracle (our port of the reference path, oracle/) timed on this box's host cores as
    BASELINE.md section 4 prescribes (physical cores stated; the benched model and the C1 model; 3 warm-up + 10 timed, median).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy rate)
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 / fp16 MFMA peak
MFMA_F32_PEAK_TF = 157.3


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)      # C3: ~80 ms per step
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--model", default="yolov8l.yaml")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--pipeline", default="resident", choices=["resident", "device", "host"],
                    help="resident: pre-staged batches in HBM (the contract's timed region, default); device: decoded dataset in HBM, "
                         "mosaic / affine / HSV / flip rendered on the device every step; host: decoded dataset in pinned host memory, "
                         "source images uploaded every step (PCIe-inclusive), rendered on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` on its own: start the N ranks as children of a fresh torch.distributed.run process.  Nothing in
    this process has touched the GPU yet (no torch.cuda call), so no exec / fork of a GPU-initialised process happens."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def synth_batch(seed, B, S, nc, device):
    """SURVEY 8(d) synthetic law: clean ~ U[0,1) stored as uint8, 1..8 boxes/image, cls ~ U{0..nc-1}, centres U(.2,.8),
    sizes U(.05,.35) clipped to the image; gamma ~ U(5,10) per batch.  Reference batch schema only (img, batch_idx, cls, bboxes)
    plus the darkening exponent of the batch."""
    import numpy as np
    import torch
    g = np.random.default_rng(seed)
    img = torch.from_numpy((g.random((B, 3, S, S), dtype=np.float32) * 255).astype(np.uint8))
    bi, cls, bb = [], [], []
    for b in range(B):
        for _ in range(int(g.integers(1, 9))):
            cx, cy = g.uniform(0.2, 0.8, 2)
            w, h = g.uniform(0.05, 0.35, 2)
            x1, y1, x2, y2 = max(cx - w / 2, 0), max(cy - h / 2, 0), min(cx + w / 2, 1), min(cy + h / 2, 1)
            bi.append(b)
            cls.append(int(g.integers(0, nc)))
            bb.append([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1])
    return dict(img=img.to(device), batch_idx=torch.tensor(bi, dtype=torch.float32).to(device),
                cls=torch.tensor(cls, dtype=torch.float32).view(-1, 1).to(device),
                bboxes=torch.tensor(bb, dtype=torch.float32).view(-1, 4).to(device), gamma=float(g.uniform(5.0, 10.0)))


def split_model_name(name):
    """'yolov8n-lowlight.yaml' -> ('yolov8-lowlight.yaml', 'n') (reference tasks.py:935 guess_model_scale)."""
    import re
    m = re.match(r"^(yolov8)([nsmlx])(.*\.yaml)$", os.path.basename(name))
    return (m.group(1) + m.group(3), m.group(2)) if m else (os.path.basename(name), "n")


def physical_cores():
    """Physical cores among the CPUs this process may run on (distinct (package, core id) pairs of /proc/cpuinfo)."""
    cpus = os.sched_getaffinity(0)
    seen, cur = set(), {}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f.read().split("\n") + [""]:
                if ":" in line:
                    k, v = [t.strip() for t in line.split(":", 1)]
                    cur[k] = v
                elif cur:
                    if int(cur.get("processor", -1)) in cpus:
                        seen.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                    cur = {}
    except OSError:
        pass
    return max(1, len(seen)) if seen else max(1, len(cpus)), len(cpus)


def cpu_leg(model_name, nc, S, B, threads, warm=3, timed=10, budget_s=40.0):
    """One CPU-oracle leg: fwd + loss + bwd + SGD of `model_name` in fp32 with `threads` threads; median of the timed steps."""
    import torch
    from oracle import loss as oloss
    from oracle import model as om
    from util import load_yaml
    torch.set_num_threads(threads)
    yaml_name, scale = split_model_name(model_name)
    plan, save = om.build_plan(load_yaml(yaml_name), scale=scale, nc=nc)
    sd = om.rng_fill(om.param_shapes(plan), 0)
    params = []
    for k, v in sd.items():
        if v.is_floating_point() and v.ndim > 0 and ".dfl." not in k and "running_" not in k:
            v.requires_grad_(True)
            params.append(v)
    b = synth_batch(99, B, S, nc, "cpu")
    hyp = oloss.default_hyp()
    low = "lowlight" in yaml_name or yaml_name == "yolov8.yaml"

    def step():
        img, clean, rec = oloss.preprocess_batch(b["img"], b["gamma"], low, low)
        maps = om.forward(plan, save, sd, img, True)
        strides = [float(S // m.shape[2]) for m in maps]
        batch = dict(batch_idx=b["batch_idx"], cls=b["cls"], bboxes=b["bboxes"], recovery_loss_batch=rec)
        loss, _ = oloss.recovery_detection_loss(maps, batch, strides, nc, hyp)
        loss.backward()
        with torch.no_grad():
            for p in params:
                p -= 0.01 * p.grad
                p.grad = None

    for _ in range(warm):
        step()
    ts, t_all = [], time.perf_counter()
    for _ in range(timed):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s and len(ts) >= 3:
            break
    ts.sort()
    med = ts[len(ts) // 2]
    return dict(model=model_name, batch=B, imgsz=S, threads=threads, img_per_s=round(B / med, 3), s_per_step=round(med, 4),
                warmup=warm, timed=len(ts))


def cpu_baseline(model_name, nc, S):
    """the oracle (CPU port of the reference path) on the host cores of this box.  Legs: the benched model
    (B = 1 for the L-scale graphs, 4 otherwise) and the C1 model (stock YOLOv8n, B = 4) with all physical cores, plus C1 with 8
    threads for comparability with the survey container.  `value` = the benched model's leg."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    phys, logical = physical_cores()
    _, scale = split_model_name(model_name)
    big = scale in "mlx"
    legs = [cpu_leg(model_name, nc, S, 1 if big else 4, phys, budget_s=45.0)]
    if os.path.basename(model_name) != "yolov8nori.yaml":
        legs.append(cpu_leg("yolov8nori.yaml", nc, 640, 4, phys, budget_s=20.0))
    if phys != 8:
        legs.append(cpu_leg("yolov8nori.yaml", nc, 640, 4, min(8, logical), budget_s=20.0))
    main = legs[0]
    return dict(value=main["img_per_s"], unit="img/s", cores=phys, kind="port", logical_cpus=logical, legs=legs,
                sample=f"oracle (CPU port of the reference path, fp32, torch {torch.__version__}): fwd+loss+bwd+SGD of {model_name} at "
                       f"{S}x{S}, batch {main['batch']}, {main['timed']} timed steps after {main['warmup']} warm-up (median), "
                       f"{phys} threads = physical cores of this box's CPU share ({logical} logical); legs: benched model, C1 "
                       "(yolov8nori.yaml@n B=4) at all cores and at 8 threads")


def kernel_profile(trainer, batches, steps=3):
    """Per-launch timing with events on the launch stream (torch's current stream) over `steps` extra steps, keyed by the GPU
    kernel symbol each C-ABI entry reports (dy_last_kernel)."""
    import torch
    from dedark_yolo_amd import _C, ops
    # one stream while instrumenting: a kernel that shares the chip with the side streams' kernels (weight gradients, Detect
    # levels) has an elapsed time that says nothing about its own efficiency
    ops.enable_wgrad_stream(False)
    ops.enable_branch_streams(False)
    _C._prof = []
    for i in range(steps):
        b = dict(batches[i % len(batches)])
        trainer.args.dark_param = b["gamma"]
        trainer.train_step(b)
    torch.cuda.synchronize()
    rec, _C._prof = _C._prof, None
    ops.enable_wgrad_stream(True)
    ops.enable_branch_streams(True)
    agg = {}
    for name, e0, e1, meta, kern in rec:
        key = kern or name
        a = agg.setdefault(key, dict(ms=0.0, n=0, flops=0.0, bytes=0.0, meta_n=0, entries=set(), dtype=None))
        a["ms"] += e0.elapsed_time(e1)
        a["n"] += 1
        a["entries"].add(name)
        if meta:
            a["flops"] += meta["flops"]
            a["bytes"] += meta["bytes"]
            a["meta_n"] += 1
            a["dtype"] = meta["dtype"].replace("torch.", "")
    return agg, steps


def workload_tag(args):
    """Which BASELINE.json config the command line is."""
    key = (os.path.basename(args.model), args.imgsz, args.batch, args.dtype)
    if key == ("yolov8n-lowlight.yaml", 640, 32, "bf16"):
        return "BASELINE configs[1] (C2: YOLOv8n + lowlight_recovery front-end)", "c2"
    if key == ("yolov8l.yaml", 640, 64, "bf16"):
        return "BASELINE configs[2] (C3: repo yolov8.yaml@L = lowlight_recovery + ASFF neck)", "c3"
    if key == ("yolov8nori.yaml", 640, 4, "fp32"):
        return "BASELINE configs[0] (C1: plain YOLOv8n graph yolov8ori.yaml@n, the reference's CPU-runnable case)", "c1"
    if key == ("yolov8l.yaml", 1280, 16, "fp16"):
        return "BASELINE configs[4] (C5: repo yolov8.yaml@L at 1280x1280, fp16)", "c5"
    return "custom workload", None


def pmc_traffic(tag, kernel, calls_per_step):
    """HBM-side bytes of the kernel symbol from the committed rocprofv3 PMC passes of this same command (tools/pmc_traffic.py:
    2 x FETCH_SIZE + WRITE_SIZE, separate passes), in the unit of `algorithmic_bytes_per_launch`: per C-ABI CALL of the symbol
    (one call can be several dispatches: a stride-2 data gradient used to be four).  Returns (bytes per call, dispatches per call,
    file) or (None, None, None) when no summary names the kernel."""
    if tag is None:
        return None, None, None
    for rnd in ("r03", "r02", "r01"):
        f = f"{rnd}_{tag}_pmc_traffic.json"
        path = os.path.join(ROOT, "profiles", f)
        if not os.path.exists(path):
            continue
        with open(path) as fh:
            js = json.load(fh)
        ks = js.get("kernels", {})
        steps = js.get("steps", 5)                 # tools/profile_round.sh: --steps 3 --warmup 2
        base, _, tail = kernel.partition("+")
        match = lambda k, b: k == b or k.startswith(b + "<") or k.startswith(b + "(")
        hit = [v for k, v in ks.items() if match(k, base)]
        if hit:
            tot = sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in hit)
            n = sum(v["launches"] for v in hit)
            if tail:        # "ns::main_kernel+reduce_kernel": the split reduction of the same namespace belongs to the launch
                ns = base.split("::")[0] + "::" if "::" in base else ""
                red = [v for k, v in ks.items() if match(k, ns + tail)]
                share = 1.0
                if ns:      # the reduce kernel is shared by all instantiations of the namespace: prorate by launches
                    allmain = sum(v["launches"] for k, v in ks.items() if k.startswith(ns) and not match(k, ns + tail))
                    share = n / max(allmain, 1)
                tot += share * sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in red)
            calls = max(calls_per_step * steps, 1)
            return round(tot / calls), round(n / calls, 3), "profiles/" + f
    return None, None, None


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("DY_SINGLE_DEVICE"):      # rehearsal of the N>1 code path on a one-GPU box (tests): every rank on cuda:0, gloo
        local = 0
        os.environ["LOCAL_RANK"] = "0"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    backend = os.environ.get("DY_DIST_BACKEND", "nccl")
    if world > 1:
        dist.init_process_group(backend)

    from dedark_yolo_amd.engine.trainer import DetectionTrainer, get_cfg
    from dedark_yolo_amd.nn.tasks import DetectionModel
    nc = 20
    torch.manual_seed(0)
    cfg = get_cfg(dict(model=args.model, dtype=args.dtype, optimizer="SGD", batch=args.batch * world, imgsz=args.imgsz,
                       lowlight_FLAG=True, dedark_FLAG=True, deterministic=False))       # throughput schedule: side streams on
    trainer = DetectionTrainer(cfg)
    trainer.setup(DetectionModel(args.model, nc=nc))
    batches = [synth_batch(1234 + 17 * rank + i, args.batch, args.imgsz, nc, device) for i in range(4)]
    loader = None
    if args.pipeline != "resident":
        # the step BEFORE the hot path: a synthetic decoded dataset (uint8 HWC at its load_image size, 1..8 boxes per
        # image) goes through the device input pipeline every step instead of pre-staged batches
        from dedark_yolo_amd.data import DeviceAugmentLoader
        g = np.random.default_rng(4321 + rank)
        n_img = max(4 * args.batch, 256)
        ims, labs = [], []
        for i in range(n_img):
            h, w = (args.imgsz, int(g.integers(args.imgsz * 5 // 8, args.imgsz + 1))) if i % 2 else (int(g.integers(args.imgsz * 5 // 8, args.imgsz + 1)), args.imgsz)
            ims.append(g.integers(0, 256, (h, w, 3), dtype=np.uint8))
            k = int(g.integers(1, 9))
            labs.append(dict(cls=g.integers(0, nc, (k, 1)).astype(np.float32),
                             bboxes=np.concatenate((g.uniform(0.2, 0.8, (k, 2)), g.uniform(0.05, 0.35, (k, 2))), 1).astype(np.float32)))
        loader = DeviceAugmentLoader(ims, labs, args.imgsz, args.batch, device=device, resident=args.pipeline == "device", seed=99 + rank)
        gammas = [b["gamma"] for b in batches]
        state = dict(it=iter(loader), k=0)

    def next_loader_batch():
        try:
            b = next(state["it"])
        except StopIteration:
            state["it"] = iter(loader)
            b = next(state["it"])
        state["k"] += 1
        return b, gammas[state["k"] % len(gammas)]

    def run(n):
        for i in range(n):
            if loader is None:
                b = dict(batches[i % len(batches)])
                trainer.args.dark_param = b["gamma"]
            else:
                b, trainer.args.dark_param = next_loader_batch()
            loss, items = trainer.train_step(b)
        return loss

    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        trainer.buckets.exposed_events = []       # events around the compute stream's waits for the bucketed all-reduce
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = run(args.steps)
    torch.cuda.synchronize()
    el_own = time.perf_counter() - t0             # this rank, before it waits for the others
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t)
    per_rank = None
    if world > 1:
        ev, trainer.buckets.exposed_events = trainer.buckets.exposed_events, None
        exposed = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
        mine = torch.tensor([1000 * el_own / args.steps, exposed], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = dict(ms_per_step=[round(float(x[0]), 3) for x in allr], allreduce_exposed_ms_per_step=[round(float(x[1]), 3) for x in allr],
                        gradient_bytes=int(trainer.flat.n) * 4, buckets=len(trainer.buckets.buckets))
    final_loss = float(loss)
    if not np.isfinite(final_loss):
        raise SystemExit(f"non-finite loss {final_loss}")

    wl, tag = workload_tag(args)
    out = dict(metric=f"training img/s at {args.imgsz}x{args.imgsz}", value=round(args.batch * world * args.steps / el, 2), unit="img/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1000 * el / args.steps, 3),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype=args.dtype, data="synthetic",
               config=dict(workload=f"{wl}: {args.model}, {args.imgsz}x{args.imgsz}, "
                                    f"{args.dtype}, batch {args.batch}/GPU, nc={nc}, gamma~U(5,10), full train step "
                                    "(preprocess+fwd+loss+assigner+bwd+clip+SGD+EMA), deterministic=False (side streams on), random-init weights, " +
                                    {"resident": "inputs resident in HBM",
                                     "device": "decoded dataset resident in HBM, mosaic/affine/HSV/flip rendered on the device every step",
                                     "host": "decoded dataset in pinned host memory: source images uploaded every step (PCIe-inclusive), "
                                             "rendered on the device"}[args.pipeline],
                           global_batch=args.batch * world, parallelism=f"dp{world}"),
               final_loss=round(final_loss, 4))
    if loader is not None:
        out["pipeline"] = dict(mode=args.pipeline, dataset_images=len(loader.shapes),
                               uploaded_mb_per_step=round(loader.uploaded_bytes / max(state["k"], 1) / 1e6, 1))
    if world > 1:                                 # replicas must stay bit-identical: same start state, same summed gradients
        h = trainer.flat.p.double().sum().reshape(1)
        hs = [torch.zeros_like(h) for _ in range(world)]
        dist.all_gather(hs, h)
        out["replicas_in_sync"] = bool(all(torch.equal(hs[0], x) for x in hs))
        out["dist_backend"] = backend
        out["rccl_ranks"] = world if backend == "nccl" else 0
        out["per_rank"] = per_rank                # own step time of every rank + the all-reduce time backward did not hide

    agg = None
    if not args.no_roofline:                      # every rank runs the instrumented steps: they contain the gradient all-reduce
        agg, psteps = kernel_profile(trainer, batches)
    if rank == 0 and agg is not None:
        tot = sum(a["ms"] for a in agg.values())
        top = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
        # the kernel with the largest share among those whose every launch carries its algorithmic work
        name, a = next(((k, v) for k, v in top if v["meta_n"] == v["n"] and v["n"] > 0), top[0])
        avg_s = a["ms"] / a["n"] / 1e3
        intensity = a["flops"] / max(a["bytes"], 1.0)
        f32 = a["dtype"] == "float32"
        peak_tf = MFMA_F32_PEAK_TF if f32 else MFMA_BF16_PEAK_TF
        ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
        if a["flops"] > 0 and intensity > ridge:
            ach = a["flops"] / a["n"] / avg_s / 1e12
            roof = dict(bound="mfma", achieved=round(ach, 2), peak=peak_tf, unit="TFLOP/s", frac=round(ach / peak_tf, 4))
        else:
            ach = a["bytes"] / a["n"] / avg_s / 1e9
            roof = dict(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4))
        traffic, disp, src = pmc_traffic(tag, name, a["n"] // psteps)
        roof.update(kernel=name, c_abi_entries=sorted(a["entries"]), launches_per_step=a["n"] // psteps, dispatches_per_launch=disp,
                    traffic_over_algorithmic=None if traffic is None else round(traffic / max(a["bytes"] / a["n"], 1.0), 3),
                    avg_launch_us=round(avg_s * 1e6, 2), share_of_kernel_time=round(a["ms"] / tot, 3), flop_per_byte=round(intensity, 1),
                    algorithmic_bytes_per_launch=round(a["bytes"] / a["n"]), algorithmic_flops_per_launch=round(a["flops"] / a["n"]),
                    traffic=traffic, traffic_source=src,
                    units="a launch = one C-ABI call of the symbol; achieved, algorithmic bytes / flops and traffic are all per call",
                    streams="kernel durations taken on ONE stream (DY_WGRAD_STREAM=0 DY_BRANCH_STREAMS=0 equivalent); the timed "
                            "region runs the weight gradients and the coarser Detect levels on side streams")
        out["roofline"] = roof
        out["kernel_time_breakdown_ms_per_step"] = {k: round(v["ms"] / psteps, 3) for k, v in top[:12]}
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.model, nc, args.imgsz)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
