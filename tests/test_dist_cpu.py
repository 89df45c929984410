"""world_size-2 gloo tests (CPU) of the data-parallel host logic: flat gradient storage, layer-aligned buckets issued from
the backward hooks, sum semantics (reference trainer.py:334-335 `loss *= world_size` + DDP mean == sum of per-rank
gradients), start-state broadcast (trainer.py:219-223) and rank-distinct synthetic shards of bench.py.
No compute kernels run here: gradients are filled by hand, the collective is gloo."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Toy(nn.Module):
    """stands in for DetectionModel: `.model` is the layer list, some layers have no parameters."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.model = nn.ModuleList([
            nn.Sequential(nn.Conv2d(3, 8, 3, bias=False), nn.BatchNorm2d(8)),
            nn.Upsample(scale_factor=2),
            nn.Sequential(nn.Conv2d(8, 16, 3, bias=False), nn.BatchNorm2d(16)),
            nn.Sequential(nn.Conv2d(16, 16, 1, bias=True)),
            nn.Identity(),
            nn.Sequential(nn.Conv2d(16, 5, 1, bias=True), nn.BatchNorm2d(5)),
        ])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dedark_yolo_amd.engine.trainer import FlatState, GradBuckets
        model = _Toy()
        if rank == 1:                                  # a diverged start state, repaired by the broadcast
            for p in model.parameters():
                p.data.add_(1.0)
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
        flat = FlatState(model, with_ema=True)
        # groups: 0 decayed weights, 1 norm weights, 2 biases (reference trainer.py:626-634)
        gid = flat.gid.numpy()
        counts = [int((gid == k).sum()) for k in range(3)]
        buckets = GradBuckets(flat, model, bucket_bytes=2048)      # small -> several buckets
        nb = len(buckets.buckets)
        cover = sorted((b["start"], b["end"]) for b in buckets.buckets)
        contiguous = cover[0][0] == 0 and cover[-1][1] == flat.n and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
        res = []
        for step in range(2):                          # two steps: the per-bucket counters must re-arm
            g = torch.Generator().manual_seed(100 * step + rank)
            flat.g.copy_(torch.randn(flat.n, generator=g))
            mine = flat.g.clone()
            issued = []
            for li in reversed(range(len(model.model))):        # backward visits layers in reverse order
                hook = getattr(model.model[li], "_dy_after_backward", None)
                if hook is not None:
                    hook()
                issued.append(len(buckets.works))
            buckets.finish()
            res.append((mine.numpy(), flat.g.clone().numpy(), issued))
        p0 = flat.p.clone().numpy()
        q.put((rank, counts, nb, contiguous, res, p0, [p.grad.data_ptr() == flat.g[o:o + n].data_ptr() for p, o, n, _ in flat.slots]))
    finally:
        dist.destroy_process_group()


def test_grad_buckets_gloo_ws2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(2):
        r = q.get(timeout=240)
        out[r[0]] = r
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, c0, nb0, cont0, res0, p0, al0), (_, c1, nb1, cont1, res1, p1, al1) = out[0], out[1]
    assert c0 == c1 and c0[2] == 16 + 5 + 8 + 16 + 5 and c0[1] == 8 + 16 + 5            # biases incl. BN biases; norm weights
    assert nb0 == nb1 and nb0 >= 2 and cont0 and cont1
    assert all(al0) and all(al1)                       # p.grad aliases the flat buffer (direct gradient placement)
    assert np.array_equal(p0, p1)                      # broadcast start state
    for s in range(2):
        want = res0[s][0] + res1[s][0]
        assert np.array_equal(res0[s][1], want) and np.array_equal(res1[s][1], want)
        issued = res0[s][2]
        assert issued[-1] == nb0                       # every bucket went out from a hook, none left for finish()
        assert issued[0] <= 1 and sorted(issued) == issued   # and they go out progressively while backward is still running


def test_bench_shards_are_rank_distinct():
    import bench
    a = bench.synth_batch(1234 + 0, 2, 64, 20, "cpu")
    b = bench.synth_batch(1234 + 17, 2, 64, 20, "cpu")
    a2 = bench.synth_batch(1234 + 0, 2, 64, 20, "cpu")
    assert a["img"].dtype == torch.uint8 and a["img"].shape == (2, 3, 64, 64)
    assert torch.equal(a["img"], a2["img"]) and not torch.equal(a["img"], b["img"])
    assert a["bboxes"].shape[1] == 4 and a["cls"].shape[1] == 1 and a["batch_idx"].shape[0] == a["cls"].shape[0]
    assert float(a["bboxes"].min()) >= 0 and float(a["bboxes"].max()) <= 1
    assert bench.split_model_name("yolov8n-lowlight.yaml") == ("yolov8-lowlight.yaml", "n")
    assert bench.split_model_name("yolov8l.yaml") == ("yolov8.yaml", "l")


def _epoch_worker(rank, world, port, q):
    """The epoch loop of DetectionTrainer.train with the device work stubbed out (no GPU here): what is under test is the
    reference's epoch-boundary protocol (engine/trainer.py:366-395) -- rank 0 validates / stops, the flag reaches every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        from types import SimpleNamespace
        from dedark_yolo_amd.engine.trainer import DetectionTrainer
        tr = DetectionTrainer.__new__(DetectionTrainer)
        tr.args = SimpleNamespace(epochs=10, warmup_epochs=0.0, nbs=64, batch=64, patience=2, val=True, save=True, lrf=0.01, cos_lr=False,
                                  warmup_bias_lr=0.1, warmup_momentum=0.8)
        tr.rank, tr.world_size, tr.device = rank, world, "cpu"
        tr.lr0, tr.momentum, tr.accumulate, tr.last_opt_step = 0.01, 0.9, 1, -1
        log = dict(steps=0, vals=[], saves=[])
        fitness = iter([0.30, 0.40, 0.35, 0.39, 0.20, 0.10, 0.10])          # best at epoch 2 -> patience 2 stops after epoch 4
        tr._batches = lambda loader: iter(loader)

        def train_step(batch, lr=None, mom=None, step_optimizer=True):
            log["steps"] += 1
            return torch.tensor(1.0), torch.tensor([1.0, 2.0, 3.0])

        def validate(val_loader):
            time.sleep(0.3)                                # rank 0 is busy; the others must wait at the broadcast, not run ahead
            f = next(fitness)
            log["vals"].append(f)
            return {"fitness": f}, f
        tr.train_step, tr.validate = train_step, validate
        tr.save_model = lambda wdir, epoch=0, fitness=None: log["saves"].append(epoch)
        t0 = time.time()
        hist = tr.train([{}] * 3, epochs=10, val_loader=[{}], save_dir="/tmp/unused")
        q.put((rank, len(hist), log["steps"], log["vals"], log["saves"], bool(tr.stop), time.time() - t0))
    finally:
        dist.destroy_process_group()


def test_epoch_boundary_stop_broadcast_gloo_ws2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_epoch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(2):
        r = q.get(timeout=240)
        out[r[0]] = r
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, n0, s0, v0, sv0, stop0, t0), (_, n1, s1, v1, sv1, stop1, t1) = out[0], out[1]
    assert n0 == n1 == 4 and s0 == s1 == 12           # both ranks ran exactly 4 epochs of 3 steps: the stop reached rank 1
    assert v0 == [0.30, 0.40, 0.35, 0.39] and v1 == [] and sv0 == [0, 1, 2, 3] and sv1 == []      # validation / checkpoints on rank 0 only
    assert stop0 and stop1
    assert t1 >= 0.9 * 4 * 0.3                        # rank 1 waited for rank 0's validation at every epoch boundary
