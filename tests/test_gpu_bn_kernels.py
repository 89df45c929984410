"""GPU unit tests of the BatchNorm + activation streaming kernels through the C-ABI (dy_bn_act_fwd / dy_bn_act_bwd_reduce /
dy_bn_act_bwd_apply: nn.BatchNorm2d + SiLU / LeakyReLU(0.1) + residual add of the reference, ultralytics/nn/modules/conv.py:49-51,
block.py:553-565) against the same formulas in torch fp32 on the device -- at a size below and a size above the 128 MB threshold where
the kernels switch to their streaming variants (more pixels in flight, non-temporal loads / stores), for every activation, with and
without the residual, in all three dtypes.  The whole-graph tests only reach the large variants at the benchmark sizes."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [(torch.float32, 64, 4096 + 3), (torch.bfloat16, 128, 8192 + 5), (torch.float16, 48, 4096 + 1),
         (torch.float32, 64, 540_001), (torch.bfloat16, 64, 1_100_003), (torch.float16, 128, 530_007)]


@pytest.mark.parametrize("dt,C,px", CASES, ids=lambda v: str(v).replace("torch.", ""))
def test_bn_act_kernels_vs_torch(dt, C, px):
    from dedark_yolo_amd import ops
    from dedark_yolo_amd._C import BN_BWD_REPLICAS, call
    from dedark_yolo_amd.ops import ptr, stream
    torch.manual_seed(px)
    dev = "cuda"
    z = (torch.randn(px, C, device=dev) * 1.5).to(dt)
    res = torch.randn(px, C, device=dev).to(dt)
    dy = torch.randn(px, C, device=dev).to(dt)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
    mean, inv, gamma = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5
    tol = {torch.float32: 2e-5, torch.bfloat16: 6e-3, torch.float16: 8e-4}[dt]          # storage rounding of the outputs
    did = ops.dt_id(dt)
    u = z.float() * sc + sh
    zh = (z.float() - mean) * inv
    for act in (1, 2, 0):
        a = u * torch.sigmoid(u) if act == 1 else (torch.where(u > 0, u, 0.1 * u) if act == 2 else u)
        for use_res in (False, True):
            y = torch.empty_like(z)
            call("dy_bn_act_fwd", ptr(z), C, ptr(sc), ptr(sh), act, ptr(res) if use_res else None, C, ptr(y), C, px, C, did, stream())
            ref = a + res.float() if use_res else a
            assert float((y.float() - ref).abs().max()) <= tol * float(ref.abs().max()), (act, use_res)
        s = torch.sigmoid(u)
        d = s * (1 + u * (1 - s)) if act == 1 else (torch.where(u > 0, 1.0, 0.1) if act == 2 else torch.ones_like(u))
        g = dy.float() * d
        gz = g * zh
        r1, r2 = g.sum(0, dtype=torch.float64), gz.sum(0, dtype=torch.float64)
        sums = torch.zeros(BN_BWD_REPLICAS * 2 * C, dtype=torch.float64, device=dev)
        call("dy_bn_act_bwd_reduce", ptr(dy), C, ptr(z), C, ptr(sc), ptr(sh), ptr(mean), ptr(inv), act, 1, ptr(sums), px, C, did, stream())
        t = sums.view(BN_BWD_REPLICAS, 2, C).sum(0)
        assert float((t[0] - r1).abs().max()) <= 1e-5 * float(g.abs().sum(0, dtype=torch.float64).max())
        assert float((t[1] - r2).abs().max()) <= 1e-5 * float(gz.abs().sum(0, dtype=torch.float64).max())
        dz = torch.empty_like(z)
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        call("dy_bn_act_bwd_apply", ptr(dy), C, ptr(z), C, ptr(sc), ptr(sh), ptr(mean), ptr(inv), ptr(gamma), act, 1, ptr(sums), ptr(dz), C,
             ptr(dg), ptr(db), px, C, did, stream())
        k1 = gamma * inv
        refdz = k1 * g - (k1 * (t[1].float() / px) * zh + k1 * (t[0].float() / px))
        assert float((dz.float() - refdz).abs().max()) <= tol * float(refdz.abs().max()), act
        assert torch.allclose(db, t[0].float(), rtol=1e-6) and torch.allclose(dg, t[1].float(), rtol=1e-6)
