#!/usr/bin/env python3
"""GPU debugging aid (not a test): raw kernels vs torch ops on the same device, printing error tables."""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dedark_yolo_amd as dy  # noqa: E402
from dedark_yolo_amd import ops  # noqa: E402
from dedark_yolo_amd._C import call  # noqa: E402


class T:
    def __init__(self):
        self.stack, self.pgrads = [], {}

    def push(self, c):
        self.stack.append(c)

    def pop(self):
        return self.stack.pop()


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max()), float(b.abs().max())


def conv_case(dtype, B, Cin, Cout, H, W, k, s, p, d=1):
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, W, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") * (1.0 / (Cin * k * k) ** 0.5)
    w.requires_grad_(True)
    bias = torch.randn(Cout, device="cuda").requires_grad_(True)
    xq = x.to(dtype).float()
    wq = w.detach().to(dtype).float().requires_grad_(True)
    xr = xq.clone().requires_grad_(True)
    ref = F.conv2d(xr, wq, bias, s, p, d)
    xin = ops.as_nhwc(x, dtype)
    tape = T()
    y = ops.conv_forward(tape, xin, w, bias, None, 0, s, p, d, False)
    e = rel(y, ref)
    if os.environ.get("DY_DUMP") and Cin == 8:
        torch.set_printoptions(precision=4, linewidth=200)
        print("y  ", y.float()[0, :6, 0, :4].flatten().tolist())
        print("ref", ref[0, :6, 0, :4].flatten().tolist())
        # which (pixel, channel) of ref does each y value equal?
        yf, rf = y.float().permute(0, 2, 3, 1).reshape(-1, Cout), ref.permute(0, 2, 3, 1).reshape(-1, Cout)
        for pi in range(2):
            for ci in range(4):
                dd = (rf - yf[pi, ci]).abs()
                j = int(dd.argmin())
                print(f"   y[px{pi},c{ci}]={float(yf[pi, ci]):.4f} ~ ref[px{j // Cout},c{j % Cout}] (err {float(dd.min()):.2e})")
    gy = torch.randn_like(ref)
    ref.backward(gy)
    gyq = ops.as_nhwc(gy, dtype)
    dx = ops.conv_backward(tape, gyq, need_dx=True)
    e_dx = rel(dx, xr.grad)
    e_dw = rel(tape.pgrads[w], wq.grad)
    e_db = rel(tape.pgrads[bias], bias.grad)
    print(f"{str(dtype):16s} B{B} {Cin:4d}->{Cout:4d} {H}x{W} k{k} s{s} p{p} d{d}: fwd {e[0]:.3e}/{e[1]:.2e}  dx {e_dx[0]:.3e}/{e_dx[1]:.2e}"
          f"  dw {e_dw[0]:.3e}/{e_dw[1]:.2e}  db {e_db[0]:.3e}/{e_db[1]:.2e}", flush=True)


def pack_case(dtype):
    w = torch.arange(2 * 3 * 2 * 2, dtype=torch.float32, device="cuda").view(2, 3, 2, 2)
    out = torch.full((4 * 2 * 2 * 4,), -1.0, dtype=dtype, device="cuda")
    call("dy_pack_weight", w.data_ptr(), out.data_ptr(), 2, 4, 3, 4, 2, 2, 0, ops.dt_id(dtype), ops.stream())
    exp = torch.zeros(4, 2, 2, 4, device="cuda")
    exp[:2, :, :, :3] = w.permute(0, 2, 3, 1)
    print("pack", dtype, rel(out.view(4, 2, 2, 4), exp), flush=True)
    call("dy_pack_weight", w.data_ptr(), out.data_ptr(), 2, 4, 3, 4, 2, 2, 1, ops.dt_id(dtype), ops.stream())
    exp = torch.zeros(4, 2, 2, 4, device="cuda")          # [Cin_pad][KH][KW][Cout_pad]
    exp[:3, :, :, :2] = w.permute(1, 2, 3, 0)
    print("pack_t", dtype, rel(out.view(4, 2, 2, 4), exp), flush=True)


def bn_case(dtype):
    from dedark_yolo_amd.nn.modules import Conv
    torch.manual_seed(1)
    dy.set_compute_dtype(dtype)
    m = Conv(16, 32, 3, 1).cuda().train()
    m.bn.eps, m.bn.momentum = 1e-3, 0.03
    with torch.no_grad():
        m.bn.weight.uniform_(0.5, 1.5)
        m.bn.bias.uniform_(-0.2, 0.2)
    x = torch.randn(2, 16, 20, 24, device="cuda")
    ref_conv = torch.nn.Conv2d(16, 32, 3, 1, 1, bias=False).cuda()
    ref_bn = torch.nn.BatchNorm2d(32, eps=1e-3, momentum=0.03).cuda()
    ref_conv.weight.data.copy_(m.conv.weight.data.to(dtype).float())
    ref_bn.weight.data.copy_(m.bn.weight.data)
    ref_bn.bias.data.copy_(m.bn.bias.data)
    xr = x.to(dtype).float().clone().requires_grad_(True)
    ref = F.silu(ref_bn(ref_conv(xr)))
    xi = x.detach().clone().requires_grad_(True)
    y = m(xi)
    print("conv+bn+silu", dtype, "fwd", rel(y, ref), flush=True)
    gy = torch.randn_like(ref)
    ref.backward(gy)
    y.backward(gy.to(y.dtype))
    print("   dx", rel(xi.grad, xr.grad), "dw", rel(m.conv.weight.grad, ref_conv.weight.grad), "dgamma",
          rel(m.bn.weight.grad, ref_bn.weight.grad), "dbeta", rel(m.bn.bias.grad, ref_bn.bias.grad), "rmean",
          rel(m.bn.running_mean, ref_bn.running_mean), "rvar", rel(m.bn.running_var, ref_bn.running_var), flush=True)


if __name__ == "__main__":
    for dt in (torch.float32, torch.bfloat16):
        pack_case(dt)
    for dt in (torch.float32, torch.bfloat16):
        conv_case(dt, 1, 8, 32, 4, 4, 1, 1, 0)
        conv_case(dt, 2, 24, 16, 9, 7, 1, 1, 0)
        conv_case(dt, 2, 16, 32, 12, 20, 3, 2, 1)
        conv_case(dt, 2, 32, 64, 16, 16, 3, 1, 1)
        conv_case(dt, 1, 64, 128, 20, 20, 3, 1, 1)
        conv_case(dt, 1, 32, 8, 10, 9, 3, 1, 2, 2)
        conv_case(dt, 2, 256, 256, 8, 8, 3, 1, 1)
        conv_case(dt, 2, 8, 16, 33, 31, 5, 1, 2)
        conv_case(dt, 2, 16, 32, 13, 21, 3, 2, 1)       # stride-2 dgrad = 4 parity classes, odd sizes
        conv_case(dt, 2, 8, 16, 12, 14, 1, 2, 0)        # k1 s2: empty parity classes -> masked fallback
        conv_case(dt, 1, 8, 16, 17, 16, 5, 2, 2)        # k5 s2: unequal class pads -> masked fallback
        if dt == torch.bfloat16:      # shapes routed to the pipelined kernel (conv_v2.hip): M >= 2048, Cin % 64 == 0
            conv_case(dt, 2, 64, 128, 40, 40, 3, 1, 1)
            conv_case(dt, 3, 128, 64, 32, 32, 3, 2, 1)
            conv_case(dt, 2, 64, 128, 65, 63, 3, 2, 1)  # parity-split dgrad on the pipelined kernel (strided destination)
            # shapes routed to the pipelined weight-gradient kernel (wgrad_v2.hip): M >= 4096, Cout >= 64, K >= 128
            conv_case(dt, 3, 64, 128, 40, 40, 3, 1, 1)
            conv_case(dt, 8, 96, 80, 37, 41, 3, 1, 1)   # ragged k' and co tiles
            conv_case(dt, 16, 64, 128, 40, 40, 3, 2, 1)
            conv_case(dt, 4, 128, 96, 40, 40, 3, 1, 3, 3)  # dilation 3 (RFB)
            conv_case(dt, 9, 64, 64, 128, 128, 3, 1, 1)    # Cout 64: pipelined kernel only for long pixel loops (M >= 131072)
            conv_case(dt, 2, 384, 256, 48, 48, 1, 1, 0)
            conv_case(dt, 3, 256, 8, 37, 23, 1, 1, 0)     # thin 1x1 dgrad (ASFF weight_level convs)
            conv_case(dt, 2, 3, 16, 64, 64, 3, 2, 1)      # stem: direct dot2 dgrad, planar dx
            conv_case(dt, 2, 3, 64, 33, 47, 3, 2, 1)
            conv_case(dt, 2, 256, 192, 48, 48, 1, 1, 0)
            conv_case(dt, 1, 64, 64, 50, 47, 3, 1, 1)
            conv_case(dt, 2, 192, 320, 24, 24, 3, 1, 1)
    for dt in (torch.float32, torch.bfloat16):
        bn_case(dt)
