"""Raw conv kernels (forward, data gradient, weight gradient, bias gradient) against torch's own fp32 conv on the same
GPU, per dispatch route of csrc/: generic implicit GEMM (f32 exact MFMA and bf16), pipelined v2, 3x3 band kernel v3,
parity-split stride-2 dgrad, direct stem / thin kernels, pipelined weight gradient.  Inputs of the bf16 cases are rounded to
bf16 before the reference runs, so the difference is output rounding and accumulation order only:
tolerance 1e-5 * max|ref| (f32) and 1e-2 * max|ref| (bf16)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


class _Tape:
    def __init__(self):
        self.stack, self.pgrads = [], {}

    def push(self, c):
        self.stack.append(c)

    def pop(self):
        return self.stack.pop()


def _err(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def _case(dtype, B, Cin, Cout, H, W, k, s, p, d=1, routes=None):
    """`routes`: dict that receives {C-ABI entry: GPU kernel symbol it launched} (dy_last_kernel) for the route assertions."""
    from dedark_yolo_amd import _C, ops
    if routes is not None:
        _C._prof = []
    torch.manual_seed(B * 1000 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, device="cuda")
    kh, kw = k if isinstance(k, tuple) else (k, k)
    w = (torch.randn(Cout, Cin, kh, kw, device="cuda") * (1.0 / (Cin * kh * kw) ** 0.5)).requires_grad_(True)
    bias = torch.randn(Cout, device="cuda").requires_grad_(True)
    xr = x.to(dtype).float().clone().requires_grad_(True)
    wq = w.detach().to(dtype).float().requires_grad_(True)
    ref = F.conv2d(xr, wq, bias, s, p, d)
    tape = _Tape()
    y = ops.conv_forward(tape, ops.as_nhwc(x, dtype), w, bias, None, 0, s, p, d, False)
    gy = torch.randn_like(ref)
    ref.backward(gy)
    dx = ops.conv_backward(tape, ops.as_nhwc(gy, dtype), need_dx=True)
    # the reference saw gy in f32; the kernel saw it rounded to `dtype`: compare against a reference fed the rounded gy
    xr2 = xr.detach().clone().requires_grad_(True)
    wq2 = wq.detach().clone().requires_grad_(True)
    b2 = bias.detach().clone().requires_grad_(True)
    F.conv2d(xr2, wq2, b2, s, p, d).backward(gy.to(dtype).float())
    torch.cuda.synchronize()
    if routes is not None:
        for name, _e0, _e1, _meta, kern in _C._prof:
            if kern:
                routes[name] = kern
        _C._prof = None
    return dict(fwd=_err(y, ref), dx=_err(dx, xr2.grad), dw=_err(tape.pgrads[w], wq2.grad), db=_err(tape.pgrads[bias], b2.grad))


GENERIC = [(1, 8, 32, 4, 4, 1, 1, 0), (2, 24, 16, 9, 7, 1, 1, 0), (2, 16, 32, 12, 20, 3, 2, 1), (2, 32, 64, 16, 16, 3, 1, 1),
           (1, 64, 128, 20, 20, 3, 1, 1), (1, 32, 8, 10, 9, 3, 1, 2, 2), (2, 256, 256, 8, 8, 3, 1, 1), (2, 8, 16, 33, 31, 5, 1, 2),
           (2, 16, 32, 13, 21, 3, 2, 1),      # stride-2 dgrad as 4 parity classes, odd sizes
           (2, 8, 16, 12, 14, 1, 2, 0),       # k1 s2: empty parity classes -> masked fallback
           (1, 8, 16, 17, 16, 5, 2, 2),       # k5 s2: unequal class pads -> masked fallback
           (5, 32, 64, 8, 8, 8, 1, 0),        # whole-input window = fully connected layer (dense.hip)
           (3, 16, 24, 4, 6, (4, 6), 1, 0)]   # ... non-square
PIPELINED = [(2, 64, 128, 40, 40, 3, 1, 1), (3, 128, 64, 32, 32, 3, 2, 1), (2, 64, 128, 65, 63, 3, 2, 1),
             (2, 256, 192, 48, 48, 1, 1, 0), (1, 64, 64, 50, 47, 3, 1, 1), (2, 192, 320, 24, 24, 3, 1, 1),
             (3, 64, 128, 40, 40, 3, 1, 1), (8, 96, 80, 37, 41, 3, 1, 1), (16, 64, 128, 40, 40, 3, 2, 1),
             (4, 128, 96, 40, 40, 3, 1, 3, 3), (9, 64, 64, 128, 128, 3, 1, 1), (2, 384, 256, 48, 48, 1, 1, 0),
             (4, 32, 32, 40, 40, 3, 1, 1), (4, 16, 16, 48, 40, 3, 1, 1), (4, 48, 32, 40, 40, 1, 1, 0),   # narrow layers, M >= 2048
             (4, 16, 32, 41, 39, 3, 2, 1), (2, 8, 16, 64, 64, 3, 2, 1), (4, 24, 40, 33, 35, 3, 1, 1),
             (2, 3, 64, 64, 96, 3, 2, 1), (1, 3, 64, 33, 47, 3, 2, 1), (2, 8, 64, 40, 42, 3, 2, 1),     # the L stem: MFMA data gradient (planar / NHWC8)
             (4, 32, 64, 40, 40, 3, 1, 1), (4, 48, 96, 40, 40, 1, 1, 0), (4, 16, 72, 41, 40, 3, 2, 1),   # narrow source, wide destination
             (2, 128, 272, 48, 48, 3, 1, 1), (4, 64, 512, 33, 31, 3, 2, 1),   # >= 256 output channels: 256x256 wgrad / conv tiles
             (6, 60, 62, 150, 147, 3, 1, 1),   # band weight gradient (64-channel 3x3, long pixel loop), ragged width / channels
             (24, 128, 128, 80, 72, 3, 1, 1), (24, 124, 64, 80, 72, 3, 1, 1), (24, 64, 128, 80, 72, 3, 1, 1),   # ... 128-channel variants
             (3, 256, 8, 37, 23, 1, 1, 0),     # thin 1x1 dgrad (ASFF weight_level convs)
             (12, 16, 32, 41, 39, 3, 2, 1), (5, 32, 24, 37, 35, 3, 1, 1), (4, 16, 16, 40, 40, 1, 1, 0), (6, 32, 16, 36, 38, 3, 2, 1),
             (3, 48, 16, 33, 30, 3, 1, 1),     # thin-layer kernel (A fragments straight from global memory): s1 / s2 / parity classes / ragged
             (2, 3, 16, 64, 64, 3, 2, 1), (2, 3, 64, 33, 47, 3, 2, 1)]   # stem: direct dot2 dgrad with planar dx


@pytest.mark.parametrize("shape", GENERIC, ids=lambda s: "x".join(map(str, s)))
def test_conv_f32_exact_mfma(shape):
    r = _case(torch.float32, *shape)
    assert max(r.values()) < 1e-5, r


@pytest.mark.parametrize("shape", GENERIC + PIPELINED, ids=lambda s: "x".join(map(str, s)))
def test_conv_bf16(shape):
    r = _case(torch.bfloat16, *shape)
    assert max(r.values()) < 1e-2, r


# the large-tile kernels only take layers that fill the chip (>= 192 tiles of 256 x 256, >= 256 tiles of 256 x 128 / 256 x 64,
# >= 16384 pixels for the weight gradient): shape, expected kernel of (forward, data gradient, weight gradient)
ROUTED = [
    ((32, 256, 256, 40, 40, 3, 1, 1), ("v4::conv_kernel", "v4::conv_kernel", "wg4::wgrad_kernel")),
    ((31, 192, 232, 41, 43, 3, 1, 1), ("v4::conv_kernel", None, "wg4::wgrad_kernel")),                         # ragged pixels / channels
    ((13, 512, 256, 80, 80, 1, 1, 0), ("v4::conv_kernel", "v5::conv_kernel<128>", "wg4::wgrad_kernel")),        # 1x1: K = 512 / K = 256
    ((13, 128, 256, 80, 80, 1, 1, 0), ("v5::conv_kernel<128>", None, None)),                                    # K = 128: two co-resident blocks
    ((20, 256, 512, 80, 80, 3, 2, 1), ("v4::conv_kernel", "v4::conv_kernel", "wg4::wgrad_kernel")),            # stride 2 (dgrad: the four parity classes in ONE conv_v4 launch)
    ((9, 256, 320, 81, 79, 3, 2, 1), (None, "v4::conv_kernel", None)),                                          # ... odd extents: classes of different grid sizes, ragged channels
    ((40, 512, 512, 40, 40, 3, 2, 1), (None, "v4::conv_kernel", None)),                                         # ... two channel tiles; 100 row tiles per class
    ((10, 128, 256, 161, 159, 3, 2, 1), (None, "v5::conv_kernel<128>", None)),                                  # ... on conv_v5 (128 gradient channels), odd extents
    ((5, 64, 128, 320, 320, 3, 2, 1), (None, "v5::conv_kernel<64>", None)),                                     # ... 64 gradient channels
    ((24, 128, 128, 80, 80, 3, 1, 1), ("v5::band_kernel<128>", "v5::band_kernel<128>", "wg3::wgrad_kernel<128>")), # 3x3 s1 p1: activation band; band weight gradient
    ((128, 128, 128, 33, 16, 3, 1, 1), ("v5::band_kernel<128>", "v5::band_kernel<128>", None)),                 # narrowest image the band takes
    ((120, 64, 64, 33, 17, 3, 1, 1), ("v5::band_kernel<64>", "v5::band_kernel<64>", None)),                     # ragged: a tile spans 15 image rows
    ((20, 128, 128, 57, 61, 3, 1, 2, 2), ("v5::conv_kernel<128>", "v5::conv_kernel<128>", None)),               # dilated, ragged
    ((20, 96, 224, 57, 61, 3, 1, 1), ("v5::band_kernel<128>", None, "wg4::wgrad_kernel")),                      # 96 = 3 x 32 source channels, ragged
    ((12, 64, 64, 160, 160, 3, 1, 1), ("v5::band_kernel<64>", "v5::band_kernel<64>", "wg3::wgrad_kernel<64>")),
    # outputs beyond 128 MiB: the plain epilogue stores of conv_v4 / conv_v5 / the band kernel go non-temporal (conv_epilogue.h: store_rows)
    ((42, 64, 64, 160, 160, 3, 1, 1), ("v5::band_kernel<64>", "v5::band_kernel<64>", "wg3::wgrad_kernel<64>")),
    ((42, 256, 256, 80, 80, 3, 1, 1), ("v4::conv_kernel", "v4::conv_kernel", "wg4::wgrad_kernel")),
    ((44, 256, 64, 40, 40, 3, 1, 1), ("v5::band_kernel<64>", None, None)),                                      # Detect stem: 8 channel chunks
    ((8, 320, 128, 160, 160, 1, 1, 0), ("v5::conv_kernel<128>", None, None)),
    ((6, 64, 256, 63, 65, 5, 1, 2), (None, None, "wg4::wgrad_kernel")),                                         # 5x5 taps in the mixed-radix walk
    ((8, 1024, 256, 48, 48, 1, 1, 0), (None, None, "wg4::wgrad_kernel")),                                       # pointwise variant
    ((16, 256, 64, 80, 80, 3, 1, 1), (None, None, "wg4::wgrad_kernel")),                                        # quarter-filled tile, long pixel loop
    ((12, 320, 128, 80, 80, 1, 1, 0), (None, None, "wg4::wgrad_kernel")),                                       # half-filled, K = 320 (2 tiles, 37 % pad)
    # band weight gradient on rows too wide for its LDS row buffers (1280x1280 inputs, BASELINE configs[4]): two column strips with halo
    ((6, 128, 128, 160, 160, 3, 1, 1), (None, None, "wg3::wgrad_kernel<128>")),
    ((2, 64, 64, 320, 320, 3, 1, 1), (None, None, "wg3::wgrad_kernel<64>")),
    ((6, 128, 128, 150, 171, 3, 1, 1), (None, None, "wg3::wgrad_kernel<128>")),                                  # ragged second strip
]


@pytest.mark.parametrize("shape,want", ROUTED, ids=lambda v: "x".join(map(str, v)) if isinstance(v[0], int) else None)
def test_conv_bf16_large_tile_routes(shape, want):
    """Every large-tile kernel (conv_v4 256x256, conv_v5 256x128 / 256x64, wgrad_v4) on shapes big enough for the dispatch to pick
    it -- and the test checks that it did (the symbol each C-ABI entry reports), so a routing change cannot silently move these
    shapes back onto the older kernels.  Same bound as test_conv_bf16."""
    routes = {}
    r = _case(torch.bfloat16, *shape, routes=routes)
    assert max(r.values()) < 1e-2, (r, routes)
    for entry, sym in zip(("dy_conv2d_fwd", "dy_conv2d_dgrad", "dy_conv2d_wgrad"), want):
        if sym is not None:
            assert routes.get(entry, "").startswith(sym), (entry, sym, routes)


# + the direct dot2 kernels in f16 (v_dot2_f32_f16): thin 1x1 dgrad, stem dgrad with planar dx
F16_SHAPES = GENERIC[:9] + PIPELINED[::3] + [r[0] for r in ROUTED[:10]] + [r[0] for r in ROUTED[-3:-1]] + [
    (3, 256, 8, 37, 23, 1, 1, 0), (2, 3, 16, 64, 64, 3, 2, 1), (2, 3, 64, 33, 47, 3, 2, 1)]


@pytest.mark.parametrize("shape", F16_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_conv_f16(shape):
    """IEEE half (BASELINE configs[4]: the reference's AMP dtype) through the same dispatch: the generic kernel and every pipelined
    kernel are instantiated for the f16 MFMA.  11-bit mantissa: bound 2e-3 * max|ref| (bf16: 1e-2)."""
    r = _case(torch.float16, *shape)
    assert max(r.values()) < 2e-3, r


THIN = [(12, 16, 32, 41, 39, 3, 2, 1), (5, 32, 24, 37, 35, 3, 1, 1), (4, 16, 16, 40, 40, 1, 1, 0), (6, 32, 16, 36, 38, 3, 2, 1),
        (3, 48, 16, 33, 30, 3, 1, 1), (4, 32, 32, 40, 40, 3, 1, 1), (4, 48, 32, 40, 40, 1, 1, 0), (4, 16, 16, 48, 40, 3, 1, 1)]


@pytest.mark.parametrize("shape", THIN, ids=lambda s: "x".join(map(str, s)))
def test_conv_bf16_thin_route_all_shapes(shape):
    """Every instantiation of the thin-layer kernel (16 / 32 / 48 source channels, forward and stride-1 data gradient): the default
    dispatch only routes the shapes where it wins, DY_CONV_THIN_ALL=1 (read once per process) routes all eligible ones."""
    import os
    if os.environ.get("DY_CONV_THIN_ALL") is None:
        pytest.skip("runs in the DY_CONV_THIN_ALL=1 subprocess of test_thin_kernel_subprocess")
    r = _case(torch.bfloat16, *shape)
    assert max(r.values()) < 1e-2, r


def test_thin_kernel_subprocess():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DY_CONV_THIN_ALL="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_conv_kernels.py"), "-q", "-x", "-k",
                        "thin_route_all_shapes", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dtype,B,Cout,H,W", [(torch.bfloat16, 2, 64, 64, 96), (torch.bfloat16, 3, 16, 33, 47), (torch.float16, 2, 32, 40, 41),
                                             (torch.bfloat16, 1, 64, 641, 17)], ids=lambda v: str(v).replace("torch.", ""))
def test_stem_forward_kernel(dtype, B, Cout, H, W):
    """The direct MFMA kernel of the network stem (Conv(3, c, 3, 2) on the zero-padded NHWC8 image, raw output + BatchNorm sums -- the
    training forward of yolov8*.yaml layer 1): output against F.conv2d on the rounded operands, per-channel sum / sum of squares of the
    f32 accumulators against the same reference, odd extents (padding on all four borders, a ragged last 16-pixel group)."""
    import ctypes as C
    from dedark_yolo_amd import _C, ops
    from dedark_yolo_amd.ops import ptr, stream
    torch.manual_seed(H * 7 + W)
    x = torch.randn(B, 3, H, W, device="cuda")
    w = torch.randn(Cout, 3, 3, 3, device="cuda") * 0.2
    xn = ops.as_nhwc(x, dtype)                                   # [B, 8 (3 used), H, W] view of an NHWC8 buffer
    assert ops.padded_channels(xn) == 8
    wp = ops._pack(w, Cout, 8, False, dtype)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = ops.empty_nhwc(B, Cout, Ho, Wo, dtype, x.device)
    stats = torch.zeros(_C.STATS_REPLICAS * 2 * Cout, dtype=torch.float64, device="cuda")
    d = ops._conv_desc(xn, wp, y, B, H, W, 8, Ho, Wo, Cout, 3, 3, 2, 1, 1, None, None, 0, stats, False, dtype)
    _C.lib().dy_clear_last_kernel()
    _C.call("dy_conv2d_fwd", C.byref(d), stream())
    torch.cuda.synchronize()
    assert _C.lib().dy_last_kernel().decode() == "stem_fwd_kernel"
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), None, 2, 1)
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3
    assert _err(y.float(), ref) < tol
    t = stats.view(_C.STATS_REPLICAS, 2, Cout).sum(0)
    s1, s2 = ref.double().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))
    assert float((t[0] - s1).abs().max()) <= 1e-4 * float(ref.abs().double().sum((0, 2, 3)).max())
    assert float((t[1] - s2).abs().max()) <= 1e-4 * float(s2.max())


@pytest.mark.parametrize("dtype,K,N,mode", [(torch.bfloat16, 128, 128, "stats"), (torch.bfloat16, 320, 128, "stats"), (torch.float16, 64, 64, "stats"),
                                          (torch.bfloat16, 128, 320, "plain"), (torch.bfloat16, 128, 320, "acc"), (torch.float16, 128, 128, "add"),
                                          (torch.bfloat16, 64, 128, "add"),
                                          # K = 256: output channels in chunks of 256 (the wide data gradients of the 80x80 stage; 640 = 2.5 chunks)
                                          (torch.bfloat16, 256, 1024, "plain"), (torch.bfloat16, 256, 640, "acc"), (torch.float16, 256, 256, "add")],
                         ids=lambda v: str(v).replace("torch.", ""))
def test_px1x1_kernel(dtype, K, N, mode):
    """The pixel-streaming 1x1 kernel (weights resident in LDS, pixel operands straight from memory; the short-K layers of the 160x160
    stage and their data gradients): y[m][n] = sum_k x[m][k] w[n][k] against torch on the rounded operands, with BatchNorm sums (forward),
    accumulating into the destination and with an addend view (data gradients), on a pixel count that is no multiple of 32."""
    import ctypes as C
    from dedark_yolo_amd import _C, ops
    from dedark_yolo_amd.ops import ptr, stream
    torch.manual_seed(K + N)
    B, H, W = 7, 197, 191                                       # 263,389 pixels
    x = torch.randn(B, K, H, W, device="cuda") * 0.5
    w = torch.randn(N, K, 1, 1, device="cuda") * (1.0 / K ** 0.5)
    xn = ops.as_nhwc(x, dtype)
    wp = ops._pack(w, N, K, False, dtype)
    y = ops.empty_nhwc(B, N, H, W, dtype, x.device)
    old = torch.randn(B, N, H, W, device="cuda")
    addend = ops.as_nhwc(torch.randn(B, N, H, W, device="cuda"), dtype)
    if mode == "acc":
        y.copy_(old.to(dtype))
    stats = torch.zeros(_C.STATS_REPLICAS * 2 * N, dtype=torch.float64, device="cuda") if mode == "stats" else None
    d = ops._conv_desc(xn, wp, y, B, H, W, K, H, W, N, 1, 1, 1, 0, 1, None, None, 0, stats, mode == "acc", dtype)
    if mode == "add":
        d.add_src, d.add_src_ld = addend.data_ptr(), ops.ld_of(addend)
    _C.lib().dy_clear_last_kernel()
    _C.call("dy_conv2d_fwd" if mode == "stats" else "dy_conv2d_dgrad", C.byref(d), stream())
    torch.cuda.synchronize()
    assert _C.lib().dy_last_kernel().decode() == "px1x1_kernel"
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float())
    raw = ref
    if mode == "acc":
        ref = ref + old.to(dtype).float()
    if mode == "add":
        ref = ref + addend.float()
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3
    assert _err(y.float(), ref) < tol
    if mode == "stats":
        t = stats.view(_C.STATS_REPLICAS, 2, N).sum(0)
        s1, s2 = raw.double().sum((0, 2, 3)), (raw.double() ** 2).sum((0, 2, 3))
        assert float((t[0] - s1).abs().max()) <= 1e-4 * float(raw.abs().double().sum((0, 2, 3)).max())
        assert float((t[1] - s2).abs().max()) <= 1e-4 * float(s2.max())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plain", "acc", "add"])
def test_big_output_epilogue_non_temporal_stores(mode):
    """An output beyond 128 MiB leaves the tiled kernels through non-temporal stores (conv_epilogue.h: store_rows, bit 1 of its
    `accumulate` argument) -- also when the epilogue adds into the destination (fan-out data gradients, `accumulate`) or adds a second
    view on the way out (`add_src`, the Bottleneck shortcut gradient).  1x1 512 -> 256 at 80x80, B = 42: 137.6 MB of output."""
    import ctypes as C
    from dedark_yolo_amd import _C, ops
    from dedark_yolo_amd.ops import stream
    dtype = torch.bfloat16
    torch.manual_seed(5)
    B, K, N, H, W = 42, 512, 256, 80, 80
    assert B * H * W * N * 2 > (128 << 20)
    x = torch.randn(B, K, H, W, device="cuda") * 0.5
    w = torch.randn(N, K, 1, 1, device="cuda") * (1.0 / K ** 0.5)
    xn = ops.as_nhwc(x, dtype)
    wp = ops._pack(w, N, K, False, dtype)
    y = ops.empty_nhwc(B, N, H, W, dtype, x.device)
    old = torch.randn(B, N, H, W, device="cuda")
    addend = ops.as_nhwc(torch.randn(B, N, H, W, device="cuda"), dtype)
    if mode == "acc":
        y.copy_(old.to(dtype))
    d = ops._conv_desc(xn, wp, y, B, H, W, K, H, W, N, 1, 1, 1, 0, 1, None, None, 0, None, mode == "acc", dtype)
    if mode == "add":
        d.add_src, d.add_src_ld = addend.data_ptr(), ops.ld_of(addend)
    _C.lib().dy_clear_last_kernel()
    _C.call("dy_conv2d_dgrad", C.byref(d), stream())
    torch.cuda.synchronize()
    assert _C.lib().dy_last_kernel().decode().startswith(("v4::conv_kernel", "v5::conv_kernel"))
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float())
    if mode == "acc":
        ref = ref + old.to(dtype).float()
    if mode == "add":
        ref = ref + addend.float()
    assert _err(y.float(), ref) < 1e-2
