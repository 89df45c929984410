"""ctypes binding of libdedark_yolo.so (the C-ABI declared in include/dedark_yolo.h).

The product path has NO fallback: if the HIP library is missing or a kernel call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DY_LIB_DIR (diagnostics: same-box A/B of two builds of the library, tools/gpu/step_ab.sh) overrides the in-tree directory
LIB_PATH = os.path.join(os.environ.get("DY_LIB_DIR") or os.path.join(_HERE, "lib"), "libdedark_yolo.so")

DY_F32, DY_BF16, DY_F16 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_LEAKY = 0, 1, 2
STATS_REPLICAS = 64          # DY_STATS_REPLICAS of include/dedark_yolo.h
BN_BWD_REPLICAS = 8          # DY_BN_BWD_REPLICAS

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float


class ConvDesc(C.Structure):
    _fields_ = [("src", vp), ("src_ld", i64), ("N", i32), ("Hs", i32), ("Ws", i32), ("Cs", i32), ("w", vp), ("dst", vp),
                ("dst_ld", i64), ("Hd", i32), ("Wd", i32), ("Cd", i32), ("KH", i32), ("KW", i32), ("stride", i32),
                ("pad", i32), ("dil", i32), ("scale", vp), ("shift", vp), ("act", i32), ("stats", vp), ("accumulate", i32),
                ("dtype", i32), ("dst_row_stride", i64), ("dst_img_stride", i64), ("KHf", i32), ("KWf", i32), ("kh0", i32),
                ("kh_step", i32), ("kw0", i32), ("kw_step", i32), ("dst_valid_channels", i32), ("dst_planar", vp),
                ("add_src", vp), ("add_src_ld", i64)]


class PackItem(C.Structure):
    _fields_ = [("w", vp), ("packed", vp), ("Cout", i32), ("Cout_pad", i32), ("Cin", i32), ("Cin_pad", i32), ("KH", i32),
                ("KW", i32), ("transposed", i32), ("dtype", i32), ("first_block", i64)]


class DetMaps(C.Structure):
    _fields_ = [("map", vp * 3), ("map_ld", i64 * 3), ("h", i32 * 3), ("w", i32 * 3), ("stride", f32 * 3), ("B", i32),
                ("nc", i32), ("n_levels", i32), ("dtype", i32)]


class AugSample(C.Structure):
    """dy_aug_sample (include/dedark_yolo.h)"""
    _fields_ = [("src", vp * 4), ("sh", i32 * 4), ("sw", i32 * 4), ("pitch", i64 * 4), ("rect", (i32 * 6) * 4), ("n_src", i32),
                ("canvas_h", i32), ("canvas_w", i32), ("hsv", i32), ("flipud", i32), ("fliplr", i32), ("minv", C.c_double * 6),
                ("lut", (C.c_uint8 * 256) * 3)]


_SIGS = {
    "dy_version": [],
    "dy_frontend_init": [],
    "dy_conv2d_fwd": [C.POINTER(ConvDesc), vp],
    "dy_conv2d_dgrad": [C.POINTER(ConvDesc), vp],
    "dy_conv2d_wgrad": [vp, i64, i32, i32, i32, i32, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, vp,
                        i32, vp],
    "dy_conv2d_wgrad_forked": [vp, vp, i64, i32, i32, i32, i32, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64,
                               vp, i32, vp],
    "dy_conv2d_bn_act_fwd": [C.POINTER(ConvDesc), i64, vp, vp, vp, vp, f32, f32, vp, i32, vp, i64, vp, i64, vp],
    "dy_bn_act_bwd": [vp, i64, vp, i64, vp, vp, i32, vp, vp, i64, vp, vp, i64, i32, i32, vp],
    "dy_pack_weight": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "dy_pack_item_blocks": [i32, i32, i32, i32],
    "dy_pack_weights_multi": [vp, i32, i64, vp],
    "dy_unpack_wgrad": [vp, vp, i32, i32, i32, i32, i32, vp],
    "dy_bn_finalize": [vp, i64, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, i32, vp],
    "dy_bn_fold_eval": [vp, vp, vp, vp, f32, vp, vp, i32, vp],
    "dy_bn_act_fwd": [vp, i64, vp, vp, i32, vp, i64, vp, i64, i64, i32, i32, vp],
    "dy_bn_act_bwd_reduce": [vp, i64, vp, i64, vp, vp, vp, vp, i32, i32, vp, i64, i32, i32, vp],
    "dy_bn_act_bwd_apply": [vp, i64, vp, i64, vp, vp, vp, vp, vp, i32, i32, vp, vp, i64, vp, vp, i64, i32, i32, vp],
    "dy_maxpool_fwd": [vp, i64, vp, i64, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "dy_maxpool_bwd": [vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "dy_upsample_nearest_fwd": [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, vp],
    "dy_upsample_nearest_bwd": [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp],
    "dy_copy2d": [vp, i64, vp, i64, i64, i32, i32, i32, vp],
    "dy_cast": [vp, i32, vp, i32, i64, vp],
    "dy_asff_fuse_fwd": [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i64, i32, i32, vp],
    "dy_asff_fuse_bwd": [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i64, i32, i32, i32,
                         i32, i32, vp],
    "dy_image_to_nhwc8": [vp, i32, i32, i32, vp, i32, i32, i32, vp],
    "dy_resize_bwd": [vp, i32, i32, i32, i32, i32, i32, vp, vp],
    "dy_filter_params_fwd": [vp, i32, vp, i32, vp],
    "dy_filter_params_bwd": [vp, i32, vp, vp, i32, vp],
    "dy_filters_pointwise_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "dy_filters_pointwise_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "dy_usm_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "dy_usm_bwd": [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "dy_loss_prepare_targets": [vp, vp, vp, i32, i32, i32, f32, f32, vp, vp, vp],
    "dy_loss_decode": [C.POINTER(DetMaps), vp, vp],
    "dy_tal_assign": [C.POINTER(DetMaps), vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "dy_tal_assign_decoded": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "dy_chan_moments": [vp, i64, i32, i64, i32, vp, i32, vp],
    "dy_sru_fwd": [vp, i64, vp, i64, i32, i64, i32, i32, vp, vp, vp, f32, i32, vp],
    "dy_sru_bwd": [vp, i64, vp, i64, vp, i64, i32, i64, i32, i32, vp, vp, vp, f32, vp, i32, vp],
    "dy_cru_fuse_fwd": [vp, i64, vp, i64, i32, i64, i32, vp, i32, vp],
    "dy_cru_fuse_bwd": [vp, i64, vp, i64, vp, i64, i32, i64, i32, vp, vp, i32, vp],
    "dy_bbox_ciou": [vp, vp, i64, vp, vp, vp],
    "dy_bbox_iou": [vp, vp, i64, i32, i32, f32, vp, vp, vp],
    "dy_aug_resize_u8": [vp, i32, i32, i64, vp, i32, i32, i64, vp],
    "dy_aug_letterbox": [vp, i32, i32, i64, i32, i32, i32, i32, i32, i32, vp, vp],
    "dy_aug_mosaic_warp": [vp, i32, i32, i32, vp, vp],
    "dy_dark_channel_prior": [vp, i32, i32, i32, vp, vp, vp],
    "dy_dfl_loss": [vp, vp, i64, vp, vp, vp],
    "dy_loss_fwd": [C.POINTER(DetMaps), vp, vp, vp, vp, vp, vp, vp],
    "dy_loss_finish": [vp, vp, f32, f32, f32, f32, i32, vp, vp, vp],
    "dy_loss_bwd": [C.POINTER(DetMaps), vp * 3, i64 * 3, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, vp],
    "dy_detect_decode": [C.POINTER(DetMaps), vp, vp],
    "dy_nms_candidates": [vp, i32, i32, i32, f32, i32, vp, vp, i64, vp],
    "dy_nms_sort": [vp, vp, vp, i32, i64, vp, vp, vp],
    "dy_nms_greedy": [vp, vp, vp, i32, i32, i32, i64, C.c_double, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp],
    "dy_preprocess_batch": [vp, vp, vp, f32, i32, i32, vp, i64, vp],
    "dy_sumsq": [vp, i64, vp, vp],
    "dy_sgd_step": [vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, i32, f32, vp, f32, f32, i64, vp],
    "dy_sgd_step_scaled": [vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, i32, f32, vp, f32, f32, vp, i64, vp],
    "dy_adamw_step_scaled": [vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, f32, f32, i32, f32, vp, f32, f32, vp, i64, vp],
    "dy_loss_scale_update": [vp, vp, f32, f32, i32, vp],
    "dy_adamw_step": [vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, f32, f32, i32, f32, vp, f32, f32, i64, vp],
    "dy_ema_lerp": [vp, vp, f32, i64, vp],
    "dy_grad_accumulate": [vp, vp, i64, vp],
    "dy_stream_fork": [vp, vp],
}

_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"dedark_yolo_amd: HIP library not built: {LIB_PATH} (run __graft_entry__.build()); "
                               "there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.dy_last_error.restype = C.c_char_p
        L.dy_last_error.argtypes = []
        L.dy_last_kernel.restype = C.c_char_p
        L.dy_last_kernel.argtypes = []
        L.dy_clear_last_kernel.restype = None
        L.dy_clear_last_kernel.argtypes = []
        for name, sig in _SIGS.items():
            fn = getattr(L, name)          # AttributeError if the .so does not export a declared symbol
            fn.argtypes = sig
            fn.restype = C.c_int64 if name == "dy_pack_item_blocks" else C.c_int
        _lib = L
    return _lib


def exported_symbols():
    return ["dy_last_error", "dy_last_kernel", "dy_clear_last_kernel"] + list(_SIGS)


_prof = None          # list of (name, start_event, end_event, meta, kernel symbol) while bench.py's per-kernel timing is active
_next_meta = None
_UNTIMED = frozenset(["dy_stream_fork"])        # stream plumbing: events around it would time the wait, not a kernel


def call(name, *args):
    global _next_meta
    L = lib()
    if _prof is None or name in _UNTIMED:
        rc = getattr(L, name)(*args)
    else:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L.dy_clear_last_kernel()
        e0.record()                      # kernels are launched on torch's current stream, so these events bracket them
        rc = getattr(L, name)(*args)
        e1.record()
        kern = L.dy_last_kernel().decode()          # GPU kernel symbol the entry launched ("" = the entry does not report one)
        _prof.append((name, e0, e1, _next_meta, kern))
        _next_meta = None
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {L.dy_last_error().decode()}")


def set_meta(**kw):
    """Algorithmic work of the NEXT call (flops / bytes), recorded only while profiling."""
    global _next_meta
    if _prof is not None:
        _next_meta = kw
