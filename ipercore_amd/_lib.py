"""ctypes binding of liblwg_hip.so (C ABI: include/lwg_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C ipercore_amd/csrc`` and travels with
the source tree.  There is NO fallback: if the shared object is missing or a symbol is absent, importing the
binding raises, and every op raises on non-CUDA tensors.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblwg_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lwg_hip.h")

LWG_MAX_TAPS = 52
EPI_NONE, EPI_RESIDUAL, EPI_SPADE = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID, ACT_LRELU = 0, 1, 2, 3, 4
ACT_RELU_MASK = 5       # conv launches with EPI_RESIDUAL only: y = res > 0 ? acc + bias : 0 (include/lwg_hip.h)
DT_F32, DT_BF16, DT_F32_Q4 = 0, 1, 2      # DT_F32_Q4: fp32 output written as channel-quad planes (B, C/4, H, W, 4)

c_f = ctypes.c_void_p  # device pointers travel as void*
c_i = ctypes.c_int


class LwgConvArgs(ctypes.Structure):
    """Mirror of ``struct LwgConvArgs`` (include/lwg_hip.h) - field order and types must match."""
    _fields_ = [
        ("x0", c_f), ("x1", c_f),
        ("C0", c_i), ("C1", c_i),
        ("B", c_i), ("H", c_i), ("W", c_i),
        ("OH", c_i), ("OW", c_i),
        ("M", c_i), ("stride", c_i), ("ntaps", c_i), ("cshift", c_i),
        ("w", c_f), ("N", c_i), ("bias", c_f),
        ("y", c_f), ("YH", c_i), ("YW", c_i), ("YC", c_i), ("ycoff", c_i),
        ("omul", c_i), ("ooy", c_i), ("oox", c_i),
        ("epi", c_i), ("act", c_i),
        ("res", c_f), ("xn", c_f), ("mean", c_f), ("rstd", c_f),
        ("xdt", c_i), ("ydt", c_i),
        ("dy", ctypes.c_byte * LWG_MAX_TAPS), ("dx", ctypes.c_byte * LWG_MAX_TAPS),
    ]


class LwgWinoDesc(ctypes.Structure):
    """Mirror of include/lwg_hip.h LwgWinoDesc (one record of lwg_winograd_panels_f32's device table)."""
    _fields_ = [("wpanel", ctypes.c_void_p), ("upk", ctypes.c_void_p), ("Cin", c_i), ("N", c_i), ("first_block", c_i), ("tap9", c_i * 9)]


class LwgPackDesc(ctypes.Structure):
    """Mirror of ``struct LwgPackDesc`` (include/lwg_hip.h): one panel of lwg_pack_panels_f32."""
    _fields_ = [("w", c_f), ("out", c_f)] + [(n, c_i) for n in ("D1", "KHW", "transposed", "ntaps", "cin", "cin_pad", "nout", "n_pad", "Kp",
                                                               "first_block")] + [("kidx", c_i * LWG_MAX_TAPS)]


_SIGS = {
    "lwg_abi_version": (c_i, []),
    "lwg_device_cu_count": (c_i, []),
    "lwg_conv2d_nhwc_f32": (c_i, [ctypes.POINTER(LwgConvArgs), c_f]),
    "lwg_conv2d_ws_floats": (ctypes.c_size_t, [ctypes.POINTER(LwgConvArgs)]),
    "lwg_conv2d_nhwc_f32_ws": (c_i, [ctypes.POINTER(LwgConvArgs), c_f, c_f]),
    "lwg_conv2d_nhwc_bf16": (c_i, [ctypes.POINTER(LwgConvArgs), c_f]),
    "lwg_conv2d_winograd_f32": (c_i, [ctypes.POINTER(LwgConvArgs), c_f]),
    "lwg_conv2d_winograd_ws_floats": (ctypes.c_size_t, [ctypes.POINTER(LwgConvArgs)]),
    "lwg_conv2d_winograd_f32_ws": (c_i, [ctypes.POINTER(LwgConvArgs), c_f, c_f]),
    "lwg_conv2d_winograd_plan": (c_i, [ctypes.POINTER(LwgConvArgs), c_i, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(c_i), ctypes.POINTER(c_i),
                                       ctypes.POINTER(c_i)]),
    "lwg_winograd_panel_f32": (c_i, [c_f, c_f, c_i, c_i, ctypes.POINTER(c_i), c_f]),
    "lwg_winograd_panels_f32": (c_i, [c_f, c_i, c_i, c_f]),
    "lwg_conv2d_winograd4_f32": (c_i, [ctypes.POINTER(LwgConvArgs), c_f]),
    "lwg_winograd4_panel_f32": (c_i, [c_f, c_f, c_i, c_i, ctypes.POINTER(c_i), c_f]),
    "lwg_conv2d_nhwc_bf16_hr": (c_i, [ctypes.POINTER(LwgConvArgs), c_f]),
    "lwg_conv2d_nhwc_c8_bf16": (c_i, [ctypes.POINTER(LwgConvArgs), c_f]),
    "lwg_conv_transpose4_nhwc_bf16": (c_i, [ctypes.POINTER(LwgConvArgs), c_f]),
    "lwg_conv_transpose4_nhwc_f32": (c_i, [ctypes.POINTER(LwgConvArgs), c_f]),
    "lwg_conv_transpose4_winograd_f32": (c_i, [ctypes.POINTER(LwgConvArgs), c_f]),
    "lwg_conv_transpose4_is_one_grid": (c_i, [ctypes.POINTER(LwgConvArgs)]),
    "lwg_conv_slice_count": (c_i, [ctypes.POINTER(LwgConvArgs)]),
    "lwg_conv2d_nhwc_f32_split": (c_i, [ctypes.POINTER(LwgConvArgs), c_f]),
    "lwg_conv2d_wgrad_ws_floats": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "lwg_conv2d_wgrad_nhwc_f32": (c_i, [ctypes.POINTER(LwgConvArgs), c_f, c_f, c_f, c_f]),
    "lwg_conv2d_wgrad_unpacked_f32": (c_i, [ctypes.POINTER(LwgConvArgs), c_f, c_f, c_f] + [c_i] * 5 + [ctypes.POINTER(ctypes.c_int), c_i, c_i, c_f, c_f]),
    "lwg_colsum_nhwc_f32": (c_i, [c_f, ctypes.c_size_t, c_i, c_f, c_f, c_f]),
    "lwg_act_bwd_f32": (c_i, [c_f, c_f, ctypes.c_size_t, c_i, c_f, c_f]),
    "lwg_norm_fwd_nhwc_f32": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "lwg_norm_bwd_nhwc_f32": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f]),
    "lwg_adam_step_f32": (c_i, [c_f, c_f, c_f, c_f, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_i, c_f]),
    "lwg_adam_step_dev_f32": (c_i, [c_f, c_f, c_f, c_f, ctypes.c_size_t, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_f, c_f]),
    "lwg_pack_panel_f32": (c_i, [c_f] + [c_i] * 5 + [ctypes.POINTER(ctypes.c_int)] + [c_i] * 5 + [c_f, c_f]),
    "lwg_pack_panels_f32": (c_i, [c_f, c_i, c_i, c_f]),
    "lwg_unpack_wgrad_f32": (c_i, [c_f] + [c_i] * 5 + [ctypes.POINTER(ctypes.c_int)] + [c_i] * 5 + [c_f, c_f]),
    "lwg_maxpool2_fwd_nhwc_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "lwg_maxpool2_bwd_nhwc_f32": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "lwg_crop_resize_bilinear_f32": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "lwg_crop_resize_bilinear_bwd_f32": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "lwg_prelu_f32": (c_i, [c_f, c_f, c_f, ctypes.c_size_t, c_i, c_f, c_f]),
    "lwg_prelu_bwd_f32": (c_i, [c_f, c_f, c_f, ctypes.c_size_t, c_i, c_f, c_f]),
    "lwg_instnorm_stats_nhwc_f32": (c_i, [c_f, c_i, c_i, c_i, ctypes.c_float, c_f, c_f, c_f, c_i, c_f]),
    "lwg_instnorm_apply_nhwc_f32": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "lwg_lwb_attention_f32": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "lwg_lwb_attention_bf16": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "lwg_instnorm_stats_nhwc_bf16": (c_i, [c_f, c_i, c_i, c_i, ctypes.c_float, c_f, c_f, c_f, c_i, c_f]),
    "lwg_head_compose_bf16": (c_i, [c_f, c_f, c_f, ctypes.c_size_t, c_i, c_i, c_i, c_f, c_f, c_f, c_f]),
    "lwg_up4_head_compose_bf16": (c_i, [ctypes.POINTER(LwgConvArgs), c_f, c_f, ctypes.c_size_t, c_f, c_f, c_f, c_f]),
    "lwg_flow_resize_f32": (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_f]),
    "lwg_lwb_fuse_f32": (c_i, [c_f] * 5 + [c_i] * 7 + [ctypes.c_float, ctypes.c_float, c_f]),
    "lwg_lwb_attention_x_f32": (c_i, [c_f] * 8 + [c_i] * 6 + [c_f]),
    "lwg_lwb_attention_x_bf16": (c_i, [c_f] * 8 + [c_i] * 6 + [c_f]),
    "lwg_lwb_attention_x_records": (c_i, [c_i, c_i, c_i, c_i]),
    "lwg_instnorm_finalize_ws_floats": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "lwg_instnorm_finalize_f32": (c_i, [c_f, c_i, c_i, c_i, ctypes.c_float, c_f, c_f, c_f]),
    "lwg_lwb_attention_bwd_f32": (c_i, [c_f] * 10 + [c_i] * 7 + [c_f]),
    "lwg_lwb_attention_kv_f32": (c_i, [c_f] * 6 + [c_i] * 7 + [c_f]),
    "lwg_lwb_attention_kv_bwd_f32": (c_i, [c_f] * 8 + [c_i] * 7 + [c_f]),
    "lwg_rasterize_ws_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "lwg_project_faces_f32": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, ctypes.c_float, c_f, c_f, c_f]),
    "lwg_rasterize_fim_wim_f32": (c_i, [c_f, c_i, c_i, c_i, ctypes.c_float, ctypes.c_float, c_f, c_f, c_f, c_f]),
    "lwg_texture_sample_f32": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, ctypes.c_float, ctypes.POINTER(ctypes.c_float), c_f, c_f]),
    "lwg_flow_compose_f32": (c_i, [c_f, c_f, c_i, c_i, c_f, c_i, c_f, c_f, c_i, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_f]),
    "lwg_bc_transform_f32": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f]),
    "lwg_encode_fim_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f]),
    "lwg_smpl_lbs_ws_floats": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "lwg_smpl_lbs_f32": (c_i, [c_f, c_i, c_f, c_i, c_i, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i,
                               c_i, c_f, c_f, c_f, c_f, c_f]),
    "lwg_head_compose_f32": (c_i, [c_f, c_f, c_f, ctypes.c_size_t, c_i, c_i, c_i, c_f, c_f, c_f, c_f]),
    "lwg_head_compose_q4_f32": (c_i, [c_f, c_f, c_f, ctypes.c_size_t, c_i, c_i, c_i, c_f, c_f, c_f, c_f]),
    "lwg_thin_conv_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f]),
    "lwg_nchw_to_nhwc_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "lwg_nhwc_to_nchw_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "lwg_frames_to_u8": (c_i, [c_f, c_i, c_i, c_i, c_f, c_f]),
    "lwg_morph_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "lwg_canny_f32": (c_i, [c_f, c_i, c_i, c_i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                            ctypes.c_float, ctypes.c_float, c_f, c_f, c_f]),
    "lwg_boundary_fill_f32": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_f]),
    "lwg_grid_sample_nchw_f32": (c_i, [c_f, ctypes.c_size_t, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "lwg_uv_merge_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f, c_f]),
    "lwg_uv_merge_parts_f32": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f, c_f]),
    "lwg_pack_inputs_f32": (c_i, [c_f, c_i, c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_f, c_f]),
}

_lib = None


def header_symbols():
    """Function names declared in include/lwg_hip.h (used by the CPU test that checks the exports)."""
    with open(HEADER_PATH) as fp:
        src = re.sub(r"/\*.*?\*/", "", fp.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(lwg_[a-z0-9_]+)\s*\(", src)))


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP kernels are not built (run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C ipercore_amd/csrc`).  There is no CPU fallback.")
        # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so) while this library is linked against /opt/rocm's: the
        # first one loaded serves every later request for that SONAME.  torch must win - its streams and allocations are what the
        # kernels run on; loaded the other way round (build() then smoke() in one process) every launch fails with hipErrorNoDevice.
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)      # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if handle.lwg_abi_version() != 10:
            raise RuntimeError("liblwg_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(err, what):
    if err != 0:
        raise RuntimeError(f"{what} failed with hipError_t {err}")
