"""ctypes binding of libynet_hip.so (the C ABI declared in include/ynet_hip.h).

There is NO fallback: if the library is missing or a symbol is absent, importing the ops fails
loudly with the build command.  Loading needs no GPU (the CPU test-suite checks every symbol).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YNET_HIP_LIB") or os.path.join(_HERE, "csrc", "libynet_hip.so")   # override: A/B builds (tools/)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ynet_hip.h")

c_fp = ctypes.c_void_p       # device pointers travel as plain addresses
c_i = ctypes.c_int
c_ll = ctypes.c_longlong
c_f = ctypes.c_float
PP = ctypes.POINTER(ctypes.c_void_p)
PI = ctypes.POINTER(ctypes.c_int)
PLL = ctypes.POINTER(ctypes.c_longlong)

class ConvAuto(ctypes.Structure):
    """YnetConvAuto of include/ynet_hip.h (the descriptor of ynet_conv2d_auto)."""
    _fields_ = [("src", c_fp * 4), ("src_c", c_i * 4), ("src_bs", c_ll * 4), ("src_bmod", c_i * 4), ("nsrc", c_i),
                ("mask", c_fp), ("mask_bs", c_ll), ("wp", c_fp), ("bias", c_fp),
                ("dst", c_fp * 4), ("dst_c", c_i * 4), ("dst_bs", c_ll * 4), ("ndst", c_i),
                ("B", c_i), ("H", c_i), ("W", c_i), ("K", c_i), ("relu", c_i), ("upsample2x", c_i),
                ("relu_of", c_fp), ("relu_of_bs", c_ll), ("pooled", c_fp), ("pooled_bs", c_ll), ("pool_code", c_fp),
                ("addend", c_fp), ("addend_bs", c_ll), ("addend_bmod", c_i),
                ("bits_out", c_fp), ("relu_bits", c_fp), ("wbits_out", c_fp), ("relu_wbits", c_fp),
                ("cache", c_fp), ("cache_floats", c_ll), ("cache_tag", ctypes.POINTER(ctypes.c_ulonglong)), ("wp_version", ctypes.c_ulonglong),
                ("workspace", c_fp), ("workspace_floats", c_ll), ("flags", ctypes.c_uint), ("dst_s2d", c_i * 4)]


class ConvTaken(ctypes.Structure):
    """YnetConvTaken: what ynet_conv2d_auto ran."""
    _fields_ = [("family", c_i), ("variant", c_i), ("nlaunch", c_i), ("tmpl", (c_i * 3) * 4), ("wrote_wbits", c_i), ("wrote_pool_code", c_i),
                ("transformed", c_i), ("wrote_s2d", c_i)]


AUTO_NO_WINOGRAD, AUTO_NO_WINOGRAD16, AUTO_WINOGRAD16_FOR_16, AUTO_NO_POOL_CODE, AUTO_NO_RELU_WBITS, AUTO_NO_SPLIT48 = 1, 2, 4, 8, 16, 32

# name -> (restype, argtypes); must list every function of include/ynet_hip.h
SIGNATURES = {
    "ynet_abi_version": (c_i, []),
    "ynet_last_error": (ctypes.c_char_p, []),
    "ynet_packed_weight_floats": (c_ll, [c_i, c_i, c_i, c_i]),
    "ynet_pack_weight": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_workspace_floats": (c_ll, [c_i, c_i, c_i, c_i]),
    "ynet_conv2d": (c_i, [PP, PI, PLL, PI, c_i, c_fp, c_ll, c_fp, c_fp, PP, PI, PLL, c_i,
                          c_i, c_i, c_i, c_i, c_i, c_fp, c_ll, c_fp]),
    "ynet_conv2d_pool_supported": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "ynet_conv2d_pool": (c_i, [PP, PI, PLL, c_i, c_fp, c_fp, c_fp, c_i, c_ll, c_fp, c_ll, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_dgrad_relu_supported": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "ynet_conv2d_dgrad_relu": (c_i, [c_fp, c_i, c_ll, c_fp, c_ll, c_fp, c_fp, c_i, c_ll, c_fp, c_ll, c_i, c_i, c_i, c_i, c_fp, c_ll, c_fp]),
    "ynet_conv2d_relu_bits_words": (c_ll, [c_i, c_i, c_i, c_i, c_i]),
    "ynet_conv2d_relu_bits": (c_i, [PP, PI, PLL, c_i, c_fp, c_fp, c_fp, c_i, c_ll, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_dgrad_relu_bits": (c_i, [c_fp, c_i, c_ll, c_fp, c_ll, c_fp, c_fp, c_i, c_ll, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_winograd_supported": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "ynet_winograd_filter_floats": (c_ll, [c_i, c_i]),
    "ynet_winograd_filter": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_winograd": (c_i, [c_fp, c_ll, c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_winograd_s2d": (c_i, [c_fp, c_ll, c_fp, c_fp, c_ll, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_winograd_pred_bce_supported": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "ynet_conv2d_winograd_pred_bce_blob": (c_i, [c_fp, c_ll, c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_ll, c_fp, c_i, c_i, c_i,
                                                 c_f, c_fp]),
    "ynet_conv2d_winograd_split_supported": (c_i, [c_i, c_i, c_i, c_i]),
    "ynet_conv2d_winograd_split": (c_i, [c_fp, c_ll, c_fp, c_fp, c_ll, c_i, c_fp, c_ll, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_upconv_dgrad_ring": (c_i, [c_fp, c_ll, c_fp, c_fp, c_ll, c_fp, c_ll, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_winograd_dgrad_relu": (c_i, [c_fp, c_ll, c_fp, c_fp, c_ll, c_fp, c_ll, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_winograd_cat_supported": (c_i, [c_i, c_i, c_i, PI, c_i, c_i, c_i]),
    "ynet_winograd_filter_cat_floats": (c_ll, [PI, c_i, c_i]),
    "ynet_winograd_filter_cat": (c_i, [c_fp, c_fp, PI, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_winograd_cat": (c_i, [PP, PI, PLL, c_i, c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_winograd_cat_add": (c_i, [PP, PI, PLL, c_i, c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_i, c_i, c_i, c_fp, c_ll, c_i, c_fp]),
    "ynet_conv2d_winograd_cat_pool": (c_i, [PP, PI, PLL, c_i, c_fp, c_fp, c_fp, c_ll, c_fp, c_ll, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_winograd_cat_pool_code": (c_i, [PP, PI, PLL, c_i, c_fp, c_fp, c_fp, c_ll, c_fp, c_ll, c_fp, c_i, c_i, c_i, c_fp]),
    "ynet_winograd_relu_bits_words": (c_ll, [c_i, c_i, c_i]),
    "ynet_conv2d_winograd_relu_bits": (c_i, [c_fp, c_ll, c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "ynet_conv2d_winograd_cat_relu_bits": (c_i, [PP, PI, PLL, c_i, c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_i, c_fp, c_ll, c_i, c_fp, c_fp]),
    "ynet_conv2d_winograd_dgrad_relu_bits": (c_i, [c_fp, c_ll, c_fp, c_fp, c_ll, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_upsample2x_conv2d_winograd_supported": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "ynet_upsample2x_conv2d_winograd": (c_i, [c_fp, c_ll, c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_winograd16_supported": (c_i, [c_i, c_i, c_i, PI, c_i, c_i, c_i]),
    "ynet_winograd16_filter_floats": (c_ll, [PI, c_i, c_i]),
    "ynet_winograd16_filter": (c_i, [c_fp, c_fp, PI, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_conv2d_winograd16": (c_i, [PP, PI, PLL, c_i, c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_i, c_i, c_i, c_fp, c_ll, c_fp, c_ll, c_i, c_fp, c_ll, c_fp]),
    "ynet_conv2d_auto_cache_floats": (c_ll, [ctypes.POINTER(ConvAuto)]),
    "ynet_conv2d_auto_workspace_floats": (c_ll, [ctypes.POINTER(ConvAuto)]),
    "ynet_conv2d_auto": (c_i, [ctypes.POINTER(ConvAuto), ctypes.POINTER(ConvTaken), c_fp]),
    "ynet_conv2d_auto_plan": (c_i, [ctypes.POINTER(ConvAuto), ctypes.POINTER(ConvTaken)]),
    "ynet_conv2d_add_supported": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "ynet_conv2d_add": (c_i, [PP, PI, PLL, PI, c_i, c_fp, c_fp, c_fp, c_i, c_ll, c_i, c_i, c_i, c_i, c_i, c_fp, c_ll, c_i, c_fp]),
    "ynet_conv2d_plan": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "ynet_conv2d_wgrad_workspace_floats": (c_ll, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "ynet_conv2d_wgrad": (c_i, [PP, PI, PLL, c_i, c_fp, c_ll, c_fp, c_ll, c_fp, c_fp, c_fp,
                                c_i, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_lora_compose": (c_i, [c_fp, c_fp, c_fp, c_f, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_lora_grad": (c_i, [c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_lora_conv2d_wgrad_supported": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "ynet_lora_conv2d_wgrad_preferred": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "ynet_lora_conv2d_wgrad_workspace_floats": (c_ll, [c_i, c_i]),
    "ynet_lora_conv2d_wgrad": (c_i, [PP, PI, PLL, c_i, c_fp, c_ll, c_fp, c_ll, c_fp, c_fp, c_f, c_fp, c_fp, c_fp,
                                     c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_lora_compose_pack": (c_i, [c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_lora_compose_pack_multi": (c_i, [c_i, PP, PP, PP, ctypes.POINTER(c_f), PP, PP, PI, PI, PI, PI, c_fp]),
    "ynet_batch_sum": (c_i, [c_fp, c_fp, c_i, c_ll, c_ll, c_fp]),
    "ynet_adam_step": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                             c_i, c_fp]),
    "ynet_maxpool2_fwd": (c_i, [c_fp, c_fp, c_ll, c_i, c_i, c_fp]),
    "ynet_maxpool2_bwd": (c_i, [c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_fp]),
    "ynet_maxpool2_bwd_add": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_i, c_fp]),
    "ynet_maxpool2_bwd_add_code": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_i, c_fp]),
    "ynet_upsample2x_fwd": (c_i, [c_fp, c_fp, c_ll, c_i, c_i, c_fp]),
    "ynet_upsample2x_bwd": (c_i, [c_fp, c_fp, c_ll, c_i, c_i, c_fp]),
    "ynet_upsample2x_bwd_relu": (c_i, [c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_fp]),
    "ynet_avgpool_pyramid": (c_i, [c_fp, PP, c_i, c_ll, c_i, c_i, c_fp]),
    "ynet_bce_workspace_bytes": (c_ll, []),
    "ynet_bce_logits_fwd": (c_i, [c_fp, c_fp, c_ll, c_fp, c_fp, c_fp]),
    "ynet_bce_logits_bwd": (c_i, [c_fp, c_fp, c_fp, c_fp, c_ll, c_fp]),
    "ynet_bce_logits_fwd_grad": (c_i, [c_fp, c_fp, c_ll, c_f, c_fp, c_fp, c_fp, c_fp]),
    "ynet_bce_grad_rescale": (c_i, [c_fp, c_fp, c_f, c_ll, c_fp]),
    "ynet_pred_bce_workspace_bytes": (c_ll, []),
    "ynet_pred_bce": (c_i, [c_fp, c_ll, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_ll, c_f, c_i, c_fp]),
    "ynet_pred_bce_blob": (c_i, [c_fp, c_ll, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_f, c_i, c_fp]),
    "ynet_softargmax2d": (c_i, [c_fp, c_fp, c_ll, c_i, c_ll, c_i, c_i, c_fp]),
    "ynet_train_readout": (c_i, [c_fp, c_ll, c_fp, c_ll, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_f, c_fp]),
    "ynet_pred_softargmax_supported": (c_i, [c_i, c_i, c_i, c_i]),
    "ynet_pred_softargmax_workspace_floats": (c_ll, [c_ll, c_i, c_i]),
    "ynet_pred_softargmax": (c_i, [c_fp, c_ll, c_fp, c_fp, c_fp, c_fp, c_ll, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_pad2d": (c_i, [c_fp, c_fp, c_ll, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_seg_onehot_pad": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_resize_nearest": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, ctypes.c_double, ctypes.c_double, c_fp]),
    "ynet_batchnorm_workspace_doubles": (c_ll, [c_i]),
    "ynet_batchnorm2d_fwd": (c_i, [c_fp] * 9 + [c_i, c_i, c_ll, c_i, ctypes.c_double, ctypes.c_double, c_fp]),
    "ynet_batchnorm2d_bwd": (c_i, [c_fp] * 9 + [c_i, c_i, c_ll, c_i, c_fp]),
    "ynet_add_relu": (c_i, [c_fp, c_fp, c_fp, c_ll, c_i, c_fp]),
    "ynet_relu_bwd": (c_i, [c_fp, c_fp, c_fp, c_ll, c_fp]),
    "ynet_rot90_flip": (c_i, [c_fp, c_fp, c_ll, c_i, c_i, c_i, c_i, c_fp]),
    "ynet_rot_coords": (c_i, [c_fp, c_ll] + [ctypes.c_double] * 8 + [c_fp]),
    "ynet_sigmoid_temp": (c_i, [c_fp, c_fp, c_ll, c_i, c_ll, PI, c_i, c_f, c_fp]),
    "ynet_gather_patch": (c_i, [c_fp, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp]),
    "ynet_heatmap_analytic": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, ctypes.c_double, c_fp, c_i, c_fp, c_fp]),
    "ynet_kmeans2d": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, ctypes.c_float, c_i, c_fp]),
    "ynet_comm_handle_bytes": (c_ll, []),
    "ynet_comm_create": (c_i, [c_i, c_i, c_ll, PP]),
    "ynet_comm_export": (c_i, [c_fp, c_fp]),
    "ynet_comm_connect": (c_i, [c_fp, c_fp]),
    "ynet_allreduce_sum": (c_i, [c_fp, c_fp, c_ll, c_fp]),
    "ynet_comm_status": (c_i, [c_fp]),
    "ynet_comm_destroy": (c_i, [c_fp]),
    "ynet_multinomial": (c_i, [c_fp, c_ll, c_ll, c_i, c_i, c_i, c_f, ctypes.c_ulonglong, c_fp, c_fp, c_fp]),
    "ynet_multinomial_devseed": (c_i, [c_fp, c_ll, c_ll, c_i, c_i, c_i, c_f, c_fp, c_fp, c_fp, c_fp]),
    "ynet_cws_prior": (c_i, [c_fp, c_ll, c_i, c_fp, c_fp, c_i, c_i, c_i, c_f, c_f, c_i, c_fp, c_fp, c_fp]),
}

_lib = None


def header_symbols():
    """Function names declared in include/ynet_hip.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ynet_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load the library and bind every symbol; raises RuntimeError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch FIRST: it brings its own HIP runtime (torch/lib/libamdhip64.so); a library loaded before it binds the system's one, and a process with two
    # HIP runtimes launches this library's kernels on a runtime that never saw torch's device ("no ROCm-capable device is detected" -- found by
    # running __graft_entry__.build() and smoke() in one process).  Loaded after torch the dependency resolves to the runtime already in the process.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built and there is no fallback path. "
            f"Build it with `make -C {os.path.dirname(LIB_PATH)}` (or `python -c 'import __graft_entry__ as g; g.build()'`).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load()
        raise RuntimeError("libynet_hip: " + lib.ynet_last_error().decode(errors="replace"))
