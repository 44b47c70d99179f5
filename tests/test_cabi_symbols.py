"""The C-ABI library loads on a CPU-only host and exports every symbol include/ynet_hip.h declares."""
import ctypes

from conftest import pkg


def test_library_exports_header():
    L = pkg("_lib")
    names = L.header_symbols()
    assert len(names) >= 20 and "ynet_conv2d" in names
    assert set(names) == set(L.SIGNATURES), set(names) ^ set(L.SIGNATURES)
    lib = ctypes.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ynet_hip.h but not exported"


def test_load_binds_and_reports_errors():
    L = pkg("_lib")
    lib = L.load()
    assert lib.ynet_abi_version() == 1
    assert lib.ynet_packed_weight_floats(32, 14, 3, 0) == (16 + 16) * 9 * 64      # rows: cin -> 16, + one chunk of slack
    assert lib.ynet_packed_weight_floats(32, 14, 3, 1) == (32 + 16) * 9 * 64
    # argument validation happens before any launch, so it is testable without a GPU
    rc = lib.ynet_pack_weight(None, None, 4, 4, 3, 0, None)
    assert rc != 0 and b"null" in lib.ynet_last_error()
    rc = lib.ynet_softargmax2d(None, None, 1, 1, 1, 8, 8, None)
    assert rc != 0


def test_ops_refuse_host_tensors():
    import pytest
    import torch
    ops = pkg("ops")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.max_pool2(torch.zeros(1, 1, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv2d(torch.zeros(1, 2, 8, 8), torch.zeros(3, 2, 3, 3), None, True, {})
    with pytest.raises(ValueError):
        ops.softargmax2d(torch.zeros(4, 4))


def test_argument_validation_without_gpu():
    """Every entry point validates its arguments before touching the device (status != 0 + message)."""
    import ctypes
    L = pkg("_lib")
    lib = L.load()
    vp = ctypes.c_void_p
    one = (vp * 1)(vp(16))
    ci = (ctypes.c_int * 1)(4)
    cl = (ctypes.c_longlong * 1)(64)
    pp = ctypes.cast(one, L.PP)
    # 5 sources: too many
    assert lib.ynet_conv2d(pp, ci, cl, None, 5, None, 0, vp(16), None, pp, ci, cl, 1, 1, 8, 8, 3, 0, None, 0, None) != 0
    assert b"sources" in lib.ynet_last_error()
    # kernel size 7 is not supported
    assert lib.ynet_conv2d(pp, ci, cl, None, 1, None, 0, vp(16), None, pp, ci, cl, 1, 1, 8, 8, 7, 0, None, 0, None) != 0
    assert b"kernel size" in lib.ynet_last_error()
    # a ReLU mask needs a single source
    two_p = ctypes.cast((vp * 2)(vp(16), vp(32)), L.PP)
    two_c, two_b = (ctypes.c_int * 2)(2, 2), (ctypes.c_longlong * 2)(128, 128)
    assert lib.ynet_conv2d(two_p, two_c, two_b, None, 2, vp(16), 0, vp(16), None, pp, ci, cl, 1, 1, 8, 8, 3, 0, None, 0, None) != 0
    assert lib.ynet_avgpool_pyramid(vp(16), pp, 1, 1, 48, 64, None) != 0 and b"multiples of 32" in lib.ynet_last_error()
    assert lib.ynet_avgpool_pyramid(vp(16), pp, 6, 1, 64, 64, None) != 0
    assert lib.ynet_sigmoid_temp(vp(16), vp(16), 1, 4, 64, (ctypes.c_int * 1)(9), 1, 1.0, None) != 0
    assert b"out of range" in lib.ynet_last_error()
    assert lib.ynet_sigmoid_temp(vp(16), vp(16), 1, 4, 64, (ctypes.c_int * 1)(0), 1, 0.0, None) != 0
    assert lib.ynet_gather_patch(vp(16), 16, 16, vp(16), vp(16), 1, 32, 32, vp(16), None) != 0
    assert lib.ynet_bce_logits_fwd(vp(4), vp(16), 8, vp(16), vp(16), None) != 0 and b"aligned" in lib.ynet_last_error()
    assert lib.ynet_conv2d_wgrad_workspace_floats(2, 16, 32, 8, 4, 3) > 0
    assert lib.ynet_conv2d_workspace_floats(32, 8, 8, 128) == 8 * 32 * 128 * 64
    assert lib.ynet_conv2d_workspace_floats(32, 256, 256, 32) == 0
    plan = lib.ynet_conv2d_plan(32, 256, 256, 32, 3)
    assert plan & 255 == 4 and (plan >> 8) & 255 == 2 and (plan >> 16) & 1 == 1      # 4 rows, two 16-wide tiles
    assert (plan >> 21) == 4 and (lib.ynet_conv2d_plan(32, 16, 16, 64, 3) >> 21) == 8   # chunk depth: large tiles 4, small tiles 8


def test_conv2d_auto_plans_the_benchmarked_layers_without_a_gpu():
    """ynet_conv2d_auto_plan is host code (csrc/conv_auto.cpp): which kernel family the dispatcher takes for the layers of a C2 step (B 32, 256^2;
    models/ynet.py:192-211,420-467) is decided -- and checked here -- without a device.  Families: 0 implicit GEMM, 1 conv_wino_kernel,
    2 conv_wino_cat_kernel, 3 conv_wino16_kernel, 4 / 5 the up-convolution forms."""
    import ctypes
    L = pkg("_lib")
    lib = L.load()

    def plan(B, H, W, srcs, dsts, K=3, relu=True, flags=0, **ops):
        d, tk = L.ConvAuto(), L.ConvTaken()
        d.nsrc, d.ndst = len(srcs), len(dsts)
        for i, c in enumerate(srcs):
            d.src[i], d.src_c[i], d.src_bs[i] = 256, c, c * H * W
        for i, c in enumerate(dsts):
            d.dst[i], d.dst_c[i], d.dst_bs[i] = (256 if c > 0 else None), abs(c), abs(c) * H * W
        d.wp, d.B, d.H, d.W, d.K, d.relu, d.flags = 256, B, H, W, K, int(relu), flags
        for k, v in ops.items():
            setattr(d, k, v)
        assert lib.ynet_conv2d_auto_plan(ctypes.byref(d), ctypes.byref(tk)) == 0, lib.ynet_last_error()
        need = lib.ynet_conv2d_auto_cache_floats(ctypes.byref(d))
        assert (need > 0) == (tk.family != 0)
        return tk.family, tk.variant, tk.nlaunch

    B = 32
    assert plan(B, 256, 256, [6, 8], [32], pooled=256)[0] == 2                       # encoder.0 + ReLU + MaxPool (conv_wino_cat_kernel<2, 3 | 6>)
    assert plan(B, 128, 128, [32], [32])[0] == 1                                     # encoder stage 1
    assert plan(B, 64, 64, [32], [64])[0] == 3 and plan(B, 64, 64, [64], [64], pooled=256)[0] == 3      # stage 2: the slice form
    assert plan(B, 32, 32, [64], [64])[0] == 0 and plan(B, 8, 8, [64], [128])[0] == 0                   # small maps: implicit GEMM
    assert plan(B, 256, 256, [32], [16], relu=False, upsample2x=1)[0] == 4          # bilinear x2 + upsample_conv[4]
    assert plan(B, 128, 128, [64], [32], relu=False, upsample2x=1)[0] == 5
    assert plan(B, 256, 256, [16, 32, 1], [32])[0] == 2                              # decoder[4][0] over cat(up, skip, way-point map)
    assert plan(B, 128, 128, [32, 32, 1], [32])[:2] == (2, 40)                       # 65 channels: 32 first, the rest added in place
    assert plan(B, 256, 256, [32], [32])[0] == 1                                     # decoder[4][2]
    assert plan(B, 256, 256, [32], [32], relu=False, relu_of=256) == (1, 21, 1)      # its data gradient through the ReLU backward
    assert plan(B, 256, 256, [32], [16, 32, -1], relu=False) == (1, 23, 1)           # decoder[4][0]'s data gradient: 16 + 32 in ONE launch (ynet_conv2d_winograd_split), the way-point map's not wanted
    assert plan(B, 256, 256, [32], [32, 16, -1], relu=False) == (1, 21, 2)           # (the other order: two launches)
    assert plan(B, 128, 128, [32], [64], relu=False) == (3, 22, 1)                   # a 64-channel destination: one launch of the slice form
    assert plan(B, 256, 256, [32], [12], K=1, relu=False)[0] == 0                    # the 1x1 predictor
    assert plan(B, 256, 256, [32], [32], flags=L.AUTO_NO_WINOGRAD)[0] == 0
    assert plan(10, 128, 128, [32], [32])[0] == 1 and plan(4, 64, 64, [32], [32])[0] == 0      # batch 10 still Winograd at 128^2; too few pixels -> implicit GEMM
