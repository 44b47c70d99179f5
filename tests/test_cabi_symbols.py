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
