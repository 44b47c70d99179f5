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
