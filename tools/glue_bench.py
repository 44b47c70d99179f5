#!/usr/bin/env python3
"""Launch time and HBM rate of the plane-wise glue kernels (development tool, GPU only).
    python tools/glue_bench.py [B C H W]       (H, W: the LARGE side of the pool / upsample pair)"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ops = importlib.import_module("motion-style-transfer_amd.ops")
L = ops.L
lib = ops._lib()
B, C, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (32, 32, 256, 256)
dev = torch.device("cuda:0")
torch.manual_seed(0)
big = torch.randn(B, C, H, W, device=dev)
big2, big3, bigo = torch.randn_like(big), torch.randn_like(big), torch.empty_like(big)
small = torch.randn(B, C, H // 2, W // 2, device=dev)
smallo = torch.empty_like(small)
N = B * C
st = ops._stream


def timeit(name, fn, nbytes):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:18s} {B}x{C}x{H}x{W}: {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s")


nb, ns = big.numel() * 4, small.numel() * 4
timeit("maxpool2_fwd", lambda: L.check(lib.ynet_maxpool2_fwd(big.data_ptr(), smallo.data_ptr(), N, H, W, st()), lib), nb + ns)
timeit("maxpool2_bwd_add", lambda: L.check(lib.ynet_maxpool2_bwd_add(big.data_ptr(), small.data_ptr(), big2.data_ptr(), big3.data_ptr(),
                                                                     bigo.data_ptr(), N, H, W, 0, st()), lib), 4 * nb + ns)
timeit("maxpool2_bwd_add+relu", lambda: L.check(lib.ynet_maxpool2_bwd_add(big.data_ptr(), small.data_ptr(), big2.data_ptr(), big3.data_ptr(),
                                                                          bigo.data_ptr(), N, H, W, 1, st()), lib), 4 * nb + ns)
timeit("upsample2x_fwd", lambda: L.check(lib.ynet_upsample2x_fwd(small.data_ptr(), bigo.data_ptr(), N, H // 2, W // 2, st()), lib), nb + ns)
timeit("upsample2x_bwd", lambda: L.check(lib.ynet_upsample2x_bwd(big.data_ptr(), smallo.data_ptr(), N, H // 2, W // 2, st()), lib), nb + ns)
S = 1050
tmpl = torch.randn(S, S, device=dev)
xy = (torch.rand(B * 12, 2, device=dev) * 200 + 20).contiguous()
pout = torch.empty(B * 12, H, W, device=dev)
stat = torch.zeros(1, dtype=torch.int32, device=dev)
timeit("gather_patch", lambda: L.check(lib.ynet_gather_patch(tmpl.data_ptr(), S, S, xy.data_ptr(), pout.data_ptr(), B * 12, H, W,
                                                             stat.data_ptr(), st()), lib), pout.numel() * 4)
# fused predictor + BCE (+ predictor dgrad) against the three launches it replaces
for cout in (12, 30):
    x32 = torch.randn(B, 32, H, W, device=dev).relu_()
    w = torch.randn(cout, 32, 1, 1, device=dev) * 0.2
    bias = torch.randn(cout, device=dev) * 0.1
    tgt = torch.rand(B, cout, H, W, device=dev) * 0.01
    wp, wpd = ops.pack_weight(w, 0), ops.pack_weight(w, 1)
    y, dxo, dyo = torch.empty(B, cout, H, W, device=dev), torch.empty_like(x32), torch.empty(B, cout, H, W, device=dev)
    loss = torch.empty((), device=dev)
    ws = torch.zeros(lib.ynet_pred_bce_workspace_bytes() // 8 + 1, device=dev, dtype=torch.float64)
    planes = (32 + cout + cout + 32) * B * H * W * 4
    timeit(f"pred_bce cout={cout}", lambda: L.check(lib.ynet_pred_bce(x32.data_ptr(), 32 * H * W, wp.data_ptr(), bias.data_ptr(), tgt.data_ptr(), y.data_ptr(),
                                                                    loss.data_ptr(), dxo.data_ptr(), None, ws.data_ptr(), B, 32, cout, H * W, 1000.0, 0, st()), lib), planes)
    timeit(f"pred_bce cout={cout} + relu mask of dx", lambda: L.check(lib.ynet_pred_bce(x32.data_ptr(), 32 * H * W, wp.data_ptr(), bias.data_ptr(), tgt.data_ptr(), y.data_ptr(),
                                                                    loss.data_ptr(), dxo.data_ptr(), None, ws.data_ptr(), B, 32, cout, H * W, 1000.0, 1, st()), lib), planes)
    bws = torch.empty(lib.ynet_bce_workspace_bytes() // 8, device=dev, dtype=torch.float64)

    def unfused():
        ops.conv2d_raw([(x32.data_ptr(), 32, 32 * H * W)], None, wp, bias, [(y.data_ptr(), cout, cout * H * W)], B, H, W, 1, False)
        L.check(lib.ynet_bce_logits_fwd_grad(y.data_ptr(), tgt.data_ptr(), y.numel(), 1000.0, loss.data_ptr(), dyo.data_ptr(), bws.data_ptr(), st()), lib)
        ops.conv2d_raw([(dyo.data_ptr(), cout, cout * H * W)], None, wpd, None, [(dxo.data_ptr(), 32, 32 * H * W)], B, H, W, 1, False)
    timeit(f"unfused   cout={cout}", unfused, (32 + cout + 3 * cout + cout + 32) * B * H * W * 4)
