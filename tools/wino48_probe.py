"""Experiment: the 48-channel data gradient of a decoder's first convolution (32 -> [16 up-sampled, 32 skip]) as ONE Winograd launch with three
output blocks per wave (192 accumulator registers: four waves per workgroup, one per SIMD; YNET_WINOGRAD48=1) against today's two launches (16 + 32), which read
and transform the input twice.  gpurun --timeout 600 -- 'YNET_WINOGRAD48=1 python3 tools/wino48_probe.py'"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("motion-style-transfer_amd.ops")

dev = torch.device("cuda:0")
B, H, W, cin = 32, 256, 256, 32
for (B, H, W) in ((32, 256, 256), (32, 128, 128), (16, 512, 512)):
    g = torch.Generator().manual_seed(1)
    dy = (torch.randn(B, cin, H, W, generator=g) * (torch.rand(B, cin, H, W, generator=g) > 0.3)).to(dev)
    w = (torch.randn(cin, 49, 3, 3, generator=g) * 0.2).to(dev)           # the forward filter [cout = 32][cin = 49]: its data gradient maps 32 -> 49
    wp = ops.pack_weight(w, 1)
    u16, u32, u48 = (ops.winograd_filter(wp, cin, n, c0, 49) for n, c0 in ((16, 0), (32, 16), (48, 0)))
    o16, o32, o48 = (torch.empty(B, n, H, W, device=dev) for n in (16, 32, 48))

    def two():
        ops.conv2d_winograd_raw((dy.data_ptr(), cin * H * W), u16, None, (o16.data_ptr(), 16 * H * W), cin, 16, B, H, W, False)
        ops.conv2d_winograd_raw((dy.data_ptr(), cin * H * W), u32, None, (o32.data_ptr(), 32 * H * W), cin, 32, B, H, W, False)

    def one():
        ops.conv2d_winograd_raw((dy.data_ptr(), cin * H * W), u48, None, (o48.data_ptr(), 48 * H * W), cin, 48, B, H, W, False)

    two()
    one()
    torch.cuda.synchronize()
    same = torch.equal(o48[:, :16], o16) and torch.equal(o48[:, 16:], o32)
    res = {}
    for name, fn in (("16 + 32 (two launches)", two), ("48 (one launch, 4 waves)", one)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 30 * 1e3
    print(f"B {B} {H}x{W} 32 -> 48: bit-identical {same}; " + "; ".join(f"{k} {v:.1f} us" for k, v in res.items()))
