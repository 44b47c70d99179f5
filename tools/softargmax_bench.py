#!/usr/bin/env python3
"""Soft-argmax launch time / HBM rate and error against fp64 (development tool, GPU only).
    python tools/softargmax_bench.py [B C H W]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ops = importlib.import_module("motion-style-transfer_amd.ops")
B, C, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (128, 30, 256, 256)
dev = torch.device("cuda:0")
torch.manual_seed(0)
for scale in (1.0, 8.0):
    x = torch.randn(B, C, H, W, device=dev) * scale
    for _ in range(3):
        out = ops.softargmax2d(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = ops.softargmax2d(x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    xs = x[:4].double()
    p = torch.softmax(xs.reshape(4, C, -1), dim=-1).reshape(4, C, H, W)
    ex = (p.sum(2) * torch.arange(W, device=dev, dtype=torch.float64)).sum(-1)
    ey = (p.sum(3) * torch.arange(H, device=dev, dtype=torch.float64)).sum(-1)
    err = max(float((out[:4, :, 0].double() - ex).abs().max()), float((out[:4, :, 1].double() - ey).abs().max()))
    print(f"scale {scale}: {us:8.1f} us  {x.numel() * 4 / us / 1e3:7.1f} GB/s   max |err| vs fp64 {err:.3e} px")
