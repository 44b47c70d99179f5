#!/usr/bin/env python3
"""Isolated timing of ynet_pred_softargmax against the two launches it replaces (C5 shapes).  YNET_PRED_SOFT_PT=4: the
four-tile variant.   gpurun -- 'python tools/pred_softargmax_bench.py'"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ops = bench.pkg("ops")
dev = torch.device("cuda", 0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B in (128, 256):
    for pred in (30, 12):
        x = torch.relu(torch.randn(B, 32, 256, 256, device=dev))
        w, b = torch.randn(pred, 32, 1, 1, device=dev) * 0.2, torch.zeros(pred, device=dev)
        fused = timed(lambda: ops.pred_softargmax(x, w, b))
        cache = {}
        two = timed(lambda: ops.softargmax2d(ops.conv2d(x, w, b, False, cache)))
        gb = x.numel() * 4 / 1e9
        print(f"B {B} pred {pred}: fused {fused:7.1f} us = {gb / fused * 1e3:5.2f} TB/s on the input   two launches {two:7.1f} us")
