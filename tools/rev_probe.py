#!/usr/bin/env python3
"""Development probe (GPU only): per-launch time of a chain of six dependent 32 -> 32 Winograd convolutions at 256^2, B 32.  Round 5 used it to price a tile walk that
alternates direction from launch to launch (an experimental YNET_WINO_REV switch in conv_wino.hip, not kept: -2 % on this chain, nothing in the step -- DESIGN.md section 4.23)."""
import importlib, os, sys
sys.path.insert(0, "/root/repo")
import torch
ops = importlib.import_module("motion-style-transfer_amd.ops")
dev = torch.device("cuda:0")
B, H, W = 32, 256, 256
torch.manual_seed(0)
x = torch.randn(B, 32, H, W, device=dev)
w = torch.randn(32, 32, 3, 3, device=dev) * 0.05
wp = ops.pack_weight(w, 0)
ys = [torch.empty(B, 32, H, W, device=dev) for _ in range(3)]
cache = {}
def conv(src, dst):
    return ops.conv2d_raw([(src.data_ptr(), 32, 32 * H * W)], None, wp, None, [(dst.data_ptr(), 32, 32 * H * W)], B, H, W, 3, True, wino=(cache, "fwd"))
for _ in range(3):
    conv(x, ys[0]); conv(ys[0], ys[1])
torch.cuda.synchronize()
# chain of 6 dependent convs, timed as a whole and the 2nd..6th individually
evs = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
tot = []
for rep in range(5):
    src = x
    evs[0].record()
    for i in range(6):
        dst = ys[i % 3]
        conv(src, dst)
        evs[i + 1].record()
        src = dst
    torch.cuda.synchronize()
    tot.append([evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(6)])
print(os.environ.get("YNET_WINO_REV"), [round(sum(t[i] for t in tot[1:]) / 4, 1) for i in range(6)])
ref = ys[2].clone()
print("checksum", float(ref.double().sum()))
