"""Isolated timing of the last decoder convolution with the predictor + criterion in its epilogue (ynet_conv2d_winograd_pred_bce_blob) against the two launches it replaces.
gpurun --timeout 600 -- 'python3 tools/conv_pred_bce_probe.py'   (YNET_HIP_LIB=<development build> for the ablations of tools/ab_conv_pred_bce_epi.sh)"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("motion-style-transfer_amd.ops")
iu = importlib.import_module("motion-style-transfer_amd.utils.image_utils")
lib = ops._lib()
dev = torch.device("cuda:0")
B, H, W, cout, S = (int(v) for v in os.environ.get("PROBE_SHAPE", "32,256,256,12,800").split(","))      # (C4's tail: PROBE_SHAPE=16,512,512,30,1600)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 32, H, W, generator=g).relu().to(dev)
w3, b3 = (torch.randn(32, 32, 3, 3, generator=g) * 0.1).to(dev), (torch.randn(32, generator=g) * 0.1).to(dev)
w1, b1 = (torch.randn(cout, 32, 1, 1, generator=g) * 0.3).to(dev), (torch.randn(cout, generator=g) * 0.1).to(dev)
tmpl = iu.analytic_gaussian_template(S, 31, 4, True, dev)
pos = (torch.rand(B * cout, 2, generator=g) * torch.tensor([W * 1.0, H * 1.0])).to(dev)
u = ops.winograd_filter(ops.pack_weight(w3, 0), 32, 32, 0, 32)
wp1 = ops.pack_weight(w1, 0)
y, logits, dx = torch.empty(B, 32, H, W, device=dev), torch.empty(B, cout, H, W, device=dev), torch.empty(B, 32, H, W, device=dev)
loss = torch.empty((), device=dev)
ws = torch.zeros(lib.ynet_pred_bce_workspace_bytes() // 8 + 1, device=dev, dtype=torch.float64)
L = importlib.import_module("motion-style-transfer_amd._lib")


def conv():
    ops.conv2d_winograd_raw((x.data_ptr(), 32 * H * W), u, b3, (y.data_ptr(), 32 * H * W), 32, 32, B, H, W, True)


def pred():
    L.check(lib.ynet_pred_bce_blob(y.data_ptr(), 32 * H * W, wp1.data_ptr(), b1.data_ptr(), pos.data_ptr(), tmpl.blob.data_ptr(), tmpl.blob.shape[0], tmpl.size, H, W,
                                   logits.data_ptr(), loss.data_ptr(), dx.data_ptr(), None, ws.data_ptr(), B, 32, cout, 1000.0, 1, None), lib)


def fused():
    ops.conv2d_winograd_pred_bce_raw((x.data_ptr(), 32, 32 * H * W), u, b3, wp1, b1, cout, pos, tmpl, logits, loss, dx, ws, B, H, W, 1000.0)


for name, fn in (("conv + ReLU", conv), ("pred_bce_blob", pred), ("fused", fused)):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:16s} {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us   (loss {float(loss):.6f})")
