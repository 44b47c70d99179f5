"""Development probe: the up-convolution 32 -> 16 with and without the bilinear x2 inside (tools, not tests)."""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import pkg
ops = pkg("ops")
dev = torch.device("cuda:0")
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (B, H, W) in [(32, 256, 256), (256, 256, 256), (16, 512, 512), (32, 128, 128)]:
    cin, cout = 32, 16
    xl = torch.relu(torch.randn(B, cin, H // 2, W // 2, device=dev))
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.2
    wp = ops.pack_weight(w, 0)
    u = ops.winograd_filter(wp, cin, cout)
    y = torch.empty(B, cout, H, W, device=dev)
    up = ops.upsample2x(xl)
    t_up = bench(lambda: ops.upsample2x(xl))
    t_conv = bench(lambda: ops.conv2d_winograd_raw((up.data_ptr(), cin * H * W), u, None, (y.data_ptr(), cout * H * W), cin, cout, B, H, W, False))
    t_fused = bench(lambda: ops.upsample2x_conv2d_raw((xl.data_ptr(), cin * H * W // 4), u, None, (y.data_ptr(), cout * H * W), cin, cout, B, H, W))
    print(f"B={B} {H}x{W}: upsample {t_up:.1f} + conv {t_conv:.1f} = {t_up + t_conv:.1f} us; fused {t_fused:.1f} us")
for (B, H, W, cin, cout) in [(32, 128, 128, 64, 32), (256, 128, 128, 64, 32), (32, 64, 64, 64, 32), (256, 64, 64, 64, 32)]:
    xl = torch.relu(torch.randn(B, cin, H // 2, W // 2, device=dev))
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.2
    wp = ops.pack_weight(w, 0)
    u = ops._wino16_filter(({}, "fwd"), wp, 0, (cin,), cout, 0, cout)[1]
    y = torch.empty(B, cout, H, W, device=dev)
    up = ops.upsample2x(xl)
    cache = {}
    t_up = bench(lambda: ops.upsample2x(xl))
    t_conv = bench(lambda: ops.conv2d_raw([(up.data_ptr(), cin, cin * H * W)], None, wp, None, [(y.data_ptr(), cout, cout * H * W)], B, H, W, 3, False, wino=(cache, "fwd")))
    t_fused = bench(lambda: ops.upsample2x_conv2d_raw((xl.data_ptr(), cin * H * W // 4), u, None, (y.data_ptr(), cout * H * W), cin, cout, B, H, W))
    print(f"B={B} {H}x{W} {cin}->{cout}: upsample {t_up:.1f} + conv {t_conv:.1f} = {t_up + t_conv:.1f} us; fused (slice form) {t_fused:.1f} us")
