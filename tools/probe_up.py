import sys, torch, time
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
for (B, H, W, cin, cout) in [(32, 256, 256, 32, 16), (32, 256, 256, 16, 32), (32, 256, 256, 32, 32), (256, 256, 256, 32, 16)]:
    x = torch.relu(torch.randn(B, cin, H, W, device=dev))
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.2
    wp = ops.pack_weight(w, 0)
    u = ops.winograd_filter(wp, cin, cout)
    y = torch.empty(B, cout, H, W, device=dev)
    t_full = bench(lambda: ops.conv2d_winograd_raw((x.data_ptr(), cin * H * W), u, None, (y.data_ptr(), cout * H * W), cin, cout, B, H, W, False))
    t_in0 = bench(lambda: ops.conv2d_winograd_raw((x.data_ptr(), 0), u, None, (y.data_ptr(), cout * H * W), cin, cout, B, H, W, False))
    # quarter of the input bytes: every image reads one of B/4 images
    print(f"B={B} {H}x{W} {cin}->{cout}: full {t_full:.1f} us, input batch-stride 0 (L2-resident input) {t_in0:.1f} us")
    # upsample kernel alone
    xl = torch.relu(torch.randn(B, cin, H // 2, W // 2, device=dev))
    t_up = bench(lambda: ops.upsample2x(xl))
    print(f"   upsample2x {cin}ch {H//2}->{H}: {t_up:.1f} us")
