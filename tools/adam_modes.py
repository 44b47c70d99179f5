"""Development aid: torch.optim.Adam update of one step in its foreach / capturable-foreach / fused forms (same grads)."""
import torch
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [(3, 42), (96, 3), (3, 96), (192, 3)]
base = [torch.randn(*s, generator=g) * 0.1 for s in shapes]
grads = [[torch.randn(*s, generator=g) * (10.0 ** -k) for s in shapes] for k in range(3)]
res = {}
for mode in ("foreach", "capturable", "fused", "fused+capturable", "cpu"):
    d = torch.device("cpu") if mode == "cpu" else dev
    ps = [torch.nn.Parameter(b.clone().to(d)) for b in base]
    kw = dict(foreach=True) if mode == "foreach" else dict(capturable=True) if mode == "capturable" else \
        dict(fused=True) if mode == "fused" else dict(fused=True, capturable=True) if mode == "fused+capturable" else {}
    opt = torch.optim.Adam(ps, lr=1e-3, **kw)
    for k in range(3):
        for p, gg in zip(ps, grads[k]):
            p.grad = gg.clone().to(d)
        opt.step()
    res[mode] = [p.detach().cpu() for p in ps]
for mode in res:
    print(mode, [float((a - b).abs().max()) / 1e-3 for a, b in zip(res[mode], res["cpu"])], "(max |dW| vs cpu, in units of lr)")
