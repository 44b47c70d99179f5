"""Development aid: fused / foreach capturable Adam inside a captured graph against the eager update."""
import torch
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [(3, 42), (96, 3), (3, 96), (192, 3), (32, 14, 3, 3), (32,)]
base = [torch.randn(*s, generator=g) * 0.1 for s in shapes]
grads = [[torch.randn(*s, generator=g) * (10.0 ** -k) for s in shapes] for k in range(4)]

def run(mode, graph):
    ps = [torch.nn.Parameter(b.clone().to(dev)) for b in base]
    opt = torch.optim.Adam(ps, lr=1e-3)          # the user's optimizer: default (foreach, host step counters)
    static = [torch.zeros_like(p) for p in ps]
    for p, s in zip(ps, static):
        p.grad = s
    def step(k):
        for s, gg in zip(static, grads[k]):
            s.copy_(gg.to(dev))
        opt.step()
    step(0)                                       # eager warm-up with the default form
    if graph:
        for grp in opt.param_groups:
            if mode == "fused":
                grp["fused"], grp["foreach"], grp["capturable"] = True, False, True
            else:
                grp["capturable"] = True
        for p, st in opt.state.items():
            st["step"] = st["step"].to(device=dev, dtype=torch.float32)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg, stream=s):
                opt.step()
            for k in range(1, 4):
                for st, gg in zip(static, grads[k]):
                    st.copy_(gg.to(dev))
                cg.replay()
        torch.cuda.current_stream().wait_stream(s)
    else:
        for k in range(1, 4):
            step(k)
    torch.cuda.synchronize()
    return [p.detach().cpu() for p in ps], [float(opt.state[p]["step"]) for p in ps][:2]

ref, st = run("foreach", False)
print("eager steps", st)
for mode in ("foreach", "fused"):
    got, st = run(mode, True)
    print(mode, "graph: steps", st, [round(float((a - b).abs().max()) / 1e-3, 6) for a, b in zip(got, ref)], "(max |dW| vs eager, units of lr)")
