import torch
dev=torch.device('cuda:0')
def t(fn,n=20):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/n
for mb in (268, 1072):
    x=torch.empty(mb*1024*1024//4,device=dev); y=torch.empty_like(x)
    us=t(lambda: x.fill_(1.0)); print(f"fill {mb}MB: {us:.1f} us {x.numel()*4/us/1e3:.0f} GB/s")
    us=t(lambda: y.copy_(x)); print(f"copy {mb}MB: {us:.1f} us r+w {2*x.numel()*4/us/1e3:.0f} GB/s")
    us=t(lambda: x.sum()); print(f"sum {mb}MB: {us:.1f} us {x.numel()*4/us/1e3:.0f} GB/s")
