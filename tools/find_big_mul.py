"""Development aid: which autograd node launches the large aten::mul of a C1 step?  (torch.profiler over one eager step)"""
import os, sys, importlib
sys.path.insert(0, os.getcwd())
os.environ["YNET_STEP_GRAPH"] = "0"
import torch
from torch.profiler import profile, ProfilerActivity
sys.argv = ["bench.py", "--config", sys.argv[1] if len(sys.argv) > 1 else "C1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-c5", "--no-roofline", "--no-repeats"]
import runpy
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    try:
        runpy.run_path("bench.py", run_name="__main__")
    except SystemExit:
        pass
for e in prof.events():
    dt = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    if e.name.startswith("aten::mul") and dt > 100:
        chain, q = [], e
        while q is not None and len(chain) < 6:
            chain.append(q.name)
            q = q.cpu_parent
        print(e.name, e.input_shapes, "device us", dt, "<-", " <- ".join(chain[1:]), "|", (e.stack or [])[:6])
