import sys, os, importlib
sys.path.insert(0, os.getcwd())
import torch
ops = importlib.import_module("motion-style-transfer_amd.ops")
orig = ops._Conv2dFn.forward
cnt = {"pool_req": 0, "pool_used": 0}
raw = ops.conv2d_raw
def spy(srcs, mask, wp, bias, dsts, B, H, W, K, relu, relu_of=None, pooled=None):
    if pooled is not None: cnt["pool_used"] += 1
    return raw(srcs, mask, wp, bias, dsts, B, H, W, K, relu, relu_of=relu_of, pooled=pooled)
ops.conv2d_raw = spy
c2 = ops.conv2d
def spy2(x, weight, bias, relu, cache, lora_a=None, lora_b=None, scale=1.0, pool=False):
    if pool: cnt["pool_req"] += 1
    return c2(x, weight, bias, relu, cache, lora_a, lora_b, scale, pool)
ops.conv2d = spy2
sys.argv = ["bench.py", "--config", sys.argv[1], "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-c5", "--no-roofline", "--no-repeats"]
os.environ["YNET_STEP_GRAPH"] = "0"
import runpy
try: runpy.run_path("bench.py", run_name="__main__")
except SystemExit: pass
print(sys.argv[2], cnt)
