"""Development aid: the captured (hipGraph) training step against the eager one on a tiny config."""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dp_worker import case_inputs, run_epochs

dev = torch.device("cuda:0")
cfg, sd, scene, traj = case_inputs(12)
res = {}
for mode in (False, True):
    print("=== graph", mode, flush=True)
    res[mode] = run_epochs(cfg, sd, scene, traj, 4, dev, lambda m: None, n_epochs=2, graph=mode)
    print(res[mode]["results"], flush=True)
for n in res[False]["weights"]:
    d = float((res[False]["weights"][n] - res[True]["weights"][n]).abs().max())
    g = float((res[False]["grads"][n] - res[True]["grads"][n]).abs().max())
    print(n, "dW", d, "dG", g, float(res[False]["grads"][n].abs().max()))
