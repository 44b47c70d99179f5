"""The N-rank step on ONE rank with every collective issued for real (started by tests/test_gpu_dp.py through the pre-GPU launcher).

A 1-GPU box cannot run two RCCL ranks, and gloo (the multi-rank tests' fall-back there) is host-synchronous: it says nothing
about how an RCCL collective, enqueued asynchronously on torch's internal NCCL stream, orders itself against the two hipGraphs
it sits between.  Here a process group of ONE rank is created on backend nccl (= RCCL) and dist.DataParallel(force=True) makes
every collective of the step run: the all-reduce of the flat gradient buffer (RCCL between graph A and graph B of the split
step, or ynet_allreduce_sum recorded inside the single graph under YNET_ALLREDUCE=oneshot), the epoch-end scalar sum, the seed
broadcast.  The sums of one rank are its inputs, so three steps [eager, capture + replay, replay] at C2 B = 32 must equal the
dp=None run BIT FOR BIT: loss, ADE / FDE, every gradient of the last step, every weight after the three Adam updates.
Reference: utils/train_epoch.py:109-115 (backward + optimizer.step, between which the collective sits); SURVEY 8(e)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from conftest import build_model, pkg  # noqa: E402
from oracle import ynet_oracle as O  # noqa: E402  (input generators / state dict only)


def loader_for(traj):
    return [(traj.clone(), [pd.DataFrame({"metaId": np.arange(traj.shape[0])})], "scene0")]


def three_steps(cfg, sd, scene, trajs, dev, dp_factory, in_t, gt_t, B):
    te, trn, sg = pkg("utils.train_epoch"), pkg("models.trainer"), pkg("utils.step_graph")
    model = build_model(cfg, sd, dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    dp = dp_factory(model)
    crit = trn.HipBCEWithLogitsLoss()
    images = {"scene0": scene[0].to(dev)}
    res, launched = [], []
    for i, traj in enumerate(trajs):
        res.append(te.train_epoch(model, loader_for(traj), images, opt, crit, cfg.loss_scale, dev, "sdd", None, gt_t, in_t,
                                  list(cfg.waypoints), i, cfg.obs_len, cfg.pred_len, B, 10000, cfg.resize_factor, cfg.network, False, dp=dp))
        entries = [e for c in sg._caches.get(model, {}).values() for e in c.entries.values()]
        launched.append("replay" if any(e.ready for e in entries) else "eager")
    entries = [e for c in sg._caches.get(model, {}).values() for e in c.entries.values() if e.ready]
    named = {n: p for n, p in model.named_parameters() if p.requires_grad}
    out = {"results": res, "launched": launched,
           "graphs_per_step": max([len(e.graphs) for e in entries] or [0]),
           "collective_in_graph": any(getattr(e, "collective_in_graph", False) for e in entries),
           "failed": sum(1 for c in sg._caches.get(model, {}).values() for e in c.entries.values() if e.failed),
           "weights": {n: p.detach().cpu().clone() for n, p in named.items()},
           "grads": {n: p.grad.detach().cpu().clone() for n, p in named.items()}}
    if dp is not None:
        out["seed_ok"] = isinstance(dp.shared_seed(), int)
        out["collective"], out["note"] = dp.collective, dp.transport_note
        split = [e.profile_split(10) for e in entries if getattr(e, "split", False)]
        out["split_ms"] = split[0] if split else None
        dp.check()
        dp.close()
    return out


def main():
    out_path, B = sys.argv[1], int(sys.argv[2])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    D = pkg("dist")
    cfg = O.sdd_short(train_net="mosa_1", position=["0", "1", "2", "3", "4"])
    H = W = int(os.environ.get("YNET_TEST_RASTER", "256"))
    sd = O.make_state_dict(cfg, seed=0, lora_b_std=0.05)
    scene = O.synthetic_scene(cfg, H, W, 0)
    trajs = [O.synthetic_trajectories(cfg, B, H, W, 41 + i) for i in range(3)]
    S = cfg.template_size
    in_t, gt_t = O.dist_template(S).to(dev), O.gaussian_template(S, cfg.kernlen, cfg.nsig).to(dev)
    plain = three_steps(cfg, sd, scene, trajs, dev, lambda m: None, in_t, gt_t, B)
    forced = three_steps(cfg, sd, scene, trajs, dev, lambda m: D.DataParallel(m.parameters(), force=True), in_t, gt_t, B)
    verdict = {
        "backend": dist.get_backend(), "world_size": dist.get_world_size(),
        "collective": forced["collective"], "transport_note": forced["note"], "seed_ok": forced["seed_ok"],
        "launched": [plain["launched"], forced["launched"]],
        "graphs_per_step": [plain["graphs_per_step"], forced["graphs_per_step"]],
        "collective_in_graph": forced["collective_in_graph"], "failed": [plain["failed"], forced["failed"]],
        "split_ms": forced["split_ms"],
        "results": [plain["results"], forced["results"]],
        "results_equal": plain["results"] == forced["results"],
        "weights_differ": [n for n in plain["weights"] if not torch.equal(plain["weights"][n], forced["weights"][n])],
        "grads_differ": [n for n in plain["grads"] if not torch.equal(plain["grads"][n], forced["grads"][n])],
        "n_tensors": len(plain["weights"]),
        "weights_moved": all(not torch.equal(plain["weights"][n], sd[n]) for n in plain["weights"]),
    }
    with open(out_path, "w") as f:
        json.dump(verdict, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
