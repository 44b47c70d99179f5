"""One rank of the multi-rank product-path test (started by tests/test_gpu_dp.py through torch.distributed.run).

Runs utils/train_epoch.train_epoch(dp=DataParallel) -- the product's own data-parallel branch: shard, expected_grad
scaling of the one-pass BCE, empty shards, ONE all-reduce per step, epoch-end metric reduction -- on the HIP kernels.
With YNET_DIST_BACKEND=gloo YNET_BENCH_SINGLE_DEVICE=1 every rank drives cuda:0 (a 1-GPU box); on a multi-GPU node the
defaults give one GPU per rank over RCCL.  Rank 0 writes {result, weights, last gradients, world info} to argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from conftest import build_model, pkg  # noqa: E402
from oracle import ynet_oracle as O  # noqa: E402  (input generators / state dict only)


def case_inputs(n_rows, H=64, W=64):
    cfg = O.sdd_short(train_net="mosa_2", position=["0", "1", "2", "3", "4"], enc=(8, 8, 16, 16, 16), dec=(16, 16, 16, 8, 8))
    sd = O.make_state_dict(cfg, seed=5, lora_b_std=0.05)
    scene = O.synthetic_scene(cfg, H, W, 5)
    traj = O.synthetic_trajectories(cfg, n_rows, H, W, 5)
    return cfg, sd, scene, traj


def run_epochs(cfg, sd, scene, traj, batch_size, dev, dp_factory, n_epochs=2, graph=None):
    import numpy as np
    import pandas as pd
    te, trn = pkg("utils.train_epoch"), pkg("models.trainer")
    model = build_model(cfg, sd, dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    dp = dp_factory(model)
    transport = None if dp is None else {"collective": dp.collective, "note": dp.transport_note}
    S = cfg.template_size
    in_t, gt_t = O.dist_template(S).to(dev), O.gaussian_template(S, cfg.kernlen, cfg.nsig).to(dev)
    loader = [(traj.clone(), [pd.DataFrame({"metaId": np.arange(traj.shape[0])})], "scene0")]
    results = []
    kw = {} if graph is None else {"graph": graph}
    for e in range(n_epochs):
        results.append(te.train_epoch(model, loader, {"scene0": scene[0]}, opt, trn.HipBCEWithLogitsLoss(), cfg.loss_scale,
                                      dev, "sdd", None, gt_t, in_t, list(cfg.waypoints), e, cfg.obs_len, cfg.pred_len,
                                      batch_size, 10000, cfg.resize_factor, cfg.network, False, dp=dp, **kw))
    named = dict(model.named_parameters())
    names = [n for n, p in named.items() if p.requires_grad]
    out = {"results": results, "weights": {n: named[n].detach().cpu().clone() for n in names},
           "grads": {n: named[n].grad.detach().cpu().clone() for n in names}}
    # the evaluation sweep under the same sharding (utils/evaluate.py dp= branch: shard, per-rank decoder passes, gather_rows);
    # the sampled way-points are injected so that every world size sees the same draws
    ev = pkg("utils.evaluate")
    g = torch.Generator().manual_seed(3)
    H, W = scene.shape[-2:]
    K, n = 4, traj.shape[0]
    forced = {b: torch.stack([torch.rand(K, min(batch_size, n - b), len(cfg.waypoints), generator=g) * (W - 1),
                              torch.rand(K, min(batch_size, n - b), len(cfg.waypoints), generator=g) * (H - 1)], dim=-1).round()
              for b in range(0, n, batch_size)}
    ade, fde, df, _ = ev.evaluate(model, loader, {"scene0": scene[0]}, dev, "sdd", None, in_t, list(cfg.waypoints), "test", K, 1,
                                  cfg.obs_len, batch_size, cfg.resize_factor, cfg.temperature, forced_samples=forced, dp=dp)
    out["eval"] = (ade, fde, df["ade"].to_numpy().copy(), df["fde"].to_numpy().copy())
    out["transport"] = transport
    if dp is not None:
        dp.check()          # a timed-out one-shot all-reduce would raise here (it also poisons the loss with NaN)
        dp.close()
    return out


def main():
    out, n_rows, batch_size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    D = pkg("dist")
    rank, local, world = D.init_from_env()
    if os.environ.get("YNET_BENCH_SINGLE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg, sd, scene, traj = case_inputs(n_rows)
    res = run_epochs(cfg, sd, scene, traj, batch_size, dev, lambda m: D.DataParallel(m.parameters()))
    info = [None] * world
    dist.all_gather_object(info, {"rank": rank, "device": str(dev), "pid": os.getpid()})
    if rank == 0:
        res["world"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": info,
                        "collective": res["transport"]["collective"], "requested": os.environ.get("YNET_ALLREDUCE", "rccl"),
                        "transport_note": res["transport"]["note"]}
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
