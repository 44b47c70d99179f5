"""Data-parallel sharding (dist.DataParallel) on 2 CPU processes over gloo: the all-reduced, shard-
weighted gradient equals the single-process gradient of the whole batch, incl. ragged and empty shards.
The arithmetic is the CPU oracle here (tests may use it); the product uses the same class over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import pkg, ROOT
from oracle import ynet_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, out):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    D = pkg("dist")
    r, l, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    cfg = O.sdd_short(train_net="mosa_2", position=["0", "1", "2", "3", "4"], enc=(8, 8, 16, 16, 16), dec=(16, 16, 16, 8, 8))
    sd = O.make_state_dict(cfg, seed=0, lora_b_std=0.05)
    names = O.trainable_names(cfg, sd)
    scene = O.synthetic_scene(cfg, 32, 32, 0)
    traj = O.synthetic_trajectories(cfg, n_rows, 32, 32, 0)
    S = cfg.template_size
    in_t, gt_t = O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig)
    params = [torch.nn.Parameter(sd[n].clone()) for n in names]
    dp = D.DataParallel(params)
    lo, hi = dp.shard(n_rows)
    dp.zero_grad()
    loss = torch.zeros(())
    ade = torch.zeros(0)
    if hi > lo:
        res = O.train_step(sd, cfg, scene, traj[lo:hi], in_t, gt_t, names, loss_weight=(hi - lo) / n_rows)
        for p, n in zip(params, names):
            p.grad.copy_(res["grads"][n])            # what autograd's AccumulateGrad does into the bound views
        loss = res["loss"] * ((hi - lo) / n_rows)
        ade = res["ade"]
    loss = dp.allreduce_grads(loss)
    ade = dp.gather_rows(ade, dp.shard_sizes(n_rows))
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in params], "loss": loss, "ade": ade, "flat": dp.flat.numel()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [4, 3, 1])
def test_two_rank_gradient_equals_single_process(tmp_path, n_rows):
    out = str(tmp_path / "r0.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_rows, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=False)
    cfg = O.sdd_short(train_net="mosa_2", position=["0", "1", "2", "3", "4"], enc=(8, 8, 16, 16, 16), dec=(16, 16, 16, 8, 8))
    sd = O.make_state_dict(cfg, seed=0, lora_b_std=0.05)
    names = O.trainable_names(cfg, sd)
    S = cfg.template_size
    ref = O.train_step(sd, cfg, O.synthetic_scene(cfg, 32, 32, 0), O.synthetic_trajectories(cfg, n_rows, 32, 32, 0),
                       O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig), names)
    assert got["flat"] == sum(sd[n].numel() for n in names) + 1        # + the loss slot
    for g, n in zip(got["grads"], names):
        w = ref["grads"][n]
        assert float((g - w).abs().max()) <= 1e-5 * float(w.abs().max()) + 1e-7, n
    assert abs(float(got["loss"]) - float(ref["loss"])) <= 1e-5 * abs(float(ref["loss"]))
    assert torch.allclose(got["ade"], ref["ade"], rtol=1e-5, atol=1e-5)


def test_shard_arithmetic():
    D = pkg("dist")

    class Fake(D.DataParallel):
        def __init__(self, rank, world):
            self.rank, self.world = rank, world
    rows = []
    for n in (0, 1, 5, 32, 33):
        cover = []
        for r in range(4):
            lo, hi = Fake(r, 4).shard(n)
            cover += list(range(lo, hi))
            assert hi - lo == Fake(r, 4).shard_sizes(n)[r]
        assert cover == list(range(n))


def _loader_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    import numpy as np
    import pandas as pd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D = pkg("dist")
    D.init_from_env("gloo")
    torch.manual_seed(1000 + 17 * rank)          # the ranks' default generators DIFFER (as in real launches)
    trn = pkg("models.trainer")
    params = dict(obs_len=8, pred_len=12, segmentation_model_fp=None, use_features_only=False, n_semantic_classes=6,
                  encoder_channels=[8, 8, 16, 16, 16], decoder_channels=[16, 16, 16, 8, 8], waypoints=[11],
                  train_net="mosa_1", position=["0"], network="original", n_fusion=None, resize_factor=0.25)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        t = trn.YNetTrainer(params, device=torch.device("cpu"))
    t.dp = D.DataParallel([torch.nn.Parameter(torch.zeros(3))])
    n_scene, per = 7, 2
    df = pd.DataFrame({"sceneId": np.repeat([f"s{i}" for i in range(n_scene)], per * 20),
                       "metaId": np.repeat(np.arange(n_scene * per), 20),
                       "x": np.arange(n_scene * per * 20, dtype=np.float32), "y": 0.0})
    images = {f"s{i}": torch.zeros(6, 32, 32) for i in range(n_scene)}
    _, loader, _ = t.prepare_data(df, images, "sdd", "train", 8, 12, 0.25, False)
    order = [[scene for _, _, scene in loader] for _epoch in range(3)]
    _, val_loader, _ = t.prepare_data(df, images, "sdd", "val", 8, 12, 0.25, False)
    torch.save({"order": order, "val": [scene for _, _, scene in val_loader]}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_walk_the_scenes_in_the_same_shuffled_order(tmp_path):
    """under data parallelism train_epoch shards trajectory[i:i+bs] of ONE scene over the ranks, so
    every rank's shuffled DataLoader must yield the scenes in the same order although the ranks' default RNGs differ."""
    out = str(tmp_path / "order.pt")
    mp.spawn(_loader_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = torch.load(out + ".0", weights_only=False), torch.load(out + ".1", weights_only=False)
    assert a["order"] == b["order"]
    assert all(sorted(e) == [f"s{i}" for i in range(7)] for e in a["order"])
    assert len({tuple(e) for e in a["order"]}) > 1, "epochs should be shuffled differently"
    assert a["val"] == b["val"] == [f"s{i}" for i in range(7)]
