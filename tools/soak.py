#!/usr/bin/env python3
"""Soak run on the GPU box: N epochs of C2 training on fresh synthetic batches; prints loss / ADE, step time and
allocator high-water marks per epoch (they must not grow), and checks that every parameter stays finite."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    pkg = bench.pkg
    trainer, te = pkg("models.trainer"), pkg("utils.train_epoch")
    dev = torch.device("cuda", 0)
    cfg, H, W, workload = bench.make_cfg(args.config)
    model = bench.build_model(cfg, dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = trainer.HipBCEWithLogitsLoss()
    in_t, gt_t = bench.templates(cfg, dev)
    images = {"scene0": bench.synthetic_scene(cfg, H, W, 0)[0].to(dev)}
    traj = bench.synthetic_trajectories(cfg, args.batch * args.steps, H, W, 7)      # the same data every epoch: the loss must fall
    peaks = []
    for ep in range(args.epochs):
        torch.cuda.reset_peak_memory_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ade, fde, loss = te.train_epoch(model, bench.loader_for(traj), images, opt, crit, cfg.loss_scale, dev, "sdd", None,
                                        gt_t, in_t, list(cfg.waypoints), ep, cfg.obs_len, cfg.pred_len, args.batch, 10000,
                                        cfg.resize_factor, cfg.network, False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
        peaks.append(torch.cuda.max_memory_allocated())
        print(f"epoch {ep}: loss {float(loss):.4f} ADE {float(ade):.3f} FDE {float(fde):.3f}  {dt / args.steps * 1e3:.2f} ms/step  "
              f"peak {peaks[-1] / 2**30:.2f} GiB reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB finite={finite}", flush=True)
        assert finite
    assert peaks[-1] <= peaks[1] * 1.01, "allocator high-water mark keeps growing"


if __name__ == "__main__":
    main()
