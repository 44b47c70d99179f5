"""How far a training step at the headline batch sizes moves against the CPU oracle with and without the Winograd convolutions
(development aid; the checks themselves are tests/test_gpu_headline.py): prints the largest deviation of the loss, the batch ADE / FDE,
every trajectory's read-out and the gradients for the chosen configuration with ops._wino_allowed on and off.
    gpurun --timeout 900 -- 'python tests/wino_margin.py C2_B32 C4_B16'
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))      # (lives in tests/: it runs the CPU oracle, which only tests / smoke / bench's baseline leg may)
import test_gpu_headline as T       # noqa: E402
from conftest import build_model, pkg      # noqa: E402
from oracle import ynet_oracle as O        # noqa: E402


def main(tags):
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    dev = torch.device("cuda:0")
    ops = pkg("ops")
    for tag in tags:
        mk, H, W, B = T.HEADLINE[tag]
        cfg = mk()
        sd = O.make_state_dict(cfg, seed=0, lora_b_std=0.05)
        scene, traj = O.synthetic_scene(cfg, H, W, 0), O.synthetic_trajectories(cfg, B, H, W, 21)
        S = cfg.template_size
        in_t, gt_t = O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig)
        names = O.trainable_names(cfg, sd)
        want = O.train_step(sd, cfg, scene, traj, in_t, gt_t, names)
        for allowed in (True, False):
            ops._wino_allowed = allowed
            n0 = ops.wino_stats["launches"]
            model = build_model(cfg, sd, dev)
            te, trn = pkg("utils.train_epoch"), pkg("models.trainer")
            caught = []
            h = model.softargmax_.register_forward_hook(lambda m, i, o: caught.append(o.detach().cpu()))
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
            ade, fde, loss = te.train_epoch(
                model, T.loader_for(traj), {"scene0": scene[0]}, opt, trn.HipBCEWithLogitsLoss(), cfg.loss_scale, dev, "sdd", None,
                gt_t.to(dev), in_t.to(dev), list(cfg.waypoints), 0, cfg.obs_len, cfg.pred_len, B, 10000, cfg.resize_factor,
                cfg.network, False)
            h.remove()
            named = dict(model.named_parameters())
            gerr = max(float((named[n].grad.detach().cpu().double() - want["grads"][n].double()).abs().max()) /
                       (float(want["grads"][n].abs().max()) + 1e-30) for n in names)
            print(f"{tag} winograd={'on ' if allowed else 'off'} ({ops.wino_stats['launches'] - n0} launches): loss rel {abs(loss - float(want['loss'])) / abs(float(want['loss'])):.2e}  "
                  f"ADE {abs(ade - float(want['ade'].mean())):.2e}  FDE {abs(fde - float(want['fde'].mean())):.2e}  "
                  f"read-out traj {float((caught[0] - want['pred_traj']).abs().max()):.2e} goal {float((caught[1] - want['pred_goal']).abs().max()):.2e} px  "
                  f"grads {gerr:.2e} of max", flush=True)


if __name__ == "__main__" and (len(sys.argv) < 2 or sys.argv[1] != "sweep"):
    main(sys.argv[1:] or ["C2_B32"])


def sweep(K=20):
    """The C5 sweep (tests/test_gpu_headline.py::test_eval_sweep_at_headline_batch_matches_oracle) with the Winograd kernels forced on
    under no_grad, and off: python tests/wino_margin.py sweep"""
    import pandas as pd      # noqa: F401
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    dev = torch.device("cuda:0")
    ops = pkg("ops")
    cfg = O.sdd_long(train_net="train")
    H = W = 256
    B = 128
    sd = O.make_state_dict(cfg, seed=0)
    scene, traj = O.synthetic_scene(cfg, H, W, 0), O.synthetic_trajectories(cfg, B, H, W, 22)
    in_t = O.dist_template(cfg.template_size)
    gen = torch.Generator().manual_seed(5)
    want = O.eval_batch(sd, cfg, scene, traj, in_t, n_goal=K, n_traj=1, generator=gen)
    ev = pkg("utils.evaluate")
    for on in (True, False):
        ops._wino_eval = on
        n0 = ops.wino_stats["launches"]
        model = build_model(cfg, sd, dev)
        caught = []
        h = model.softargmax_.register_forward_hook(lambda m, i, o: caught.append(o.detach().cpu()))
        ade, fde, df, _ = ev.evaluate(
            model, T.loader_for(traj), {"scene0": scene[0]}, dev, "sdd", None, in_t.to(dev), list(cfg.waypoints), "test", K, 1,
            cfg.obs_len, B, cfg.resize_factor, cfg.temperature, forced_samples={0: want["waypoint_samples"]})
        h.remove()
        got = torch.cat(caught).view(K, B, cfg.pred_len, 2)
        dc = (got - want["trajs"]).abs()
        print(f"sweep K={K} winograd={'on ' if on else 'off'} ({ops.wino_stats['launches'] - n0} launches): coordinates max {float(dc.max()):.2e} px, "
              f"{int((dc > 1e-4).sum())} of {dc.numel()} beyond 1e-4; per-trajectory ADE max {np.abs(df['ade'].to_numpy() - want['ade'].numpy()).max():.2e} "
              f"FDE max {np.abs(df['fde'].to_numpy() - want['fde'].numpy()).max():.2e}; mean ADE {abs(ade - float(want['ade'].mean())):.2e} "
              f"FDE {abs(fde - float(want['fde'].mean())):.2e}", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sweep":
    sweep(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
