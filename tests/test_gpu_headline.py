"""Parity at the BENCHMARKED sizes: the HIP path at BASELINE.json's per-GPU batch sizes -- C2
B = 32 (256^2), C4 B = 16 (512^2), C5 B = 128 with the K goal samples folded into the batch -- against the CPU oracle
computed on the box.  Kernel dispatch (rows per wave, split-K thresholds, persistent-grid sizes, K-fold group size)
depends on B, so the small-B fixtures do not cover these launches.  Tolerances: loss 2e-5 relative, ADE / FDE 1e-4
(north star), gradients 5e-4 of the tensor's maximum (fp32 sums over B*H*W = 2-4 M pixels in a different order than
MKL-DNN).  Reference: utils/train_epoch.py:44-126, utils/evaluate.py:248-291."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import build_model, pkg
from oracle import ynet_oracle as O

pytestmark = pytest.mark.gpu


def loader_for(traj):
    return [(traj.clone(), [pd.DataFrame({"metaId": np.arange(traj.shape[0])})], "scene0")]


@pytest.fixture(autouse=True)
def _cpu_threads():
    # torch's intra-op pool collapses when oversubscribed (256 threads on the GPU box's host ran the oracle 30x slower)
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, n)))
    yield
    torch.set_num_threads(old)


HEADLINE = {
    "C2_B32": (lambda: O.sdd_short(train_net="mosa_1", position=["0", "1", "2", "3", "4"]), 256, 256, 32),
    "C3_B32_rank4": (lambda: O.sdd_short(train_net="mosa_4", position=["0", "1", "2", "3", "4"]), 256, 256, 32),
    "C4_B16": (lambda: O.ind_long(network="fusion", n_fusion=2, train_net="mosa_3", position=["scene"]), 512, 512, 16),
    "C2_B10_reference_scripts": (lambda: O.sdd_short(train_net="mosa_1", position=["0", "1", "2", "3", "4"]), 256, 256, 10),
    # C1 as `bench.py --config C1` runs it: every weight trainable, all 46 filter gradients at the production shapes
    "C1_B32_all_weights": (lambda: O.sdd_short(train_net="train"), 256, 256, 32),
}


@pytest.mark.parametrize("tag", list(HEADLINE))
def test_train_step_at_headline_batch_matches_oracle(dev, tag):
    mk, H, W, B = HEADLINE[tag]
    cfg = mk()
    sd = O.make_state_dict(cfg, seed=0, lora_b_std=0.05)
    scene, traj = O.synthetic_scene(cfg, H, W, 0), O.synthetic_trajectories(cfg, B, H, W, 21)
    S = cfg.template_size
    in_t, gt_t = O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig)
    names = O.trainable_names(cfg, sd)
    want = O.train_step(sd, cfg, scene, traj, in_t, gt_t, names)

    model = build_model(cfg, sd, dev)
    te, trn = pkg("utils.train_epoch"), pkg("models.trainer")
    caught = []
    h = model.softargmax_.register_forward_hook(lambda m, i, o: caught.append(o.detach().cpu()))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    ade, fde, loss = te.train_epoch(
        model, loader_for(traj), {"scene0": scene[0]}, opt, trn.HipBCEWithLogitsLoss(), cfg.loss_scale, dev, "sdd", None,
        gt_t.to(dev), in_t.to(dev), list(cfg.waypoints), 0, cfg.obs_len, cfg.pred_len, B, 10000, cfg.resize_factor,
        cfg.network, False)
    h.remove()
    assert abs(loss - float(want["loss"])) <= 2e-5 * abs(float(want["loss"])), (loss, float(want["loss"]))
    assert abs(ade - float(want["ade"].mean())) <= 1e-4, (ade, float(want["ade"].mean()))
    assert abs(fde - float(want["fde"].mean())) <= 1e-4, (fde, float(want["fde"].mean()))
    # every trajectory's soft-argmax read-out, not only the batch mean
    np.testing.assert_allclose(caught[0].numpy(), want["pred_traj"].numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(caught[1].numpy(), want["pred_goal"].numpy(), rtol=1e-5, atol=1e-4)
    named = dict(model.named_parameters())
    for n in names:
        g, w = named[n].grad.detach().cpu().double(), want["grads"][n].double()
        err, tol = float((g - w).abs().max()), 5e-4 * float(w.abs().max()) + 1e-7
        assert err <= tol, f"grad {n}: max err {err:.3e} > {tol:.3e}"


TIMED_PATH = {
    "C2_B32": (lambda: O.sdd_short(train_net="mosa_1", position=["0", "1", "2", "3", "4"]), 256, 256, 32),
    "C4_B16": (lambda: O.ind_long(network="fusion", n_fusion=2, train_net="mosa_3", position=["scene"]), 512, 512, 16),
    # the replayed graph of `bench.py --config C1` carries 46 filter gradients (a side branch per decoder) and
    # ynet_adam_step over 1.64 M parameters; C3 is the rank-4 adapter chain (wgrad -> reduce -> rank-r GEMMs on the adapter branch)
    "C1_B32_all_weights": (lambda: O.sdd_short(train_net="train"), 256, 256, 32),
    "C3_B32_rank4": (lambda: O.sdd_short(train_net="mosa_4", position=["0", "1", "2", "3", "4"]), 256, 256, 32),
}


@pytest.mark.parametrize("tag", list(TIMED_PATH))
def test_timed_path_eager_capture_replay_matches_oracle(dev, tag):
    """what bench.py TIMES is the replay of a captured step (hipGraph with four concurrent branches, fused
    multi-tensor Adam), not the eager first step.  Three batches at the benchmarked size go through train_epoch -- its step
    sequence for one shape is [eager, capture + replay, replay] -- and every step is compared with three oracle steps
    (train_step + adam_update) computed on the box: each step's loss (2e-5 relative), ADE / FDE (1e-4), the last step's
    gradients (5e-4 of the tensor's maximum) and the weights after the three Adam updates.  Kernel dispatch, workspace
    sizes and stream joins all depend on B, so the tiny-case captured-vs-eager test does not cover this."""
    mk, H, W, B = TIMED_PATH[tag]
    cfg = mk()
    lr = 1e-3
    sd0 = O.make_state_dict(cfg, seed=0, lora_b_std=0.05)
    scene = O.synthetic_scene(cfg, H, W, 0)
    trajs = [O.synthetic_trajectories(cfg, B, H, W, 31 + i) for i in range(3)]
    S = cfg.template_size
    in_t, gt_t = O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig)
    names = O.trainable_names(cfg, sd0)

    # ---- oracle: three steps with Adam in between
    sd = {k: v.clone() for k, v in sd0.items()}
    ms = {n: torch.zeros_like(sd[n]) for n in names}
    vs = {n: torch.zeros_like(sd[n]) for n in names}
    want, want_w = [], [{n: sd[n].clone() for n in names}]
    for i, traj in enumerate(trajs):
        r = O.train_step(sd, cfg, scene, traj, in_t, gt_t, names)
        want.append(r)
        for n in names:
            sd[n], ms[n], vs[n] = O.adam_update(sd[n], r["grads"][n], ms[n], vs[n], i + 1, lr)
        want_w.append({n: sd[n].clone() for n in names})

    # ---- product: one epoch per batch -> eager, capture + replay, replay
    model = build_model(cfg, sd0, dev)
    te, trn, sg = pkg("utils.train_epoch"), pkg("models.trainer"), pkg("utils.step_graph")
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    crit = trn.HipBCEWithLogitsLoss()
    gt_d, in_d = gt_t.to(dev), in_t.to(dev)
    images = {"scene0": scene[0].to(dev)}
    launched = []
    named = dict(model.named_parameters())
    got_w = [{n: named[n].detach().cpu().clone() for n in names}]
    for i, traj in enumerate(trajs):
        ade, fde, loss = te.train_epoch(
            model, loader_for(traj), images, opt, crit, cfg.loss_scale, dev, "sdd", None, gt_d, in_d, list(cfg.waypoints), i,
            cfg.obs_len, cfg.pred_len, B, 10000, cfg.resize_factor, cfg.network, False)
        entries = [e for c in sg._caches.get(model, {}).values() for e in c.entries.values()]
        launched.append("replay" if any(e.ready for e in entries) else "eager")
        got_w.append({n: named[n].detach().cpu().clone() for n in names})
        w = want[i]
        assert abs(loss - float(w["loss"])) <= 2e-5 * abs(float(w["loss"])), (i, launched[-1], loss, float(w["loss"]))
        assert abs(ade - float(w["ade"].mean())) <= 1e-4, (i, launched[-1], ade, float(w["ade"].mean()))
        assert abs(fde - float(w["fde"].mean())) <= 1e-4, (i, launched[-1], fde, float(w["fde"].mean()))
    if sg.enabled(None, dev):
        assert launched == ["eager", "replay", "replay"], launched      # the third step is a pure replay of the captured graph
        assert not any(e.failed for c in sg._caches.get(model, {}).values() for e in c.entries.values())
    # ---- the last (replayed) step's gradients, and EVERY step's weight update (the tiny-fixture form, tests/test_gpu_model.py:
    # the update of a step within 2 % of lr).  Adam normalises every entry's gradient, so an entry whose gradient is at rounding
    # level moves by +-lr whatever its sign: the entries compared are those whose gradient stood clear (>= 1 % of the tensor's
    # maximum, twenty times the gradient tolerance) in this and every earlier step -- the moments carry the earlier ones.
    for n in names:
        g, w = named[n].grad.detach().cpu().double(), want[-1]["grads"][n].double()
        err, tol = float((g - w).abs().max()), 5e-4 * float(w.abs().max()) + 1e-7
        assert err <= tol, f"grad {n} of the replayed step: max err {err:.3e} > {tol:.3e}"
        clear = torch.ones_like(sd0[n], dtype=torch.bool)
        for i, r in enumerate(want):
            clear &= r["grads"][n].abs() >= 1e-2 * r["grads"][n].abs().max()
            if not bool(clear.any()):
                break
            upd_got = got_w[i + 1][n] - got_w[i][n]
            upd_want = want_w[i + 1][n] - want_w[i][n]
            d = float((upd_got - upd_want)[clear].abs().max())
            assert d <= 0.02 * lr, f"weight update of {n} in step {i} ({launched[i]}): off by {d:.3e} = {d / lr:.4f} lr"


@pytest.mark.parametrize("K", [4, 20])
def test_eval_sweep_at_headline_batch_matches_oracle(dev, K):
    """C5 shape at B = 128: the K decoder passes run folded into the batch, G = max_effective_batch // B = 2 goal samples
    per pass (256 virtual batch items, encoder features read in place through the batch modulus).  K = 20 is BASELINE.json's
    configs[4] itself -- ten folded passes alternating between the two sweep streams, the full B = 128 batch against the host
    oracle (its cost is linear in K: about a minute on the box's host) --, K = 4 the quick form of it.

    Bounds (VERDICT r4 item 1 -- round 4 had loosened the per-coordinate bound to "1 % beyond 1e-4, none beyond 2e-2 px" for the
    Winograd launches; the fp64 adjudication, tests/wino_fp64.py -> profiles/r05_wino_fp64.log, showed that was not rounding but a
    defect: all 487 outliers sat in virtual image 0 of a folded pass, whose 256 x 32 x 256^2 tensor is exactly 2 GiB, so the zero-fill
    offset 0x80000000 of conv_wino_kernel was INSIDE its whole-tensor buffer descriptor and the top padding row read real memory.  With
    one image per descriptor the Winograd path is within the reference's own fp32 noise: |HIP - fp64| max 2.6e-5 px against
    |oracle32 - fp64| 3.8e-5):
      * every coordinate of every goal sample within 1e-4 px of the fp32 oracle, whichever convolution kernels ran;
      * every SAMPLE's ADE [K, B] (before the best-of-K minimum) and every trajectory's ADE / FDE within 1e-4;
      * against the oracle run in fp64 (the exact value of the same function of the same fp32 weights and inputs; sixteen of the 128
        trajectories incl. the first and the last virtual image, fp64 convolutions on the host being ~8x slower):
        |HIP - fp64| <= max(1e-4, 3 |oracle32 - fp64|) per coordinate -- the HIP path is inside the reference's own rounding noise."""
    cfg = O.sdd_long(train_net="train")
    H = W = 256
    B = 128
    sd = O.make_state_dict(cfg, seed=0)
    scene, traj = O.synthetic_scene(cfg, H, W, 0), O.synthetic_trajectories(cfg, B, H, W, 22)
    in_t = O.dist_template(cfg.template_size)
    gen = torch.Generator().manual_seed(5)
    want = O.eval_batch(sd, cfg, scene, traj, in_t, n_goal=K, n_traj=1, generator=gen)
    sub = list(range(8)) + list(range(B - 8, B))
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    want64 = O.eval_batch(sd64, cfg, scene.double(), traj[sub], in_t.double(), n_goal=K, n_traj=1,
                          waypoint_samples=want["waypoint_samples"][:, sub].double())["trajs"]
    model = build_model(cfg, sd, dev)
    ev = pkg("utils.evaluate")
    ops = pkg("ops")
    fut = traj[:, cfg.obs_len:]

    def run():
        caught = []
        h = model.softargmax_.register_forward_hook(lambda m, i, o: caught.append(o.detach().cpu()))
        ade, fde, df, _ = ev.evaluate(
            model, loader_for(traj), {"scene0": scene[0]}, dev, "sdd", None, in_t.to(dev), list(cfg.waypoints), "test", K, 1,
            cfg.obs_len, B, cfg.resize_factor, cfg.temperature, forced_samples={0: want["waypoint_samples"]})
        h.remove()
        assert len(caught) == K // 2 and caught[0].shape[0] == 256, [c.shape for c in caught]      # K / 2 folded passes of 2 x 128
        return torch.cat(caught).view(K, B, cfg.pred_len, 2), ade, fde, df

    def check(got, ade, fde, df, what):
        np.testing.assert_allclose(df["ade"].to_numpy(), want["ade"].numpy(), rtol=0, atol=1e-4, err_msg=what)
        np.testing.assert_allclose(df["fde"].to_numpy(), want["fde"].numpy(), rtol=0, atol=1e-4, err_msg=what)
        assert abs(ade - float(want["ade"].mean())) <= 1e-4 and abs(fde - float(want["fde"].mean())) <= 1e-4, what
        np.testing.assert_allclose(got.numpy(), want["trajs"].numpy(), rtol=0, atol=1e-4, err_msg=what + ": coordinates vs the fp32 oracle")
        ade_k = O.displacement_error(fut, got, cfg.resize_factor).mean(dim=2)             # [K, B]: every sample, not the best of K
        np.testing.assert_allclose(ade_k.numpy(), want["ade_k"].numpy(), rtol=0, atol=1e-4, err_msg=what + ": per-sample ADE")
        e32 = (want["trajs"][:, sub].double() - want64).abs()
        d64 = (got[:, sub].double() - want64).abs()
        bad = d64 > torch.clamp(3.0 * e32, min=1e-4)
        assert not bool(bad.any()), (what, float(d64.max()), float(e32.max()), int(bad.sum()))

    n0 = ops.wino_stats["launches"]
    check(*run(), "default kernels")
    assert ops.wino_stats["launches"] > n0 or not ops._wino_allowed          # (the Winograd launches ARE what ran)
    if K == 4:
        old = ops._wino_allowed
        ops._wino_allowed = False
        try:
            check(*run(), "implicit GEMM only")
        finally:
            ops._wino_allowed = old
