"""The CPU oracle (oracle/ynet_oracle.py) against the fixtures generated FROM THE REFERENCE
(oracle/gen_goldens.py): this is what pins the oracle.  Runs without a GPU and without /root/reference."""
import numpy as np
import pytest
import torch

from conftest import Golden, TINY_CASES
from oracle import ynet_oracle as O

TIGHT = dict(rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", TINY_CASES)
def test_train_step(case):
    g = Golden(case)
    cfg, m = g.cfg(), g.meta
    sd = g.state_dict()
    S = cfg.template_size
    in_t, gt_t = O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig)
    names = O.trainable_names(cfg, sd)
    assert names == [str(s) for s in g.z["step/trainable"]]
    assert sum(sd[n].numel() for n in names) == int(g.z["step/n_trainable"])
    st = O.train_step(sd, cfg, g.t("scene"), g.t("traj"), in_t, gt_t, names, keep_maps=True)
    for i, f in enumerate(st["features"]):
        g.compare(f"step/features{i}", f, **TIGHT)
    g.compare("step/goal_map", st["goal_map"], **TIGHT)
    g.compare("step/traj_map", st["traj_map"], **TIGHT)
    g.compare("step/pred_traj", st["pred_traj"], rtol=1e-5, atol=2e-5)
    g.compare("step/pred_goal", st["pred_goal"], rtol=1e-5, atol=2e-5)
    assert abs(float(st["loss"]) - float(g.z["step/loss"])) <= 1e-6 * abs(float(g.z["step/loss"]))
    assert abs(float(st["ade"].mean()) - float(g.z["step/ade"])) <= 1e-5
    assert abs(float(st["fde"].mean()) - float(g.z["step/fde"])) <= 1e-5
    for n in names:
        g.compare("step/grad/" + n, st["grads"][n], rtol=1e-4, atol=1e-6)
        z = torch.zeros_like(sd[n])
        p1, _, _ = O.adam_update(sd[n], st["grads"][n], z, z.clone(), 1, m["lr"])
        g.compare("step/after/" + n, p1, rtol=1e-5, atol=1e-7)
    for n in g.keys("step/buffers/"):       # BatchNorm running statistics of serial adapters after the step
        g.compare("step/buffers/" + n, st["buffers"][n], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", [c for c in TINY_CASES if "epoch/ade" in Golden(c).z.files])
def test_ragged_epoch(case):
    g = Golden(case)
    cfg, m = g.cfg(), g.meta
    sd = g.state_dict()
    S = cfg.template_size
    in_t, gt_t = O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig)
    names = O.trainable_names(cfg, sd)
    ms = {n: torch.zeros_like(sd[n]) for n in names}
    vs = {n: torch.zeros_like(sd[n]) for n in names}
    traj, B = g.t("epoch/traj"), m["B"]
    ades, fdes, tot = [], [], 0.0
    for step, i in enumerate(range(0, traj.shape[0], B), 1):
        r = O.train_step(sd, cfg, g.t("scene"), traj[i:i + B], in_t, gt_t, names)
        for n in names:
            sd[n], ms[n], vs[n] = O.adam_update(sd[n], r["grads"][n], ms[n], vs[n], step, m["lr"])
        ades.append(r["ade"]); fdes.append(r["fde"]); tot += float(r["loss"])
    assert abs(float(torch.cat(ades).mean()) - float(g.z["epoch/ade"])) <= 1e-4
    assert abs(float(torch.cat(fdes).mean()) - float(g.z["epoch/fde"])) <= 1e-4
    assert abs(tot - float(g.z["epoch/loss"])) <= 1e-5 * abs(float(g.z["epoch/loss"]))
    for n in names:
        g.compare("epoch/after/" + n, sd[n], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("case", [c for c in TINY_CASES if "eval/trajs" in Golden(c).z.files])
def test_eval_sweep(case):
    g = Golden(case)
    cfg, m = g.cfg(), g.meta
    sd = g.state_dict()
    in_t = O.dist_template(cfg.template_size)
    ev = O.eval_batch(sd, cfg, g.t("scene"), g.t("traj"), in_t, n_goal=m["n_goal"],
                      waypoint_samples=g.t("eval/waypoint_samples"))
    g.compare("eval/goal_map", ev["goal_map"], **TIGHT)
    np.testing.assert_allclose(ev["trajs"].numpy(), g.z["eval/trajs"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(ev["ade"].numpy(), g.z["eval/ade_per_traj"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(ev["fde"].numpy(), g.z["eval/fde_per_traj"], rtol=1e-5, atol=2e-5)
    # the sampler: same seed -> the reference's own draws
    torch.manual_seed(int(g.z["eval/seed"]))
    ev2 = O.eval_batch(sd, cfg, g.t("scene"), g.t("traj"), in_t, n_goal=m["n_goal"])
    assert torch.equal(ev2["waypoint_samples"], g.t("eval/waypoint_samples"))


TRAINED_CASES = ["trained_tiny_long", "trained_short_full"]


@pytest.mark.parametrize("case", TRAINED_CASES)
def test_trained_weights_step_and_sweep(case):
    """weights after a few hundred Adam steps OF THE REFERENCE (peaked heat-maps: the largest soft-max probability
    of a decoded plane is 20-1000x a flat map's) -- one training step and one K = 20 sweep by the reference, reproduced by the oracle."""
    g = Golden(case)
    cfg, m = g.cfg(), g.meta
    sd = g.state_dict()
    H, W = m["H"], m["W"]
    S = cfg.template_size
    in_t, gt_t = O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig)
    names = O.trainable_names(cfg, sd)
    assert names == [str(s) for s in g.z["step/trainable"]]
    st = O.train_step(sd, cfg, g.t("scene"), g.t("traj"), in_t, gt_t, names, keep_maps=True)
    assert float(np.median(g.z["step/peak_prob"])) >= 10.0 / (H * W)          # the maps ARE peaked
    g.compare("step/goal_map", st["goal_map"], rtol=1e-5, atol=1e-5)
    g.compare("step/traj_map", st["traj_map"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(st["pred_traj"].numpy(), g.z["step/pred_traj"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(st["pred_goal"].numpy(), g.z["step/pred_goal"], rtol=0, atol=1e-4)
    assert abs(float(st["loss"]) - float(g.z["step/loss"])) <= 1e-6 * abs(float(g.z["step/loss"]))
    np.testing.assert_allclose(st["ade"].numpy(), g.z["step/ade_per_traj"], rtol=0, atol=1e-4)
    for n in g.keys("step/grad/"):
        g.compare("step/grad/" + n, st["grads"][n], rtol=1e-4, atol=1e-5 * float(np.abs(g.z["step/grad/" + n]).max()) + 1e-7)
    # the sweep runs on the base weights (the step fixture may carry adapters on top of them)
    ecfg = O.Cfg(**{**cfg.__dict__, "train_net": m["trained_mode"], "position": []})
    esd = {k: v for k, v in sd.items() if "lora_" not in k}
    ev = O.eval_batch(esd, ecfg, g.t("scene"), g.t("traj"), in_t, n_goal=m["n_goal"], waypoint_samples=g.t("eval/waypoint_samples"))
    np.testing.assert_allclose(ev["trajs"].numpy(), g.z["eval/trajs"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(ev["ade"].numpy(), g.z["eval/ade_per_traj"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(ev["fde"].numpy(), g.z["eval/fde_per_traj"], rtol=0, atol=1e-4)
    torch.manual_seed(int(g.z["eval/seed"]))
    ev2 = O.eval_batch(esd, ecfg, g.t("scene"), g.t("traj"), in_t, n_goal=m["n_goal"])
    assert torch.equal(ev2["waypoint_samples"], g.t("eval/waypoint_samples"))


TTST_CWS_CASES = ["tiny_short_ttst", "tiny_long_cws", "tiny_long_ttst_cws_ntraj2"]


@pytest.mark.parametrize("case", TTST_CWS_CASES)
def test_eval_sweep_ttst_cws(case):
    """TTST (k-means of 10000 goal samples) / CWS (Gaussian waypoint prior): same RNG streams -> the reference's numbers."""
    g = Golden(case)
    cfg, m = g.cfg(), g.meta
    in_t = O.dist_template(cfg.template_size)
    torch.manual_seed(int(g.z["eval/seed"]))
    np.random.seed(int(g.z["eval/seed"]))
    ev = O.eval_batch(g.state_dict(), cfg, g.t("scene"), g.t("traj"), in_t, n_goal=m["n_goal"], n_traj=m["n_traj"],
                      use_ttst=m["use_ttst"], use_cws=m["use_cws"], cws_params=m["cws_params"] or None,
                      rel_thresh=m["rel_thresh"])
    np.testing.assert_allclose(ev["waypoint_samples"].numpy(), g.z["eval/waypoint_samples"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(ev["trajs"].numpy(), g.z["eval/trajs"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(ev["ade"].numpy(), g.z["eval/ade_per_traj"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(ev["fde"].numpy(), g.z["eval/fde_per_traj"], rtol=1e-5, atol=1e-4)


def test_leaf_vectors():
    g = Golden("kernels")
    np.testing.assert_allclose(O.softargmax2d(g.t("softargmax/x")).numpy(), g.z["softargmax/out"], rtol=1e-6, atol=1e-6)
    dm, gm = O.dist_template(210), O.gaussian_template(210, 31, 4)
    assert abs(float(dm.double().sum()) - float(g.z["template/dist_S210_sum"])) < 1e-6
    assert abs(float(gm.double().sum()) - float(g.z["template/gauss_S210_sum"])) < 1e-9
    assert float(gm.max()) == float(g.z["template/gauss_S210_peak"])
    for S in (1050, 1386):
        d = O.dist_template(S)
        assert abs(float(d.double().sum()) - float(g.z[f"template/dist_S{S}_sum"])) < 1e-3
        assert np.array_equal(torch.diagonal(d).numpy(), g.z[f"template/dist_S{S}_diag"])
    xy = g.z["patch/xy"]
    assert np.array_equal(O.crop_patches(dm, xy, 24, 40).numpy(), g.z["patch/dist"])
    assert np.array_equal(O.crop_patches(gm, xy, 24, 40).numpy(), g.z["patch/gauss"])


def test_known_answers():
    """identity at init, one-hot soft-argmax, patch peak position, BCE(0,0) = ln2."""
    cfg = O.sdd_short(train_net="mosa_2", position=["0", "1", "2", "3", "4"], enc=(8, 8, 16, 16, 16), dec=(16, 16, 16, 8, 8))
    sd = O.make_state_dict(cfg, seed=0, lora_b_std=0.0)
    base = {k: v for k, v in sd.items() if "lora" not in k}
    sc, mo = torch.rand(2, 6, 32, 32), torch.rand(2, 8, 32, 32)
    for a, b in zip(O.encoder(sd, cfg, sc, mo), O.encoder(base, cfg, sc, mo)):
        assert torch.equal(a, b)
    x = torch.full((1, 1, 16, 24), -30.0)
    x[0, 0, 9, 13] = 50.0
    assert torch.allclose(O.softargmax2d(x)[0, 0], torch.tensor([13.0, 9.0]), atol=1e-4)
    p = O.crop_patches(O.gaussian_template(210, 31, 4), np.array([[7.4, 11.6]], dtype=np.float32), 32, 32)[0]
    assert divmod(int(p.argmax()), 32) == (12, 7)
    assert abs(float(O.bce_logits_mean(torch.zeros(4, 4), torch.zeros(4, 4))) * 1000 - 1000 * np.log(2)) < 1e-3


def test_fullsize_weight_checksums():
    """The seeded full-size weights used by the GPU-side scalar goldens are reproducible here."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "fullsize_scalars.npz"))
    cfg = O.sdd_short(train_net="mosa_1", position=["0", "1", "2", "3", "4"])
    sd = O.make_state_dict(cfg, seed=3, lora_b_std=0.05)
    chk = sum(float(v.double().abs().sum()) for v in sd.values())
    assert abs(chk - float(z["C2_sdd_short_mosa1/weight_checksum"])) <= 1e-9 * chk
    assert sum(sd[n].numel() for n in O.trainable_names(cfg, sd)) == 8190
    assert sum(v.numel() for k, v in sd.items() if "lora" not in k) == 1641381


def test_device_sampler_restatement_known_answers():
    """Philox4x32-10 of the product's documented sampler against the Random123 known-answer vectors, and the sampler's
    frequencies (oracle.device_multinomial is what the GPU kernel ynet_multinomial is compared with, draw for draw)."""
    import numpy as np
    x0, x1 = O._philox4x32_10(np.array([0]), np.array([0]), np.array([0]), np.array([0]), 0, 0)
    assert (int(x0[0]), int(x1[0])) == (0x6627E8D5, 0xE169C58D)
    f = 0xFFFFFFFF
    x0, x1 = O._philox4x32_10(np.array([f]), np.array([f]), np.array([f]), np.array([f]), f, f)
    assert (int(x0[0]), int(x1[0])) == (0x408F276D, 0x41C83B0E)
    x0, x1 = O._philox4x32_10(np.array([0x243F6A88]), np.array([0x85A308D3]), np.array([0x13198A2E]), np.array([0x03707344]),
                              0xA4093822, 0x299F31D0)
    assert (int(x0[0]), int(x1[0])) == (0xD16CFE09, 0x94FDCCEB)
    q = torch.tensor([[0.1, 0.2, 0.0, 0.3, 0.4]])
    c = O.device_multinomial(q, 20000, True, None, 7)[0].numpy()
    np.testing.assert_allclose(np.bincount(c, minlength=5) / 20000, q[0].numpy(), atol=0.012)
    first = np.array([int(O.device_multinomial(q, 1, False, None, s)[0, 0]) for s in range(1500)])
    np.testing.assert_allclose(np.bincount(first, minlength=5) / 1500, q[0].numpy(), atol=0.04)
    w = O.device_multinomial(q, 4, False, None, 3)[0].tolist()
    assert sorted(w) == [0, 1, 3, 4]                          # without replacement: the zero entry never wins
    thr = O.device_multinomial(q, 3000, True, 0.6, 5)[0].numpy()   # entries below 0.6 * max = 0.24 are zeroed: 0.3 and 0.4 remain
    assert set(thr.tolist()) == {3, 4}
