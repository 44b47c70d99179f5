"""Generate tests/golden/*.npz from the REFERENCE itself (run only in the build container).

TEST INFRASTRUCTURE ONLY.  Imports /root/reference (read-only) with three import stubs from
oracle/_stubs (cv2, seaborn: never called on this path; loralib: restated 0.1.1 Conv2d, PARITY
UNPINNED), drives the reference's own ``YNetTrainer._train`` / ``train_epoch`` / ``evaluate`` on
seeded synthetic scenes, checks that oracle/ynet_oracle.py reproduces every captured quantity, and
writes inputs + expected outputs as fixtures.  Fixtures are data only: no reference source travels.

    python oracle/gen_goldens.py            # rewrites tests/golden/
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "_stubs"), REF, ROOT]

from models.trainer import YNetTrainer                                   # noqa: E402  (reference)
from utils.train_epoch import train_epoch as ref_train_epoch             # noqa: E402  (reference)
from utils.evaluate import evaluate as ref_evaluate                      # noqa: E402  (reference)
from utils.image_utils import (create_dist_mat, create_gaussian_heatmap_template,   # noqa: E402
                               get_patch as ref_get_patch)
from utils.softargmax import SoftArgmax2D as RefSoftArgmax               # noqa: E402
from utils import data_utils as ref_data_utils                            # noqa: E402  (reference: rot / fliplr / augment_data)

from oracle import ynet_oracle as O                                      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FULL_LIMIT = 40000
STRIDE = 5


def pack(store, name, t, stride=STRIDE):
    a = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    if a.size <= FULL_LIMIT:
        store[name] = a
    else:
        flat = a.reshape(-1)
        store[name + "__strided"] = flat[::stride].copy()
        if stride != STRIDE:
            store[name + "__stride"] = np.array(stride)
        store[name + "__shape"] = np.array(a.shape)
        store[name + "__sum"] = np.array(flat.astype(np.float64).sum())
        store[name + "__sqsum"] = np.array((flat.astype(np.float64) ** 2).sum())


def check(name, got, want, rtol=1e-5, atol=1e-6):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = want.detach().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max() if got.size else 0.0
    ok = np.allclose(got, want, rtol=rtol, atol=atol)
    print(f"    oracle vs reference {name:42s} max|d|={err:.3e} {'ok' if ok else 'MISMATCH'}")
    assert ok, name
    return err


def ref_params(cfg, batch_size, lr=1e-3, n_goal=20, n_traj=1):
    return dict(
        obs_len=cfg.obs_len, pred_len=cfg.pred_len, segmentation_model_fp=None, use_features_only=False,
        n_semantic_classes=cfg.n_classes, encoder_channels=list(cfg.enc), decoder_channels=list(cfg.dec),
        waypoints=list(cfg.waypoints), train_net=cfg.train_net, position=list(cfg.position),
        network=cfg.network, n_fusion=cfg.n_fusion, resize_factor=cfg.resize_factor,
        ckpt_path=None, dataset_name="sdd", batch_size=batch_size, lr=lr, n_epoch=1, n_goal=n_goal,
        n_traj=n_traj, kernlen=cfg.kernlen, nsig=cfg.nsig, e_unfreeze=10000, loss_scale=cfg.loss_scale,
        temperature=cfg.temperature, use_raw_data=False, save_every_n=1000, fine_tune=True,
        augment=False, ynet_bias=False, use_CWS=False, CWS_params=None, steps=[20], n_round=1,
        rel_threshold=0.01, use_TTST=False)


def loader_for(traj):
    meta = pd.DataFrame({"metaId": np.arange(traj.shape[0])})
    return [(traj.clone(), [meta], "scene0")]


class Capture:
    """Forward hooks on the reference model; keeps the FIRST call of each sub-network."""

    def __init__(self, model):
        self.data, self.soft = {}, []
        def keep(key, fn):
            def hook(m, i, o):
                if key not in self.data:
                    self.data[key] = fn(o)
            return hook

        def soft(m, i, o):
            self.soft.append(o.detach().clone())

        self.h = [
            model.encoder.register_forward_hook(keep("features", lambda o: [t.detach().clone() for t in o])),
            model.goal_decoder.register_forward_hook(keep("goal_map", lambda o: o.detach().clone())),
            model.traj_decoder.register_forward_hook(keep("traj_map", lambda o: o.detach().clone())),
            model.softargmax_.register_forward_hook(soft),
        ]

    def close(self):
        for h in self.h:
            h.remove()


def make_case(tag, cfg, H, W, B, seed, lora_b_std=0.05, lr=1e-3, n_goal=20, do_epoch=True, do_eval=True, adapter_std=0.05):
    if ONLY and not any(o in tag for o in ONLY):
        return
    print(f"[{tag}] network={cfg.network} train_net={cfg.train_net} position={list(cfg.position)} {H}x{W} B={B}")
    store = {}
    sd0 = O.make_state_dict(cfg, seed=seed, lora_b_std=lora_b_std, adapter_std=adapter_std)
    scene = O.synthetic_scene(cfg, H, W, seed)
    traj = O.synthetic_trajectories(cfg, B, H, W, seed)
    S = cfg.template_size
    meta = dict(obs_len=cfg.obs_len, pred_len=cfg.pred_len, waypoints=list(cfg.waypoints), enc=list(cfg.enc),
                dec=list(cfg.dec), network=cfg.network, n_fusion=cfg.n_fusion or 0, train_net=cfg.train_net,
                position=list(cfg.position), resize_factor=cfg.resize_factor, temperature=cfg.temperature,
                loss_scale=cfg.loss_scale, H=H, W=W, B=B, seed=seed, lr=lr, n_goal=n_goal, adapter_std=adapter_std,
                lora_source="oracle/_stubs/loralib (restated 0.1.1, PARITY UNPINNED)")
    store["meta"] = np.array(repr(meta))
    for k, v in sd0.items():
        store["sd/" + k] = v.numpy()
    store["scene"] = scene.numpy()
    store["traj"] = traj.numpy()

    # -------- one step through the reference's YNetTrainer._train (freeze policy + Adam + ckpt) --------
    params = ref_params(cfg, batch_size=B, lr=lr, n_goal=n_goal)
    with contextlib.redirect_stdout(io.StringIO()) as log:
        trainer = YNetTrainer(params, device=torch.device("cpu"))
    missing = trainer.model.load_state_dict(sd0, strict=True)
    images = {"scene0": scene[0].clone()}
    trainer.prepare_data = lambda *a, **k: (images, loader_for(traj), None)
    cap = Capture(trainer.model)
    with tempfile.TemporaryDirectory() as tmp:
        params["ckpt_path"] = tmp
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()) as log:
            trainer._train(None, None, None, None, "exp", **{k: v for k, v in params.items()})
        ck = torch.load(os.path.join(tmp, "exp.pt"), weights_only=False)
        ck_keys = list(ck.keys())
        ck_is_param = [isinstance(v, torch.nn.Parameter) for v in ck.values()]
    cap.close()
    out = log.getvalue()
    n_train_line = [l for l in out.splitlines() if "number of trainable parameters" in l][0]
    n_trainable = int(n_train_line.split(":")[1])
    model = trainer.model
    tr_names = [n for n, p in model.named_parameters() if p.requires_grad]
    # oracle
    in_t, gt_t = O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig)
    ref_in_t = torch.Tensor(create_dist_mat(size=S))
    ref_gt_t = torch.Tensor(create_gaussian_heatmap_template(size=S, kernlen=cfg.kernlen, nsig=cfg.nsig, normalize=False))
    assert torch.equal(in_t, ref_in_t), "dist template not bit-exact"
    assert torch.equal(gt_t, ref_gt_t), "gaussian template not bit-exact"
    o_names = O.trainable_names(cfg, sd0)
    assert o_names == tr_names, (o_names, tr_names)
    assert n_trainable == sum(sd0[n].numel() for n in o_names)
    st = O.train_step(sd0, cfg, scene, traj, in_t, gt_t, o_names, keep_maps=True)
    for i, (a, b) in enumerate(zip(st["features"], cap.data["features"])):
        check(f"features[{i}]", a, b)
        pack(store, f"step/features{i}", b)
    check("goal_map", st["goal_map"], cap.data["goal_map"])
    check("traj_map", st["traj_map"], cap.data["traj_map"])
    pack(store, "step/goal_map", cap.data["goal_map"])
    pack(store, "step/traj_map", cap.data["traj_map"])
    check("softargmax(traj)", st["pred_traj"], cap.soft[0])
    check("softargmax(goal)", st["pred_goal"], cap.soft[1])
    store["step/pred_traj"], store["step/pred_goal"] = cap.soft[0].numpy(), cap.soft[1].numpy()
    named = dict(model.named_parameters())
    for n in tr_names:
        check("grad " + n, st["grads"][n], named[n].grad, rtol=1e-4, atol=1e-6)
        store["step/grad/" + n] = named[n].grad.numpy()
        m0 = torch.zeros_like(sd0[n])
        p1, _, _ = O.adam_update(sd0[n], st["grads"][n], m0, m0.clone(), 1, lr)
        # _train reloads nothing when best_epoch == 0, so the model holds the post-step weights
        check("adam " + n, p1, named[n].detach(), rtol=1e-5, atol=1e-7)
        store["step/after/" + n] = named[n].detach().numpy()
    for n, b in model.named_buffers():       # BatchNorm statistics of serial adapters after the step
        if n in sd0:
            check("buffer " + n, st["buffers"][n], b, rtol=1e-5, atol=1e-6)
            store["step/buffers/" + n] = b.detach().numpy()
    store["step/trainable"] = np.array(tr_names)
    store["step/n_trainable"] = np.array(n_trainable)
    store["step/ckpt_keys"] = np.array(ck_keys)
    store["step/ckpt_is_parameter"] = np.array(ck_is_param)
    # scalar losses straight from the reference's train_epoch on a fresh copy of the weights
    tr2 = fresh_reference(cfg, sd0, B, lr, o_names)
    opt = torch.optim.Adam(tr2.parameters(), lr=lr)
    ade, fde, loss = ref_train_epoch(
        tr2, loader_for(traj), images, opt, torch.nn.BCEWithLogitsLoss(), cfg.loss_scale, torch.device("cpu"),
        "sdd", None, ref_gt_t, ref_in_t, list(cfg.waypoints), 0, cfg.obs_len, cfg.pred_len, B, 10000,
        cfg.resize_factor, cfg.network, False)
    check("step loss", st["loss"], loss, rtol=1e-6)
    check("step ADE", st["ade"].mean(), ade, rtol=1e-6)
    check("step FDE", st["fde"].mean(), fde, rtol=1e-6)
    store["step/loss"], store["step/ade"], store["step/fde"] = np.array(loss), np.array(ade), np.array(fde)
    store["step/goal_loss"], store["step/traj_loss"] = st["goal_loss"].numpy(), st["traj_loss"].numpy()

    # -------- a ragged epoch: N = 2B+1 trajectories, batch_size B -> 3 Adam steps --------
    if do_epoch:
        N = 2 * B + 1
        traj_e = O.synthetic_trajectories(cfg, N, H, W, seed + 7)
        tr3 = fresh_reference(cfg, sd0, B, lr, o_names)
        opt = torch.optim.Adam(tr3.parameters(), lr=lr)
        ade, fde, loss = ref_train_epoch(
            tr3, loader_for(traj_e), images, opt, torch.nn.BCEWithLogitsLoss(), cfg.loss_scale, torch.device("cpu"),
            "sdd", None, ref_gt_t, ref_in_t, list(cfg.waypoints), 0, cfg.obs_len, cfg.pred_len, B, 10000,
            cfg.resize_factor, cfg.network, False)
        # oracle epoch
        sd = {k: v.clone() for k, v in sd0.items()}
        ms = {n: torch.zeros_like(sd[n]) for n in o_names}
        vs = {n: torch.zeros_like(sd[n]) for n in o_names}
        ades, fdes, tot = [], [], 0.0
        for step, i in enumerate(range(0, N, B), 1):
            r = O.train_step(sd, cfg, scene, traj_e[i:i + B], in_t, gt_t, o_names)
            for n in o_names:
                sd[n], ms[n], vs[n] = O.adam_update(sd[n], r["grads"][n], ms[n], vs[n], step, lr)
            ades.append(r["ade"]); fdes.append(r["fde"]); tot = tot + r["loss"]
        check("epoch ADE", torch.cat(ades).mean(), ade, rtol=1e-5)
        check("epoch FDE", torch.cat(fdes).mean(), fde, rtol=1e-5)
        check("epoch loss", tot, loss, rtol=1e-5)
        named3 = dict(tr3.named_parameters())
        for n in o_names:
            check("epoch param " + n, sd[n], named3[n].detach(), rtol=1e-4, atol=1e-6)
            store["epoch/after/" + n] = named3[n].detach().numpy()
        store["epoch/traj"] = traj_e.numpy()
        store["epoch/ade"], store["epoch/fde"], store["epoch/loss"] = np.array(ade), np.array(fde), np.array(loss)

    # -------- evaluation sweep (K = n_goal) with the reference's own sampling --------
    if do_eval:
        tr4 = fresh_reference(cfg, sd0, B, lr, o_names)
        cap = Capture(tr4)
        torch.manual_seed(seed + 3)
        ade, fde, df, td = ref_evaluate(
            tr4, loader_for(traj), images, torch.device("cpu"), "sdd", None, ref_in_t, list(cfg.waypoints), "test",
            n_goal, 1, cfg.obs_len, B, cfg.resize_factor, cfg.temperature, False, False, 0.01, None,
            return_preds=True, return_samples=True, network=cfg.network)
        cap.close()
        wps = torch.from_numpy(td["waypoint_sample"]).permute(2, 0, 1, 3).contiguous()   # [K,B,nwp,2]
        ev = O.eval_batch(sd0, cfg, scene, traj, in_t, n_goal=n_goal, waypoint_samples=wps)
        check("eval goal_map", ev["goal_map"], td["goal_map"])
        check("eval trajs", ev["trajs"], torch.stack(cap.soft), rtol=1e-5, atol=2e-5)
        check("eval ade/traj", ev["ade"], df["ade"].to_numpy(), rtol=1e-5, atol=2e-5)
        check("eval fde/traj", ev["fde"], df["fde"].to_numpy(), rtol=1e-5, atol=2e-5)
        # the sampler itself, re-seeded: multinomial on the oracle's sigmoid map must redraw the same points
        torch.manual_seed(seed + 3)
        ev2 = O.eval_batch(sd0, cfg, scene, traj, in_t, n_goal=n_goal)
        assert torch.equal(ev2["waypoint_samples"], wps), "sampling restatement diverges from the reference"
        store["eval/waypoint_samples"] = wps.numpy()
        pack(store, "eval/goal_map", torch.from_numpy(td["goal_map"]))
        store["eval/trajs"] = torch.stack(cap.soft).numpy()
        store["eval/ade_per_traj"], store["eval/fde_per_traj"] = df["ade"].to_numpy(), df["fde"].to_numpy()
        store["eval/ade"], store["eval/fde"] = np.array(ade), np.array(fde)
        store["eval/seed"] = np.array(seed + 3)

    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **store)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def ttst_cws_case(tag, cfg, H, W, B, seed, n_goal, n_traj, use_ttst, use_cws, cws_params, rel_thresh=0.002):
    """Evaluation sweep with the test-time sampling trick and / or conditioned waypoint sampling
    (utils/evaluate.py:134-161, 172-224), RNG streams shared between the reference and the oracle."""
    if ONLY and not any(o in tag for o in ONLY):
        return
    print(f"[{tag}] ttst={use_ttst} cws={use_cws} n_goal={n_goal} n_traj={n_traj} {H}x{W} B={B}")
    store = {}
    sd0 = O.make_state_dict(cfg, seed=seed)
    scene = O.synthetic_scene(cfg, H, W, seed)
    traj = O.synthetic_trajectories(cfg, B, H, W, seed)
    S = cfg.template_size
    meta = dict(obs_len=cfg.obs_len, pred_len=cfg.pred_len, waypoints=list(cfg.waypoints), enc=list(cfg.enc),
                dec=list(cfg.dec), network=cfg.network, n_fusion=cfg.n_fusion or 0, train_net=cfg.train_net,
                position=list(cfg.position), resize_factor=cfg.resize_factor, temperature=cfg.temperature,
                loss_scale=cfg.loss_scale, H=H, W=W, B=B, seed=seed, lr=1e-3, n_goal=n_goal, n_traj=n_traj,
                use_ttst=use_ttst, use_cws=use_cws, cws_params=cws_params or {}, rel_thresh=rel_thresh)
    store["meta"] = np.array(repr(meta))
    for k, v in sd0.items():
        store["sd/" + k] = v.numpy()
    store["scene"], store["traj"] = scene.numpy(), traj.numpy()
    in_t = O.dist_template(S)
    model = fresh_reference(cfg, sd0, B, 1e-3, O.trainable_names(cfg, sd0))
    images = {"scene0": scene[0].clone()}
    cap = Capture(model)
    torch.manual_seed(seed + 3)
    np.random.seed(seed + 3)
    ade, fde, df, td = ref_evaluate(
        model, loader_for(traj), images, torch.device("cpu"), "sdd", None, in_t, list(cfg.waypoints), "test",
        n_goal, n_traj, cfg.obs_len, B, cfg.resize_factor, cfg.temperature, use_ttst, use_cws, rel_thresh, cws_params,
        return_preds=True, return_samples=True, network=cfg.network)
    cap.close()
    wps = torch.from_numpy(td["waypoint_sample"]).permute(2, 0, 1, 3).contiguous()       # [K,B,nwp,2]
    n_first = 1 if use_ttst else 0          # TTST calls the soft-argmax once more, before the trajectory passes
    ref_trajs = torch.stack(cap.soft[n_first:])
    torch.manual_seed(seed + 3)
    np.random.seed(seed + 3)
    ev = O.eval_batch(sd0, cfg, scene, traj, in_t, n_goal=n_goal, n_traj=n_traj, use_ttst=use_ttst, use_cws=use_cws,
                      cws_params=cws_params, rel_thresh=rel_thresh)
    check("waypoint samples", ev["waypoint_samples"], wps, rtol=1e-5, atol=1e-4)
    check("eval trajs", ev["trajs"], ref_trajs, rtol=1e-5, atol=1e-4)
    check("eval ade/traj", ev["ade"], df["ade"].to_numpy(), rtol=1e-5, atol=1e-4)
    check("eval fde/traj", ev["fde"], df["fde"].to_numpy(), rtol=1e-5, atol=1e-4)
    if use_ttst:
        # the 10000-sample draw itself, so that device runs can be fed the very same points
        torch.manual_seed(seed + 3)
        goal_map = torch.from_numpy(td["goal_map"])
        sig = torch.sigmoid(goal_map[:, list(cfg.waypoints)] / cfg.temperature)
        draw = O.sample_coords(sig[:, -1:], 10000, rel_threshold=rel_thresh, replacement=True).permute(2, 0, 1, 3)
        store["eval/ttst_samples"] = draw.numpy().astype(np.int16)
    store["eval/waypoint_samples"] = wps.numpy()
    store["eval/trajs"] = ref_trajs.numpy()
    store["eval/ade_per_traj"], store["eval/fde_per_traj"] = df["ade"].to_numpy(), df["fde"].to_numpy()
    store["eval/ade"], store["eval/fde"] = np.array(ade), np.array(fde)
    store["eval/seed"] = np.array(seed + 3)
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **store)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def trained_case(tag, cfg, H, W, B, seed, steps, lr, step_cfg=None, n_goal=20):
    """fixtures on TRAINED weights.  Default-initialised weights decode diffuse heat-maps; SURVEY section 7 measured the
    reference's own fp32-vs-fp64 soft-argmax gap at 1.7e-5 px on flat maps and 3.0e-4 px on peaky ones (utils/softargmax.py:55-81).  The
    REFERENCE's train_epoch (utils/train_epoch.py:44-126, every weight trainable, models/trainer.py:222-235) runs `steps` Adam steps of
    batch B on the synthetic scene; from the resulting state dict: one more training step in `step_cfg`'s mode (default: cfg's own) and
    one K-sample evaluation sweep, both by the reference, both reproduced by the oracle, stored with the weights."""
    if ONLY and not any(o in tag for o in ONLY):
        return
    import time
    print(f"[{tag}] {steps} reference Adam steps of batch {B} at {H}x{W}, lr {lr}")
    store = {}
    sd0 = O.make_state_dict(cfg, seed=seed)
    scene = O.synthetic_scene(cfg, H, W, seed)
    images = {"scene0": scene[0].clone()}
    S = cfg.template_size
    in_t, gt_t = O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig)
    names0 = O.trainable_names(cfg, sd0)
    model = fresh_reference(cfg, sd0, B, lr, names0)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    t0 = time.time()
    traj_all = O.synthetic_trajectories(cfg, steps * B, H, W, seed + 50)
    ade, fde, loss = ref_train_epoch(
        model, loader_for(traj_all), images, opt, torch.nn.BCEWithLogitsLoss(), cfg.loss_scale, torch.device("cpu"), "sdd", None,
        gt_t, in_t, list(cfg.waypoints), 0, cfg.obs_len, cfg.pred_len, B, 10000, cfg.resize_factor, cfg.network, False)
    print(f"  trained in {time.time() - t0:.0f} s: mean loss per step {float(loss) / steps:.3f}, ADE {float(ade):.3f}, FDE {float(fde):.3f}")
    trained = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith("semantic_segmentation")}
    assert set(trained) == set(sd0)
    # ---- the step fixture (step_cfg's adapters on top of the trained base weights)
    scfg = step_cfg or cfg
    sd1 = O.make_state_dict(scfg, seed=seed + 1, lora_b_std=0.05)
    for k, v in trained.items():
        assert sd1[k].shape == v.shape, k
        sd1[k] = v.clone()
    traj = O.synthetic_trajectories(scfg, B, H, W, seed + 60)
    names = O.trainable_names(scfg, sd1)
    meta = dict(obs_len=scfg.obs_len, pred_len=scfg.pred_len, waypoints=list(scfg.waypoints), enc=list(scfg.enc), dec=list(scfg.dec),
                network=scfg.network, n_fusion=scfg.n_fusion or 0, train_net=scfg.train_net, position=list(scfg.position),
                resize_factor=scfg.resize_factor, temperature=scfg.temperature, loss_scale=scfg.loss_scale, H=H, W=W, B=B, seed=seed,
                lr=lr, n_goal=n_goal, adapter_std=0.0, trained_steps=steps, trained_mode=cfg.train_net,
                lora_source="oracle/_stubs/loralib (restated 0.1.1, PARITY UNPINNED)")
    store["meta"] = np.array(repr(meta))
    for k, v in sd1.items():
        store["sd/" + k] = v.numpy()
    store["scene"], store["traj"] = scene.numpy(), traj.numpy()
    m1 = fresh_reference(scfg, sd1, B, lr, names)
    cap = Capture(m1)
    opt1 = torch.optim.SGD(m1.parameters(), lr=0.0)          # (gradients only: the weights stay the fixture's)
    ade, fde, loss = ref_train_epoch(
        m1, loader_for(traj), images, opt1, torch.nn.BCEWithLogitsLoss(), scfg.loss_scale, torch.device("cpu"), "sdd", None,
        gt_t, in_t, list(scfg.waypoints), 0, scfg.obs_len, scfg.pred_len, B, 10000, scfg.resize_factor, scfg.network, False)
    cap.close()
    st = O.train_step(sd1, scfg, scene, traj, in_t, gt_t, names, keep_maps=True)
    check("step loss", st["loss"], loss, rtol=1e-6)
    check("step ADE", st["ade"].mean(), ade, rtol=1e-6)
    check("step FDE", st["fde"].mean(), fde, rtol=1e-6)
    check("goal_map", st["goal_map"], cap.data["goal_map"], rtol=1e-5, atol=1e-5)
    check("traj_map", st["traj_map"], cap.data["traj_map"], rtol=1e-5, atol=1e-5)
    check("softargmax(traj)", st["pred_traj"], cap.soft[0], rtol=1e-6, atol=1e-4)
    check("softargmax(goal)", st["pred_goal"], cap.soft[1], rtol=1e-6, atol=1e-4)
    big = H * W * B > 100000            # (full-size case: the weights are 6.6 MB already -- maps every 23rd element, no feature maps)
    if not big:
        for i, b in enumerate(cap.data["features"]):
            pack(store, f"step/features{i}", b)
    pack(store, "step/goal_map", cap.data["goal_map"], 23 if big else STRIDE)
    pack(store, "step/traj_map", cap.data["traj_map"], 23 if big else STRIDE)
    store["step/pred_traj"], store["step/pred_goal"] = cap.soft[0].numpy(), cap.soft[1].numpy()
    named = dict(m1.named_parameters())
    for n in names:
        check("grad " + n, st["grads"][n], named[n].grad, rtol=1e-4, atol=1e-5 * float(named[n].grad.abs().max()) + 1e-7)
        if sd1[n].numel() <= FULL_LIMIT:
            store["step/grad/" + n] = named[n].grad.numpy()
    store["step/trainable"] = np.array(names)
    store["step/loss"], store["step/ade"], store["step/fde"] = np.array(loss), np.array(ade), np.array(fde)
    store["step/ade_per_traj"], store["step/fde_per_traj"] = st["ade"].numpy(), st["fde"].numpy()
    # how peaked the decoded maps are: the largest soft-max probability of every plane (a flat 256^2 map has 1.5e-5)
    tm = cap.data["traj_map"]
    pk = torch.softmax(tm.flatten(2), dim=2).max(dim=2)[0]
    store["step/peak_prob"] = pk.numpy()
    print(f"  largest soft-max probability per plane: median {float(pk.median()):.4f}, max {float(pk.max()):.4f} (flat: {1.0 / (H * W):.2e})")
    # ---- the sweep fixture on the trained weights of cfg itself
    m2 = fresh_reference(cfg, trained, B, lr, names0)
    cap = Capture(m2)
    torch.manual_seed(seed + 3)
    ade, fde, df, td = ref_evaluate(
        m2, loader_for(traj), images, torch.device("cpu"), "sdd", None, in_t, list(cfg.waypoints), "test", n_goal, 1, cfg.obs_len, B,
        cfg.resize_factor, cfg.temperature, False, False, 0.01, None, return_preds=True, return_samples=True, network=cfg.network)
    cap.close()
    wps = torch.from_numpy(td["waypoint_sample"]).permute(2, 0, 1, 3).contiguous()
    ev = O.eval_batch(trained, cfg, scene, traj, in_t, n_goal=n_goal, waypoint_samples=wps)
    check("eval goal_map", ev["goal_map"], td["goal_map"], rtol=1e-5, atol=1e-5)
    check("eval trajs", ev["trajs"], torch.stack(cap.soft), rtol=1e-6, atol=1e-4)
    check("eval ade/traj", ev["ade"], df["ade"].to_numpy(), rtol=1e-5, atol=1e-4)
    check("eval fde/traj", ev["fde"], df["fde"].to_numpy(), rtol=1e-5, atol=1e-4)
    torch.manual_seed(seed + 3)
    ev2 = O.eval_batch(trained, cfg, scene, traj, in_t, n_goal=n_goal)
    assert torch.equal(ev2["waypoint_samples"], wps), "sampling restatement diverges from the reference"
    store["eval/waypoint_samples"] = wps.numpy()
    pack(store, "eval/goal_map", torch.from_numpy(td["goal_map"]), 23 if big else STRIDE)
    store["eval/trajs"] = torch.stack(cap.soft).numpy()
    store["eval/ade_per_traj"], store["eval/fde_per_traj"] = df["ade"].to_numpy(), df["fde"].to_numpy()
    store["eval/ade"], store["eval/fde"] = np.array(ade), np.array(fde)
    store["eval/seed"] = np.array(seed + 3)
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **store)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def augment_case():
    """SURVEY 8(f)-4, the pinnable piece: the REFERENCE's rot / fliplr / augment_data (utils/data_utils.py:113-233)
    on small label maps and a 3-channel image, with cv2.rotate / cv2.flip / cv2.imread served by oracle/_stubs/cv2.py as their NumPy
    equivalents (np.rot90 / np.fliplr / an in-memory dict) -- the permutations are OpenCV's documented ones, but OpenCV itself is absent:
    PARITY UNPINNED at that boundary, and the fixture says so."""
    import cv2 as stub
    rng = np.random.RandomState(7)
    scenes = {"sceneA": rng.randint(0, 6, size=(6, 10)).astype(np.uint8), "sceneB": rng.randint(0, 6, size=(8, 4)).astype(np.uint8)}
    rows = []
    meta = 0
    for sid, im in scenes.items():
        for _ in range(3):
            for f in range(4):
                rows.append({"frame": f, "trackId": meta, "x": float(rng.uniform(0, im.shape[1])), "y": float(rng.uniform(0, im.shape[0])),
                             "sceneId": sid, "metaId": meta})
            meta += 1
    df = pd.DataFrame(rows)
    stub.FILES.clear()
    for sid, im in scenes.items():
        stub.FILES[os.path.join("mem", sid, "oracle.png")] = im
    out_df, out_images = ref_data_utils.augment_data(df.copy(), image_path="mem", images={}, image_file="oracle.png", seg_mask=True)
    store = {"meta": np.array(repr({"cv2": "oracle/_stubs/cv2.py: rotate = np.rot90(., 1), flip = np.fliplr (NumPy equivalents, PARITY UNPINNED against OpenCV)"})),
             "in/x": df["x"].to_numpy(), "in/y": df["y"].to_numpy(), "in/metaId": df["metaId"].to_numpy(), "in/frame": df["frame"].to_numpy(),
             "in/sceneId": np.array(df["sceneId"].tolist()),
             "out/x": out_df["x"].to_numpy(), "out/y": out_df["y"].to_numpy(), "out/metaId": out_df["metaId"].to_numpy(),
             "out/sceneId": np.array(out_df["sceneId"].tolist()), "out/keys": np.array(list(out_images.keys()))}
    for sid, im in scenes.items():
        store["in/image/" + sid] = im
    for key, im in out_images.items():
        store["out/image/" + key] = im
    # rot / fliplr on their own, on a 3-channel image, every k
    img3 = rng.randint(0, 255, size=(5, 7, 3)).astype(np.uint8)
    pts = pd.DataFrame({"x": rng.uniform(0, 7, 6), "y": rng.uniform(0, 5, 6)})
    store["rot/image"], store["rot/x"], store["rot/y"] = img3, pts["x"].to_numpy(), pts["y"].to_numpy()
    for k in (1, 2, 3):
        d, im = ref_data_utils.rot(pts.copy(), img3.copy(), k)
        store[f"rot/k{k}/image"], store[f"rot/k{k}/x"], store[f"rot/k{k}/y"] = im, d["x"].to_numpy(), d["y"].to_numpy()
        assert np.array_equal(im, np.rot90(img3, k))
    d, im = ref_data_utils.fliplr(pts.copy(), img3.copy())
    store["flip/image"], store["flip/x"], store["flip/y"] = im, d["x"].to_numpy(), d["y"].to_numpy()
    path = os.path.join(OUT, "augment.npz")
    np.savez_compressed(path, **store)
    print(f"  wrote {path}: {len(df)} -> {len(out_df)} rows, {len(out_images)} scenes")


class RecordingBCE(torch.nn.Module):
    """nn.BCEWithLogitsLoss that remembers every value it returned (train_epoch calls it twice per step: goal, trajectory)."""

    def __init__(self):
        super().__init__()
        self.inner = torch.nn.BCEWithLogitsLoss()
        self.values = []

    def forward(self, x, t):
        v = self.inner(x, t)
        self.values.append(float(v.detach()))
        return v


def trajectory_case(tag, cfg, H, W, B, seed, epochs, steps_per_epoch, lr, milestones, lora_b_std=0.0, n_eval=32, n_goal=20, init_from=None):
    """a multi-step TRAINING TRAJECTORY of the reference (every other fixture is <= 3 consecutive steps).
    The reference's own loop (models/trainer.py:222-235: per epoch train_epoch(...), then lr_scheduler.step(); Adam + MultiStepLR
    as models/trainer.py:197-201) runs `epochs` epochs of `steps_per_epoch` steps of batch B; stored: every step's loss
    (goal + trajectory, x loss_scale, as utils/train_epoch.py:94,106-107 sums them), every epoch's (ADE, FDE, loss) return value,
    the trajectories, and a K-sample sweep of the FINAL weights over `n_eval` held-out trajectories (mean best-of-K ADE / FDE).
    The run is done TWICE, with 8 and with 3 intra-op threads: MKL-DNN then sums in a different order, and the difference between
    the two runs is the reference's OWN fp32 noise along the trajectory -- the yardstick the product's deviation is held against.
    The initial weights are O.make_state_dict(cfg, seed, lora_b_std) (stored as a checksum; the trajectories are stored)."""
    if ONLY and not any(o in tag for o in ONLY):
        return
    import time
    n_steps = epochs * steps_per_epoch
    print(f"[{tag}] {epochs} x {steps_per_epoch} reference Adam steps of batch {B} at {H}x{W}, lr {lr}, milestones {milestones}")
    sd0 = O.make_state_dict(cfg, seed=seed, lora_b_std=lora_b_std)
    if init_from is not None:
        # start from the weights of another fixture (trained_short_full: 200 reference Adam steps from scratch): the run then is a smooth
        # continuation -- a from-scratch run at lr 1e-3 oscillates in its first dozen steps and amplifies every rounding difference
        z0 = np.load(os.path.join(OUT, init_from + ".npz"), allow_pickle=False)
        for k in sd0:
            sd0[k] = torch.from_numpy(np.array(z0["sd/" + k]))
        seed = int(eval(str(z0["meta"]))["seed"])            # (the scene those weights were trained on)
    names = O.trainable_names(cfg, sd0)
    scene = O.synthetic_scene(cfg, H, W, seed)
    images = {"scene0": scene[0].clone()}
    S = cfg.template_size
    in_t, gt_t = O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig)
    trajs = [O.synthetic_trajectories(cfg, steps_per_epoch * B, H, W, seed + 100 + e) for e in range(epochs)]
    eval_traj = O.synthetic_trajectories(cfg, n_eval, H, W, seed + 90)

    def run(threads):
        torch.set_num_threads(threads)
        model = fresh_reference(cfg, sd0, B, lr, names)
        opt = torch.optim.Adam(model.parameters(), lr=lr)
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=list(milestones), gamma=0.1)
        crit = RecordingBCE()
        rets, lrs = [], []
        t0 = time.time()
        for e in range(epochs):
            lrs.append(opt.param_groups[0]["lr"])
            rets.append(ref_train_epoch(model, loader_for(trajs[e]), images, opt, crit, cfg.loss_scale, torch.device("cpu"), "sdd", None,
                                        gt_t, in_t, list(cfg.waypoints), e, cfg.obs_len, cfg.pred_len, B, 10000, cfg.resize_factor,
                                        cfg.network, False))
            sched.step()
        v = np.array(crit.values, dtype=np.float64).reshape(n_steps, 2)
        losses = v.sum(axis=1) * cfg.loss_scale
        print(f"  {threads} threads: {time.time() - t0:.0f} s; loss step 1 {losses[0]:.3f} -> step {n_steps} {losses[-1]:.3f}; "
              f"epoch returns {[tuple(round(float(x), 3) for x in r) for r in rets]}")
        torch.manual_seed(seed + 3)
        ade, fde, df, td = ref_evaluate(model, loader_for(eval_traj), images, torch.device("cpu"), "sdd", None, in_t, list(cfg.waypoints),
                                        "test", n_goal, 1, cfg.obs_len, n_eval, cfg.resize_factor, cfg.temperature, False, False,
                                        0.01, None, return_preds=True, return_samples=True, network=cfg.network)
        wps = torch.from_numpy(td["waypoint_sample"]).permute(2, 0, 1, 3).contiguous()       # [K, n_eval, nwp, 2]: the draws of this sweep
        final = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith("semantic_segmentation")}
        return (losses, np.array([[float(x) for x in r] for r in rets]), np.array(lrs), (float(ade), float(fde)), final, wps.numpy(),
                df["ade"].to_numpy(), df["fde"].to_numpy())

    a = run(8)
    b = run(3)
    torch.set_num_threads(8)
    noise = np.abs(a[0] - b[0]) / np.abs(a[0])
    print(f"  the reference against itself (8 vs 3 threads): per-step loss differs by max {noise[:10].max():.2e} in steps 1-10, "
          f"{noise.max():.2e} overall, {noise[-1]:.2e} at the last step; sweep ADE {a[3][0]:.4f} vs {b[3][0]:.4f}")
    meta = dict(obs_len=cfg.obs_len, pred_len=cfg.pred_len, waypoints=list(cfg.waypoints), enc=list(cfg.enc), dec=list(cfg.dec),
                network=cfg.network, n_fusion=cfg.n_fusion or 0, train_net=cfg.train_net, position=list(cfg.position),
                resize_factor=cfg.resize_factor, temperature=cfg.temperature, loss_scale=cfg.loss_scale, H=H, W=W, B=B, seed=seed, lr=lr,
                epochs=epochs, steps_per_epoch=steps_per_epoch, milestones=list(milestones), lora_b_std=lora_b_std, n_eval=n_eval, init_from=init_from or "",
                n_goal=n_goal, lora_source="oracle/_stubs/loralib (restated 0.1.1, PARITY UNPINNED)")
    store = {"meta": np.array(repr(meta)),
             "weight_checksum": np.array(sum(float(v.double().abs().sum()) for v in sd0.values())),
             "traj": torch.stack(trajs).numpy(), "eval_traj": eval_traj.numpy(),
             "loss_per_step": a[0], "loss_per_step_other_threads": b[0],
             "epoch_returns": a[1], "epoch_returns_other_threads": b[1], "lr_per_epoch": a[2],
             "sweep_ade_fde": np.array(a[3]), "sweep_ade_fde_other_threads": np.array(b[3]),
             "sweep_waypoint_samples": a[5], "sweep_ade_per_traj": a[6], "sweep_fde_per_traj": a[7],
             "final_weight_l2": np.array([float(a[4][n].double().norm()) for n in names]),
             "final_weight_moved_l2": np.array([float((a[4][n] - sd0[n]).double().norm()) for n in names]),
             "final_weight_self_noise_l2": np.array([float((a[4][n] - b[4][n]).double().norm()) for n in names]),
             "trainable": np.array(names)}
    path = os.path.join(OUT, tag + ".npz")
    np.savez_compressed(path, **store)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def fresh_reference(cfg, sd0, B, lr, trainable):
    with contextlib.redirect_stdout(io.StringIO()):
        t = YNetTrainer(ref_params(cfg, B, lr), device=torch.device("cpu"))
    t.model.load_state_dict(sd0, strict=True)
    for n, p in t.model.named_parameters():
        p.requires_grad = n in set(trainable)
    return t.model


def kernel_vectors():
    """Op-level vectors from the reference's leaf functions (soft-argmax, get_patch, templates)."""
    store = {}
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 4, 24, 40, generator=g) * 3
    x[0, 0] = -20
    x[0, 0, 5, 7] = 50.0                                   # known answer: (7, 5)
    x[1, 1] *= 10                                          # peaky
    store["softargmax/x"] = x.numpy()
    store["softargmax/out"] = RefSoftArgmax(normalized_coordinates=False)(x).numpy()
    check("softargmax", O.softargmax2d(x), store["softargmax/out"])
    S = 210
    dm = torch.Tensor(create_dist_mat(size=S))
    gm = torch.Tensor(create_gaussian_heatmap_template(size=S, kernlen=31, nsig=4, normalize=False))
    assert torch.equal(dm, O.dist_template(S)) and torch.equal(gm, O.gaussian_template(S, 31, 4))
    store["template/dist_S210_sum"] = np.array(float(dm.double().sum()))
    store["template/gauss_S210_sum"] = np.array(float(gm.double().sum()))
    store["template/gauss_S210_peak"] = np.array(float(gm.max()))
    for S_full in (1050, 1386):
        dmf = torch.Tensor(create_dist_mat(size=S_full))
        assert torch.equal(dmf, O.dist_template(S_full))
        store[f"template/dist_S{S_full}_sum"] = np.array(float(dmf.double().sum()))
        store[f"template/dist_S{S_full}_diag"] = torch.diagonal(dmf).numpy()
    xy = np.array([[10.5, 3.5], [11.5, 4.5], [0.0, 0.0], [39.49, 23.51], [12.3, 7.8], [2.5, 0.5]], dtype=np.float32)
    patches = torch.stack(ref_get_patch(dm, xy, 24, 40))
    assert torch.equal(patches, O.crop_patches(dm, xy, 24, 40))
    store["patch/xy"] = xy
    store["patch/dist"] = patches.numpy()
    store["patch/gauss"] = torch.stack(ref_get_patch(gm, xy, 24, 40)).numpy()
    np.savez_compressed(os.path.join(OUT, "kernels.npz"), **store)
    print("  wrote kernels.npz")


def fullsize_scalars():
    """Loss / ADE / FDE / grad norms at the BASELINE.json shapes (small B) for the GPU-side tests."""
    rows = {}
    pos5 = ["0", "1", "2", "3", "4"]
    cases = {
        "C1_sdd_short_train": (O.sdd_short(train_net="train"), 256, 256, 2),
        "C2_sdd_short_mosa1": (O.sdd_short(train_net="mosa_1", position=pos5), 256, 256, 2),
        "C3_sdd_short_mosa4": (O.sdd_short(train_net="mosa_4", position=pos5), 256, 256, 2),
        "C4_ind_long_fusion_mosa3_scene": (O.ind_long(network="fusion", n_fusion=2, train_net="mosa_3", position=["scene"]), 512, 512, 1),
        "C5_sdd_long_eval": (O.sdd_long(train_net="train"), 256, 256, 2),
    }
    store = {}
    for tag, (cfg, H, W, B) in cases.items():
        print(f"[{tag}]")
        sd0 = O.make_state_dict(cfg, seed=3, lora_b_std=0.05)
        scene, traj = O.synthetic_scene(cfg, H, W, 3), O.synthetic_trajectories(cfg, B, H, W, 3)
        S = cfg.template_size
        in_t, gt_t = O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig)
        names = O.trainable_names(cfg, sd0)
        model = fresh_reference(cfg, sd0, B, 1e-3, names)
        images = {"scene0": scene[0].clone()}
        store[tag + "/weight_checksum"] = np.array(sum(float(v.double().abs().sum()) for v in sd0.values()))
        if tag.startswith("C5"):
            cap = Capture(model)
            torch.manual_seed(5)
            ade, fde, df, td = ref_evaluate(
                model, loader_for(traj), images, torch.device("cpu"), "sdd", None, in_t, list(cfg.waypoints), "test",
                20, 1, cfg.obs_len, B, cfg.resize_factor, cfg.temperature, False, False, 0.01, None,
                return_preds=True, return_samples=True, network=cfg.network)
            cap.close()
            wps = torch.from_numpy(td["waypoint_sample"]).permute(2, 0, 1, 3).contiguous()
            ev = O.eval_batch(sd0, cfg, scene, traj, in_t, waypoint_samples=wps)
            check("eval trajs", ev["trajs"], torch.stack(cap.soft), rtol=1e-5, atol=5e-5)
            store[tag + "/waypoint_samples"] = wps.numpy()
            store[tag + "/trajs"] = torch.stack(cap.soft).numpy()
            store[tag + "/ade_per_traj"], store[tag + "/fde_per_traj"] = df["ade"].to_numpy(), df["fde"].to_numpy()
            store[tag + "/ade"], store[tag + "/fde"] = np.array(ade), np.array(fde)
            continue
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        ade, fde, loss = ref_train_epoch(
            model, loader_for(traj), images, opt, torch.nn.BCEWithLogitsLoss(), cfg.loss_scale, torch.device("cpu"),
            "sdd", None, gt_t, in_t, list(cfg.waypoints), 0, cfg.obs_len, cfg.pred_len, B, 10000,
            cfg.resize_factor, cfg.network, False)
        st = O.train_step(sd0, cfg, scene, traj, in_t, gt_t, names)
        check("loss", st["loss"], loss, rtol=1e-6)
        check("ADE", st["ade"].mean(), ade, rtol=1e-6)
        named = dict(model.named_parameters())
        gn = []
        for n in names:
            check("grad " + n, st["grads"][n], named[n].grad, rtol=1e-4, atol=1e-6)
            gn.append(float(named[n].grad.double().norm()))
        store[tag + "/loss"], store[tag + "/ade"], store[tag + "/fde"] = np.array(loss), np.array(ade), np.array(fde)
        store[tag + "/grad_names"], store[tag + "/grad_norms"] = np.array(names), np.array(gn)
        small = [n for n in names if sd0[n].numel() <= 4096][:12]
        for n in small:
            store[tag + "/grad/" + n] = named[n].grad.numpy()
    np.savez_compressed(os.path.join(OUT, "fullsize_scalars.npz"), **store)
    print("  wrote fullsize_scalars.npz")


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]      # optional substrings of the case tags to (re)generate


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    tiny = dict(enc=(8, 8, 16, 16, 16), dec=(16, 16, 16, 8, 8))
    pos5 = ["0", "1", "2", "3", "4"]
    if not ONLY or "kernels" in ONLY:
        kernel_vectors()
    if not ONLY or "augment" in ONLY:
        augment_case()
    make_case("tiny_short_train", O.sdd_short(train_net="train", **tiny), 32, 64, 2, seed=1)
    make_case("tiny_short_mosa1", O.sdd_short(train_net="mosa_1", position=pos5, **tiny), 32, 64, 2, seed=2)
    make_case("tiny_short_mosa4_partial", O.sdd_short(train_net="mosa_4", position=["0", "2", "4"], **tiny), 64, 32, 3, seed=3, do_eval=False)
    make_case("tiny_long_fusion_mosa3_scene", O.ind_long(network="fusion", n_fusion=2, train_net="mosa_3", position=["scene"], **tiny), 32, 64, 2, seed=4)
    make_case("tiny_long_train", O.sdd_long(train_net="train", **tiny), 64, 64, 2, seed=5, do_epoch=False)
    make_case("tiny_short_encoder_pos", O.sdd_short(train_net="encoder", position=["1", "3"], **tiny), 32, 32, 2, seed=6, do_eval=False)
    make_case("tiny_fusion_scene_only", O.sdd_short(network="fusion", n_fusion=2, train_net="scene", **tiny), 32, 32, 2, seed=7, do_eval=False, do_epoch=False)
    make_case("tiny_short_bias", O.sdd_short(train_net="bias", **tiny), 32, 32, 2, seed=8, do_eval=False, do_epoch=False)
    # adapters (models/ynet.py:15-131, 237-283) and the embedding network (154-167); SURVEY 8(f)-2
    make_case("tiny_short_serial_blocks", O.sdd_short(train_net="serial", position=["1", "3"], **tiny), 32, 32, 3, seed=9, do_epoch=False)
    make_case("tiny_short_parallel3_blocks", O.sdd_short(train_net="parallel_3x3", position=["0", "2"], **tiny), 32, 32, 2, seed=10, do_eval=False)
    make_case("tiny_short_parallel5_blocks", O.sdd_short(train_net="parallel_5x5", position=["1"], **tiny), 32, 32, 2, seed=11, do_eval=False, do_epoch=False)
    make_case("tiny_short_parallelLayer3", O.sdd_short(train_net="parallelLayer_3x3", position=pos5, **tiny), 32, 32, 2, seed=12, do_epoch=False)
    make_case("tiny_short_parallelLayer_multi", O.sdd_short(train_net="parallelLayer_1x1_3x3", position=["1", "2"], **tiny), 32, 32, 2, seed=13, do_eval=False, do_epoch=False)
    make_case("tiny_short_serialLayer", O.sdd_short(train_net="serialLayer", position=["0", "4"], **tiny), 32, 32, 3, seed=14, do_eval=False)
    make_case("tiny_short_embed_train", O.sdd_short(network="embed", train_net="train", **tiny), 32, 32, 2, seed=15, do_epoch=False)
    # TTST (k-means of 10000 goal samples) and CWS (Gaussian prior on intermediate waypoints); SURVEY 8(f)-1
    make_eval = ttst_cws_case
    make_eval("tiny_short_ttst", O.sdd_short(train_net="train", **tiny), 32, 32, 3, seed=21, n_goal=5, n_traj=1,
              use_ttst=True, use_cws=False, cws_params=None)
    make_eval("tiny_long_cws", O.sdd_long(train_net="train", **tiny), 32, 32, 2, seed=22, n_goal=4, n_traj=1,
              use_ttst=False, use_cws=True, cws_params={"sigma_factor": 6.0, "ratio": 2.0, "rot": True})
    make_eval("tiny_long_ttst_cws_ntraj2", O.sdd_long(train_net="train", **tiny), 32, 32, 2, seed=23, n_goal=3, n_traj=2,
              use_ttst=True, use_cws=True, cws_params={"sigma_factor": 6.0, "ratio": 2.0, "rot": False})
    # trained (peaky-map) weights: the reference trains, then one step + one K = 20 sweep
    trained_case("trained_tiny_long", O.sdd_long(train_net="train", **tiny), 64, 64, 4, seed=31, steps=300, lr=2e-3)
    trained_case("trained_short_full", O.sdd_short(train_net="train"), 256, 256, 4, seed=32, steps=200, lr=1e-3,
                 step_cfg=O.sdd_short(train_net="mosa_1", position=pos5))
    # multi-step training trajectories: 4 epochs x 50 steps, the learning rate x 0.1 from epoch 2 on
    trajectory_case("trajectory_tiny_long", O.sdd_long(train_net="train", **tiny), 64, 64, 4, seed=41, epochs=4, steps_per_epoch=50, lr=2e-3,
                    milestones=[2])
    trajectory_case("trajectory_short_mosa1", O.sdd_short(train_net="mosa_1", position=pos5), 256, 256, 4, seed=43, epochs=4, steps_per_epoch=50,
                    lr=1e-3, milestones=[2], lora_b_std=0.05)
    trajectory_case("trajectory_short_full", O.sdd_short(train_net="train"), 256, 256, 4, seed=42, epochs=4, steps_per_epoch=50, lr=1e-3,
                    milestones=[2])
    trajectory_case("trajectory_short_full_continued", O.sdd_short(train_net="train"), 256, 256, 4, seed=44, epochs=4, steps_per_epoch=50, lr=1e-4,
                    milestones=[2], init_from="trained_short_full")
    if not ONLY or "fullsize" in ONLY:
        fullsize_scalars()


if __name__ == "__main__":
    main()
