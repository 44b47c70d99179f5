"""Evaluation sweep (mirror of utils/evaluate.py:37-315): encoder + goal decoder once per batch, sigmoid(x/T) and
multinomial goal/waypoint sampling (optionally TTST: k-means of 10000 goal samples on the device, and CWS: Gaussian
prior on the intermediate waypoints), then K = n_goal*n_traj passes of {gather_patch, waypoint pyramid, trajectory
decoder, soft-argmax}; best-of-K ADE/FDE.  Same signature and return value as the reference;
``forced_samples`` (not in the reference) teacher-forces the sampled way-points for parity tests,
``dp`` shards every batch over ranks, ``max_effective_batch`` bounds the K-folding of the decoder passes.
"""
import contextlib
import os
import weakref

import numpy as np
import pandas as pd
import torch

from .. import ops
from . import step_graph
from .image_utils import SAMPLER, draw_seed, gather_patches, image2world, sampling, swap_pavement_terrain


# YNET_SWEEP_STREAMS=0: the K-sample decoder passes of a batch run back to back on one stream
SWEEP_STREAMS = os.environ.get("YNET_SWEEP_STREAMS", "1") != "0"
_last_sweep_launch = "eager"


def last_sweep_launch() -> str:
    """How the most recent evaluate() call launched its K-sample decoder passes: "eager" or "hipGraph replay" (bench.py)."""
    return _last_sweep_launch


def _decoder_passes(model, features, waypoint_samples, input_template, n_local, n_wp, H, W, max_effective_batch, device):
    """The K = n_goal * n_traj trajectory-decoder passes of utils/evaluate.py:248-266, folded into the batch: G samples at a time run
    as ONE pass over G * n_local virtual batch items whose encoder features repeat along the batch (read in place by the conv
    kernels, never replicated).  -> [K, n_local, pred_len, 2]"""
    K = waypoint_samples.shape[0]
    G = max(1, min(K, max_effective_batch // max(n_local, 1)))
    trajs_samples = []
    # The sample groups are independent of each other: they alternate between two HIP streams, so that the
    # HBM-bound launches of one pass (bilinear x2, patch gather, pyramid, read-out: ~15 % of a pass) and its
    # latency-bound 8^2 .. 32^2 layers run beside the other pass's MFMA-bound convolutions.
    two = SWEEP_STREAMS and K > G and torch.device(device).type == "cuda"
    global _last_sweep_launch
    _last_sweep_launch = "eager, sample groups alternate between two streams" if two else "eager"
    main = torch.cuda.current_stream(device) if two else None
    lanes = ops.side_streams(device) if two else None
    # (the skip-feature part of each decoder level's first conv is the same for all K samples: once per batch)
    with model.traj_decoder.share_skip_features(features):
        if two:
            for st in lanes:
                st.wait_stream(main)          # features, shared terms and way-point samples are ready
        for idx, k0 in enumerate(range(0, K, G)):
            g = min(G, K - k0)
            with (torch.cuda.stream(lanes[idx & 1]) if two else contextlib.nullcontext()):
                coords = waypoint_samples[k0:k0 + g].reshape(-1, 2)            # [g * n_local * n_wp, 2]
                waypoint_map = gather_patches(input_template, coords, H, W).view(g * n_local, n_wp, H, W)
                pyramid = ops.avgpool_pyramid(waypoint_map, len(features))
                traj_input = [ops.lazy_cat([ops.batch_repeat(f, g), p]) for f, p in zip(features, pyramid)]
                pred_traj = model.pred_traj_coords(traj_input)                  # [g * n_local, pred, 2] = softargmax(pred_traj(.))
                trajs_samples.append(pred_traj.view(g, n_local, -1, 2))
        if two:
            for st in lanes:
                main.wait_stream(st)
            for t in trajs_samples:
                t.record_stream(main)
    return torch.cat(trajs_samples)


def _plain_sweep(model, coords, scene_image, input_template, waypoints, n_goal, n_traj, obs_len, temperature, resize_factor, network,
                 max_effective_batch, device, seeds=None):
    """One batch of the sweep as every shipped configuration runs it (no TTST, no CWS, nothing forced, utils/evaluate.py:109-291):
    encoder + goal decoder, sigmoid(x / T), goal and way-point draws, the K decoder passes, best-of-K ADE / FDE.  `coords`
    [n_local, obs + pred, 2]: host tensor (eager: window checks on the host) or device tensor (captured sweep).  `seeds`: None --
    each draw takes its seed from torch's CPU generator, in this order -- or a device int64 tensor with one seed per draw.
    -> (ade [n_local], fde [n_local])"""
    _, _, H, W = scene_image.shape
    n_local, n_wp = coords.shape[0], len(waypoints)
    observed_map = gather_patches(input_template, coords[:, :obs_len].reshape(-1, 2), H, W).view(-1, obs_len, H, W)
    gt_future = coords[:, obs_len:].to(device)
    if network == "embed":      # utils/evaluate.py:119-121
        observed_map = model.motion_embedding(observed_map)
    features = model.pred_features(scene_image.expand(n_local, -1, -1, -1), observed_map)
    pred_goal_map = model.pred_goal(features)
    wp_sigmoid = ops.sigmoid_temp(pred_goal_map, waypoints, temperature)
    goal_samples = sampling(wp_sigmoid[:, -1:], num_samples=n_goal, seed=None if seeds is None else seeds[0:1]).permute(2, 0, 1, 3)
    if n_wp > 1:
        waypoint_samples = sampling(wp_sigmoid[:, :-1], num_samples=n_goal * n_traj,
                                    seed=None if seeds is None else seeds[1:2]).permute(2, 0, 1, 3)
        waypoint_samples = torch.cat([waypoint_samples, goal_samples.repeat(n_traj, 1, 1, 1)], dim=2)
    else:
        waypoint_samples = goal_samples
    trajs_samples = _decoder_passes(model, features, waypoint_samples, input_template, n_local, n_wp, H, W, max_effective_batch, device)
    gt_goal = gt_future[:, -1:]
    ade_batch = ((((gt_future - trajs_samples) / resize_factor) ** 2).sum(dim=3) ** 0.5).mean(dim=2)
    fde_batch = ((((gt_goal - waypoint_samples[:, :, -1:]) / resize_factor) ** 2).sum(dim=3) ** 0.5)
    return ade_batch.min(dim=0)[0], fde_batch.min(dim=0)[0][:, 0]


# ------------------------------------------------------------------------------------------------
# The sweep as a hipGraph: at the reference scripts' batch of 10 a batch of the K = 20 sweep is ~200 short
# launches behind ~1 ms of Python / ctypes per pass.  Like the training step (utils/step_graph.py) a batch of the PLAIN sweep is
# captured on the second sighting of its shape and replayed afterwards: per batch the host copies the coordinates and the seeds
# its draws would have taken (drawn from torch's CPU generator in the same order, so a replayed sweep takes exactly the draws of
# the eager one) into static buffers and issues one graph launch.  YNET_EVAL_GRAPH=0 keeps the sweep eager.
# ------------------------------------------------------------------------------------------------
EVAL_GRAPH = os.environ.get("YNET_EVAL_GRAPH", "1") != "0"
_sweep_graphs = weakref.WeakKeyDictionary()      # model -> {"token": weights token, "pool": graph memory pool, "entries": {key: _CapturedSweep}}


def _weights_token(model):
    """A captured sweep bakes in the addresses of the packed / Winograd-domain filters written before the capture (and of the buffers
    a torch op of it reads in place), so it belongs to ONE state of the weights: per parameter and buffer its identity, its storage
    address and its in-place version -- an optimizer step or load_state_dict bumps the version, `p.data = ...`,
    load_state_dict(assign=True), model.to(...) or a replaced Parameter changes the address / identity (a sum of versions
    misses those, and can collide across parameters)."""
    return tuple((id(t), t.data_ptr(), t._version) for t in list(model.parameters()) + list(model.buffers()))


class _CapturedSweep:
    MAX_ENTRIES = 8

    def __init__(self):
        self.seen = self.ready = self.failed = False

    def capture(self, cache, batch, scene_image, n_seeds, body):
        import gc
        dev = scene_image.device
        gc.collect()
        was = gc.isenabled()
        gc.disable()          # (no hipGraphExecDestroy / hipFree of unrelated objects inside an open capture: utils/step_graph.py)
        try:
            stream = torch.cuda.current_stream(dev)
            if stream == torch.cuda.default_stream(dev):
                raise RuntimeError("capture needs a non-default stream")
            if cache.get("pool") is None:
                cache["pool"] = torch.cuda.graph_pool_handle()
            self.coords = torch.empty(tuple(batch.shape), device=dev, dtype=torch.float32)
            self.coords.copy_(batch)
            self.seeds = torch.zeros(max(n_seeds, 1), device=dev, dtype=torch.int64)
            self.scene = scene_image.detach().clone()
            self.scene_src = None
            pg = torch.distributed.is_available() and torch.distributed.is_initialized()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=cache["pool"], stream=stream, capture_error_mode="thread_local" if pg else "global"):
                self.ade, self.fde = body(self.coords, self.scene, self.seeds)
            self.graph = g
            self.ready = True
        except Exception as e:      # noqa: BLE001 -- any capture failure means: this shape stays eager
            import warnings
            self.failed, self.ready = True, False
            cache["pool"] = None
            warnings.warn(f"hipGraph capture of the evaluation sweep failed ({type(e).__name__}: {e}); running it eagerly")
        finally:
            if was:
                gc.enable()

    def replay(self, batch, scene_image, seeds_host):
        self.coords.copy_(batch)
        if seeds_host is not None:
            self.seeds.copy_(seeds_host)
        if self.scene_src is not scene_image:
            self.scene.copy_(scene_image)
            self.scene_src = scene_image
        self.graph.replay()
        return self.ade.clone(), self.fde.clone()


def _sweep_cache(model):
    c = _sweep_graphs.get(model)
    token = _weights_token(model)
    if c is None or c["token"] != token:
        c = _sweep_graphs[model] = {"token": token, "pool": None if c is None else c.get("pool"), "entries": {}}
    return c


def evaluate(model, val_loader, val_images, device, dataset_name, homo_mat, input_template, waypoints, mode,
             n_goal, n_traj, obs_len, batch_size, resize_factor=0.25, temperature=1, use_TTST=False, use_CWS=False,
             rel_thresh=0.002, CWS_params=None, return_preds=False, return_samples=False, network=None,
             swap_semantic=False, forced_samples=None, dp=None, max_effective_batch=256, forced_goals=None,
             forced_ttst_samples=None):
    """utils/evaluate.py:37-315.  Extra keyword arguments (tests / data-parallel runs): ``forced_samples`` [K,B,nwp,2]
    per batch replaces every random draw, ``forced_goals`` [n_goal,B,1,2] per batch only the goal draw,
    ``forced_ttst_samples`` [10000,B,1,2] per batch the TTST draw; ``dp`` shards each batch over ranks."""
    model.eval()
    waypoints = list(waypoints)
    n_wp = len(waypoints)
    counter = 0
    ade_list, fde_list, meta_id_list, scene_id_list = [], [], [], []
    if return_preds:
        keys = ["groundtruth", "prediction"] + (["waypoint_sample", "goal_map", "goal_sigmoid_map"] if return_samples else [])
        trajs_dict = {k: [] for k in keys}
    else:
        trajs_dict = None

    with torch.no_grad():
        # Every packed filter is (re)written HERE, on the caller's stream: the K-sample passes below run the trajectory decoder on two
        # side streams, and a filter packed lazily by the first pass on one of them (after a training epoch bumped the parameter
        # versions: in place, into the layer's persistent buffers) would be read by the other stream's cache hit with no
        # dependency on the pack kernel.  The side streams wait for this stream before their first launch.
        ops.refresh_filters(model)
        plain = (forced_samples is None and forced_goals is None and not use_TTST and not use_CWS and not return_preds
                 and not return_samples and dataset_name != "eth")
        sa = getattr(model, "softargmax_", None)
        hooked = sa is not None and (sa._forward_hooks or sa._forward_pre_hooks)
        graphs = None
        if plain and EVAL_GRAPH and not hooked and step_graph.enabled(None, device) and SAMPLER == "device":
            graphs = _sweep_cache(model)
        global _last_sweep_launch
        # (captures need a non-default stream: with the graphs on, the whole sweep -- eager first batch, capture, replays -- runs on
        # the persistent side stream of utils/step_graph.py; the caller's stream waits for it at the end)
        sweep_stream = step_graph.enter_stream(device) if graphs is not None else None
        try:
            for trajectory, df_batch, scene_id in val_loader:
                scene_image = model.segmentation(val_images[scene_id].to(device).unsqueeze(0))
                scene_image = model.adapt_semantic(scene_image)
                meta_ids = df_batch[0].metaId.unique()
                n_data = trajectory.shape[0]
                if swap_semantic:
                    scene_image = swap_pavement_terrain(scene_image)
                if network == "embed":          # utils/evaluate.py:98-100
                    scene_image = model.scene_embedding(scene_image)
                if dataset_name == "eth":
                    print(counter)
                    counter += batch_size
                    if counter > 30 and mode == "val":
                        break
                _, _, H, W = scene_image.shape

                for b in range(0, len(trajectory), batch_size):
                    batch = trajectory[b:b + batch_size]
                    n_global = len(batch)
                    lo = 0
                    if dp is not None:
                        lo, hi = dp.shard(n_global)
                        batch = batch[lo:hi]
                    n_local = len(batch)
                    if n_local > 0 and plain:
                        # ---- every shipped configuration: nothing forced, no TTST / CWS, only ADE / FDE wanted
                        def body(coords, scene, seeds):
                            return _plain_sweep(model, coords, scene, input_template, waypoints, n_goal, n_traj, obs_len, temperature,
                                                resize_factor, network, max_effective_batch, device, seeds)
                        entry = None
                        if graphs is not None:
                            key = (tuple(scene_image.shape), n_local, obs_len, tuple(waypoints), n_goal, n_traj, float(temperature),
                                   float(resize_factor), network, input_template.data_ptr(), max_effective_batch, SWEEP_STREAMS)
                            entry = graphs["entries"].get(key)
                            if entry is None:
                                while len(graphs["entries"]) >= _CapturedSweep.MAX_ENTRIES:
                                    graphs["entries"].pop(next(iter(graphs["entries"])))
                                entry = graphs["entries"][key] = _CapturedSweep()
                        n_seeds = 1 + (1 if n_wp > 1 else 0)
                        if entry is not None and not entry.ready and entry.seen and not entry.failed:
                            ops.check_patch_windows(input_template.shape, batch[:, :obs_len], H, W)
                            entry.capture(graphs, batch, scene_image, n_seeds, body)      # (the capture itself launches nothing and draws nothing)
                        if entry is not None and entry.ready:
                            ops.check_patch_windows(input_template.shape, batch[:, :obs_len], H, W)
                            # the seeds the eager draws would take, in their order (torch.manual_seed makes the sweep reproducible either way)
                            seeds_host = torch.tensor([draw_seed() for _ in range(n_seeds)], dtype=torch.int64)
                            ade, fde = entry.replay(batch, scene_image, seeds_host)
                            _last_sweep_launch = "hipGraph replay"
                        else:
                            if entry is not None:
                                entry.seen = True
                            ade, fde = body(batch, scene_image, None)
                    elif n_local > 0:
                        observed_map = gather_patches(input_template, batch[:, :obs_len].reshape(-1, 2), H, W).view(-1, obs_len, H, W)
                        gt_future = batch[:, obs_len:].to(device)
                        if network == "embed":      # utils/evaluate.py:119-121
                            observed_map = model.motion_embedding(observed_map)
                        features = model.pred_features(scene_image.expand(n_local, -1, -1, -1), observed_map)
                        pred_goal_map = model.pred_goal(features)
                        # sigmoid(pred_goal_map[:, waypoints] / T): channel gather + scale + sigmoid in one pass
                        wp_sigmoid = ops.sigmoid_temp(pred_goal_map, waypoints, temperature)

                        if forced_samples is not None:
                            waypoint_samples = forced_samples[b][:, lo:lo + n_local].to(device)
                        else:
                            if forced_goals is not None:
                                goal_samples = forced_goals[b][:, lo:lo + n_local].to(device)
                            elif use_TTST:
                                draw = None if forced_ttst_samples is None else forced_ttst_samples[b][:, lo:lo + n_local].to(device)
                                goal_samples = ttst_goals(model, wp_sigmoid[:, -1:], pred_goal_map[:, waypoints[-1:]], n_goal,
                                                          rel_thresh, draw)
                            else:
                                goal_samples = sampling(wp_sigmoid[:, -1:], num_samples=n_goal).permute(2, 0, 1, 3)
                            if use_CWS and n_wp > 1:
                                last_observed = batch[:, obs_len - 1].to(device)
                                waypoint_samples = cws_waypoints(model, wp_sigmoid, goal_samples, last_observed, n_goal, n_traj,
                                                                 CWS_params["sigma_factor"], CWS_params["ratio"], CWS_params["rot"])
                            elif n_wp > 1:
                                waypoint_samples = sampling(wp_sigmoid[:, :-1], num_samples=n_goal * n_traj).permute(2, 0, 1, 3)
                                waypoint_samples = torch.cat([waypoint_samples, goal_samples.repeat(n_traj, 1, 1, 1)], dim=2)
                            else:
                                waypoint_samples = goal_samples

                        if return_samples:
                            trajs_dict["goal_map"].append(pred_goal_map.cpu().numpy())
                            trajs_dict["goal_sigmoid_map"].append(model.sigmoid(pred_goal_map / temperature).cpu().numpy())
                            trajs_dict["waypoint_sample"].append(waypoint_samples.permute(1, 2, 0, 3).cpu().numpy())

                        trajs_samples = _decoder_passes(model, features, waypoint_samples, input_template, n_local, n_wp, H, W,
                                                        max_effective_batch, device)
                        gt_goal = gt_future[:, -1:]
                        if dataset_name == "eth":
                            waypoint_samples = image2world(waypoint_samples, scene_id, homo_mat, resize_factor)
                            gt_future = image2world(gt_future, scene_id, homo_mat, resize_factor)
                        ade_batch = ((((gt_future - trajs_samples) / resize_factor) ** 2).sum(dim=3) ** 0.5).mean(dim=2)
                        fde_batch = ((((gt_goal - waypoint_samples[:, :, -1:]) / resize_factor) ** 2).sum(dim=3) ** 0.5)
                        if return_preds:
                            if b == 0:
                                trajs_dict["groundtruth"].append(trajectory.cpu().numpy() / resize_factor)
                            best = ade_batch.argmin(dim=0)
                            trajs_dict["prediction"].append(
                                (trajs_samples[best, torch.arange(trajs_samples.shape[1], device=device)] / resize_factor).cpu().numpy())
                        ade = ade_batch.min(dim=0)[0]
                        fde = fde_batch.min(dim=0)[0][:, 0]
                    else:
                        ade = fde = torch.zeros(0, device=device)
                    if dp is not None:
                        sizes = dp.shard_sizes(n_global)
                        ade, fde = dp.gather_rows(ade, sizes), dp.gather_rows(fde, sizes)
                    ade_list.append(ade.cpu().numpy())
                    fde_list.append(fde.cpu().numpy())
                    ops.check_patch_status()      # (the batch is synchronised by the copies above: a window that left the template raises here)
                meta_id_list.append(meta_ids)
                scene_id_list.append([scene_id] * n_data)
        finally:
            step_graph.leave_stream(sweep_stream)

    ops.check_patch_status()
    val_ade_arr, val_fde_arr = np.concatenate(ade_list), np.concatenate(fde_list)
    df_out = pd.DataFrame({"metaId": np.concatenate(meta_id_list), "sceneId": sum(scene_id_list, []),
                           "ade": val_ade_arr, "fde": val_fde_arr})
    if return_preds:
        for key, value in trajs_dict.items():
            trajs_dict[key] = np.concatenate(value, axis=0)
        trajs_dict["metaId"] = df_out["metaId"].to_numpy()
        trajs_dict["sceneId"] = list(df_out["sceneId"])
    return val_ade_arr.mean(), val_fde_arr.mean(), df_out, trajs_dict


def _kmeans_host_path(X, k, init_idx, tol=1e-3, iter_limit=1000):
    """The reference's Lloyd iteration with device tensors (utils/kmeans.py:22-108), used only for a point set in
    which a cluster went empty (it then needs the reference's torch.randint re-seed)."""
    X = X.float()
    c = X[init_idx.to(X.device).long()].clone()
    it = 0
    while True:
        d = ((X.unsqueeze(1) - c.unsqueeze(0)) ** 2.0).sum(dim=-1)
        assign = torch.argmin(d, dim=1)
        prev = c.clone()
        for j in range(k):
            sel = X[assign == j]
            if sel.shape[0] == 0:
                sel = X[torch.randint(len(X), (1,))]
            c[j] = sel.mean(dim=0)
        shift = torch.sum(torch.sqrt(torch.sum((c - prev) ** 2, dim=1)))
        it += 1
        if float(shift) ** 2 < tol or (iter_limit != 0 and it >= iter_limit):
            return c


def ttst_goals(model, wp_sigmoid_last, wp_logits_last, n_goal, rel_thresh, draw=None):
    """Test-time sampling trick (utils/evaluate.py:134-161): 10000 thresholded goal samples per person (with
    replacement) clustered into n_goal - 1 centres by ``ynet_kmeans2d`` (one workgroup per person; the reference
    loops over persons in Python); the first goal is the soft-argmax of the logits.  The initial centres are drawn
    with np.random.choice per person, in order, like the reference.  -> [n_goal, B, 1, 2]."""
    if draw is None:
        draw = sampling(wp_sigmoid_last, num_samples=10000, replacement=True, rel_threshold=rel_thresh).permute(2, 0, 1, 3)
    first = model.softargmax(wp_logits_last)                     # [B,1,2]
    n_people, k = draw.shape[1], n_goal - 1
    if k == 0:          # n_goal == 1: the soft-argmax goal alone (the reference's cluster loop has nothing to add)
        return first.unsqueeze(0)
    points = draw[:, :, 0].permute(1, 0, 2).contiguous()         # [B, 10000, 2]
    init = torch.from_numpy(np.stack([np.random.choice(points.shape[1], k, replace=False) for _ in range(n_people)]).astype(np.int32))
    centers, status = ops.kmeans2d(points, init, tol=0.001, iter_limit=1000)
    for person in torch.nonzero(status.cpu() & 1).flatten().tolist():
        centers[person] = _kmeans_host_path(points[person], k, init[person])
    goals = centers.permute(1, 0, 2).unsqueeze(2)                # [k, B, 1, 2]
    return torch.cat([first.unsqueeze(0), goals], dim=0)


def cws_gaussians(mean_xy, H, W, dist, sigma_factor, ratio, rot):
    """torch_multivariate_gaussian_heatmap (utils/evaluate.py:9-34) for all N persons at once: anisotropic Gaussians
    centred at mean_xy [N,2], long axis along dist [N,2], std (|dist| + 5) / sigma_factor along it and that / ratio
    across, on linspace(0, H, H) x linspace(0, W, W), each normalised to sum 1.  -> [N,H,W]  (ynet_cws_prior with an
    all-ones sigmoid map)."""
    ones = torch.ones((1, H, W), device=mean_xy.device)
    return ops.cws_prior(ones, mean_xy, dist, sigma_factor, ratio, rot, want_map=True, want_xy=False)[0]


def cws_waypoints(model, wp_sigmoid, goal_samples, last_observed, n_goal, n_traj, sigma_factor, ratio, rot):
    """Conditioned waypoint sampling (utils/evaluate.py:172-224): waypoints are drawn backwards from the goal; the
    sigmoid map of waypoint w is multiplied by a Gaussian centred at goal + (last_observed - goal) / (w + 2) and
    renormalised; the first n_goal trajectories take its expectation, later ones one thresholded sample.
    The reference loops over goal samples and persons in Python; here the n_goal expectation chains run as ONE
    ynet_cws_prior launch per waypoint (rows = n_goal * B, fp64 per pixel), the sampled chains one launch per goal
    sample (their draws stay in the reference's order).  -> [n_goal * n_traj, B, n_waypoints, 2]."""
    B, nwp, H, W = wp_sigmoid.shape
    goals = goal_samples.repeat(n_traj, 1, 1, 1).squeeze(2)           # [K, B, 2]
    K = goals.shape[0]
    chains = [None] * K
    # ---- trajectories of the first set (g_num < n_goal): expectations, batched over the goal samples
    n_det = min(n_goal, K)
    wp = goals[:n_det].reshape(-1, 2)                                  # row r = g * B + person
    last = last_observed.repeat(n_det, 1)
    det = [wp]
    for w in reversed(range(nwp - 1)):
        distance = last - wp
        _, wp = ops.cws_prior(wp_sigmoid[:, w], wp + distance * (1 / (w + 2)), distance, sigma_factor, ratio, rot)
        det.append(wp)
    det = torch.stack(det[::-1]).view(nwp, n_det, B, 2).permute(1, 2, 0, 3)      # [n_det, B, nwp, 2]
    for g in range(n_det):
        chains[g] = det[g]
    # ---- later sets: one thresholded sample per waypoint, in the reference's order
    for g_num in range(n_det, K):
        wp = goals[g_num]
        traj_idx = g_num // n_goal
        chain = [wp]
        for w in reversed(range(nwp - 1)):
            distance = last_observed - wp
            m, _ = ops.cws_prior(wp_sigmoid[:, w], wp + distance * (1 / (w + 2)), distance, sigma_factor - traj_idx, ratio, rot,
                                 want_map=True, want_xy=False)
            wp = sampling(m.unsqueeze(1), num_samples=1, rel_threshold=0.05).permute(2, 0, 1, 3).squeeze(2).squeeze(0)
            chain.append(wp)
        chains[g_num] = torch.stack(chain[::-1]).permute(1, 0, 2)
    return torch.stack(chains)
