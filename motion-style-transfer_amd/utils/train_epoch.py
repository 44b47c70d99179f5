"""One training epoch (mirror of utils/train_epoch.py:8-136, same signature and return value).

Per batch: heat-maps by one gather launch each (instead of a Python list of slices + torch.stack),
encoder -> goal decoder -> BCE, waypoint pyramid in one pass, trajectory decoder (concat fused into
the convs) -> BCE, backward through the hand-written dgrad/wgrad/LoRA kernels, optimizer step,
soft-argmax ADE/FDE.  ``dp`` (optional, not in the reference) shards every batch over ranks.

The step is launch-bound on the host at the reference's own batch sizes (its scripts train with batch_size 10: ~250
kernel launches of a few microseconds each behind ~6 ms of Python / ctypes / autograd bookkeeping), so by default the
whole step -- gather, forward, backward, optimizer, read-out -- is captured ONCE per (scene size, shard size, learning
rate) into a hipGraph and replayed: per step the host copies the [B, T, 2] coordinates into a static buffer and issues
one graph launch (``utils/step_graph.py``).  The first step of a shape runs eagerly (it is also the warm-up the capture
needs), the second is captured, later ones replay.  ``graph=False`` / ``YNET_STEP_GRAPH=0`` keeps everything eager; the
arithmetic is the same kernels in the same order either way.
"""
import os

import numpy as np
import torch

from .. import ops
from . import step_graph
from ..models.ynet import announce_bce_target
from .image_utils import gather_patches, swap_pavement_terrain


# YNET_TRAIN_READOUT=0: the step's read-out runs module by module (two soft-argmax launches + the reference's elementwise chain)
FUSED_READOUT = os.environ.get("YNET_TRAIN_READOUT", "1") != "0"


def _step_forward_backward(model, criterion, coords, scene_image, gt_template, input_template, waypoints, obs_len,
                           pred_len, loss_scale, network, swap_semantic, device, n_local, n_global, dp, overlap,
                           resize_factor=1.0, branches=False):
    """utils/train_epoch.py:54-110 for one (shard of a) batch: heat-maps, forward, both losses, backward.
    ``coords`` [n_local, obs+pred, 2]: a host tensor (eager; window checks on the host) or a device tensor (captured
    step).  Returns (loss as the reference sums it -- shard-weighted under dp --, pred_goal_map, pred_traj_map, gt_future,
    (ade, fde) or None).  ``branches`` (captured steps): besides the two decoders, the read-out runs on a forked stream,
    i.e. as a parallel branch of the hipGraph beside the backward pass."""
    semantic_img = model.adapt_semantic(scene_image)
    if swap_semantic:
        semantic_img = swap_pavement_terrain(semantic_img)
    _, _, H, W = scene_image.shape
    with ops.fold_skip_gradients():      # skip-connection gradients summed inside the max-pool backward
        # heat-maps: distance map per observed step, Gaussian blob per future step, distance map per waypoint
        def target_maps():
            gt_map = gather_patches(gt_template, coords[:, obs_len:].reshape(-1, 2), H, W).view(-1, pred_len, H, W)
            if coords.is_cuda:      # (a list index would upload an index tensor: not allowed inside a capture)
                wps = coords[:, obs_len:].index_select(1, step_graph.waypoint_index(device, waypoints))
            else:
                wps = coords[:, obs_len:][:, waypoints]
            return gt_map, gather_patches(input_template, wps.reshape(-1, 2), H, W).view(-1, len(waypoints), H, W)

        side = None
        if overlap and branches:
            # Off the encoder's critical path, on a forked stream (a branch of the captured step): the target / way-point
            # maps, needed by the decoders and the losses only.
            main0 = torch.cuda.current_stream(device)
            side = ops.side_streams(device, 2)
            side.wait_stream(main0)
            with torch.cuda.stream(side):
                gt_future_map, gt_waypoint_map = target_maps()
            for t in (gt_future_map, gt_waypoint_map):
                t.record_stream(main0)
        ops.refresh_filters(model)      # every changed filter (W + BA*s of the adapted convs, W of the trainable plain ones) packed in one launch
        observed_map = gather_patches(input_template, coords[:, :obs_len].reshape(-1, 2), H, W).view(-1, obs_len, H, W)
        gt_future = coords[:, obs_len:].to(device)
        if side is None:
            gt_future_map, gt_waypoint_map = target_maps()
        join_targets = side is not None
        sem1 = semantic_img
        if network == "embed":      # utils/train_epoch.py:80-83 (before the expand)
            sem1 = model.scene_embedding(semantic_img)
            observed_map = model.motion_embedding(observed_map)

        semantic_map = sem1.expand(n_local, -1, -1, -1)
        features = model.pred_features(semantic_map, observed_map)
        if join_targets:
            torch.cuda.current_stream(device).wait_stream(side)      # the decoders and the losses need the target maps
        if hasattr(criterion, "expected_grad"):
            # d(loss)/d(criterion output) as autograd will compute it (fp32): lets the criterion emit the
            # gradient of its logits in the same pass as the loss
            up = np.float32(1.0) if dp is None else np.float32(n_local / n_global)
            criterion.expected_grad = float(np.float32(up * np.float32(loss_scale)))
        if overlap:
            # The goal and the trajectory decoder are independent given the features: run them on two
            # HIP streams so the launch-latency-bound small maps (8^2 .. 32^2) of one overlap the other
            # (autograd replays each backward op on the stream of its forward op).
            main = torch.cuda.current_stream(device)
            s_goal, s_traj = ops.side_streams(device)
            for f in features:
                for t in ops._parts(f):
                    t.record_stream(s_goal)
                    t.record_stream(s_traj)
            gt_future_map.record_stream(s_goal)
            gt_future_map.record_stream(s_traj)
            gt_waypoint_map.record_stream(s_traj)
            s_goal.wait_stream(main)
            s_traj.wait_stream(main)
            with torch.cuda.stream(s_goal):
                with announce_bce_target(model.goal_decoder, criterion, gt_future_map):
                    pred_goal_map = model.pred_goal(features)
                goal_loss = criterion(pred_goal_map, gt_future_map) * loss_scale
            with torch.cuda.stream(s_traj):
                pyramid = ops.avgpool_pyramid(gt_waypoint_map, len(features))
                traj_input = [ops.lazy_cat([f, g]) for f, g in zip(features, pyramid)]   # concat fused into the convs
                with announce_bce_target(model.traj_decoder, criterion, gt_future_map):
                    pred_traj_map = model.pred_traj(traj_input)
                traj_loss = criterion(pred_traj_map, gt_future_map) * loss_scale
            main.wait_stream(s_goal)
            main.wait_stream(s_traj)
            for t in (pred_goal_map, goal_loss, pred_traj_map, traj_loss):
                t.record_stream(main)
        else:
            # (the decoders are told their BCE target: predictor, loss and the predictor's dgrad run as one kernel)
            with announce_bce_target(model.goal_decoder, criterion, gt_future_map):
                pred_goal_map = model.pred_goal(features)
            goal_loss = criterion(pred_goal_map, gt_future_map) * loss_scale
            pyramid = ops.avgpool_pyramid(gt_waypoint_map, len(features))
            traj_input = [ops.lazy_cat([f, g]) for f, g in zip(features, pyramid)]
            with announce_bce_target(model.traj_decoder, criterion, gt_future_map):
                pred_traj_map = model.pred_traj(traj_input)
            traj_loss = criterion(pred_traj_map, gt_future_map) * loss_scale

        loss = goal_loss + traj_loss
        if dp is not None:
            loss = loss * (n_local / n_global)      # BCE is a mean: weight by the shard's share
        early = None
        if overlap and branches:
            # the read-out depends on the forward pass only: a third branch beside the backward pass
            s_m = ops.side_streams(device, 2)
            s_m.wait_stream(torch.cuda.current_stream(device))
            for t in (pred_goal_map, pred_traj_map, gt_future):
                t.record_stream(s_m)
            with torch.cuda.stream(s_m), torch.no_grad():
                early = _step_metrics(model, pred_goal_map, pred_traj_map, gt_future, resize_factor)
        # (adapter gradients of the mosa_* modes run on a fourth branch beside the encoder's dgrad chain: ops.wgrad_branch.  FULL filter
        # gradients there were measured twice: every one of them in round 2 -- 5 % slower at C1 -- and those of the <= 32^2 / 64^2 / 128^2
        # maps only in round 3 -- 14.09 -> 14.34 / 14.29 / 14.43 ms: one more branch couples the two decoders' backward streams)
        loss.backward()
        # the adapter-gradient branch joins HERE (not only at the context's exit): anything that reads lora_A.grad / lora_B.grad
        # after backward() -- dp.stage, gradient clipping, the optimizer -- is then ordered behind it
        ops.join_wgrad_branch()
        if early is not None:
            cur = torch.cuda.current_stream(device)
            cur.wait_stream(s_m)
            for t in early:
                t.record_stream(cur)
    return loss, pred_goal_map, pred_traj_map, gt_future, early


def _step_metrics(model, pred_goal_map, pred_traj_map, gt_future, resize_factor):
    """utils/train_epoch.py:118-126: soft-argmax read-out, per-trajectory ADE / FDE."""
    sa = getattr(model, "softargmax_", None)
    if (FUSED_READOUT and sa is not None and not sa.normalized_coordinates and not sa._forward_hooks and not sa._forward_pre_hooks
            and pred_traj_map.is_cuda and pred_goal_map.shape[0] == pred_traj_map.shape[0] and (pred_traj_map.shape[3] % 4 == 0)):
        # both soft-argmax calls in one launch, the ADE / FDE arithmetic in a second (ynet_train_readout); a forward hook on
        # the model's SoftArgmax2D (tests, visualisation) keeps the module-by-module path below
        _, _, ade, fde = ops.train_readout(pred_traj_map, pred_goal_map, gt_future, resize_factor)
        return ade, fde
    pred_traj = model.softargmax(pred_traj_map)
    pred_goal = model.softargmax(pred_goal_map[:, -1:])
    ade = ((((gt_future - pred_traj) / resize_factor) ** 2).sum(dim=2) ** 0.5).mean(dim=1)
    fde = ((((gt_future[:, -1:] - pred_goal[:, -1:]) / resize_factor) ** 2).sum(dim=2) ** 0.5).mean(dim=1)
    return ade, fde


def train_epoch(model, train_loader, train_images, optimizer, criterion, loss_scale, device, dataset_name, homo_mat,
                gt_template, input_template, waypoints, epoch, obs_len, pred_len, batch_size, e_unfreeze,
                resize_factor, network=None, swap_semantic=False, dp=None, graph=None):
    device = torch.device(device)
    train_loss = 0
    train_ADE, train_FDE = [], []
    model.train()
    waypoints = list(waypoints)
    graphs = step_graph.cache_for(model, optimizer, device) if step_graph.enabled(graph, device) else None
    model_token = step_graph.model_state_token(model) if graphs is not None else None
    # Captured steps need a non-default stream, and autograd's gradient accumulators remember the stream of their first
    # backward pass: every epoch -- captured or eager -- runs on the same persistent side stream (the caller's stream
    # waits for it at the end), so that eager and captured epochs of one model can alternate freely.
    epoch_stream = step_graph.enter_stream(device) if device.type == "cuda" and torch.cuda.is_available() else None

    try:
        for trajectory, meta, scene in train_loader:
            if epoch < e_unfreeze:
                model.eval()
                scene_image = model.segmentation(train_images[scene].to(device).unsqueeze(0))
                model.train()

            for i in range(0, len(trajectory), batch_size):
                if epoch >= e_unfreeze:
                    scene_image = model.segmentation(train_images[scene].to(device).unsqueeze(0))

                batch = trajectory[i:i + batch_size]
                n_global = len(batch)
                if dp is not None:
                    lo, hi = dp.shard(n_global)
                    batch = batch[lo:hi]
                n_local = len(batch)

                step = None
                if graphs is not None and n_local > 0 and not swap_semantic and not scene_image.requires_grad:
                    step = graphs.lookup(step_graph.step_key(
                        scene_image, n_local, n_global, obs_len, pred_len, waypoints, loss_scale, resize_factor, network,
                        criterion, gt_template, input_template, optimizer, dp, model_token))

                if step is not None and step.ready:
                    # ---- replay: coordinates (and the scene's semantic map) into the static inputs, one graph launch
                    _, _, H, W = scene_image.shape
                    ops.check_patch_windows(input_template.shape, batch, H, W)
                    loss, ade, fde = step.replay(batch, scene_image)
                else:
                    def forward_backward(coords, scene_img, overlap):
                        if dp is not None:
                            dp.zero_grad()
                        else:
                            optimizer.zero_grad()
                        return _step_forward_backward(model, criterion, coords, scene_img, gt_template, input_template,
                                                      waypoints, obs_len, pred_len, loss_scale, network, swap_semantic,
                                                      device, n_local, n_global, dp, overlap, resize_factor,
                                                      branches=graphs is not None and step_graph.OVERLAP_DECODERS)

                    def finish(fb):
                        if fb[4] is not None:
                            return fb[4]
                        with torch.no_grad():
                            return _step_metrics(model, fb[1], fb[2], fb[3], resize_factor)

                    if step is not None and step.seen and not step.failed:
                        # ---- second sighting of this shape: capture, then run the captured step
                        _, _, H, W = scene_image.shape
                        ops.check_patch_windows(input_template.shape, batch, H, W)
                        step.capture(batch, scene_image, forward_backward, optimizer, dp, finish)
                    if step is not None and step.ready:
                        loss, ade, fde = step.replay(batch, scene_image)
                    else:
                        # ---- eager step (also the warm-up of a later capture)
                        if step is not None:
                            step.seen = True
                        if n_local > 0:
                            if graphs is not None:
                                # same code path as the capture: coordinates on the device, windows checked on the host
                                _, _, H, W = scene_image.shape
                                ops.check_patch_windows(input_template.shape, batch, H, W)
                                fb = forward_backward(batch.to(device), scene_image, step_graph.OVERLAP_DECODERS)
                            else:
                                fb = forward_backward(batch, scene_image, ops.overlap_decoders)
                            loss = fb[0]
                        else:
                            if dp is not None:
                                dp.zero_grad()
                            else:
                                optimizer.zero_grad()
                            loss = torch.zeros((), device=device)
                        if dp is not None:
                            loss = dp.allreduce_grads(loss)          # ONE collective per step: gradients + loss
                        optimizer.step()
                        # Always: a fused multi-tensor Adam (the form a capture switches the caller's optimizer to, or a
                        # user's Adam(fused=True)) updates the weights without bumping Parameter._version, and the caches
                        # of packed / LoRA-composed filters are keyed on it -- also when this epoch runs with graph=False.
                        step_graph.mark_parameters_changed(p for g in optimizer.param_groups for p in g["params"])
                        if n_local > 0:
                            ade, fde = finish(fb)
                        else:
                            ade = fde = torch.zeros(0, device=device)

                with torch.no_grad():
                    train_loss += loss
                    train_ADE.append(ade)
                    train_FDE.append(fde)

        train_ADE, train_FDE = torch.cat(train_ADE), torch.cat(train_FDE)
        if dp is not None and dp.active:
            # per-trajectory errors stay local during the epoch; one (sum, sum, count) reduction at its end
            stats = torch.stack([train_ADE.sum(), train_FDE.sum(),
                                 torch.tensor(float(train_ADE.numel()), device=train_ADE.device)])
            stats = dp.sum_scalar(stats)
            train_ADE, train_FDE = stats[0] / stats[2], stats[1] / stats[2]
        else:
            train_ADE, train_FDE = train_ADE.mean(), train_FDE.mean()
    finally:
        if epoch_stream is not None:
            step_graph.leave_stream(epoch_stream)
    ops.check_patch_status()
    return train_ADE.item(), train_FDE.item(), train_loss.item()
