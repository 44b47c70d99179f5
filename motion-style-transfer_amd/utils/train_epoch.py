"""One training epoch (mirror of utils/train_epoch.py:8-136, same signature and return value).

Per batch: heat-maps by one gather launch each (instead of a Python list of slices + torch.stack),
encoder -> goal decoder -> BCE, waypoint pyramid in one pass, trajectory decoder (concat fused into
the convs) -> BCE, backward through the hand-written dgrad/wgrad/LoRA kernels, optimizer step,
soft-argmax ADE/FDE.  ``dp`` (optional, not in the reference) shards every batch over ranks.
"""
import numpy as np
import torch

from .. import ops
from .image_utils import gather_patches, swap_pavement_terrain


def train_epoch(model, train_loader, train_images, optimizer, criterion, loss_scale, device, dataset_name, homo_mat,
                gt_template, input_template, waypoints, epoch, obs_len, pred_len, batch_size, e_unfreeze,
                resize_factor, network=None, swap_semantic=False, dp=None):
    train_loss = 0
    train_ADE, train_FDE = [], []
    model.train()
    waypoints = list(waypoints)

    for trajectory, meta, scene in train_loader:
        if epoch < e_unfreeze:
            model.eval()
            scene_image = model.segmentation(train_images[scene].to(device).unsqueeze(0))
            model.train()

        for i in range(0, len(trajectory), batch_size):
            if epoch >= e_unfreeze:
                scene_image = model.segmentation(train_images[scene].to(device).unsqueeze(0))
            semantic_img = model.adapt_semantic(scene_image)
            if swap_semantic:
                semantic_img = swap_pavement_terrain(semantic_img)
            _, _, H, W = scene_image.shape

            batch = trajectory[i:i + batch_size]
            n_global = len(batch)
            if dp is not None:
                lo, hi = dp.shard(n_global)
                batch = batch[lo:hi]
            n_local = len(batch)

            if dp is not None:
                dp.zero_grad()
            else:
                optimizer.zero_grad()

            if n_local > 0:
                with ops.fold_skip_gradients():      # skip-connection gradients summed inside the max-pool backward
                    # heat-maps: distance map per observed step, Gaussian blob per future step, distance map per waypoint
                    observed_map = gather_patches(input_template, batch[:, :obs_len].reshape(-1, 2), H, W).view(-1, obs_len, H, W)
                    gt_future = batch[:, obs_len:].to(device)
                    gt_future_map = gather_patches(gt_template, batch[:, obs_len:].reshape(-1, 2), H, W).view(-1, pred_len, H, W)
                    gt_waypoints = batch[:, obs_len:][:, waypoints]
                    gt_waypoint_map = gather_patches(input_template, gt_waypoints.reshape(-1, 2), H, W).view(-1, len(waypoints), H, W)
                    sem1 = semantic_img
                    if network == "embed":      # utils/train_epoch.py:80-83 (before the expand)
                        sem1 = model.scene_embedding(semantic_img)
                        observed_map = model.motion_embedding(observed_map)

                    semantic_map = sem1.expand(n_local, -1, -1, -1)
                    features = model.pred_features(semantic_map, observed_map)
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
                    if hasattr(criterion, "expected_grad"):
                        # d(loss)/d(criterion output) as autograd will compute it (fp32): lets the criterion emit the
                        # gradient of its logits in the same pass as the loss
                        up = np.float32(1.0) if dp is None else np.float32(n_local / n_global)
                        criterion.expected_grad = float(np.float32(up * np.float32(loss_scale)))
                    with torch.cuda.stream(s_goal):
                        pred_goal_map = model.pred_goal(features)
                        goal_loss = criterion(pred_goal_map, gt_future_map) * loss_scale
                    with torch.cuda.stream(s_traj):
                        pyramid = ops.avgpool_pyramid(gt_waypoint_map, len(features))
                        traj_input = [ops.lazy_cat([f, g]) for f, g in zip(features, pyramid)]   # concat fused into the convs
                        pred_traj_map = model.pred_traj(traj_input)
                        traj_loss = criterion(pred_traj_map, gt_future_map) * loss_scale
                    main.wait_stream(s_goal)
                    main.wait_stream(s_traj)
                    for t in (pred_goal_map, goal_loss, pred_traj_map, traj_loss):
                        t.record_stream(main)

                    loss = goal_loss + traj_loss
                    if dp is not None:
                        loss = loss * (n_local / n_global)      # BCE is a mean: weight by the shard's share
                    loss.backward()
            else:
                loss = torch.zeros((), device=device)
            if dp is not None:
                loss = dp.allreduce_grads(loss)          # ONE collective per step: gradients + loss
            optimizer.step()

            with torch.no_grad():
                train_loss += loss
                if n_local > 0:
                    pred_traj = model.softargmax(pred_traj_map)
                    pred_goal = model.softargmax(pred_goal_map[:, -1:])
                    ade = ((((gt_future - pred_traj) / resize_factor) ** 2).sum(dim=2) ** 0.5).mean(dim=1)
                    fde = ((((gt_future[:, -1:] - pred_goal[:, -1:]) / resize_factor) ** 2).sum(dim=2) ** 0.5).mean(dim=1)
                else:
                    ade = fde = torch.zeros(0, device=device)
                train_ADE.append(ade)
                train_FDE.append(fde)

    train_ADE, train_FDE = torch.cat(train_ADE), torch.cat(train_FDE)
    if dp is not None and dp.world > 1:
        # per-trajectory errors stay local during the epoch; one (sum, sum, count) reduction at its end
        stats = torch.stack([train_ADE.sum(), train_FDE.sum(),
                             torch.tensor(float(train_ADE.numel()), device=train_ADE.device)])
        stats = dp.sum_scalar(stats)
        train_ADE, train_FDE = stats[0] / stats[2], stats[1] / stats[2]
    else:
        train_ADE, train_FDE = train_ADE.mean(), train_FDE.mean()
    ops.check_patch_status()
    return train_ADE.item(), train_FDE.item(), train_loss.item()
