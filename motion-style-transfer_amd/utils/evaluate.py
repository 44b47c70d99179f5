"""Evaluation sweep (mirror of utils/evaluate.py:37-315 without the TTST / CWS branches, which are
off in every shipped config): encoder + goal decoder once per batch, sigmoid(x/T) and multinomial
goal/waypoint sampling, then K = n_goal*n_traj passes of {gather_patch, waypoint pyramid, trajectory
decoder, soft-argmax}; best-of-K ADE/FDE.  Same signature and return value as the reference;
``forced_samples`` (not in the reference) teacher-forces the sampled way-points for parity tests,
``dp`` shards every batch over ranks, ``max_effective_batch`` bounds the K-folding of the decoder passes.
"""
import numpy as np
import pandas as pd
import torch

from .. import ops
from .image_utils import gather_patches, image2world, sampling, swap_pavement_terrain


def evaluate(model, val_loader, val_images, device, dataset_name, homo_mat, input_template, waypoints, mode,
             n_goal, n_traj, obs_len, batch_size, resize_factor=0.25, temperature=1, use_TTST=False, use_CWS=False,
             rel_thresh=0.002, CWS_params=None, return_preds=False, return_samples=False, network=None,
             swap_semantic=False, forced_samples=None, dp=None, max_effective_batch=256):
    if use_TTST or use_CWS:
        raise NotImplementedError("TTST / CWS are outside the MI355X hot path (disabled in every shipped config)")
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
                if n_local > 0:
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
                        goal_samples = sampling(wp_sigmoid[:, -1:], num_samples=n_goal).permute(2, 0, 1, 3)
                        if n_wp > 1:
                            waypoint_samples = sampling(wp_sigmoid[:, :-1], num_samples=n_goal * n_traj).permute(2, 0, 1, 3)
                            waypoint_samples = torch.cat([waypoint_samples, goal_samples.repeat(n_traj, 1, 1, 1)], dim=2)
                        else:
                            waypoint_samples = goal_samples

                    if return_samples:
                        trajs_dict["goal_map"].append(pred_goal_map.cpu().numpy())
                        trajs_dict["goal_sigmoid_map"].append(model.sigmoid(pred_goal_map / temperature).cpu().numpy())
                        trajs_dict["waypoint_sample"].append(waypoint_samples.permute(1, 2, 0, 3).cpu().numpy())

                    # The K = n_goal * n_traj decoder passes of the reference loop are folded into the batch:
                    # G samples at a time run as ONE pass over G * n_local virtual batch items whose encoder
                    # features repeat along the batch (read in place by the conv kernels, never replicated).
                    K = waypoint_samples.shape[0]
                    G = max(1, min(K, max_effective_batch // max(n_local, 1)))
                    trajs_samples = []
                    for k0 in range(0, K, G):
                        g = min(G, K - k0)
                        coords = waypoint_samples[k0:k0 + g].reshape(-1, 2)            # [g * n_local * n_wp, 2]
                        waypoint_map = gather_patches(input_template, coords, H, W).view(g * n_local, n_wp, H, W)
                        pyramid = ops.avgpool_pyramid(waypoint_map, len(features))
                        traj_input = [ops.lazy_cat([ops.batch_repeat(f, g), p]) for f, p in zip(features, pyramid)]
                        pred_traj = model.softargmax(model.pred_traj(traj_input))       # [g * n_local, pred, 2]
                        trajs_samples.append(pred_traj.view(g, n_local, -1, 2))
                    trajs_samples = torch.cat(trajs_samples)
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
            meta_id_list.append(meta_ids)
            scene_id_list.append([scene_id] * n_data)

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
