"""YNetTrainer on MI355X (mirror of models/trainer.py: same public methods, freeze policy,
optimizer, log lines and checkpoint format).

Reference: __init__ 46-75, _train 80-293 (freeze table 116-195), _test 299-352, prepare_data 518-584,
load_params / save_params / load_separated_params 586-614, mark_*_bias_trainable 20-42.
Differences, all outside the arithmetic: image files are not decoded here (cv2/smp are out of scope) —
``prepare_data`` takes a dict {scene_id: float tensor [C,H,W]} in place of an image directory; an
optional ``dp`` (dist.DataParallel) shards batches over the GPUs of a node.
"""
import os
import pathlib
import re
from collections import OrderedDict, deque
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader
from tqdm import tqdm

from .. import ops
from ..utils.dataloader import SceneDataset, scene_collate
from ..utils.evaluate import evaluate
from ..utils.image_utils import (analytic_dist_template, analytic_gaussian_template, create_dist_mat,
                                 create_gaussian_heatmap_template)
from ..utils.train_epoch import train_epoch
from .ynet import YNet


class HipBCEWithLogitsLoss(nn.Module):
    """nn.BCEWithLogitsLoss() (mean) on the fused HIP kernels.  `expected_grad` is the factor the caller is about to
    multiply the loss by (train_epoch sets it to its loss_scale): the gradient of the logits is then produced in the
    same pass as the loss.  It is a hint only -- results do not depend on it."""

    fuses_with_predictor = True      # a YNetDecoder told about the target (models/ynet.py: announce_bce_target) computes the
                                     # loss in its predictor kernel; forward() then only hands that value out

    def __init__(self):
        super().__init__()
        self.expected_grad = 1.0

    def forward(self, input, target):
        fused = getattr(input, "_ynet_fused_bce", None)
        if fused is not None:
            # The fused predictor returned these logits as NON-differentiable (its kernel already produced the gradients for
            # the announced target and upstream gradient): any other use would silently drop the decoder's gradient.
            same_target = fused[0] is target or (torch.is_tensor(target) and target.data_ptr() == fused[0].data_ptr()
                                                 and target.shape == fused[0].shape and target.stride() == fused[0].stride())
            if same_target and fused[2] == float(self.expected_grad):
                return fused[1]
            raise RuntimeError(
                "HipBCEWithLogitsLoss: these maps come from a decoder that was told its BCE target (announce_bce_target) "
                + ("but are compared with a different target tensor" if not same_target else
                   f"for the upstream gradient {fused[2]} but expected_grad is now {float(self.expected_grad)}")
                + "; compute the maps outside announce_bce_target (or set YNET_PRED_BCE=0) for any other use")
        return ops.bce_with_logits(input, target, self.expected_grad)


def _mark_bias(module):
    for name, p in module.named_parameters():
        if "bias" in name:
            p.requires_grad = True


def mark_encoder_bias_trainable(model):
    _mark_bias(model.encoder)
    return model


def mark_goal_bias_trainable(model):
    _mark_bias(model.goal_decoder)
    return model


def mark_traj_bias_trainable(model):
    _mark_bias(model.traj_decoder)
    return model


def mark_ynet_bias_trainable(model):
    return mark_traj_bias_trainable(mark_goal_bias_trainable(mark_encoder_bias_trainable(model)))


_FUSION_PARTS = {"scene": ("scene",), "motion": ("motion",), "fusion": ("fusion",),
                 "scene_fusion": ("scene", "fusion"), "motion_fusion": ("motion", "fusion"),
                 "scene_motion": ("scene", "motion"), "scene_motion_fusion": ("scene", "motion", "fusion")}


def apply_freeze_policy(model, train_net, position=(), network=None, ynet_bias=False):
    """Which tensors train, per ``train_net`` (models/trainer.py:113-195)."""
    for p in model.semantic_segmentation.parameters():
        p.requires_grad = False
    if train_net in ("all", "train"):
        return model
    for p in model.parameters():
        p.requires_grad = False
    position = [str(i) for i in position]
    enc = model.encoder
    if train_net == "encoder" and len(position) == 0:
        for p in enc.parameters():
            p.requires_grad = True
    elif train_net == "encoder":
        for name, p in enc.named_parameters():
            if name.split(".")[1] in position:
                p.requires_grad = True
    elif "serial" in train_net or "parallel" in train_net:
        key = "serial" if "serial" in train_net else "parallel"
        for name, p in enc.named_parameters():
            if key in name:
                p.requires_grad = True
    elif "mosa" in train_net:
        for name, p in enc.named_parameters():
            if "lora" in name:
                p.requires_grad = True
    elif "semantic" in train_net:
        for name, p in model.named_parameters():
            if "semantic_adapter" in name:
                p.requires_grad = True
    elif network == "fusion" and train_net in _FUSION_PARTS:
        for part in _FUSION_PARTS[train_net]:
            for p in getattr(enc, part + "_stages").parameters():
                p.requires_grad = True
    elif train_net == "biasEncoder":
        mark_encoder_bias_trainable(model)
    elif train_net == "biasGoal":
        mark_goal_bias_trainable(model)
    elif train_net == "biasTraj":
        mark_traj_bias_trainable(model)
    elif train_net == "bias":
        mark_ynet_bias_trainable(model)
    elif "segmentation" in train_net:
        layer = train_net.split("_")[1]
        for name, p in model.semantic_segmentation.named_parameters():
            if (layer in name) if layer in ("head", "bias", "bn") else re.search(rf"decoder.blocks.\d.{layer}", name):
                p.requires_grad = True
    else:
        raise NotImplementedError
    if ynet_bias:
        mark_ynet_bias_trainable(model)
    return model


class YNetTrainer:
    def __init__(self, params, device=None):
        self.params = params
        self.device = device if device else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.device = torch.device(self.device)
        if self.device.type == "cuda" and self.device.index is not None and torch.cuda.is_available():
            torch.cuda.set_device(self.device)      # the HIP kernels launch on the current device's streams
        print(f"Working on {self.device}")
        self.division_factor = 2 ** len(params["encoder_channels"])
        self.template_size = int(4200 * params["resize_factor"])
        self.model = YNet(
            obs_len=params["obs_len"], pred_len=params["pred_len"],
            segmentation_model_fp=params["segmentation_model_fp"], use_features_only=params["use_features_only"],
            n_semantic_classes=params["n_semantic_classes"], encoder_channels=params["encoder_channels"],
            decoder_channels=params["decoder_channels"], n_waypoints=len(params["waypoints"]),
            train_net=params["train_net"], position=params["position"], network=params["network"],
            n_fusion=params["n_fusion"])
        self.dp = None

    # ------------------------------------------------------------------------------------------
    def templates(self, kernlen=None, nsig=None):
        """models/trainer.py:60-62, 209-211.  On the GPU the templates are analytic (ops.AnalyticTemplate: the windows
        get_patch slices out of them are computed in the kernel, bit-identical, and the 4.4 / 7.7 MB arrays are never
        built); YNET_ANALYTIC_HEATMAPS=0 keeps the materialised tensors."""
        analytic = self.device.type == "cuda" and os.environ.get("YNET_ANALYTIC_HEATMAPS", "1") != "0"
        if analytic:
            input_template = analytic_dist_template(self.template_size, self.device)
        else:
            input_template = torch.Tensor(create_dist_mat(size=self.template_size)).to(self.device)
        if kernlen is None:
            return input_template
        if analytic:
            gt_template = analytic_gaussian_template(self.template_size, kernlen, nsig, False, self.device)
        else:
            gt_template = torch.Tensor(create_gaussian_heatmap_template(
                size=self.template_size, kernlen=kernlen, nsig=nsig, normalize=False)).to(self.device)
        return input_template, gt_template

    def train(self, df_train, df_val, train_image_path, val_image_path, experiment_name):
        return self._train(df_train, df_val, train_image_path, val_image_path, experiment_name, **self.params)

    def _train(self, df_train, df_val, train_image_path, val_image_path, experiment_name, ckpt_path,
               dataset_name, resize_factor, obs_len, pred_len, batch_size, lr, n_epoch,
               waypoints, n_goal, n_traj, kernlen, nsig, e_unfreeze, loss_scale, temperature,
               use_raw_data=False, save_every_n=10, train_net="all", position=[],
               fine_tune=False, augment=False, ynet_bias=False,
               use_CWS=False, resl_thresh=0.002, CWS_params=None, n_early_stop=5,
               steps=[20], lr_decay_ratio=0.1, network=None, swap_semantic=False, window_size=9,
               smooth_val=False, **kwargs):
        train_images, train_loader, self.homo_mat = self.prepare_data(
            df_train, train_image_path, dataset_name, "train", obs_len, pred_len, resize_factor, use_raw_data, augment)
        val_images, val_loader, _ = self.prepare_data(
            df_val, val_image_path, dataset_name, "val", obs_len, pred_len, resize_factor, use_raw_data, False)

        model = self.model.to(self.device)
        apply_freeze_policy(model, train_net, position, network, ynet_bias)
        optimizer = torch.optim.Adam(model.parameters(), lr=lr)
        if fine_tune:
            print("LR Schedular because finetuning")
            lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=steps, gamma=lr_decay_ratio)
        print("The number of trainable parameters: {:d}".format(
            sum(p.numel() for p in model.parameters() if p.requires_grad)))
        criterion = HipBCEWithLogitsLoss()
        input_template, gt_template = self.templates(kernlen, nsig)
        if self.dp is not None:
            self.dp.__init__(model.parameters(), self.dp.group)

        best_val_ADE, best_epoch = 99999999999999, 0
        self.val_ADE, self.val_FDE = [], []
        state_dicts = deque()
        half_window_size = (window_size // 2) + 1
        rank0 = self.dp is None or self.dp.rank == 0

        print("Start training")
        for e in tqdm(range(n_epoch), desc="Epoch"):
            train_ADE, train_FDE, train_loss = train_epoch(
                model, train_loader, train_images, optimizer, criterion, loss_scale, self.device,
                dataset_name, self.homo_mat, gt_template, input_template, waypoints,
                e, obs_len, pred_len, batch_size, e_unfreeze, resize_factor, network, swap_semantic, dp=self.dp)
            val_ADE, val_FDE, _, _ = evaluate(
                model, val_loader, val_images, self.device, dataset_name, self.homo_mat, input_template, waypoints,
                "val", n_goal, n_traj, obs_len, batch_size, resize_factor, temperature, False, use_CWS,
                resl_thresh, CWS_params, network=network, swap_semantic=swap_semantic, dp=self.dp)
            line = (f"Epoch {e}: \tTrain (Top-1) ADE: {train_ADE:.2f} FDE: {train_FDE:.2f} \t\t"
                    f"Val (Top-k) ADE: {val_ADE:.2f} FDE: {val_FDE:.2f}")
            print(line + (f"   lr={lr_scheduler.get_last_lr()[0]}" if fine_tune else ""))
            self.val_ADE.append(val_ADE)
            self.val_FDE.append(val_FDE)
            if fine_tune:
                lr_scheduler.step()

            if smooth_val:
                print("Length: ", len(state_dicts))
                if len(state_dicts) == half_window_size:
                    curr_model_dict = state_dicts.popleft()
                state_dicts.append(deepcopy(model.state_dict()))
                val_ADE = best_val_ADE + 1 if e < window_size else sum(self.val_ADE[-window_size:]) / window_size
            else:
                curr_model_dict = deepcopy(model.state_dict())

            if val_ADE < best_val_ADE:
                best_val_ADE = val_ADE
                best_epoch = e - half_window_size + 1 if smooth_val else e
                best_state_dict = curr_model_dict
                if not fine_tune and rank0:
                    print(f"Best Epoch {e}: \nVal ADE: {val_ADE} \nVal FDE: {val_FDE}")
                    pathlib.Path(ckpt_path).mkdir(parents=True, exist_ok=True)
                    torch.save(model.state_dict(), f"{ckpt_path}/{experiment_name}_weights.pt")
            if (e + 1) % save_every_n == 0 and rank0:
                pathlib.Path(ckpt_path).mkdir(parents=True, exist_ok=True)
                self.save_params(f"{ckpt_path}/{experiment_name}__epoch_{e}.pt", train_net)
            if fine_tune and (best_val_ADE < min(self.val_ADE[-n_early_stop:])):
                print(f"Early stop at epoch {e}")
                break

        print(f"Best epoch at {best_epoch}")
        if best_epoch != 0:
            model.load_state_dict(best_state_dict, strict=True)
        if rank0:
            pathlib.Path(ckpt_path).mkdir(parents=True, exist_ok=True)
            self.save_params(f"{ckpt_path}/{experiment_name}.pt", train_net)
        return self.val_ADE, self.val_FDE

    # ------------------------------------------------------------------------------------------
    def test(self, df_test, image_path, return_preds=False, return_samples=False):
        return self._test(df_test, image_path, return_preds=return_preds, return_samples=return_samples, **self.params)

    def _test(self, df_test, image_path, dataset_name, resize_factor, batch_size, n_round, obs_len, pred_len,
              waypoints, n_goal, n_traj, temperature, rel_threshold, use_TTST, use_CWS, CWS_params,
              use_raw_data=False, return_preds=False, return_samples=False, network=None, swap_semantic=False,
              **kwargs):
        test_images, test_loader, self.homo_mat = self.prepare_data(
            df_test, image_path, dataset_name, "test", obs_len, pred_len, resize_factor, use_raw_data)
        model = self.model.to(self.device)
        input_template = self.templates()
        self.eval_ADE, self.eval_FDE = [], []
        list_metrics, list_trajs = [], []
        print("TTST setting:", use_TTST)
        print("Start testing")
        for e in tqdm(range(n_round), desc="Round"):
            test_ADE, test_FDE, df_metrics, trajs_dict = evaluate(
                model, test_loader, test_images, self.device, dataset_name, self.homo_mat, input_template, waypoints,
                "test", n_goal, n_traj, obs_len, batch_size, resize_factor, temperature, use_TTST, use_CWS,
                rel_threshold, CWS_params, return_preds=return_preds, return_samples=return_samples,
                network=network, swap_semantic=swap_semantic, dp=self.dp)
            list_metrics.append(df_metrics)
            list_trajs.append(trajs_dict)
            print(f"Round {e}: \nTest ADE: {test_ADE} \nTest FDE: {test_FDE}")
            self.eval_ADE.append(test_ADE)
            self.eval_FDE.append(test_FDE)
        avg_ade = sum(self.eval_ADE) / len(self.eval_ADE)
        avg_fde = sum(self.eval_FDE) / len(self.eval_FDE)
        print(f"\nAverage performance (by {n_round}): \nTest ADE: {avg_ade} \nTest FDE: {avg_fde}")
        return avg_ade, avg_fde, list_metrics, list_trajs

    # ------------------------------------------------------------------------------------------
    def prepare_data(self, df, image_path, dataset_name, mode, obs_len, pred_len, resize_factor, use_raw_data,
                     augment=False):
        name = dataset_name.lower()
        if name not in ("sdd", "ind-dataset-v1.0", "eth"):
            raise ValueError(f"{name} dataset is not supported")
        if name == "eth":
            raise NotImplementedError("ETH/UCY homographies are read from data files: out of the MI355X hot path")
        if not isinstance(image_path, dict):
            raise ImportError("decoding scene images needs OpenCV + segmentation_models_pytorch (out of scope here): "
                              "pass a dict {scene_id: float tensor [C,H,W]} (pre-processed, padded to a multiple "
                              f"of {self.division_factor}) instead of an image directory")
        # pad (utils/image_utils.py:95-107) and, for raw label maps, the one-hot encoding of
        # preprocess_image_for_segmentation(seg_mask=True) (image_utils.py:74-81) run on the device; decoding image files,
        # cv2.resize and the RGB normalisation of the segmentation backbone stay outside (no OpenCV / smp in this image)
        images = dict(image_path)
        if augment:
            # models/trainer.py:566-571: augment_data BEFORE resize / pad / encode -- every scene rotated by 90 / 180 / 270 degrees, then everything
            # flipped (utils/data_utils.py:176-233; the index permutations and the coordinate transforms run on the device).  Raw label maps only:
            # pre-processed planes arrive padded, and the reference pads AFTER it rotates (the border stays at the bottom / right)
            if any((im.dim() if torch.is_tensor(im) else np.asarray(im).ndim) != 2 for im in images.values()):
                raise ValueError("augment=True needs the raw label maps [H, W] (the reference augments before it pads and encodes); "
                                 "pre-processed planes [C, H, W] cannot be rotated correctly here")
            from ..utils.data_utils import augment_data
            df, images = augment_data(df, images=images, seg_mask=True)
            print("Augmented data and images")
        else:
            print("No data and images augmentation")
        for k, im in images.items():
            t = im if torch.is_tensor(im) else torch.from_numpy(np.ascontiguousarray(im))
            if t.dim() == 2:            # a label map [H, W]: padded with label 0, then one-hot planes
                if self.device.type != "cuda":
                    raise RuntimeError("raw label maps are encoded by the HIP kernels: a HIP device is required")
                images[k] = ops.seg_onehot_pad(t.to(self.device), classes=self.params.get("n_semantic_classes", 6) if hasattr(self.params, "get") else 6,
                                               division_factor=self.division_factor)
            elif t.shape[-1] % self.division_factor or t.shape[-2] % self.division_factor:
                # PRE-PROCESSED planes cannot be padded correctly here: the reference pads the raw image / label map first and
                # encodes afterwards (models/trainer.py:581-582), so its border is class 0 (plane 0 == 1) for a label map and
                # (0 - mean) / std for RGB -- a zero border in every plane would be neither.  Raw label maps [H, W]
                # take the branch above (ynet_seg_onehot_pad pads before it encodes, like the reference).
                raise ValueError(f"scene {k}: pre-processed planes {tuple(t.shape)} are not padded to a multiple of "
                                 f"{self.division_factor}; pass the raw label map [H, W] (padded and encoded on the device) "
                                 f"or pad before pre-processing (utils.image_utils.pad)")
            else:
                images[k] = t
        image_path = images
        dataset = SceneDataset(df, resize=resize_factor, total_len=obs_len + pred_len)
        generator = None
        if mode == "train" and self.dp is not None:
            # data-parallel ranks must walk the scenes in the SAME order (train_epoch shards the batches of one scene
            # over the ranks): the shuffle draws from a generator seeded identically everywhere (seed from rank 0)
            generator = torch.Generator()
            generator.manual_seed(self.dp.shared_seed())
        loader = DataLoader(dataset, batch_size=1, collate_fn=scene_collate, shuffle=(mode == "train"), generator=generator)
        return image_path, loader, None

    def load_params(self, path):
        on_gpu = self.device.type == "cuda"
        self.model.load_state_dict(torch.load(path, map_location=None if on_gpu else "cpu", weights_only=False), strict=False)
        print("Loaded ynet model to GPU" if on_gpu else "Loaded ynet model to CPU")

    def save_params(self, path, train_net):
        if train_net in ("all", "train"):
            state_dict = {k: v for k, v in self.model.state_dict().items() if "segmentation" not in k}
        else:
            state_dict = OrderedDict((n, p) for n, p in self.model.named_parameters() if p.requires_grad)
        torch.save(state_dict, path)

    def load_separated_params(self, pretrained_path, tuned_path):
        on_gpu = self.device.type == "cuda"
        for path in (pretrained_path, tuned_path):
            self.model.load_state_dict(torch.load(path, map_location=None if on_gpu else "cpu", weights_only=False), strict=False)
        print("Loaded ynet model to GPU" if on_gpu else "Loaded ynet model to CPU")
