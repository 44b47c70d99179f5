"""Heat-map templates, patch gathering and goal sampling (mirror of utils/image_utils.py).

  create_dist_mat / create_gaussian_heatmap_template : float64 NumPy, bit-identical to the reference
      (utils/image_utils.py:7-37); they are built once per run, not on the hot path.
  get_patch        : same signature/return type as the reference (a list of [H,W] tensors) but the
      windows come from ONE ynet_gather_patch launch (utils/image_utils.py:40-63).
  gather_patches   : the same without the list, [N,H,W] contiguous (what train_epoch/evaluate use).
  sampling         : multinomial goal / waypoint sampling (utils/image_utils.py:110-135) by ynet_multinomial, a
      device sampler with a documented counter-based generator (YNET_SAMPLER=torch: torch.multinomial instead).
  pad / preprocess_image_for_segmentation(seg_mask=True) : the part of the scene pipeline that needs neither OpenCV nor the
      segmentation backbone (utils/image_utils.py:66-81, 95-107), on the device (ynet_pad2d, ynet_seg_onehot_pad).
  resize(seg_mask=True) : cv2.resize(..., INTER_NEAREST) of label maps (utils/image_utils.py:83-87) restated from OpenCV's
      published rule on the device (ynet_resize_nearest) -- parity unpinned, cv2 is absent from the image.
Image decoding, cv2.resize(INTER_AREA) of RGB images and their smp normalisation need cv2 + smp and stay out of scope.
"""
import os

import numpy as np
import torch

from .. import ops


def gkern(kernlen=31, nsig=4):
    half = (kernlen - 1) / 2.0
    ax = np.linspace(-half, half, kernlen)
    g = np.exp(-0.5 * (ax[None, :] ** 2 + ax[:, None] ** 2) / np.square(nsig))
    return g / np.sum(g)


def create_gaussian_heatmap_template(size, kernlen=81, nsig=4, normalize=True):
    template = np.zeros([size, size])
    kernel = gkern(kernlen=kernlen, nsig=nsig)
    m = kernel.shape[0]
    lo, hi = size // 2 - m // 2, size // 2 + (m + 1) // 2
    template[lo:hi, lo:hi] = kernel
    return template / template.max() if normalize else template


def create_dist_mat(size, normalize=True):
    off = np.arange(size, dtype=np.int64) - size // 2
    dist = np.sqrt((off[:, None] ** 2 + off[None, :] ** 2).astype(np.float64))
    return dist / dist.max() * 2 if normalize else dist


def analytic_dist_template(size, device):
    """create_dist_mat(size) as an ops.AnalyticTemplate: windows are computed in the kernel, bit-identical to slices of
    the float64 -> fp32 array, and the S x S array is never built."""
    return ops.AnalyticTemplate("dist", size, device)


def analytic_gaussian_template(size, kernlen=81, nsig=4, normalize=True, device="cuda"):
    """create_gaussian_heatmap_template(...) as an ops.AnalyticTemplate (only its kernlen x kernlen blob is kept)."""
    blob = gkern(kernlen=kernlen, nsig=nsig)
    if normalize:
        blob = blob / blob.max()
    return ops.AnalyticTemplate("gaussian", size, device, blob=blob, normalize=normalize)


def pad(images, division_factor=32):
    """utils/image_utils.py:95-107, in place on the dict like the reference.  Values: device tensors [H, W] / [C, H, W]
    (padded by ynet_pad2d) or NumPy arrays [H, W] / [H, W, C] as cv2.imread returns them (np.pad: a constant zero border at the
    bottom / right is all that cv2.copyMakeBorder(..., BORDER_CONSTANT) does)."""
    for key, im in images.items():
        if torch.is_tensor(im):
            images[key] = ops.pad_planes(im, division_factor) if im.is_floating_point() else \
                ops.pad_planes(im.float(), division_factor).to(im.dtype)
        else:
            H, W = im.shape[:2]
            Hn, Wn = int(np.ceil(H / division_factor) * division_factor), int(np.ceil(W / division_factor) * division_factor)
            images[key] = np.pad(im, ((0, Hn - H), (0, Wn - W)) + ((0, 0),) * (im.ndim - 2), mode="constant")


def preprocess_image_for_segmentation(images, encoder="resnet101", encoder_weights="imagenet", seg_mask=False, classes=6, device=None):
    """utils/image_utils.py:66-81 for segmentation MASKS (seg_mask=True): label maps [H, W] (NumPy or tensor) -> one-hot float
    tensors [classes, H, W] on the device.  The RGB branch applies smp's encoder-specific normalisation and needs
    segmentation_models_pytorch, which this image does not have."""
    if not seg_mask:
        raise ImportError("preprocess_image_for_segmentation(seg_mask=False) needs segmentation_models_pytorch (out of scope here)")
    for key, im in images.items():
        lab = im if torch.is_tensor(im) else torch.from_numpy(np.ascontiguousarray(im))
        if device is not None:
            lab = lab.to(device)
        images[key] = ops.seg_onehot_pad(lab, classes=classes, division_factor=1)


def gather_patches(template, traj, H, W):
    """template [S,S] (device), traj [N,2] (x,y) -> [N,H,W] device tensor."""
    return ops.gather_patches(template, traj, H, W)


def get_patch(template, traj, H, W):
    return list(gather_patches(template, traj, H, W).unbind(0))


# "device": ynet_multinomial (Philox4x32-10, documented in include/ynet_hip.h; a pure function of the map and a seed that
# each call draws from torch's default CPU generator, so torch.manual_seed() makes a sweep reproducible and the CPU
# oracle can replay it).  "torch": torch.multinomial on the device (torch's own Philox stream).
SAMPLER = os.environ.get("YNET_SAMPLER", "device")


def draw_seed() -> int:
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def sampling(probability_map, num_samples, rel_threshold=None, replacement=False, seed=None):
    """utils/image_utils.py:110-135: [B,C,H,W] maps -> [B,C,num_samples,2] (x, y) pixel coordinates."""
    b, c, h, w = probability_map.shape
    prob = probability_map.reshape(b * c, -1)
    if SAMPLER == "device" and probability_map.is_cuda and (replacement or num_samples <= min(48, h * w)):
        idx = ops.multinomial(prob, num_samples, replacement, rel_threshold, draw_seed() if seed is None else seed)
    else:
        if rel_threshold is not None:
            keep = prob >= prob.max(dim=1, keepdim=True)[0] * rel_threshold
            prob = prob * keep.int()
            prob = prob / prob.sum()
        idx = torch.multinomial(prob, num_samples=num_samples, replacement=replacement)
    idx = idx.view(b, c, -1).float()
    return torch.stack([idx % w, torch.floor(idx / w)], dim=3)


def image2world(image_coords, scene, homo_mat, resize):
    """ETH/UCY pixel -> world coordinates (utils/image_utils.py:138-162)."""
    pts = image_coords.clone()
    if pts.dim() == 4:
        pts = pts.reshape(-1, image_coords.shape[2], 2)
    if scene in ["eth", "hotel"]:
        pts = pts.flip(-1)
    pts = pts / resize
    pts = torch.cat([pts, torch.ones_like(pts[..., :1])], dim=-1).reshape(-1, 3)
    pts = (homo_mat[scene] @ pts.T).T
    pts = pts[:, :2] / pts[:, 2:]
    return pts.view_as(image_coords)


def swap_pavement_terrain(semantic_img):
    if semantic_img.dim() != 4:
        raise ValueError(f"semanctic image has shape {semantic_img.shape} but should have 4 dimensions")
    semantic_img[:, [1, 2]] = semantic_img[:, [2, 1]]
    return semantic_img


def _needs_cv2(name):
    def fn(*a, **k):
        raise ImportError(f"{name} needs OpenCV / segmentation_models_pytorch image I/O, which is outside the "
                          f"MI355X hot path; pre-process scene images with the reference and pass tensors")
    return fn


def resize(images, factor, seg_mask=False, device=None):
    """utils/image_utils.py:83-92, in place on the dict like the reference -- for segmentation MASKS (seg_mask=True): label maps
    [H, W] -> [round(H f), round(W f)] by nearest-neighbour (ynet_resize_nearest: OpenCV's published INTER_NEAREST rule, PARITY
    UNPINNED -- cv2 is absent from this image).  Device tensors stay tensors; NumPy arrays (what cv2.imread returns) go through
    the device and come back as NumPy arrays of the same dtype.  The RGB branch (INTER_AREA) needs OpenCV and stays out of scope."""
    if not seg_mask:
        raise ImportError("resize(seg_mask=False) is cv2.resize(..., INTER_AREA) of RGB images: OpenCV image I/O is outside the "
                          "MI355X hot path; pre-process scene images with the reference and pass tensors")
    for key, im in images.items():
        if torch.is_tensor(im):
            images[key] = ops.resize_nearest(im if device is None else im.to(device), factor)
        else:
            arr = np.ascontiguousarray(im)
            if not torch.cuda.is_available():
                raise RuntimeError("resize: label maps are resized by the HIP kernel: a HIP device is required")
            t = torch.from_numpy(arr.astype(np.int32)).to(device if device is not None else "cuda")
            images[key] = ops.resize_nearest(t, factor).cpu().numpy().astype(arr.dtype)
