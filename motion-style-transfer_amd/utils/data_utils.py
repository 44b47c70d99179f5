"""Data augmentation of scenes and trajectories (mirror of utils/data_utils.py:113-233, SURVEY 8(f)-4 -- the part that is pinnable
without OpenCV).

  rot(df, image, k)   : the image k times cv2.rotate(ROTATE_90_COUNTERCLOCKWISE), the 'x' / 'y' columns rotated about the image
                        centre by the reference's matrix R = [[c, s], [-s, c]], c = cos(-k pi / 2), s = sin(-k pi / 2).
  fliplr(df, image)   : cv2.flip(image, 1) and R = [[-1, 0], [0, 1]].
  augment_data(...)   : the reference's loop -- every scene rotated by 90 / 180 / 270 degrees, then everything flipped: 8x the data,
                        scene ids suffixed '_rot90' / '_rot180' / '_rot270' / '_fliplr', metaIds offset as the reference does.

Both image operations are index permutations (= np.rot90 / np.fliplr) and run on the device bit-exactly (ynet_rot90_flip); the
coordinates are transformed on the device in float64 (ynet_rot_coords) with the very doubles NumPy computes for c and s.  What
stays out: reading image files (cv2.imread) -- `images` must hold the scenes already (label maps or planes, NumPy or tensors);
with an empty dict the reference's disk path would be needed and an ImportError says so.
"""
import numpy as np
import pandas as pd
import torch

from .. import ops


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("data_utils: the augmentation kernels run on a HIP device (no CPU fallback exists by design)")
    return torch.device("cuda", torch.cuda.current_device())


def _image_op(image, k, flip):
    """np.rot90(image, k) / np.fliplr over the first two dimensions of an OpenCV-style array [H, W] or [H, W, C] -- or over the
    LAST two of a device tensor [H, W] / [C, H, W] (the layout the rest of the package keeps scenes in)."""
    if torch.is_tensor(image):
        t = image if image.element_size() == 4 else (image.float() if image.is_floating_point() else image.to(torch.int32))
        return ops.rot90_flip(t, k, flip).to(image.dtype)
    arr = np.ascontiguousarray(image)
    planes = arr if arr.ndim == 2 else np.moveaxis(arr, 2, 0)
    if arr.dtype.kind not in "fiub" or arr.dtype.kind == "f" and arr.dtype != np.float32 or arr.dtype.kind in "iu" and arr.dtype.itemsize > 4 \
            or arr.dtype == np.uint32:
        raise TypeError(f"image dtype {arr.dtype} does not fit the 32-bit permutation kernel")
    work = planes.astype(np.float32 if arr.dtype.kind == "f" else np.int32)
    out = ops.rot90_flip(torch.from_numpy(np.ascontiguousarray(work)).to(_device()), k, flip).cpu().numpy().astype(arr.dtype)
    return out if arr.ndim == 2 else np.ascontiguousarray(np.moveaxis(out, 0, 2))


def _shape_hw(image):
    if torch.is_tensor(image):
        return int(image.shape[-2]), int(image.shape[-1])
    return int(image.shape[0]), int(image.shape[1])


def _transform(df, center, matrix, offset):
    xy = df.copy()
    pts = torch.from_numpy(np.ascontiguousarray(xy[["x", "y"]].to_numpy(dtype=np.float64))).to(_device())
    ops.rot_coords(pts, center, matrix, offset)
    out = pts.cpu().numpy()
    xy["x"], xy["y"] = out[:, 0], out[:, 1]
    return xy


def rot(df, image, k=1):
    """utils/data_utils.py:113-142.  Returns (rotated DataFrame, rotated image)."""
    y0, x0 = _shape_hw(image)
    c, s = np.cos(-k * np.pi / 2), np.sin(-k * np.pi / 2)
    image = _image_op(image, k, False)
    y1, x1 = _shape_hw(image)
    return _transform(df, (x0 / 2, y0 / 2), [[c, s], [-s, c]], (x1 / 2, y1 / 2)), image


def fliplr(df, image):
    """utils/data_utils.py:145-171."""
    y0, x0 = _shape_hw(image)
    image = _image_op(image, 0, True)
    return _transform(df, (x0 / 2, y0 / 2), [[-1.0, 0.0], [0.0, 1.0]], (x0 / 2, y0 / 2)), image


def augment_data(data, image_path="data/SDD/train", images=None, image_file="reference.jpg", seg_mask=False, use_raw_data=False):
    """utils/data_utils.py:176-233 with the scenes GIVEN in `images` (key: sceneId); returns (augmented DataFrame, images) with the
    rotated / flipped scenes added under the reference's keys."""
    images = {} if images is None else images
    missing = [s for s in data.sceneId.unique() if s not in images]
    if missing:
        raise ImportError(f"augment_data: scenes {missing} are not in `images`; reading {image_file} from {image_path} is cv2.imread "
                          f"(OpenCV image I/O is outside the MI355X hot path): load the scenes and pass them in")
    data_ = data.copy()
    k2rot = {1: "_rot90", 2: "_rot180", 3: "_rot270"}
    for k in (1, 2, 3):
        metaId_max = data["metaId"].max()
        for scene in data_.sceneId.unique():
            data_rot, im = rot(data_[data_.sceneId == scene], images[scene], k)
            images[scene + k2rot[k]] = im
            data_rot["sceneId"] = scene + k2rot[k]
            data_rot["metaId"] = data_rot["metaId"] + metaId_max + 1
            data = pd.concat([data, data_rot], axis=0)
    metaId_max = data["metaId"].max()
    for scene in data.sceneId.unique():
        data_flip, im_flip = fliplr(data[data.sceneId == scene], images[scene])
        data_flip["sceneId"] = data_flip["sceneId"] + "_fliplr"
        data_flip["metaId"] = data_flip["metaId"] + metaId_max + 1
        data = pd.concat([data, data_flip], axis=0)
        images[scene + "_fliplr"] = im_flip
    return data, images
