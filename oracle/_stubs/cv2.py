"""Import stub used ONLY by oracle/gen_goldens.py inside the build container.

The reference imports cv2 at module top (utils/image_utils.py:1, utils/data_utils.py:2); the only call reachable without image
files on the training path is cv2.setRNGSeed (utils/data_utils.py:950).

For the augmentation fixture (tests/golden/augment.npz; utils/data_utils.py:113-233) three more names are provided as their NumPy
EQUIVALENTS -- labelled as such in the fixture's metadata: OpenCV documents ROTATE_90_COUNTERCLOCKWISE and flipCode 1 as exact index
permutations (rotate: dst(i, j) = src(j, cols - 1 - i); flip: dst(i, j) = src(i, cols - 1 - j)), which is what np.rot90(., 1) and
np.fliplr compute.  imread serves arrays registered in FILES (no image decoding exists here).  PARITY UNPINNED against OpenCV itself.
"""
import numpy as np

ROTATE_90_COUNTERCLOCKWISE = 2      # (the real module's enum value)
FILES = {}                          # path -> array, filled by oracle/gen_goldens.py


def setRNGSeed(seed):  # noqa: N802 (name dictated by the real module)
    return None


def rotate(src, rotateCode):  # noqa: N802,N803
    if rotateCode != ROTATE_90_COUNTERCLOCKWISE:
        raise NotImplementedError("stub: only ROTATE_90_COUNTERCLOCKWISE (the one the reference uses)")
    return np.ascontiguousarray(np.rot90(src, 1))


def flip(src, flipCode):  # noqa: N803
    if flipCode != 1:
        raise NotImplementedError("stub: only flipCode 1 (the one the reference uses)")
    return np.ascontiguousarray(np.fliplr(src))


def imread(path, flags=1):
    return FILES[path].copy()
