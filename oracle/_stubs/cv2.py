"""Import stub used ONLY by oracle/gen_goldens.py inside the build container.

The reference imports cv2 at module top (utils/image_utils.py:1, utils/data_utils.py:2) but the
only call reachable without image files is cv2.setRNGSeed (utils/data_utils.py:950).
"""


def setRNGSeed(seed):  # noqa: N802 (name dictated by the real module)
    return None
