"""augment_data's pinnable piece on the device (utils/data_utils.py:113-233): rot / fliplr of label maps, planes and
trajectories.  The image side is an index permutation and must be BIT-EXACT against np.rot90 / np.fliplr (what OpenCV documents for
ROTATE_90_COUNTERCLOCKWISE and flipCode 1); the coordinate side follows the reference's R matrices in float64.  tests/golden/augment.npz
holds the imported REFERENCE's augment_data / rot / fliplr outputs with cv2.rotate / cv2.flip stubbed by those NumPy equivalents
(oracle/_stubs/cv2.py, labelled in the fixture: PARITY UNPINNED against OpenCV itself)."""
import numpy as np
import pandas as pd
import pytest
import torch

from conftest import Golden, pkg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 1), (5, 7), (7, 5), (64, 96), (3, 33, 17), (2, 3, 8, 8)])
@pytest.mark.parametrize("dtype", [torch.int32, torch.float32])
def test_rot90_and_flip_are_exact_index_permutations(dev, shape, dtype):
    ops = pkg("ops")
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randint(-5, 200, shape, generator=g).to(dtype)
    if dtype == torch.float32:
        x = x + torch.rand(shape, generator=g)
        x.view(-1)[0] = float("nan")       # a permutation moves bit patterns: NaN payloads included
    a = x.numpy()
    for k in range(0, 5):
        for flip in (False, True):
            want = np.rot90(a, k, axes=(-2, -1))
            if flip:
                want = np.flip(want, axis=-1)
            got = ops.rot90_flip(x.to(dev), k, flip).cpu().numpy()
            assert got.shape == want.shape
            assert np.array_equal(got.view(np.int32), np.ascontiguousarray(want).view(np.int32)), (k, flip)


def test_rot_and_fliplr_match_the_reference_fixture(dev):
    du = pkg("utils.data_utils")
    g = Golden("augment")
    img = g.z["rot/image"]
    pts = pd.DataFrame({"x": g.z["rot/x"], "y": g.z["rot/y"]})
    for k in (1, 2, 3):
        d, im = du.rot(pts.copy(), img.copy(), k)
        assert im.dtype == img.dtype and np.array_equal(im, g.z[f"rot/k{k}/image"]) and np.array_equal(im, np.rot90(img, k))
        np.testing.assert_allclose(d["x"].to_numpy(), g.z[f"rot/k{k}/x"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(d["y"].to_numpy(), g.z[f"rot/k{k}/y"], rtol=0, atol=1e-12)
    d, im = du.fliplr(pts.copy(), img.copy())
    assert np.array_equal(im, g.z["flip/image"]) and np.array_equal(im, np.fliplr(img))
    np.testing.assert_allclose(d["x"].to_numpy(), g.z["flip/x"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(d["y"].to_numpy(), g.z["flip/y"], rtol=0, atol=1e-12)
    # known answers: a point at the image centre stays at the (new) centre; the corner (0, 0) of a 5 x 7 image goes to (0, 7) under one
    # counter-clockwise quarter turn (x' = y, y' = W - x)
    d, _ = du.rot(pd.DataFrame({"x": [3.5, 0.0], "y": [2.5, 0.0]}), img.copy(), 1)
    np.testing.assert_allclose(d[["x", "y"]].to_numpy(), [[2.5, 3.5], [0.0, 7.0]], atol=1e-12)
    # device tensors [C, H, W] rotate over their last two dimensions
    t = torch.from_numpy(np.ascontiguousarray(np.moveaxis(img, 2, 0)).astype(np.float32)).to(dev)
    _, rt = du.rot(pts.copy(), t, 3)
    assert torch.equal(rt.cpu(), torch.from_numpy(np.ascontiguousarray(np.rot90(np.moveaxis(img, 2, 0).astype(np.float32), 3, axes=(1, 2)))))


def test_augment_data_matches_the_reference_fixture(dev):
    du = pkg("utils.data_utils")
    g = Golden("augment")
    df = pd.DataFrame({"frame": g.z["in/frame"], "trackId": g.z["in/metaId"], "x": g.z["in/x"], "y": g.z["in/y"],
                       "sceneId": [str(s) for s in g.z["in/sceneId"]], "metaId": g.z["in/metaId"]})
    images = {k: g.z["in/image/" + k] for k in g.keys("in/image/")}
    out, out_images = du.augment_data(df.copy(), images=dict(images), seg_mask=True)
    assert len(out) == 8 * len(df) == len(g.z["out/x"])
    assert [str(s) for s in g.z["out/sceneId"]] == out["sceneId"].tolist()
    assert np.array_equal(out["metaId"].to_numpy(), g.z["out/metaId"])
    np.testing.assert_allclose(out["x"].to_numpy(), g.z["out/x"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out["y"].to_numpy(), g.z["out/y"], rtol=0, atol=1e-12)
    assert list(out_images.keys()) == [str(k) for k in g.z["out/keys"]]
    for k, im in out_images.items():
        want = g.z["out/image/" + k]
        assert im.dtype == want.dtype and np.array_equal(im, want), k
    with pytest.raises(ImportError, match="cv2.imread"):
        du.augment_data(df.copy(), images={})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg("ops").rot90_flip(torch.zeros(4, 4, dtype=torch.int32), 1)


def test_trainer_prepare_data_with_augmentation(dev):
    """models/trainer.py:559-583 with augment=True on raw label maps: augment_data (8x the scenes and trajectories, on the device), then pad to a
    multiple of 32 and the one-hot planes -- every scene of the dict against np.rot90 / np.fliplr of its source followed by the reference's own
    pad + one-hot rule (class 0 border)."""
    import contextlib
    import io
    trn = pkg("models.trainer")
    params = dict(obs_len=8, pred_len=12, segmentation_model_fp=None, use_features_only=False, n_semantic_classes=6,
                  encoder_channels=[8, 8, 16, 16, 16], decoder_channels=[16, 16, 16, 8, 8], waypoints=[11],
                  train_net="mosa_1", position=["0"], network="original", n_fusion=None, resize_factor=0.25)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        t = trn.YNetTrainer(params, device=dev)
    rng = np.random.RandomState(3)
    maps = {"s0": rng.randint(0, 6, size=(40, 70)).astype(np.uint8), "s1": rng.randint(0, 6, size=(33, 33)).astype(np.uint8)}
    rows = []
    for m, (sid, im) in enumerate(maps.items()):
        for f in range(20):
            rows.append({"frame": f, "trackId": m, "x": float(rng.uniform(0, im.shape[1])), "y": float(rng.uniform(0, im.shape[0])), "sceneId": sid, "metaId": m})
    df = pd.DataFrame(rows)
    with contextlib.redirect_stdout(out):
        images, loader, homo = t.prepare_data(df, dict(maps), "sdd", "train", 8, 12, 0.25, False, augment=True)
    assert "Augmented data and images" in out.getvalue() and homo is None
    assert len(images) == 16 and len(loader.dataset) == 16
    for key, planes in images.items():
        src = maps[key.split("_")[0]]
        lab = src
        for suffix in key.split("_")[1:]:
            lab = np.rot90(lab, {"rot90": 1, "rot180": 2, "rot270": 3}[suffix]) if suffix.startswith("rot") else np.fliplr(lab)
        H, W = lab.shape
        Hp, Wp = -(-H // 32) * 32, -(-W // 32) * 32
        padded = np.zeros((Hp, Wp), dtype=np.int64)
        padded[:H, :W] = lab
        want = torch.nn.functional.one_hot(torch.from_numpy(padded), 6).permute(2, 0, 1).float()
        assert planes.shape == (6, Hp, Wp) and torch.equal(planes.cpu(), want), key
    with pytest.raises(ValueError, match="raw label maps"):
        with contextlib.redirect_stdout(out):
            t.prepare_data(df, {k: torch.zeros(6, 64, 96) for k in maps}, "sdd", "train", 8, 12, 0.25, False, augment=True)
