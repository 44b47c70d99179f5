"""Host-side logic of the product (no GPU): state-dict contract, freeze policy, LoRA module
semantics/init, lazy concatenation, templates, sampling, dataset contract, checkpoint format."""
import io
import contextlib
import os
import sys

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import Golden, TINY_CASES, build_model, pkg, ROOT
from oracle import ynet_oracle as O


@pytest.mark.parametrize("case", TINY_CASES)
def test_state_dict_contract_and_freeze_policy(case):
    g = Golden(case)
    cfg = g.cfg()
    sd = g.state_dict()
    model = build_model(cfg, sd)                       # strict load: key names and shapes are the reference's
    assert list(model.state_dict().keys()) == list(sd.keys())
    trainable = [n for n, p in model.named_parameters() if p.requires_grad]
    assert trainable == [str(s) for s in g.z["step/trainable"]]
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == int(g.z["step/n_trainable"])


def test_fullsize_parameter_counts():
    cfg = O.sdd_short(train_net="mosa_4", position=["0", "1", "2", "3", "4"])
    m = build_model(cfg)
    assert sum(p.numel() for n, p in m.named_parameters() if "lora" not in n) == 1641381
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 32760
    cfg = O.ind_long(network="fusion", n_fusion=2, train_net="mosa_3", position=["scene"])
    m = build_model(cfg)
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 5346
    assert m.encoder.scene_stages[0][0].lora_A.shape == (9, 18)


def test_unknown_modes_raise_like_the_reference():
    ynet, trainer = pkg("models.ynet"), pkg("models.trainer")
    kw = dict(encoder_channels=[8, 8, 16, 16, 16], decoder_channels=[16, 16, 16, 8, 8])
    with pytest.raises(ValueError, match="No network parameter"):
        ynet.YNet(8, 12, None, train_net="train", network=None, **kw)
    m = ynet.YNet(8, 12, None, train_net="train", network="original", **kw)
    with pytest.raises(NotImplementedError):
        trainer.apply_freeze_policy(m, "nonsense")
    with pytest.raises(AssertionError):
        ynet.YNetEncoderFusion(6, 8, [8, 9, 16], train_net="train", n_fusion=1)


def test_lora_conv_matches_loralib_restatement_init_and_names():
    """Same seed -> the same parameters as the (restated) loralib 0.1.1 Conv2d the reference builds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_stubs"))
    import loralib
    ynet = pkg("models.ynet")
    torch.manual_seed(5)
    a = loralib.Conv2d(14, 32, kernel_size=3, r=2, stride=1, padding=1)
    torch.manual_seed(5)
    b = ynet.LoRAConv2d(14, 32, kernel_size=3, r=2, stride=1, padding=1)
    assert [n for n, _ in a.named_parameters()] == [n for n, _ in b.named_parameters()]
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q), n
        assert p.requires_grad == q.requires_grad, n
    assert b.scaling == 0.5 and not b.weight.requires_grad and float(b.lora_B.abs().sum()) == 0.0


def test_model_init_consumes_rng_like_the_reference_layout():
    """Two product models built under one seed are identical and depend on the seed."""
    cfg = O.sdd_short(train_net="mosa_1", position=["0", "2"], enc=(8, 8, 16, 16, 16), dec=(16, 16, 16, 8, 8))
    torch.manual_seed(3)
    a = build_model(cfg)
    torch.manual_seed(3)
    b = build_model(cfg)
    torch.manual_seed(4)
    c = build_model(cfg)
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
    assert not all(torch.equal(p, q) for p, q in zip(a.parameters(), c.parameters()))
    assert hasattr(a.encoder.stages[0][0], "lora_A") and not hasattr(a.encoder.stages[1][1], "lora_A")


def test_lazy_cat_protocol():
    ops = pkg("ops")
    a, b, c = torch.rand(2, 3, 4, 5), torch.rand(2, 1, 4, 5), torch.rand(2, 2, 4, 5)
    lc = ops.lazy_cat([a, b])
    assert isinstance(lc, ops.LazyCat) and lc.shape == (2, 4, 4, 5) and lc.dim() == 4 and lc.size(1) == 4
    nested = torch.cat([lc, c], dim=1)                     # stays lazy through torch.cat(dim=1)
    assert isinstance(nested, ops.LazyCat) and [p.shape[1] for p in nested.parts] == [3, 1, 2]
    assert torch.equal(nested.materialize(), torch.cat([a, b, c], 1))
    assert torch.equal(torch.flatten(lc, 2), torch.cat([a, b], 1).flatten(2))     # any other op materialises
    assert ops.lazy_cat([a]) is a
    with pytest.raises(ValueError):
        ops.lazy_cat([a, torch.rand(2, 1, 4, 6)])


def test_templates_and_sampling_match_oracle():
    iu = pkg("utils.image_utils")
    for S in (210, 1050):
        assert torch.equal(torch.Tensor(iu.create_dist_mat(size=S)), O.dist_template(S))
        assert torch.equal(torch.Tensor(iu.create_gaussian_heatmap_template(size=S, kernlen=31, nsig=4, normalize=False)),
                           O.gaussian_template(S, 31, 4))
    t = iu.create_gaussian_heatmap_template(size=101, kernlen=31, nsig=4, normalize=True)
    assert t.max() == 1.0 and abs(iu.gkern(31, 4).sum() - 1.0) < 1e-12
    prob = torch.rand(3, 2, 16, 24)
    torch.manual_seed(1)
    a = iu.sampling(prob, 7)
    torch.manual_seed(1)
    b = O.sample_coords(prob, 7)
    assert a.shape == (3, 2, 7, 2) and torch.equal(a, b)
    assert float(a[..., 0].max()) < 24 and float(a[..., 1].max()) < 16
    x = torch.arange(2 * 3 * 2 * 2, dtype=torch.float32).view(2, 3, 2, 2)
    y = iu.swap_pavement_terrain(x.clone())
    assert torch.equal(y[:, 1], x[:, 2]) and torch.equal(y[:, 2], x[:, 1]) and torch.equal(y[:, 0], x[:, 0])
    with pytest.raises(ImportError):
        iu.resize({}, 0.25)                      # cv2.resize(INTER_AREA) of RGB images: out of scope (nothing here can pin it)
    with pytest.raises(RuntimeError, match="HIP devices only"):
        iu.resize({"s": torch.zeros(8, 8, dtype=torch.int64)}, 0.25, seg_mask=True)      # label maps: a device op, no CPU fallback
    with pytest.raises(ImportError):
        iu.preprocess_image_for_segmentation({}, seg_mask=False)      # the RGB branch needs segmentation_models_pytorch


def test_scene_dataset_contract():
    dl = pkg("utils.dataloader")
    rows = []
    for scene, n in (("a", 3), ("b", 2)):
        for m in range(n):
            for t in range(20):
                rows.append(dict(sceneId=scene, metaId=f"{scene}{m}", x=float(4 * t + m), y=float(8 * t)))
    ds = dl.SceneDataset(pd.DataFrame(rows), resize=0.25, total_len=20)
    assert len(ds) == 2
    traj, meta, scene = dl.scene_collate([ds[0]])
    assert scene == "a" and traj.shape == (3, 20, 2) and traj.dtype == torch.float32
    assert float(traj[1, 2, 0]) == (4 * 2 + 1) * 0.25 and len(meta[0]) == 60


def test_trainer_checkpoint_format_cpu(tmp_path):
    """save_params: full state dict (minus segmentation) for train/all; only requires_grad Parameters otherwise."""
    trn = pkg("models.trainer")
    g = Golden("tiny_short_mosa1")
    cfg = g.cfg()
    params = dict(obs_len=8, pred_len=12, segmentation_model_fp=None, use_features_only=False, n_semantic_classes=6,
                  encoder_channels=list(cfg.enc), decoder_channels=list(cfg.dec), waypoints=[11],
                  train_net=cfg.train_net, position=list(cfg.position), network="original", n_fusion=None,
                  resize_factor=0.25)
    with contextlib.redirect_stdout(io.StringIO()):
        t = trn.YNetTrainer(params, device=torch.device("cpu"))
    assert t.template_size == 1050 and t.division_factor == 32
    t.model.load_state_dict(g.state_dict(), strict=True)
    trn.apply_freeze_policy(t.model, cfg.train_net, cfg.position, "original")
    t.save_params(str(tmp_path / "delta.pt"), cfg.train_net)
    ck = torch.load(tmp_path / "delta.pt", weights_only=False)
    assert list(ck.keys()) == [str(k) for k in g.z["step/ckpt_keys"]]
    assert all(isinstance(v, torch.nn.Parameter) for v in ck.values())
    t.save_params(str(tmp_path / "full.pt"), "train")
    full = torch.load(tmp_path / "full.pt", weights_only=False)
    assert list(full.keys()) == list(g.state_dict().keys())
    with contextlib.redirect_stdout(io.StringIO()):
        t2 = trn.YNetTrainer(params, device=torch.device("cpu"))
        t2.load_separated_params(str(tmp_path / "full.pt"), str(tmp_path / "delta.pt"))
    for (n, p), (_, q) in zip(t.model.named_parameters(), t2.model.named_parameters()):
        assert torch.equal(p, q), n
    with pytest.raises(ImportError):
        t.prepare_data(pd.DataFrame(), "some/dir", "sdd", "train", 8, 12, 0.25, False)
    with pytest.raises(ValueError):
        t.prepare_data(pd.DataFrame(), {}, "kitti", "train", 8, 12, 0.25, False)
    # pre-processed planes must arrive padded: their border value depends on the pre-processing (class 0 for one-hot planes, (0 - mean) / std
    # for RGB), so a zero border would be silently wrong; raw label maps [H, W] are padded and encoded on the device instead
    with pytest.raises(ValueError, match="pre-processed planes"):
        t.prepare_data(pd.DataFrame(), {"s": torch.zeros(6, 40, 64)}, "sdd", "train", 8, 12, 0.25, False)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    L = pkg("_lib")
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        L.load()


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference tree (build container only)")
def test_checkpoints_round_trip_through_the_reference():
    """oracle/check_ckpt_roundtrip.py: product-written full / delta checkpoints load into the REFERENCE's YNetTrainer
    (load_params / load_separated_params, models/trainer.py:586-614) and reference-written ones into the product,
    every tensor bit-identical (six train_net modes)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "check_ckpt_roundtrip.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all cases identical in both directions" in r.stdout


def test_step_graph_keys():
    """utils/step_graph.py host logic (no GPU): the optimizer part of a step's key ignores the flags the capture flips
    itself (fused / foreach / capturable) but follows the learning rate; the model token follows the freeze pattern and
    in-place changes of FROZEN weights only (trainable ones change every step)."""
    sg = pkg("utils.step_graph")
    g = Golden("tiny_short_mosa1")
    model = build_model(g.cfg(), g.state_dict(), "cpu")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    h0 = sg._hyper(opt)
    for grp in opt.param_groups:
        grp["fused"], grp["foreach"], grp["capturable"] = True, False, True
    assert sg._hyper(opt) == h0
    opt.param_groups[0]["lr"] = 1e-4
    assert sg._hyper(opt) != h0
    t0 = sg.model_state_token(model)
    trainable = [p for p in model.parameters() if p.requires_grad]
    frozen = [p for p in model.parameters() if not p.requires_grad]
    with torch.no_grad():
        trainable[0].add_(1.0)
    assert sg.model_state_token(model) == t0
    with torch.no_grad():
        frozen[0].add_(1.0)                       # e.g. load_state_dict between two train() calls
    assert sg.model_state_token(model) != t0
    t1 = sg.model_state_token(model)
    frozen[1].requires_grad = True                # a different freeze policy
    assert sg.model_state_token(model) != t1
    # swapped storage keeps every version but not the addresses a graph holds -- both tokens must notice
    t2 = sg.model_state_token(model)
    ev = pkg("utils.evaluate")
    w0 = ev._weights_token(model)
    versions = [p._version for p in model.parameters()]
    frozen[2].data = frozen[2].data.clone()
    assert [p._version for p in model.parameters()] == versions
    assert sg.model_state_token(model) != t2 and ev._weights_token(model) != w0
    w1 = ev._weights_token(model)
    with torch.no_grad():
        trainable[0].add_(1.0)                     # the sweep's token follows trainable weights too (an optimizer step)
    assert ev._weights_token(model) != w1
    assert not sg.enabled(None, torch.device("cpu")) and not sg.enabled(True, "cpu")
    assert sg.waypoint_index("cpu", [14, 29]).tolist() == [14, 29]


def test_step_graph_cache_is_lru_and_drops_stale_model_states():
    """a dataset cycling through more step shapes than the cache holds must keep the recently used ones (a hit
    refreshes the entry), and entries captured for an older model state are dropped when a newer state shows up."""
    sg = pkg("utils.step_graph")
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
    c = sg.GraphCache(opt)
    c.MAX_ENTRIES = 4
    keys = [("shape", i, "tokA") for i in range(4)]
    first = [c.lookup(k) for k in keys]
    assert c.lookup(keys[0]) is first[0]                 # hit: refreshed, now the youngest
    c.lookup(("shape", 4, "tokA"))                       # evicts the least recently used = keys[1]
    assert keys[1] not in c.entries and keys[0] in c.entries and len(c.entries) == 4
    assert c.lookup(keys[0]) is first[0]
    e_new = c.lookup(("shape", 0, "tokB"))               # a new model state: everything captured for tokA can never hit again
    assert list(c.entries) == [("shape", 0, "tokB")] and c.lookup(("shape", 0, "tokB")) is e_new


def test_fused_criterion_refuses_a_foreign_target():
    """logits returned by the fused predictor + criterion kernel are not differentiable; comparing them with
    another target (or under another upstream gradient) must raise instead of silently dropping the decoder's gradient."""
    trn = pkg("models.trainer")
    crit = trn.HipBCEWithLogitsLoss()
    crit.expected_grad = 1000.0
    maps, target = torch.zeros(1, 2, 4, 4), torch.ones(1, 2, 4, 4)
    maps._ynet_fused_bce = (target, torch.tensor(3.0), 1000.0)
    assert float(crit(maps, target)) == 3.0
    assert float(crit(maps, target.view(1, 2, 4, 4))) == 3.0          # a re-viewed alias of the same memory is the same target
    with pytest.raises(RuntimeError, match="different target"):
        crit(maps, target.clone())
    crit.expected_grad = 1.0
    with pytest.raises(RuntimeError, match="expected_grad"):
        crit(maps, target)


def test_pad_numpy_branch_matches_copy_make_border_semantics():
    """utils/image_utils.py:95-107 on NumPy arrays (what cv2.imread hands the reference): bottom / right zero border up to
    the division factor, 2-D label maps and H x W x C images alike, already-aligned images untouched."""
    iu = pkg("utils.image_utils")
    rgb = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    images = {"a": rgb.copy(), "b": np.ones((64, 32), dtype=np.uint8)}
    iu.pad(images, division_factor=32)
    assert images["a"].shape == (32, 32, 3) and np.array_equal(images["a"][:5, :7], rgb)
    assert images["a"][5:].sum() == 0 and images["a"][:, 7:].sum() == 0 and images["a"].dtype == np.uint8
    assert images["b"].shape == (64, 32) and images["b"].sum() == 64 * 32


def test_adam_kernel_tables_refuse_what_the_kernel_does_not_cover():
    """utils/step_graph._AdamTables.prepare: only a plain Adam / AdamW over contiguous fp32 DEVICE parameters without step hooks is
    taken over by ynet_adam_step inside a captured step; everything else keeps optimizer.step()."""
    sg = pkg("utils.step_graph")
    p = torch.nn.Parameter(torch.zeros(4))
    assert sg._AdamTables.prepare(torch.optim.SGD([p], lr=0.1)) is None
    assert sg._AdamTables.prepare(torch.optim.Adam([p], lr=0.1)) is None                      # host parameter
    assert sg._AdamTables.prepare(torch.optim.Adam([p], lr=0.1, amsgrad=True)) is None
    opt = torch.optim.Adam([p], lr=0.1)
    opt.register_step_post_hook(lambda *a, **k: None)
    assert sg._AdamTables.prepare(opt) is None

    class MyAdam(torch.optim.Adam):
        pass
    assert sg._AdamTables.prepare(MyAdam([p], lr=0.1)) is None                               # a subclass may change the rule


def test_fold_context_switches_the_gradient_branch_on_and_off():
    ops = pkg("ops")
    assert ops.wgrad_branch is False and ops.premask is False and ops.skip_fold is False
    with ops.fold_skip_gradients():
        assert ops.wgrad_branch == (ops._wgrad_branch_allowed and ops.overlap_decoders)
        assert ops.premask == ops._premask_allowed
    assert ops.wgrad_branch is False and ops.premask is False and not ops._wgrad_pending
