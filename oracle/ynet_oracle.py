"""CPU oracle for the Y-Net (+MoSA/LoRA) forward/backward path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``motion-style-transfer_amd/`` may import this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the
checker / the timed CPU baseline — never as the shipped path.

It is a *functional* restatement (a state dict of plain tensors + stock ``torch.nn.functional`` ops on
CPU — the very ATen ops the reference dispatches) of these reference symbols:

  * models/ynet.py:134-151      get_conv2d / loralib.Conv2d      -> :func:`effective_weight`, :func:`conv`
  * models/ynet.py:170-234      YNetEncoder / YNetEncoderL       -> :func:`encoder`
  * models/ynet.py:286-395      YNetEncoderFusion (Y-Net-Mod)    -> :func:`encoder`
  * models/ynet.py:15-131       Adapter / AdapterBlock / AdapterLayer (serial, parallel) -> :func:`conv`, :func:`encoder`
  * models/ynet.py:154-167      Embedding (network='embed')      -> :func:`embedding`
  * models/ynet.py:237-283      YNetEncoderB                     -> :func:`encoder`
  * models/ynet.py:398-471      YNetDecoder                      -> :func:`decoder`
  * utils/softargmax.py:55-81   SoftArgmax2D.forward             -> :func:`softargmax2d`
  * utils/image_utils.py:7-63   gkern / templates / get_patch    -> :func:`gaussian_template`,
                                                                    :func:`dist_template`, :func:`crop_patches`
  * utils/image_utils.py:110-135 sampling                        -> :func:`sample_coords`
  * utils/train_epoch.py:44-126 one training step                -> :func:`train_step`
  * utils/evaluate.py:109-291   one evaluation batch              -> :func:`eval_batch`
  * utils/evaluate.py:134-161 + utils/kmeans.py:9-108  TTST (k-means of 10000 goal samples) -> :func:`ttst_goals`, :func:`kmeans_lloyd`
  * utils/evaluate.py:9-34, 172-224  CWS (Gaussian prior on the intermediate waypoints) -> :func:`cws_waypoints`, :func:`cws_gaussian`
  * models/trainer.py:116-195   freeze policy                    -> :func:`trainable_names`
  * torch.optim.Adam (trainer.py:197)                            -> :func:`adam_update`

Pinned against the reference itself: ``oracle/gen_goldens.py`` imports /root/reference in the build
container, runs both on identical seeded inputs, asserts agreement and writes ``tests/golden/*.npz``
(checked again by ``tests/test_oracle_golden.py`` without the reference).  The LoRA boundary is
"parity unpinned": loralib 0.1.1 is absent from the tree, see ``oracle/_stubs/loralib``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------
@dataclass
class Cfg:
    obs_len: int = 8
    pred_len: int = 12
    n_classes: int = 6
    enc: Sequence[int] = (32, 32, 64, 64, 64)
    dec: Sequence[int] = (64, 64, 64, 32, 32)
    waypoints: Sequence[int] = (11,)
    network: str = "original"          # 'original' | 'fusion'
    n_fusion: Optional[int] = None
    train_net: str = "train"
    position: Sequence[str] = field(default_factory=list)
    resize_factor: float = 0.25
    temperature: float = 1.0
    loss_scale: float = 1000.0
    kernlen: int = 31
    nsig: float = 4.0

    @property
    def n_wp(self) -> int:
        return len(self.waypoints)

    @property
    def rank(self) -> Optional[int]:
        # models/ynet.py:186-189
        if "mosa" not in self.train_net:
            return None
        parts = self.train_net.split("_")
        return int(parts[1]) if len(parts) > 1 else 1

    # ---- adapters (models/ynet.py:15-131, 237-283)
    @property
    def adapter_kind(self) -> Optional[str]:
        """'serial' | 'parallel' | None (mosa / plain training modes)."""
        if "mosa" in self.train_net:
            return None
        if "serial" in self.train_net:
            return "serial"
        if "parallel" in self.train_net:
            return "parallel"
        return None

    @property
    def adapter_in_layer(self) -> bool:
        # 'Layer' in train_net: the adapter lives inside the conv (AdapterLayer, YNetEncoderL);
        # otherwise AdapterBlocks sit between the stages (YNetEncoderB)
        return "Layer" in self.train_net

    @property
    def adapter_kernels(self) -> List[int]:
        """Kernel sizes of the parallel adapter convs: 'parallel_3x3' -> [3], 'parallelLayer_1x1_3x3' -> [1, 3],
        no size -> [1] (models/ynet.py:29,36)."""
        sizes = self.train_net.split("_")[1:]
        return [int(z.split("x")[0]) for z in sizes] if sizes else [1]

    @property
    def adapter_multiple(self) -> bool:
        return len(self.train_net.split("_")[1:]) >= 2

    @property
    def template_size(self) -> int:
        return int(4200 * self.resize_factor)  # models/trainer.py:61


def sdd_short(**kw) -> Cfg:
    return Cfg(obs_len=8, pred_len=12, waypoints=(11,), resize_factor=0.25, temperature=1.0, **kw)


def sdd_long(**kw) -> Cfg:
    return Cfg(obs_len=5, pred_len=30, waypoints=(14, 29), resize_factor=0.25, temperature=1.8, **kw)


def ind_long(**kw) -> Cfg:
    return Cfg(obs_len=5, pred_len=30, waypoints=(14, 29), resize_factor=0.33, temperature=1.8, **kw)


# ----------------------------------------------------------------------------------------------
# layer tables (state-dict contract)
# ----------------------------------------------------------------------------------------------
@dataclass
class ConvSpec:
    name: str      # state-dict prefix
    cin: int
    cout: int
    k: int
    layer: str     # the 'l' passed to get_conv2d ('' for decoder convs: never adapted)


def encoder_specs(cfg: Cfg) -> List[ConvSpec]:
    ch = list(cfg.enc)
    out: List[ConvSpec] = []
    if cfg.network == "fusion":
        nsep = len(ch) - cfg.n_fusion - 1
        for br, cin0 in (("scene", cfg.n_classes), ("motion", cfg.obs_len)):
            out.append(ConvSpec(f"encoder.{br}_stages.0.0", cin0, ch[0] // 2, 3, br))
            for i in range(nsep):
                out.append(ConvSpec(f"encoder.{br}_stages.{i + 1}.1", ch[i] // 2, ch[i + 1] // 2, 3, br))
                out.append(ConvSpec(f"encoder.{br}_stages.{i + 1}.3", ch[i + 1] // 2, ch[i + 1] // 2, 3, br))
        for j, i in enumerate(range(nsep, len(ch) - 1)):
            out.append(ConvSpec(f"encoder.fusion_stages.{j}.1", ch[i], ch[i + 1], 3, "fusion"))
            out.append(ConvSpec(f"encoder.fusion_stages.{j}.3", ch[i + 1], ch[i + 1], 3, "fusion"))
    else:
        out.append(ConvSpec("encoder.stages.0.0", cfg.n_classes + cfg.obs_len, ch[0], 3, "0"))
        for i in range(len(ch) - 1):
            out.append(ConvSpec(f"encoder.stages.{i + 1}.1", ch[i], ch[i + 1], 3, str(i + 1)))
            out.append(ConvSpec(f"encoder.stages.{i + 1}.3", ch[i + 1], ch[i + 1], 3, str(i + 1)))
    return out


def decoder_specs(cfg: Cfg, which: str) -> List[ConvSpec]:
    extra = cfg.n_wp if which == "traj_decoder" else 0
    enc = [c + extra for c in cfg.enc][::-1]
    dec = list(cfg.dec)
    center = enc[0]
    out = [ConvSpec(f"{which}.center.0", center, 2 * center, 3, ""),
           ConvSpec(f"{which}.center.2", 2 * center, 2 * center, 3, "")]
    up_in = [2 * center] + dec[:-1]
    up_out = [c // 2 for c in up_in]
    for i, (a, b) in enumerate(zip(up_in, up_out)):
        out.append(ConvSpec(f"{which}.upsample_conv.{i}", a, b, 3, ""))
    for i, (e, u, d) in enumerate(zip(enc, up_out, dec)):
        out.append(ConvSpec(f"{which}.decoder.{i}.0", e + u, d, 3, ""))
        out.append(ConvSpec(f"{which}.decoder.{i}.2", d, d, 3, ""))
    out.append(ConvSpec(f"{which}.predictor", dec[-1], cfg.pred_len, 1, ""))
    return out


def all_specs(cfg: Cfg) -> List[ConvSpec]:
    return (embedding_specs(cfg) + encoder_specs(cfg) + decoder_specs(cfg, "goal_decoder")
            + decoder_specs(cfg, "traj_decoder"))


def is_adapted(cfg: Cfg, spec: ConvSpec) -> bool:
    # models/ynet.py:139-144: lora conv iff 'mosa' in train_net and str(l) in position
    return cfg.rank is not None and spec.layer != "" and spec.layer in [str(p) for p in cfg.position]


BUFFER_SUFFIXES = (".running_mean", ".running_var", ".num_batches_tracked")


def is_buffer(name: str) -> bool:
    return name.endswith(BUFFER_SUFFIXES)


def has_layer_adapter(cfg: Cfg, spec: ConvSpec) -> bool:
    # models/ynet.py:145-148: AdapterLayer iff 'Layer' in train_net and str(l) in position
    return (cfg.adapter_kind is not None and cfg.adapter_in_layer and spec.layer != ""
            and spec.layer in [str(p) for p in cfg.position])


def embedding_specs(cfg: Cfg) -> List[ConvSpec]:
    out: List[ConvSpec] = []
    if cfg.network == "embed":      # models/ynet.py:154-167, 528-531
        for which, c in (("scene_embedding", cfg.n_classes), ("motion_embedding", cfg.obs_len)):
            out += [ConvSpec(f"{which}.conv.{j}", c, c, 3, "") for j in (0, 2, 4)]
    return out


def _adapter_entries(g, prefix: str, kind: str, cin: int, cout: int, kernels: Sequence[int], multiple: bool,
                     std: float) -> Dict[str, Tensor]:
    """Parameters (and BatchNorm buffers) of one Adapter (models/ynet.py:15-56 / 72-115).  The reference
    zero-initialises the adapter convs; ``std`` > 0 draws them from N(0, std) so that tests exercise them."""
    e: Dict[str, Tensor] = {}
    if kind == "serial":
        c = cout
        e[prefix + ".serial_layer.0.weight"] = 1.0 + 0.1 * torch.randn(c, generator=g) if std > 0 else torch.ones(c)
        e[prefix + ".serial_layer.0.bias"] = 0.1 * torch.randn(c, generator=g) if std > 0 else torch.zeros(c)
        e[prefix + ".serial_layer.0.running_mean"] = torch.zeros(c)
        e[prefix + ".serial_layer.0.running_var"] = torch.ones(c)
        e[prefix + ".serial_layer.0.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
        e[prefix + ".serial_layer.1.weight"] = torch.randn(c, c, 1, 1, generator=g) * std
    else:
        for j, k in enumerate(kernels):
            name = f"{prefix}.parallel_layer.{j}.weight" if multiple else f"{prefix}.parallel_layer.weight"
            e[name] = torch.randn(cout, cin, k, k, generator=g) * std
    return e


def make_state_dict(cfg: Cfg, seed: int = 0, lora_b_std: float = 0.0, adapter_std: float = 0.0) -> Dict[str, Tensor]:
    """Deterministic random-init weights (CPU generator): conv weight/bias ~ U(+-1/sqrt(fan_in)) like
    nn.Conv2d's default, lora_A ~ U(+-1/sqrt(fan_in)), lora_B ~ N(0, lora_b_std) (0 => identity), adapter
    convs ~ N(0, adapter_std) (0 => the reference's zero init).  Key order = the reference's registration order."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def plain(s: ConvSpec):
        bound = 1.0 / math.sqrt(s.cin * s.k * s.k)
        sd[s.name + ".weight"] = (torch.rand(s.cout, s.cin, s.k, s.k, generator=g) * 2 - 1) * bound
        sd[s.name + ".bias"] = (torch.rand(s.cout, generator=g) * 2 - 1) * bound

    for s in embedding_specs(cfg):
        plain(s)
    for s in encoder_specs(cfg):
        plain(s)
        if is_adapted(cfg, s):
            r = cfg.rank
            ba = 1.0 / math.sqrt(s.cin * s.k)
            sd[s.name + ".lora_A"] = (torch.rand(r * s.k, s.cin * s.k, generator=g) * 2 - 1) * ba
            sd[s.name + ".lora_B"] = torch.randn(s.cout * s.k, r * s.k, generator=g) * lora_b_std
        if has_layer_adapter(cfg, s):
            sd.update(_adapter_entries(g, s.name, cfg.adapter_kind, s.cin, s.cout, cfg.adapter_kernels,
                                       cfg.adapter_multiple, adapter_std))
    if cfg.adapter_kind is not None and not cfg.adapter_in_layer and cfg.network != "fusion":
        # YNetEncoderB: one AdapterBlock per position, after the stages (models/ynet.py:249-256)
        ch = list(cfg.enc)
        par_in = [cfg.n_classes + cfg.obs_len] + ch[:-1]
        for j, i in enumerate(int(p) for p in cfg.position):
            sd.update(_adapter_entries(g, f"encoder.adapters.{j}", cfg.adapter_kind, par_in[i], ch[i],
                                       cfg.adapter_kernels, cfg.adapter_multiple, adapter_std))
    for s in decoder_specs(cfg, "goal_decoder") + decoder_specs(cfg, "traj_decoder"):
        plain(s)
    return sd


def trainable_names(cfg: Cfg, sd: Dict[str, Tensor], ynet_bias: bool = False) -> List[str]:
    """Freeze policy of models/trainer.py:116-195 restricted to the Y-Net parameters."""
    tn, pos = cfg.train_net, [str(p) for p in cfg.position]
    names = [n for n in sd.keys() if not is_buffer(n)]
    enc = [n for n in names if n.startswith("encoder.")]
    if tn in ("all", "train"):
        out = names
    elif tn == "encoder" and not pos:
        out = enc
    elif tn == "encoder":
        out = [n for n in enc if n[len("encoder."):].split(".")[1] in pos]
    elif "serial" in tn:        # trainer.py:128-131 (checked before 'mosa')
        out = [n for n in enc if "serial" in n]
    elif "parallel" in tn:      # trainer.py:132-135
        out = [n for n in enc if "parallel" in n]
    elif "mosa" in tn:
        # loralib freezes the adapted conv's weight; trainer.py:137-139 enables names containing 'lora'
        out = [n for n in enc if "lora" in n]
    elif cfg.network == "fusion" and tn in ("scene", "motion", "fusion", "scene_fusion", "motion_fusion",
                                            "scene_motion", "scene_motion_fusion"):
        parts = tn.split("_")
        out = [n for n in enc if n.split(".")[1].replace("_stages", "") in parts]
    elif tn == "biasEncoder":
        out = [n for n in enc if "bias" in n]
    elif tn == "biasGoal":
        out = [n for n in names if n.startswith("goal_decoder.") and "bias" in n]
    elif tn == "biasTraj":
        out = [n for n in names if n.startswith("traj_decoder.") and "bias" in n]
    elif tn == "bias":
        out = [n for n in names if "bias" in n]
    else:
        raise NotImplementedError(tn)
    if ynet_bias and tn not in ("all", "train"):
        out = out + [n for n in names if "bias" in n and n not in out]
    return [n for n in names if n in set(out)]


# ----------------------------------------------------------------------------------------------
# network
# ----------------------------------------------------------------------------------------------
def effective_weight(sd: Dict[str, Tensor], name: str) -> Tensor:
    """loralib 0.1.1 Conv2d: W + (B @ A).view(W.shape) * (lora_alpha / r), lora_alpha = 1.
    The .view is a FLAT reshape of the [Cout*k, Cin*k] matrix."""
    w = sd[name + ".weight"]
    a = sd.get(name + ".lora_A")
    if a is None:
        return w
    b = sd[name + ".lora_B"]
    k = w.shape[-1]
    r = a.shape[0] // k
    return w + (b @ a).view(w.shape) * (1.0 / r)


def _batch_norm(sd, prefix: str, x: Tensor, training: bool) -> Tensor:
    """nn.BatchNorm2d defaults (momentum 0.1, eps 1e-5); in training mode the running statistics held in
    ``sd`` are updated in place, exactly like the module's buffers."""
    nb = sd.get(prefix + ".num_batches_tracked")
    if training and nb is not None:
        nb += 1
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training, 0.1, 1e-5)


def _adapter(sd, prefix: str, x_in: Tensor, x_out: Optional[Tensor], training: bool) -> Tensor:
    """Adapter branch WITHOUT the residual.  serial: conv1x1(BN(x_out)) (models/ynet.py:24-26,64-66); parallel:
    sum of KxK convs of x_in, no bias (27-39, 57-63)."""
    if prefix + ".serial_layer.1.weight" in sd:
        z = _batch_norm(sd, prefix + ".serial_layer.0", x_out, training)
        return F.conv2d(z, sd[prefix + ".serial_layer.1.weight"], None)
    if prefix + ".parallel_layer.weight" in sd:
        w = sd[prefix + ".parallel_layer.weight"]
        return F.conv2d(x_in, w, None, padding=w.shape[-1] // 2)
    y, j = 0, 0
    while f"{prefix}.parallel_layer.{j}.weight" in sd:
        w = sd[f"{prefix}.parallel_layer.{j}.weight"]
        y = y + F.conv2d(x_in, w, None, padding=w.shape[-1] // 2)
        j += 1
    return y


def _has_adapter(sd, prefix: str) -> bool:
    return any(k in sd for k in (prefix + ".serial_layer.1.weight", prefix + ".parallel_layer.weight",
                                 prefix + ".parallel_layer.0.weight"))


def conv(sd, name: str, x: Tensor, relu: bool, training: bool = True) -> Tensor:
    w = effective_weight(sd, name)
    y = F.conv2d(x, w, sd[name + ".bias"], stride=1, padding=w.shape[-1] // 2)
    if _has_adapter(sd, name):      # AdapterLayer.forward (models/ynet.py:117-131): branch + conv output, ReLU after
        y = _adapter(sd, name, x, y, training) + y
    return F.relu(y) if relu else y


def _stage(sd, prefix: str, x: Tensor, first: bool, training: bool = True) -> Tensor:
    if first:
        return conv(sd, prefix + ".0", x, True, training)
    x = F.max_pool2d(x, 2, 2)
    x = conv(sd, prefix + ".1", x, True, training)
    return conv(sd, prefix + ".3", x, True, training)


def embedding(sd, which: str, x: Tensor) -> Tensor:
    """models/ynet.py:154-167: three 3x3 conv + ReLU."""
    for j in (0, 2, 4):
        x = conv(sd, f"{which}.conv.{j}", x, True)
    return x


def encoder(sd, cfg: Cfg, scene: Tensor, motion: Tensor, training: bool = True) -> List[Tensor]:
    ch = list(cfg.enc)
    feats: List[Tensor] = []
    if cfg.network == "fusion":
        nsep = len(ch) - cfg.n_fusion - 1
        branches = []
        for br, x in (("scene", scene), ("motion", motion)):
            fs = []
            for i in range(nsep + 1):
                x = _stage(sd, f"encoder.{br}_stages.{i}", x, i == 0, training)
                fs.append(x)
            branches.append(fs)
        feats = [torch.cat([s, m], dim=1) for s, m in zip(*branches)]
        x = feats[-1]
        for j in range(cfg.n_fusion):
            x = _stage(sd, f"encoder.fusion_stages.{j}", x, False, training)
            feats.append(x)
        feats.append(F.max_pool2d(x, 2, 2))
    else:
        x = torch.cat([scene, motion], dim=1)
        blocks = cfg.adapter_kind if (cfg.adapter_kind is not None and not cfg.adapter_in_layer) else None
        pos = [int(p) for p in cfg.position] if blocks else []
        j = 0
        for i in range(len(ch)):
            # YNetEncoderB.forward (models/ynet.py:258-283): serial blocks transform the stage output; parallel
            # blocks see the stage's (pooled) input and are added after the stage's last ReLU
            if blocks == "parallel":
                src = x if i == 0 else F.max_pool2d(x, 2, 2)
                x = _stage(sd, f"encoder.stages.{i}", x, i == 0, training)
                if i in pos:
                    x = x + _adapter(sd, f"encoder.adapters.{j}", src, None, training)
                    j += 1
            else:
                x = _stage(sd, f"encoder.stages.{i}", x, i == 0, training)
                if blocks == "serial" and i in pos:
                    x = _adapter(sd, f"encoder.adapters.{j}", None, x, training) + x
                    j += 1
            feats.append(x)
        feats.append(F.max_pool2d(x, 2, 2))
    return feats


def decoder(sd, cfg: Cfg, which: str, feats: List[Tensor]) -> Tensor:
    fr = feats[::-1]
    x = conv(sd, f"{which}.center.0", fr[0], True)
    x = conv(sd, f"{which}.center.2", x, True)
    for i, skip in enumerate(fr[1:]):
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        x = conv(sd, f"{which}.upsample_conv.{i}", x, False)   # the only 3x3 conv without ReLU
        x = torch.cat([x, skip], dim=1)
        x = conv(sd, f"{which}.decoder.{i}.0", x, True)
        x = conv(sd, f"{which}.decoder.{i}.2", x, True)
    return conv(sd, f"{which}.predictor", x, False)


def waypoint_pyramid(wp_map: Tensor, n_levels: int) -> List[Tensor]:
    # utils/train_epoch.py:97-100
    return [wp_map] + [F.avg_pool2d(wp_map, 2 ** i, 2 ** i) for i in range(1, n_levels)]


def traj_inputs(feats: List[Tensor], wp_map: Tensor) -> List[Tensor]:
    return [torch.cat([f, g], dim=1) for f, g in zip(feats, waypoint_pyramid(wp_map, len(feats)))]


# ----------------------------------------------------------------------------------------------
# soft-argmax, loss, templates, patches, sampling
# ----------------------------------------------------------------------------------------------
def softargmax2d(x: Tensor, eps: float = 1e-6) -> Tensor:
    """[B,C,H,W] -> [B,C,2] in (x, y) order, unnormalised pixel coordinates."""
    if x.dim() != 4:
        raise ValueError(f"expected BxCxHxW, got {tuple(x.shape)}")
    b, c, h, w = x.shape
    flat = x.reshape(b, c, -1)
    e = torch.exp(flat - flat.max(dim=-1, keepdim=True)[0])
    inv = 1.0 / (e.sum(dim=-1, keepdim=True) + eps)
    ys = torch.arange(h, dtype=x.dtype).view(h, 1).expand(h, w).reshape(-1)
    xs = torch.arange(w, dtype=x.dtype).view(1, w).expand(h, w).reshape(-1)
    ey = ((ys * e) * inv).sum(dim=-1, keepdim=True)
    ex = ((xs * e) * inv).sum(dim=-1, keepdim=True)
    return torch.cat([ex, ey], dim=-1)


def bce_logits_mean(x: Tensor, t: Tensor) -> Tensor:
    return F.binary_cross_entropy_with_logits(x, t)


def gaussian_kernel(kernlen: int, nsig: float) -> np.ndarray:
    ax = np.linspace(-(kernlen - 1) / 2.0, (kernlen - 1) / 2.0, kernlen)
    sq = ax[None, :] ** 2 + ax[:, None] ** 2
    k = np.exp(-0.5 * sq / (nsig ** 2))
    return k / k.sum()


def gaussian_template(size: int, kernlen: int = 31, nsig: float = 4.0, normalize: bool = False) -> Tensor:
    """float64 NumPy, then cast to fp32 (trainer.py:210-211 passes normalize=False)."""
    t = np.zeros((size, size))
    k = gaussian_kernel(kernlen, nsig)
    lo = size // 2 - kernlen // 2
    hi = size // 2 + (kernlen + 1) // 2
    t[lo:hi, lo:hi] = k
    if normalize:
        t = t / t.max()
    return torch.from_numpy(t.astype(np.float32))


def dist_template(size: int, normalize: bool = True) -> Tensor:
    r = np.arange(size, dtype=np.int64) - size // 2
    d = np.sqrt((r[:, None] ** 2 + r[None, :] ** 2).astype(np.float64))
    if normalize:
        d = d / d.max() * 2
    return torch.from_numpy(d.astype(np.float32))


def round_coords(xy: np.ndarray):
    """np.round = round-half-to-even, as in utils/image_utils.py:52-53."""
    return np.round(xy[:, 0]).astype(np.int64), np.round(xy[:, 1]).astype(np.int64)


def crop_patches(template: Tensor, xy, H: int, W: int) -> Tensor:
    """[N,2] (x,y) -> [N,H,W]; window template[S/2-y : S/2+H-y, S/2-x : S/2+W-x]."""
    xy = np.asarray(xy, dtype=np.float32).reshape(-1, 2)
    xs, ys = round_coords(xy)
    cy, cx = template.shape[0] // 2, template.shape[1] // 2
    out = torch.empty(len(xs), H, W, dtype=template.dtype)
    for n, (x, y) in enumerate(zip(xs, ys)):
        out[n] = template[cy - y:cy - y + H, cx - x:cx - x + W]
    return out


def sample_coords(prob: Tensor, num_samples: int, generator=None, rel_threshold: Optional[float] = None,
                  replacement: bool = False) -> Tensor:
    """[B,C,H,W] -> [B,C,K,2] float (x,y): multinomial on the flat plane (utils/image_utils.py:110-135).
    ``rel_threshold`` zeroes entries below that fraction of the plane's maximum; the renormalisation divides by
    the sum over ALL planes (as the reference does: multinomial only needs per-row proportions)."""
    b, c, h, w = prob.shape
    pm = prob.reshape(b * c, -1)
    if DEVICE_SAMPLER:      # the product's documented sampler (one 62-bit seed per call, drawn like utils/image_utils.py:draw_seed)
        seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator).item())
        idx = device_multinomial(pm, num_samples, replacement, rel_threshold, seed)
        idx = idx.view(b, c, num_samples).float()
        return torch.stack([idx % w, torch.floor(idx / w)], dim=-1)
    if rel_threshold is not None:
        mask = pm < pm.max(dim=1)[0].unsqueeze(1) * rel_threshold
        pm = pm * (~mask).int()
        pm = pm / pm.sum()
    idx = torch.multinomial(pm, num_samples, replacement=replacement, generator=generator)
    idx = idx.view(b, c, num_samples).float()
    return torch.stack([idx % w, torch.floor(idx / w)], dim=-1)


# ----------------------------------------------------------------------------------------------
# The product's device sampler, restated (include/ynet_hip.h: ynet_multinomial).  The REFERENCE samples with
# torch.multinomial, whose stream differs per device and build; the product documents its own counter-based generator
# so that this CPU restatement reproduces every draw from the seed alone -- that is what makes an un-forced sweep
# checkable.  Philox4x32-10, key = (seed low, seed high), counter = (element / sample, 0, row, stream).
# ----------------------------------------------------------------------------------------------
def _philox4x32_10(c0, c1, c2, c3, k0, k1):
    M = np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & M for c in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M
        n1 = p1 & M
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M
        n3 = p0 & M
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M
        k1 = (k1 + np.uint64(0xBB67AE85)) & M
    return c0, c1


def _philox_uniform(elem, row: int, stream: int, seed: int) -> np.ndarray:
    elem = np.asarray(elem, dtype=np.uint64)
    x0, x1 = _philox4x32_10(elem, np.zeros_like(elem), np.full_like(elem, row), np.full_like(elem, stream),
                            seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return ((x0 >> np.uint64(5)).astype(np.float64) * 67108864.0 + (x1 >> np.uint64(6)).astype(np.float64) + 0.5) / 9007199254740992.0


def device_multinomial(prob: Tensor, num_samples: int, replacement: bool = False, rel_threshold: Optional[float] = None,
                       seed: int = 0) -> Tensor:
    """ynet_multinomial on the CPU: prob [rows, n] -> int64 [rows, num_samples]."""
    P = prob.detach().cpu().numpy().astype(np.float32)
    rows, n = P.shape
    out = np.zeros((rows, num_samples), dtype=np.int64)
    for r in range(rows):
        p = P[r].copy()
        if rel_threshold:
            cut = np.float32(p.max()) * np.float32(rel_threshold)
            p[p < cut] = 0.0
        p[~(p > 0)] = 0.0
        p64 = p.astype(np.float64)
        if not replacement:
            u = _philox_uniform(np.arange(n), r, 0, seed)
            with np.errstate(divide="ignore"):
                key = np.where(p64 > 0, p64 / -np.log(u), -1.0)
            order = np.lexsort((np.arange(n), -key))          # descending key, ties to the smaller index
            if key[order[num_samples - 1]] <= 0:
                raise RuntimeError("invalid multinomial distribution (too few positive entries)")
            out[r] = order[:num_samples]
        else:
            seg = (n + 255) // 256
            m = np.zeros(256 * seg, dtype=np.float64)
            m[:n] = p64
            m = m.reshape(256, seg)
            totals = np.cumsum(m, axis=1)[:, -1]              # sequential sums (np.cumsum accumulates in order)
            seg_off = np.concatenate([[0.0], np.cumsum(totals)])
            total = seg_off[256]
            if not total > 0:
                raise RuntimeError("invalid multinomial distribution (no positive entry)")
            target = _philox_uniform(np.arange(num_samples), r, 1, seed) * total
            t = np.minimum(np.searchsorted(seg_off[1:], target, side="left"), 255)
            run = np.cumsum(np.concatenate([seg_off[:256, None], m], axis=1), axis=1)[:, 1:]     # seg_off[t] + p0 + p1 + ...
            for j in range(num_samples):
                tj = int(t[j])
                i = int(np.searchsorted(run[tj], target[j], side="left"))
                if i >= seg or tj * seg + i >= n:
                    pos = np.nonzero(m[tj] > 0)[0]
                    i = int(pos[-1]) if len(pos) else -tj * seg
                out[r, j] = tj * seg + i
    return torch.from_numpy(out)


DEVICE_SAMPLER = False      # True: sample_coords draws with device_multinomial (one seed per call from the torch generator)


def kmeans_lloyd(X: Tensor, k: int, tol: float = 1e-3, iter_limit: int = 1000) -> Tensor:
    """Lloyd's algorithm as utils/kmeans.py:22-108 runs it for TTST: centres initialised with k distinct points
    drawn by ``np.random.choice`` (global NumPy RNG), squared-Euclidean assignment (first minimum wins), an empty
    cluster re-seeded with one point drawn by ``torch.randint`` (global torch RNG), stop when
    (sum_k ||c_k - c_k_prev||)^2 < tol or after ``iter_limit`` iterations.  Returns the [k, 2] centres."""
    X = X.float()
    c = X[np.random.choice(len(X), k, replace=False)]
    it = 0
    while True:
        d = ((X.unsqueeze(1) - c.unsqueeze(0)) ** 2.0).sum(dim=-1).squeeze()
        assign = torch.argmin(d, dim=1)
        prev = c.clone()
        for j in range(k):
            sel = X[assign == j]
            if sel.shape[0] == 0:
                sel = X[torch.randint(len(X), (1,))]
            c[j] = sel.mean(dim=0)
        shift = torch.sum(torch.sqrt(torch.sum((c - prev) ** 2, dim=1)))
        it += 1
        if shift ** 2 < tol:
            break
        if iter_limit != 0 and it >= iter_limit:
            break
    return c


def ttst_goals(wp_sig_last: Tensor, wp_logits_last: Tensor, n_goal: int, rel_thresh: float, generator=None,
               samples: Optional[Tensor] = None) -> Tensor:
    """Test-time sampling trick (utils/evaluate.py:134-161): 10000 thresholded goal samples per person (with
    replacement), clustered into n_goal - 1 centres; the first goal is the soft-argmax of the logits.
    [B,1,H,W] x2 -> [n_goal, B, 1, 2].  ``samples`` [10000, B, 1, 2] overrides the draw (tests)."""
    if samples is None:
        samples = sample_coords(wp_sig_last, 10000, generator, rel_threshold=rel_thresh, replacement=True).permute(2, 0, 1, 3)
    first = softargmax2d(wp_logits_last)                       # [B,1,2]
    centres = [kmeans_lloyd(samples[:, person, 0], n_goal - 1) for person in range(samples.shape[1])]
    goals = torch.stack(centres).permute(1, 0, 2).unsqueeze(2)
    return torch.cat([first.unsqueeze(0), goals], dim=0)


def cws_gaussian(mean_xy: Tensor, H: int, W: int, dist: Tensor, sigma_factor: float, ratio: float, rot: bool = False) -> Tensor:
    """utils/evaluate.py:9-34: anisotropic Gaussian centred at ``mean_xy``, long axis along ``dist`` (the vector from
    the goal to the last observed position), std = (|dist| + 5) / sigma_factor along it and that / ratio across;
    sampled on linspace(0, H, H) x linspace(0, W, W) and normalised to sum 1."""
    ax = torch.linspace(0, H, H) - mean_xy[1]
    ay = torch.linspace(0, W, W) - mean_xy[0]
    xx, yy = torch.meshgrid([ax, ay], indexing="ij")
    mesh = torch.stack([yy, xx], dim=-1)
    rad = torch.atan2(dist[0], dist[1])
    c, sn = torch.cos(rad), torch.sin(rad)
    R = torch.Tensor([[c, sn], [-sn, c]])
    if rot:
        R = torch.matmul(torch.Tensor([[0, -1], [1, 0]]), R)
    norm = dist.square().sum(-1).sqrt() + 5
    cov = torch.square(torch.Tensor([[norm / sigma_factor / ratio, 0], [0, norm / sigma_factor]]))
    T = torch.matmul(torch.matmul(R, cov), R.T)
    k = torch.exp(-0.5 * (torch.matmul(mesh, torch.inverse(T)) * mesh).sum(-1))
    return k / k.sum()


def softargmax_on_map(p: Tensor) -> Tensor:
    """models/ynet.py:588-600: expectation of (x, y) under maps that already sum to 1.  [B,C,H,W] -> [B,C,2]."""
    b, c, h, w = p.shape
    ys = torch.arange(h, dtype=p.dtype).view(h, 1).expand(h, w).reshape(-1)
    xs = torch.arange(w, dtype=p.dtype).view(1, w).expand(h, w).reshape(-1)
    flat = p.flatten(2)
    return torch.cat([(xs * flat).sum(-1, keepdim=True), (ys * flat).sum(-1, keepdim=True)], dim=-1)


def cws_waypoints(wp_sig: Tensor, goals: Tensor, last_obs: Tensor, n_goal: int, n_traj: int, sigma_factor: float,
                  ratio: float, rot: bool, generator=None) -> Tensor:
    """Conditioned waypoint sampling (utils/evaluate.py:172-224).  wp_sig [B,nwp,H,W] (sigmoid maps), goals
    [n_goal,B,1,2], last_obs [B,2] -> [n_goal*n_traj, B, nwp, 2].  Waypoints are drawn backwards from the goal: the
    map of waypoint w is multiplied by a Gaussian centred at goal + (last_obs - goal) / (w + 2); the first n_goal
    trajectories take its expectation, the others one thresholded sample."""
    b, nwp, H, W = wp_sig.shape
    goals = goals.repeat(n_traj, 1, 1, 1)
    out = []
    for g_num, wp in enumerate(goals.squeeze(2)):
        chain = [wp]
        for w in reversed(range(nwp - 1)):
            distance = last_obs - wp
            traj_idx = g_num // n_goal
            maps = torch.stack([cws_gaussian(coord + d * (1 / (w + 2)), H, W, d, sigma_factor - traj_idx, ratio, rot)
                                for d, coord in zip(distance, wp)])
            m = wp_sig[:, w] * maps
            m = (m.flatten(1) / m.flatten(1).sum(-1, keepdim=True)).view_as(m)
            if traj_idx == 0:
                wp = softargmax_on_map(m.unsqueeze(0)).squeeze(0)
            else:
                wp = sample_coords(m.unsqueeze(1), 1, generator, rel_threshold=0.05).permute(2, 0, 1, 3).squeeze(2).squeeze(0)
            chain.append(wp)
        out.append(torch.stack(chain[::-1]).permute(1, 0, 2))
    return torch.stack(out)


def displacement_error(gt: Tensor, pred: Tensor, resize: float) -> Tensor:
    return ((((gt - pred) / resize) ** 2).sum(dim=-1) ** 0.5)


# ----------------------------------------------------------------------------------------------
# one training step / one evaluation batch
# ----------------------------------------------------------------------------------------------
def build_maps(cfg: Cfg, traj: Tensor, H: int, W: int, in_tmpl: Tensor, gt_tmpl: Optional[Tensor]):
    b = traj.shape[0]
    obs = traj[:, :cfg.obs_len].reshape(-1, 2).numpy()
    observed = crop_patches(in_tmpl, obs, H, W).reshape(b, cfg.obs_len, H, W)
    fut = traj[:, cfg.obs_len:]
    gt_map = wp_map = None
    if gt_tmpl is not None:
        gt_map = crop_patches(gt_tmpl, fut.reshape(-1, 2).numpy(), H, W).reshape(b, cfg.pred_len, H, W)
        wps = fut[:, list(cfg.waypoints)]
        wp_map = crop_patches(in_tmpl, wps.reshape(-1, 2).numpy(), H, W).reshape(b, cfg.n_wp, H, W)
    return observed, gt_map, wp_map


def train_step(sd: Dict[str, Tensor], cfg: Cfg, scene: Tensor, traj: Tensor, in_tmpl: Tensor,
               gt_tmpl: Tensor, trainable: Sequence[str], loss_weight: float = 1.0,
               keep_maps: bool = False) -> Dict[str, object]:
    """utils/train_epoch.py:54-126 for ONE batch (no optimizer step).  ``scene`` is [1,C,H,W].
    ``loss_weight`` scales the loss (B_local/B_global in data-parallel runs)."""
    H, W = scene.shape[-2:]
    b = traj.shape[0]
    params = {k: (v.detach().clone().requires_grad_(k in set(trainable)) if v.is_floating_point() else v.clone())
              for k, v in sd.items()}
    observed, gt_map, wp_map = build_maps(cfg, traj, H, W, in_tmpl, gt_tmpl)
    sem1 = scene
    if cfg.network == "embed":      # utils/train_epoch.py:80-83 (before the expand)
        sem1 = embedding(params, "scene_embedding", scene)
        observed = embedding(params, "motion_embedding", observed)
    sem = sem1.expand(b, -1, -1, -1)
    feats = encoder(params, cfg, sem, observed, training=True)
    goal_map = decoder(params, cfg, "goal_decoder", feats)
    goal_loss = bce_logits_mean(goal_map, gt_map) * cfg.loss_scale
    traj_map = decoder(params, cfg, "traj_decoder", traj_inputs(feats, wp_map))
    traj_loss = bce_logits_mean(traj_map, gt_map) * cfg.loss_scale
    loss = goal_loss + traj_loss
    names = [n for n in sd if n in set(trainable)]
    grads = torch.autograd.grad(loss * loss_weight, [params[n] for n in names], allow_unused=True)
    with torch.no_grad():
        fut = traj[:, cfg.obs_len:]
        pred_traj = softargmax2d(traj_map)
        pred_goal = softargmax2d(goal_map[:, -1:])
        ade = displacement_error(fut, pred_traj, cfg.resize_factor).mean(dim=1)
        fde = displacement_error(fut[:, -1:], pred_goal[:, -1:], cfg.resize_factor).mean(dim=1)
    out = {"loss": loss.detach(), "goal_loss": goal_loss.detach(), "traj_loss": traj_loss.detach(),
           "grads": {n: (g if g is not None else torch.zeros_like(sd[n])) for n, g in zip(names, grads)},
           "ade": ade, "fde": fde, "pred_traj": pred_traj, "pred_goal": pred_goal}
    out["buffers"] = {k: v.detach() for k, v in params.items() if is_buffer(k)}     # BatchNorm statistics after the step
    if keep_maps:
        out.update(features=[f.detach() for f in feats], goal_map=goal_map.detach(),
                   traj_map=traj_map.detach(), observed=observed, gt_map=gt_map, wp_map=wp_map)
    return out


def adam_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
                b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """torch.optim.Adam defaults (no weight decay, no amsgrad); returns (p, m, v)."""
    m = m * b1 + g * (1 - b1)
    v = v * b2 + g * g * (1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


@torch.no_grad()
def eval_batch(sd, cfg: Cfg, scene: Tensor, traj: Tensor, in_tmpl: Tensor, n_goal: int = 20,
               n_traj: int = 1, waypoint_samples: Optional[Tensor] = None, generator=None, use_ttst: bool = False,
               use_cws: bool = False, cws_params: Optional[dict] = None, rel_thresh: float = 0.002,
               ttst_samples: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """utils/evaluate.py:109-291 for ONE batch.  ``waypoint_samples`` [K,B,nwp,2] teacher-forces the sampled
    goals/waypoints; ``ttst_samples`` [10000,B,1,2] the TTST draw."""
    H, W = scene.shape[-2:]
    b = traj.shape[0]
    observed, _, _ = build_maps(cfg, traj, H, W, in_tmpl, None)
    fut = traj[:, cfg.obs_len:]
    sem1 = scene
    if cfg.network == "embed":      # utils/evaluate.py:98-100, 119-121
        sem1 = embedding(sd, "scene_embedding", scene)
        observed = embedding(sd, "motion_embedding", observed)
    feats = encoder(sd, cfg, sem1.expand(b, -1, -1, -1), observed, training=False)
    goal_map = decoder(sd, cfg, "goal_decoder", feats)
    wp_logits = goal_map[:, list(cfg.waypoints)]
    wp_sig = torch.sigmoid(wp_logits / cfg.temperature)
    if waypoint_samples is None:
        if use_ttst:
            goals = ttst_goals(wp_sig[:, -1:], wp_logits[:, -1:], n_goal, rel_thresh, generator, ttst_samples)
        else:
            goals = sample_coords(wp_sig[:, -1:], n_goal, generator).permute(2, 0, 1, 3)
        if use_cws and cfg.n_wp > 1:
            waypoint_samples = cws_waypoints(wp_sig, goals, traj[:, cfg.obs_len - 1], n_goal, n_traj,
                                             cws_params["sigma_factor"], cws_params["ratio"], cws_params["rot"], generator)
        elif cfg.n_wp > 1:
            wps = sample_coords(wp_sig[:, :-1], n_goal * n_traj, generator).permute(2, 0, 1, 3)
            waypoint_samples = torch.cat([wps, goals.repeat(n_traj, 1, 1, 1)], dim=2)
        else:
            waypoint_samples = goals
    trajs = []
    for wp in waypoint_samples:
        wp_map = crop_patches(in_tmpl, wp.reshape(-1, 2).numpy(), H, W).reshape(b, cfg.n_wp, H, W)
        tmap = decoder(sd, cfg, "traj_decoder", traj_inputs(feats, wp_map))
        trajs.append(softargmax2d(tmap))
    trajs = torch.stack(trajs)
    ade_k = displacement_error(fut, trajs, cfg.resize_factor).mean(dim=2)             # [K,B]
    fde_k = displacement_error(fut[:, -1:], waypoint_samples[:, :, -1:], cfg.resize_factor)  # [K,B,1]
    return {"goal_map": goal_map, "wp_sigmoid": wp_sig, "waypoint_samples": waypoint_samples,
            "trajs": trajs, "ade_k": ade_k, "fde_k": fde_k[:, :, 0],
            "ade": ade_k.min(dim=0)[0], "fde": fde_k.min(dim=0)[0][:, 0], "features": feats}


# ----------------------------------------------------------------------------------------------
# synthetic workload (SURVEY 8d / BASELINE.md section 3)
# ----------------------------------------------------------------------------------------------
def synthetic_scene(cfg: Cfg, H: int, W: int, seed: int = 0) -> Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.softmax(torch.randn(cfg.n_classes, H, W, generator=g), dim=0).unsqueeze(0)


def synthetic_trajectories(cfg: Cfg, n: int, H: int, W: int, seed: int = 0) -> Tensor:
    g = torch.Generator().manual_seed(seed + 1)
    t = cfg.obs_len + cfg.pred_len
    start = torch.rand(n, 1, 2, generator=g) * torch.tensor([0.4 * W, 0.4 * H]) + torch.tensor([0.3 * W, 0.3 * H])
    steps = torch.randn(n, t, 2, generator=g) * 2.0
    steps[:, 0] = 0
    return (start + steps.cumsum(dim=1)).float()
