"""Y-Net on MI355X: host-side mirror of the reference's models/ynet.py.

Same public classes, constructor signatures, attribute tree and state-dict keys as the reference
(``encoder.stages.{i}.{j}.{weight,bias,lora_A,lora_B}``, ``{goal,traj}_decoder.{center,
upsample_conv,decoder,predictor}...`` — SURVEY.md A.2), so reference checkpoints load unchanged.
The arithmetic runs in the hand-written gfx950 kernels behind ``..ops``:

  * nn.Conv2d + nn.ReLU            -> one fused MFMA implicit-GEMM launch (ops.conv2d)
  * loralib.Conv2d (MoSA)          -> MFMA compose of W + BA*s, then the same conv (LoRAConv2d)
  * torch.cat of skip / waypoint   -> never materialised; the conv reads its parts (ops.LazyCat)
  * MaxPool2d, bilinear x2, soft-argmax, sigmoid -> dedicated HBM-bound kernels

  * serial / parallel adapters (AdapterBlock, AdapterLayer), Embedding -> the same conv kernels (1x1, 3x3, 5x5,
    no bias); BatchNorm2d, the residual adds and the ReLU behind them on ynet_batchnorm2d_* / ynet_add_relu (round 6;
    variants outside the BASELINE configs)

Reference: models/ynet.py:15-131 (Adapter, AdapterBlock, AdapterLayer), 134-151 (get_conv2d), 154-167
(Embedding), 170-283 (YNetEncoder/L/B), 286-395 (YNetEncoderFusion), 398-471 (YNetDecoder), 474-600 (YNet).
"""
import math
import os

import torch
import torch.nn as nn

from .. import ops
from ..utils.softargmax import SoftArgmax2D, create_meshgrid

# YNET_FUSED_READOUT=0: evaluate()'s trajectory passes run predictor and soft-argmax as two launches (A/B runs)
FUSED_READOUT = os.environ.get("YNET_FUSED_READOUT", "1") != "0"
# YNET_SHARED_SCENE=0: the fusion encoder's scene branch convolves every copy of a batch-broadcast scene (as the reference does)
SHARED_SCENE_BRANCH = os.environ.get("YNET_SHARED_SCENE", "1") != "0"


class HipConv2d(nn.Conv2d):
    """nn.Conv2d (stride 1, 'same' padding, K in {1,3,5}) executed by ynet_conv2d."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=None, bias=True):
        if isinstance(kernel_size, (tuple, list)):
            if kernel_size[0] != kernel_size[1]:
                raise ValueError("only square kernels are supported")
            kernel_size = kernel_size[0]
        if isinstance(stride, (tuple, list)):
            stride = stride[0]
        if isinstance(padding, (tuple, list)):
            padding = padding[0]
        if padding is None:
            padding = kernel_size // 2
        if stride != 1 or padding != kernel_size // 2 or kernel_size not in (1, 3, 5):
            raise ValueError(f"HipConv2d supports stride 1, padding k//2, k in (1,3,5); got k={kernel_size} "
                             f"stride={stride} padding={padding}")
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)
        self._packed = {}

    def forward(self, x, relu=False, pool=False, bits=False, defer=False):
        """defer: the caller hands the (post-ReLU) result to ops.pred_bce next -- where the fused launch applies the convolution runs inside it (ops.conv2d)."""
        return ops.conv2d(x, self.weight, self.bias, relu, self._packed, pool=pool, bits=bits, defer=defer)


class LoRAConv2d(HipConv2d):
    """loralib==0.1.1 ``Conv2d`` semantics (parameters weight, bias, lora_A, lora_B on the conv
    itself; scaling = lora_alpha / r; base weight frozen).  Restated from the published algorithm:
    loralib is not part of the reference tree (requirements.txt:11) -> parity unpinned."""

    def __init__(self, in_channels, out_channels, kernel_size, r=0, lora_alpha=1, lora_dropout=0.0,
                 merge_weights=True, **kwargs):
        self._lora_ready = False
        super().__init__(in_channels, out_channels, kernel_size, **kwargs)
        assert type(kernel_size) is int
        self.r, self.lora_alpha, self.merged, self.merge_weights = r, lora_alpha, False, merge_weights
        if r > 0:
            self.lora_A = nn.Parameter(self.weight.new_zeros((r * kernel_size, in_channels * kernel_size)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((out_channels * kernel_size, r * kernel_size)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
        self._lora_ready = True
        self.reset_parameters()

    def reset_parameters(self):
        nn.Conv2d.reset_parameters(self)
        if getattr(self, "_lora_ready", False) and hasattr(self, "lora_A"):
            nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B)

    def forward(self, x, relu=False, pool=False, bits=False):
        if self.r > 0:
            return ops.conv2d(x, self.weight, self.bias, relu, self._packed, self.lora_A, self.lora_B, self.scaling, pool=pool, bits=bits)
        return ops.conv2d(x, self.weight, self.bias, relu, self._packed, pool=pool, bits=bits)


class HipMaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        if self.kernel_size != 2 or self.stride != 2 or self.padding != 0 or self.ceil_mode:
            raise ValueError("HipMaxPool2d supports kernel 2 / stride 2 / no padding only")
        return ops.max_pool2(x)


class HipBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (the serial adapters' first layer, models/ynet.py:24,64) on ynet_batchnorm2d_fwd / _bwd: same parameters, buffers and
    state-dict keys, nn.BatchNorm2d's own rule for the exponential average factor and num_batches_tracked."""

    def forward(self, x):
        self._check_input_dim(x)
        factor = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
            if self.momentum is None:      # cumulative moving average (needs the count on the host: not capturable, as in torch)
                factor = 1.0 / float(self.num_batches_tracked)
        train = self.training or (self.running_mean is None and self.running_var is None)
        return ops.batch_norm2d(x, self.weight, self.bias, self.running_mean if (not self.training or self.track_running_stats) else None,
                                self.running_var if (not self.training or self.track_running_stats) else None, train, factor, self.eps)


def _plain_conv(in_channels, out_channels=None, kernel_size=1, stride=1, padding=None, is_bias=False):
    """models/ynet.py:8-12."""
    if out_channels is None:
        out_channels = in_channels
    return HipConv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding, bias=is_bias)


class _AdapterMixin:
    """Construction / zero initialisation shared by AdapterBlock and AdapterLayer (models/ynet.py:15-56, 72-115):
    'serial...'            -> serial_layer = [BatchNorm2d, 1x1 conv]      (applied to the conv / stage output)
    'parallel..._KxK'      -> parallel_layer = KxK conv (no bias) of the input
    'parallel..._KxK_LxL'  -> parallel_layer = ModuleList of such convs, summed."""

    def _build_adapter(self, adapter_name, in_channels, out_channels, stride, is_bias, serial_channels):
        self.is_bias = is_bias
        self.adapter_name = adapter_name
        self.adapter_size = adapter_name.split("_")[1:]
        self.is_multiple = len(self.adapter_size) >= 2
        if "serial" in adapter_name:
            self.serial_layer = nn.Sequential(HipBatchNorm2d(serial_channels), _plain_conv(serial_channels, is_bias=is_bias))
        elif "parallel" in adapter_name and not self.is_multiple:
            k = int(self.adapter_size[0].split("x")[0]) if self.adapter_size else 1
            self.parallel_layer = _plain_conv(in_channels, out_channels, k, stride, is_bias=is_bias)
        elif "parallel" in adapter_name:
            self.parallel_layer = nn.ModuleList(
                _plain_conv(in_channels, out_channels, int(z.split("x")[0]), stride, is_bias=is_bias)
                for z in self.adapter_size)
        else:
            raise ValueError(f"Invalid adapter={adapter_name}")
        self.initialize()

    def initialize(self):
        if "serial" in self.adapter_name:
            nn.init.zeros_(self.serial_layer[1].weight)
            if self.is_bias:
                nn.init.zeros_(self.serial_layer[1].bias)
        elif "parallel" in self.adapter_name:
            for p in self.parallel_layer.parameters():
                nn.init.zeros_(p)

    def _branch(self, x_in, x_out):
        """The adapter branch without the residual."""
        if "serial" in self.adapter_name:
            return self.serial_layer[1](self.serial_layer[0](x_out))
        if self.is_multiple:
            y = None
            for layer in self.parallel_layer:
                z = layer(x_in)
                y = z if y is None else ops.add_relu(y, z)
            return y
        return self.parallel_layer(x_in)


class AdapterBlock(nn.Module, _AdapterMixin):
    """models/ynet.py:15-69: adapter between encoder stages (YNetEncoderB)."""

    def __init__(self, adapter_name, in_channels, out_channels=None, stride=1, is_bias=False):
        nn.Module.__init__(self)
        self._build_adapter(adapter_name, in_channels, out_channels, stride, is_bias, serial_channels=in_channels)

    def forward(self, x):
        if "serial" in self.adapter_name:
            return ops.add_relu(self._branch(None, x), x)
        return self._branch(x, None)


class AdapterLayer(HipConv2d, _AdapterMixin):
    """models/ynet.py:72-131: the conv plus a serial / parallel adapter on the same input, ReLU afterwards."""

    def __init__(self, in_channels, out_channels, kernel_size, adapter_name, adapter_dropout=0.0, stride=1,
                 is_bias=False, **kwargs):
        HipConv2d.__init__(self, in_channels, out_channels, kernel_size, stride=kwargs.pop("stride_", stride), **kwargs)
        self._build_adapter(adapter_name, in_channels, out_channels, stride, is_bias, serial_channels=out_channels)

    def forward(self, x, relu=False):
        out = ops.conv2d(x, self.weight, self.bias, False, self._packed)
        return ops.add_relu(self._branch(x, out), out, relu)


class FusedSequential(nn.Sequential):
    """nn.Sequential with the reference's child indices, executing Conv2d+ReLU pairs as one launch."""

    def forward(self, x, pool_next=False, defer_last=False):
        """pool_next: the caller feeds the result to a MaxPool2d(2, 2) next (the following encoder stage opens with one): the last
        conv + ReLU of this sequence is asked to write the pooled copy too (ops.conv2d(pool=True)).  defer_last: the caller feeds the result to the fused
        predictor + criterion next (ops.pred_bce): the last conv + ReLU may run inside that launch (ops.conv2d(defer=True))."""
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, HipConv2d):
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                last = i + (2 if fuse else 1) >= len(mods)
                # conv -> ReLU -> conv: the next conv's data gradient is written through THIS ReLU's backward; ask this launch for the
                # 1-bit form of its mask (ops.conv2d(bits=True))
                chain = (fuse and i + 2 < len(mods) and type(m) in (HipConv2d, LoRAConv2d) and type(mods[i + 2]) in (HipConv2d, LoRAConv2d)
                         and m.kernel_size[0] == 3 and mods[i + 2].kernel_size[0] == 3)
                if pool_next and last and type(m) in (HipConv2d, LoRAConv2d):
                    x = m(x, relu=fuse, pool=True)
                elif chain:
                    x = m(x, relu=fuse, bits=int(mods[i + 2].out_channels) if mods[i + 2].out_channels > 1 else True)
                elif defer_last and last and fuse and type(m) is HipConv2d:
                    x = m(x, relu=True, defer=True)
                else:
                    x = m(x, relu=fuse)
                i += 2 if fuse else 1
            else:
                if isinstance(m, nn.ReLU):
                    raise NotImplementedError("a stand-alone ReLU is not on the Y-Net path (it is fused into the conv)")
                x = m(x)
                i += 1
        return x


def get_conv2d(train_net, l, position, kernel_size, in_channels, out_channels=None, rank=None, stride=1,
               padding=None):
    """models/ynet.py:134-151."""
    if out_channels is None:
        out_channels = in_channels
    if padding is None:
        padding = kernel_size // 2
    l = str(l)
    position = [str(i) for i in position] if position is not None else []
    if "mosa" in train_net and l in position:
        assert rank != 0 and rank is not None
        return LoRAConv2d(in_channels, out_channels, kernel_size=kernel_size, r=rank, stride=stride, padding=padding)
    if "Layer" in train_net and l in position:
        return AdapterLayer(adapter_name=train_net, in_channels=in_channels, out_channels=out_channels,
                            kernel_size=kernel_size, stride=stride, padding=padding)
    return HipConv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding)


def _mosa_rank(train_net):
    if "mosa" in train_net:
        parts = train_net.split("_")
        return int(parts[1]) if len(parts) > 1 else 1
    return None


def _stage(train_net, l, position, cin, cout, rank, first):
    if first:
        return FusedSequential(get_conv2d(train_net, l, position, 3, cin, cout, rank), nn.ReLU(inplace=False))
    return FusedSequential(
        HipMaxPool2d(kernel_size=2, stride=2, padding=0, dilation=1, ceil_mode=False),
        get_conv2d(train_net, l, position, 3, cin, cout, rank), nn.ReLU(inplace=False),
        get_conv2d(train_net, l, position, 3, cout, cout, rank), nn.ReLU(inplace=False))


class YNetEncoder(nn.Module):
    def __init__(self, in_channels, channels=(64, 128, 256, 512, 512), train_net=None, position=[]):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, channels
        self.train_net, self.position = train_net, position
        self.rank = _mosa_rank(train_net)
        self.stages = nn.ModuleList([_stage(train_net, 0, position, in_channels, channels[0], self.rank, True)])
        for i in range(len(channels) - 1):
            self.stages.append(_stage(train_net, i + 1, position, channels[i], channels[i + 1], self.rank, False))
        self.stages.append(FusedSequential(HipMaxPool2d(kernel_size=2, stride=2, padding=0, dilation=1, ceil_mode=False)))

    def forward(self, x):
        features = []
        for i, stage in enumerate(self.stages):
            nxt = self.stages[i + 1] if i + 1 < len(self.stages) else None
            x = stage(x, pool_next=nxt is not None and isinstance(nxt[0], HipMaxPool2d))
            features.append(x)
        return features


class YNetEncoderL(YNetEncoder):
    pass


class Embedding(nn.Module):
    """models/ynet.py:154-167 (network='embed'): three 3x3 conv + ReLU on the scene / the motion maps."""

    def __init__(self, channels):
        super().__init__()
        self.conv = FusedSequential(
            HipConv2d(channels, channels, kernel_size=3, stride=1, padding=1), nn.ReLU(inplace=False),
            HipConv2d(channels, channels, kernel_size=3, stride=1, padding=1), nn.ReLU(inplace=False),
            HipConv2d(channels, channels, kernel_size=3, stride=1, padding=1), nn.ReLU(inplace=False))

    def forward(self, x):
        return self.conv(x)


class YNetEncoderB(YNetEncoder):
    """Reference models/ynet.py:237-283: the plain encoder, optionally with AdapterBlocks at ``position``."""

    def __init__(self, in_channels, channels=(64, 128, 256, 512, 512), train_net=None, position=[]):
        self.position = [int(i) for i in position]
        super().__init__(in_channels, channels, train_net, self.position)
        par_channels_in = [in_channels] + list(channels[:-1])
        if "serial" in self.train_net:
            self.adapters = nn.ModuleList([AdapterBlock(train_net, channels[i]) for i in self.position])
        elif "parallel" in self.train_net:
            self.adapters = nn.ModuleList([AdapterBlock(train_net, par_channels_in[i], channels[i]) for i in self.position])

    def forward(self, x):
        """The feature pyramid (one map per stage).  Adapter blocks sit at the stages listed in ``position``: a serial block
        transforms its stage's OUTPUT; a parallel block sees what the stage's convolutions see -- the stage input behind the
        stage's own leading max-pool, if it has one -- and is added to the stage output (reference models/ynet.py:258-283)."""
        serial, parallel = "serial" in self.train_net, "parallel" in self.train_net
        blocks = iter(self.adapters) if (serial or parallel) else iter(())
        pyramid = []
        for i, stage in enumerate(self.stages):
            adapted = i in self.position
            if serial:
                x = stage(x)
                x = next(blocks)(x) if adapted else x
            elif parallel:
                seen = stage[0](x) if isinstance(stage[0], nn.MaxPool2d) else x      # (pooled twice: once here, once inside the stage -- as the reference)
                x = stage(x)
                x = ops.add_relu(x, next(blocks)(seen)) if adapted else x
            else:
                following = self.stages[i + 1] if i + 1 < len(self.stages) else None
                x = stage(x, pool_next=following is not None and isinstance(following[0], HipMaxPool2d))
            pyramid.append(x)
        return pyramid


class YNetEncoderFusion(nn.Module):
    """Y-Net-Mod: separate scene / motion branches, concatenated, then fused stages (ynet.py:286-395)."""

    def __init__(self, scene_channel, motion_channel, channels, train_net=None, position=[], n_fusion=2):
        super().__init__()
        self.scene_channel, self.motion_channel, self.channels = scene_channel, motion_channel, channels
        self.train_net, self.position = train_net, position
        self.rank = _mosa_rank(train_net)
        assert not any(i % 2 for i in channels), f"Odd value in channels={channels}"
        assert n_fusion <= len(channels) - 1, "The number of fusion exceeds the total number of layer in encoder"
        n_sep = len(channels) - n_fusion - 1
        half = [c // 2 for c in channels]
        self.scene_stages = nn.ModuleList([_stage(train_net, "scene", position, scene_channel, half[0], self.rank, True)])
        self.motion_stages = nn.ModuleList([_stage(train_net, "motion", position, motion_channel, half[0], self.rank, True)])
        self.fusion_stages = nn.ModuleList()
        for i in range(n_sep):
            self.scene_stages.append(_stage(train_net, "scene", position, half[i], half[i + 1], self.rank, False))
        for i in range(n_sep):
            self.motion_stages.append(_stage(train_net, "motion", position, half[i], half[i + 1], self.rank, False))
        for i in range(n_sep, len(channels) - 1):
            self.fusion_stages.append(_stage(train_net, "fusion", position, channels[i], channels[i + 1], self.rank, False))
        self.fusion_stages.append(FusedSequential(HipMaxPool2d(kernel_size=2, stride=2, padding=0, dilation=1, ceil_mode=False)))

    def forward(self, scene_map, motion_map):
        scene, motion = [], []
        # The scene branch sees the SAME image for every trajectory of a batch (utils/train_epoch.py:87 expands one scene
        # to the batch size, and the reference convolves all the copies).  A batch-broadcast input (stride 0) goes through
        # the branch once; its features are handed on as broadcast views -- the convs read the one image in place and the
        # expand's backward sums the per-trajectory gradients before they enter the branch: the same values, 1/B of the
        # branch's forward and backward work.
        B = scene_map.shape[0]
        once = SHARED_SCENE_BRANCH and torch.is_tensor(scene_map) and B > 1 and scene_map.stride(0) == 0
        x = scene_map[:1] if once else scene_map
        def walk(stages, x, out, tail_pooled):
            # (every stage but a branch's last is followed by a stage that opens with MaxPool2d; the branch's last output is pooled by
            # the first fused stage -- as part of a concatenation, so each part is pooled on its own there)
            for i, stage in enumerate(stages):
                nxt = stages[i + 1] if i + 1 < len(stages) else None
                x = stage(x, pool_next=(isinstance(nxt[0], HipMaxPool2d) if nxt is not None else tail_pooled))
                out.append(x)
            return x

        tail = len(self.fusion_stages) > 0 and isinstance(self.fusion_stages[0][0], HipMaxPool2d)
        walk(self.scene_stages, x, scene, tail)
        last_scene = scene[-1]
        if once:
            # (training: every conv that reads a scene feature expands it itself, ops.BatchExpand -- the two decoders' gradients are
            # then summed over the batch BEFORE they are added)
            scene = [ops.BatchExpand(t, B) if (torch.is_grad_enabled() and t.requires_grad) else t.expand(B, -1, -1, -1) for t in scene]
        walk(self.motion_stages, motion_map, motion, tail)
        features = [ops.lazy_cat([s, m]) for s, m in zip(scene, motion)]
        x = features[-1]
        for i, stage in enumerate(self.fusion_stages):
            nxt = self.fusion_stages[i + 1] if i + 1 < len(self.fusion_stages) else None
            pool_next = nxt is not None and isinstance(nxt[0], HipMaxPool2d)
            if i == 0:
                # first fused stage starts with a max-pool of the concatenation = concat of the pools
                mods = list(stage)
                if once:
                    pooled = ops.lazy_cat([mods[0](last_scene).expand(B, -1, -1, -1), mods[0](x.parts[1])])
                else:
                    pooled = ops.lazy_cat([mods[0](p) for p in x.parts])
                x = pooled
                j = 1
                while j < len(mods):
                    fuse = j + 1 < len(mods) and isinstance(mods[j + 1], nn.ReLU)
                    last = j + (2 if fuse else 1) >= len(mods)
                    chain = (fuse and j + 2 < len(mods) and type(mods[j]) in (HipConv2d, LoRAConv2d) and type(mods[j + 2]) in (HipConv2d, LoRAConv2d))
                    if pool_next and last and type(mods[j]) in (HipConv2d, LoRAConv2d):
                        x = mods[j](x, relu=fuse, pool=True)
                    else:
                        x = mods[j](x, relu=fuse, bits=int(mods[j + 2].out_channels) if mods[j + 2].out_channels > 1 else True) if chain else mods[j](x, relu=fuse)
                    j += 2 if fuse else 1
                if isinstance(x, ops.LazyCat):      # n_fusion == 0: only the final pool
                    x = x.materialize()
            else:
                x = stage(x, pool_next=pool_next)
            features.append(x)
        return features


class YNetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels, output_len, traj=False):
        super().__init__()
        if traj:
            encoder_channels = [c + traj for c in encoder_channels]
        encoder_channels = encoder_channels[::-1]
        center = encoder_channels[0]
        self.center = FusedSequential(
            HipConv2d(center, center * 2, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1)), nn.ReLU(inplace=False),
            HipConv2d(center * 2, center * 2, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1)), nn.ReLU(inplace=False))
        up_in = [center * 2] + decoder_channels[:-1]
        up_out = [c // 2 for c in up_in]
        self.upsample_conv = nn.ModuleList([
            HipConv2d(a, b, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1)) for a, b in zip(up_in, up_out)])
        in_channels = [e + d for e, d in zip(encoder_channels, up_out)]
        self.decoder = nn.ModuleList([
            FusedSequential(
                HipConv2d(a, b, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1)), nn.ReLU(inplace=False),
                HipConv2d(b, b, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1)), nn.ReLU(inplace=False))
            for a, b in zip(in_channels, decoder_channels)])
        self.predictor = HipConv2d(decoder_channels[-1], output_len, kernel_size=1, stride=1, padding=0)

    # Set by `announce_bce_target` (utils/train_epoch.py): the criterion that will be applied to this decoder's output
    # and its target.  The predictor then runs fused with the loss and with its own dgrad (ynet_pred_bce: the last
    # activation is read once instead of three times); the logits it returns carry the loss for the criterion.
    _bce = None

    # utils/evaluate.py runs the K goal samples of a trajectory as a batch whose encoder features REPEAT along it
    # (ops.BatchRepeat).  A convolution is linear in its input channels, so the part of decoder[i][0] over those repeated
    # skip features is the same for all K samples: `share_skip_features` computes it once per trajectory batch
    # (ops.shared_conv_term) and the per-sample launches convolve only the up-sampled path and the way-point channels
    # (ynet_conv2d_add).  At the top level that is 32 of 50 input channels of the largest layer of the sweep.
    _shared_terms = None

    def share_skip_features(self, features):
        """Context manager: precompute the skip-feature terms of decoder[i][0] for the encoder features `features`
        (list of 6, finest first, as model.pred_features returns them) -- for the levels the additive kernels serve."""
        return _SharedSkipTerms(self, features)

    def forward(self, features, readout=False):
        """``readout=True`` (inference, the caller only wants ``softargmax`` of the result): where the shape allows it the
        heat-maps come back as an ``ops.LazyPredictor`` that ``SoftArgmax2D`` evaluates together with the read-out."""
        features = features[::-1]
        x = self.center(features[0])
        for lvl, (f, d, up) in enumerate(zip(features[1:], self.decoder, self.upsample_conv)):
            # (F.interpolate(x, scale_factor=2, mode='bilinear') + upsample_conv[lvl], reference models/ynet.py:463-464: one launch where
            #  the shape is served and the filter is frozen -- the up-sampled tensor is never written, ops.upsample2x_conv2d)
            x = ops.upsample2x_conv2d(x, up) if type(up) is HipConv2d else up(ops.upsample2x(x))
            y = self._first_conv_shared(lvl, d, x, f)
            # (the last level's second convolution may run INSIDE the fused predictor + criterion launch below: ops.conv2d(defer=True))
            defer = (lvl == len(self.decoder) - 1 and not readout and self._bce is not None and torch.is_grad_enabled() and type(self.predictor) is HipConv2d
                     and type(d) is FusedSequential and not d._forward_hooks and not d[2]._forward_hooks and not d[3]._forward_hooks)
            x = d[2](y, relu=True) if y is not None else (d(ops.lazy_cat([x, f]), defer_last=True) if defer else d(ops.lazy_cat([x, f])))
        if readout:
            if (FUSED_READOUT and not torch.is_grad_enabled() and type(self.predictor) is HipConv2d
                    and ops.pred_softargmax_supported(x, self.predictor.weight)):
                return ops.LazyPredictor(x, self.predictor.weight, self.predictor.bias)
            return self.predictor(x)
        bce = self._bce
        if (bce is not None and torch.is_grad_enabled() and type(self.predictor) is HipConv2d
                and ops.pred_bce_supported(x, self.predictor.weight)):
            target, expected = bce
            y, loss = ops.pred_bce(x, self.predictor.weight, self.predictor.bias, target, expected, self.predictor._packed)
            y._ynet_fused_bce = (target, loss, float(expected))
            return y
        return self.predictor(ops.materialize_deferred(x))


    def _first_conv_shared(self, lvl, d, x, f):
        """decoder[lvl][0] + ReLU through the shared-term kernel, or None when it does not apply to this call."""
        terms = self._shared_terms
        if terms is None or lvl not in terms or torch.is_grad_enabled() or not isinstance(f, ops.LazyCat):
            return None
        term, src = terms[lvl]
        src_parts = ops._parts(src)              # one tensor, or the (scene, motion) pair of the fusion encoder
        parts = f.parts
        if len(parts) <= len(src_parts):
            return None
        lead, rest = parts[:len(src_parts)], parts[len(src_parts):]
        times = lead[0].times if isinstance(lead[0], ops.BatchRepeat) else 1      # (one sample per pass: no repeat)
        for p_, s_ in zip(lead, src_parts):      # the leading parts must be the registered features (BatchRepeat holds a detached alias)
            t_ = p_.tensor if isinstance(p_, ops.BatchRepeat) else p_
            if not (torch.is_tensor(t_) and t_.data_ptr() == s_.data_ptr() and t_.shape == s_.shape
                    and (p_.times if isinstance(p_, ops.BatchRepeat) else 1) == times):
                return None
        conv0 = d[0]
        if not (type(conv0) is HipConv2d and torch.is_tensor(x) and all(torch.is_tensor(p) for p in rest) and isinstance(d[1], nn.ReLU)):
            return None
        B, cx, H, W = x.shape
        if B != src_parts[0].shape[0] * times or not ops.conv2d_add_supported(B, H, W, conv0.out_channels, 3):
            return None
        cf = sum(s_.shape[1] for s_ in src_parts)
        return ops.conv2d_shared_term(None, times, [x, *rest], conv0.weight, conv0.bias, True, conv0._packed, term, cx, cx + cf)


class _SharedSkipTerms:
    def __init__(self, decoder, features):
        self.decoder, self.features = decoder, features

    def __enter__(self):
        dec = self.decoder
        terms = {}
        if not torch.is_grad_enabled() and os.environ.get("YNET_SHARED_SKIP", "1") != "0":
            feats = self.features[::-1]
            for lvl, (f, d, up) in enumerate(zip(feats[1:], dec.decoder, dec.upsample_conv)):
                conv0 = d[0]
                fparts = ops._parts(f)
                if not (all(torch.is_tensor(t) and t.is_cuda for t in fparts) and type(conv0) is HipConv2d and conv0.kernel_size[0] == 3):
                    continue
                B, _, H, W = fparts[0].shape
                cf = sum(t.shape[1] for t in fparts)
                # (the per-sample launches have a multiple of B images: a level the kernels serve at B is served at k * B)
                if W % 4 or not ops.conv2d_add_supported(B, H, W, conv0.out_channels, 3):
                    continue
                cx = up.out_channels
                terms[lvl] = (ops.shared_conv_term(f, conv0.weight, cx, cx + cf, conv0._packed), f)
                ops.rest_filter(conv0.weight, cx, cx + cf, conv0._packed)      # packed HERE, on the caller's stream (see ops.rest_filter)
                # ... and its Winograd form, for per-sample launches over [the up-sampled features, what follows the skip features]
                ops.rest_filter_winograd(conv0.weight, cx, cx + cf, conv0._packed, (cx, conv0.in_channels - cx - cf), B, H, W)
        dec._shared_terms = terms or None
        return self

    def __exit__(self, *exc):
        self.decoder._shared_terms = None
        return False


class announce_bce_target:
    """``with announce_bce_target(decoder, criterion, target): maps = decoder(features)`` -- tells a YNetDecoder which
    BCE-with-logits target its output is about to be compared with, so that predictor, loss and the predictor's dgrad
    run as one kernel.  ``criterion(maps, target)`` afterwards returns the loss computed in that pass
    (models/trainer.py: HipBCEWithLogitsLoss); any other use of the maps is unaffected."""

    def __init__(self, decoder, criterion, target):
        self.decoder = decoder
        on = getattr(criterion, "fuses_with_predictor", False) and os.environ.get("YNET_PRED_BCE", "1") != "0"
        self.spec = (target, getattr(criterion, "expected_grad", 1.0)) if on else None

    def __enter__(self):
        self.decoder._bce = self.spec
        return self

    def __exit__(self, *exc):
        self.decoder._bce = None
        return False


class YNet(nn.Module):
    def __init__(self, obs_len, pred_len, segmentation_model_fp, use_features_only=False, n_semantic_classes=6,
                 encoder_channels=[], decoder_channels=[], n_waypoints=1, train_net=None, position=[],
                 network=None, n_fusion=None):
        super().__init__()
        self.train_net = train_net
        if segmentation_model_fp is not None:
            # frozen backbone, run once per scene (out of the hot path): loaded exactly as the reference does
            self.semantic_segmentation = torch.load(
                segmentation_model_fp, map_location=None if torch.cuda.is_available() else torch.device("cpu"),
                weights_only=False)
            print("Loaded segmentation model to GPU" if torch.cuda.is_available() else "Loaded segmentation model to CPU")
            if use_features_only:
                self.semantic_segmentation.segmentation_head = nn.Identity()
                n_semantic_classes = 16
        else:
            self.semantic_segmentation = nn.Identity()
        self.feature_channels = n_semantic_classes + obs_len
        self.network = network
        if "semantic" in train_net:
            kernel_size = int(train_net.split("_")[-1].split("x")[0])
            self.semantic_adapter = get_conv2d(train_net, None, None, kernel_size, n_semantic_classes, n_semantic_classes)
            nn.init.zeros_(self.semantic_adapter.weight)
            nn.init.zeros_(self.semantic_adapter.bias)
        if network == "fusion":
            assert n_fusion is not None
            self.encoder = YNetEncoderFusion(n_semantic_classes, obs_len, encoder_channels, train_net=train_net,
                                             position=position, n_fusion=n_fusion)
        elif network == "original" or network == "embed":
            if network == "embed":
                self.scene_embedding = Embedding(n_semantic_classes)
                self.motion_embedding = Embedding(obs_len)
            cls = YNetEncoderL if ("mosa" in train_net or "Layer" in train_net) else YNetEncoderB
            self.encoder = cls(in_channels=self.feature_channels, channels=encoder_channels, train_net=train_net,
                               position=position)
        else:
            raise ValueError("No network parameter is provided")
        self.goal_decoder = YNetDecoder(encoder_channels, decoder_channels, output_len=pred_len)
        self.traj_decoder = YNetDecoder(encoder_channels, decoder_channels, output_len=pred_len, traj=n_waypoints)
        self.softargmax_ = SoftArgmax2D(normalized_coordinates=False)
        self.encoder_channels = encoder_channels

    def segmentation(self, image):
        return self.semantic_segmentation(image)

    def adapt_semantic(self, semantic_img):
        if "semantic" in self.train_net:
            return self.semantic_adapter(semantic_img) + semantic_img
        return semantic_img

    def pred_goal(self, features):
        return self.goal_decoder(features)

    def pred_traj(self, features):
        return self.traj_decoder(features)

    def pred_traj_coords(self, features):
        """``softargmax(pred_traj(features))`` (utils/evaluate.py:259-262) without the heat-maps in between."""
        return self.softargmax_(self.traj_decoder(features, readout=True))

    def pred_features(self, scene_map, motion_map):
        if self.network == "fusion":
            return self.encoder(scene_map, motion_map)
        return self.encoder(ops.lazy_cat([scene_map, motion_map]))

    def softmax(self, x):
        return nn.Softmax(2)(x.view(*x.size()[:2], -1)).view_as(x)

    def softargmax(self, output):
        return self.softargmax_(output)

    def sigmoid(self, output):
        return ops.sigmoid(output)

    def softargmax_on_softmax_map(self, x):
        pos_y, pos_x = create_meshgrid(x, normalized_coordinates=False)
        x = x.flatten(2)
        ex = torch.sum(pos_x.reshape(-1) * x, dim=-1, keepdim=True)
        ey = torch.sum(pos_y.reshape(-1) * x, dim=-1, keepdim=True)
        return torch.cat([ex, ey], dim=-1)
