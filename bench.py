#!/usr/bin/env python3
"""Benchmark of the MI355X Y-Net(+LoRA) training step — BASELINE.json's metric.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (N=1): BASELINE.json configs[1] "SDD shortterm Y-Net + LoRA rank=1 on encoder[0-4],
batch=32, 256x256 raster, obs=8 pred=12".  One "step" = one batch iteration of train_epoch:
{3x gather_patch, encoder, goal decoder, BCE, waypoint pyramid, trajectory decoder, BCE, backward,
RCCL all-reduce of the adapter gradients (N>1), Adam step, 2x soft-argmax}.  N>1 is WEAK scaling:
every rank keeps 32 trajectories per step (global batch 32*N sharded by dist.DataParallel).
Inputs are synthetic and resident in HBM before the timed region, except the
[B, 20, 2] trajectory coordinates that train_epoch receives on the host like the reference does.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : the dominant kernel (the MFMA conv), timed live with HIP events on the launch stream
                 in a separate instrumented step; achieved = algorithmic FLOPs / kernel time
  cpu_baseline : the CPU oracle (torch-CPU restatement of the reference, "port") on a bounded sample.
"""
import argparse
import glob
import importlib
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "motion-style-transfer_amd"

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix = fp32 vector peak
PEAK_HBM_GBS = 8000.0


# Algorithmic work per trajectory of the whole step: conv FLOPs (2 x MACs; forward + dgrad + the
# wgrads of the trainable convs) and compulsory fp32 HBM traffic with every graph op reading its inputs / writing its
# outputs once.  C5: encoder + goal decoder once, 20 trajectory-decoder passes.
STEP_WORK = {"C1": (43.60, 1001.4), "C2": (30.66, 826.8), "C3": (30.66, 826.8), "C4": (112.15, 3545.9), "C5": (140.7, None)}


def pkg(sub):
    return importlib.import_module(PKG + "." + sub)


class WorkCfg(SimpleNamespace):
    """The workload's hyper-parameters; `oracle_cfg` turns it into the oracle's Cfg for the CPU legs."""

    @property
    def template_size(self):
        return int(4200 * self.resize_factor)      # models/trainer.py:61


def _cfg(obs_len, pred_len, waypoints, resize_factor, temperature, **kw):
    base = dict(obs_len=obs_len, pred_len=pred_len, waypoints=tuple(waypoints), resize_factor=resize_factor, temperature=temperature,
                n_classes=6, enc=(32, 32, 64, 64, 64), dec=(64, 64, 64, 32, 32), network="original", n_fusion=None,
                train_net="train", position=[], loss_scale=1000.0, kernlen=31, nsig=4.0)
    base.update(kw)
    return WorkCfg(**base)


def sdd_short(**kw):
    return _cfg(8, 12, (11,), 0.25, 1.0, **kw)


def sdd_long(**kw):
    return _cfg(5, 30, (14, 29), 0.25, 1.8, **kw)


def ind_long(**kw):
    return _cfg(5, 30, (14, 29), 0.33, 1.8, **kw)


def oracle_cfg(O, c):
    return O.Cfg(**{k: v for k, v in vars(c).items()})


def make_cfg(name):
    pos5 = ["0", "1", "2", "3", "4"]
    if name == "C2":
        return sdd_short(train_net="mosa_1", position=pos5), 256, 256, \
            "C2: SDD shortterm Y-Net + LoRA rank=1 on encoder[0-4], 256x256 raster, obs=8 pred=12"
    if name == "C1":
        return sdd_short(train_net="train"), 256, 256, "C1: SDD shortterm Y-Net, all weights trainable, 256x256"
    if name == "C3":
        return sdd_short(train_net="mosa_4", position=pos5), 256, 256, "C3: SDD ped->biker MoSA, LoRA rank=4, 256x256"
    if name == "C4":
        return ind_long(network="fusion", n_fusion=2, train_net="mosa_3", position=["scene"]), 512, 512, \
            "C4: inD longterm Y-Net-Mod, scene adapter only (LoRA r=3), 512x512, obs=5 pred=30"
    if name == "C5":
        return sdd_long(train_net="train"), 256, 256, \
            "C5: SDD longterm eval sweep, K=20 goal samples, obs=5 pred=30 waypoints [14,29], 256x256"
    raise SystemExit(f"unknown config {name}")


# Synthetic inputs of BASELINE.md section 3 (the product's own generators: the oracle is imported by the cpu_baseline /
# parity_check legs only): semantic map softmax(randn(6, H, W), 0) shared by the batch; trajectories start ~ U(0.3 W, 0.7 W)^2,
# steps ~ N(0, 2 px) cumulative, fp32 [n, obs + pred, 2].
def synthetic_scene(cfg, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.softmax(torch.randn(cfg.n_classes, H, W, generator=g), dim=0).unsqueeze(0)


def synthetic_trajectories(cfg, n, H, W, seed=0):
    g = torch.Generator().manual_seed(seed + 1)
    start = torch.rand(n, 1, 2, generator=g) * torch.tensor([0.4 * W, 0.4 * H]) + torch.tensor([0.3 * W, 0.3 * H])
    steps = torch.randn(n, cfg.obs_len + cfg.pred_len, 2, generator=g) * 2.0
    steps[:, 0] = 0
    return (start + steps.cumsum(dim=1)).float()


def build_model(cfg, dev, seed=0, lora_b_std=0.05, state_dict=None):
    """Random-init Y-Net of the config (torch's default initialisation under a fixed seed; LoRA runs set lora_B ~ N(0, 0.05):
    BASELINE.md section 3) with the freeze policy of its train_net -- or the given state dict (parity_check: the oracle's)."""
    ynet, trainer = pkg("models.ynet"), pkg("models.trainer")
    torch.manual_seed(seed)
    model = ynet.YNet(cfg.obs_len, cfg.pred_len, None, encoder_channels=list(cfg.enc), decoder_channels=list(cfg.dec),
                      n_waypoints=len(cfg.waypoints), train_net=cfg.train_net, position=list(cfg.position),
                      network=cfg.network, n_fusion=cfg.n_fusion)
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=True)
    else:
        g = torch.Generator().manual_seed(seed + 17)
        with torch.no_grad():
            for m in model.modules():
                if getattr(m, "r", 0) and hasattr(m, "lora_B"):
                    m.lora_B.copy_(torch.randn(m.lora_B.shape, generator=g) * lora_b_std)
    trainer.apply_freeze_policy(model, cfg.train_net, cfg.position, cfg.network)
    return model.to(dev)


def templates(cfg, dev):
    """(input template, ground-truth template) as YNetTrainer.templates() builds them on a HIP device (analytic: the windows
    are computed in the kernel, bit-identical to slices of the S x S arrays of utils/image_utils.py:15-37)."""
    iu = pkg("utils.image_utils")
    return (iu.analytic_dist_template(cfg.template_size, dev),
            iu.analytic_gaussian_template(cfg.template_size, cfg.kernlen, cfg.nsig, False, dev))


def loader_for(traj):
    return [(traj, [pd.DataFrame({"metaId": np.arange(traj.shape[0])})], "scene0")]


class ConvTimer:
    """Wraps ops.conv2d_raw with HIP-event pairs (same stream as the launch) for ONE instrumented step."""

    def __init__(self, ops):
        self.ops, self.orig, self.rec = ops, ops.conv2d_raw, []
        self.orig_wino = ops.conv2d_winograd_raw
        self.orig_cat = ops.conv2d_winograd_cat_raw
        self.orig_16 = ops.conv2d_winograd16_raw
        self.orig_up = ops.upsample2x_conv2d_raw
        self.orig_split = ops.conv2d_winograd_split_raw
        self.orig_predbce = ops.conv2d_winograd_pred_bce_raw
        self.orig_auto = getattr(ops, "conv_auto", False)

    def __enter__(self):
        # (the timed region runs ynet_conv2d_auto, which composes a layer's launches inside the library; the instrumented steps go through
        #  the Python twin of that dispatcher -- the same kernels, tests/test_gpu_kernels.py::test_conv2d_auto_takes_the_launches_of_the_python_dispatcher --
        #  because only there every launch of a call can be bracketed by its own event pair)
        self.ops.conv_auto = False

        def timed_wino(src, u, bias, dst, cin, cout, B, H, W, relu, relu_of=None, wbits_out=None, relu_wbits=None, s2d=False):
            # one Winograd launch (a convolution with 48 / 64 outputs is two of them): its own event pair, rocprof's kernel name
            # (third template argument: 0 plain, 1 through a ReLU backward with the float activation, 2 with the 1-bit mask, 3 plain + mask written,
            #  4 plain, stored space-to-depth: the gradient of an up-convolution's output)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.orig_wino(src, u, bias, dst, cin, cout, B, H, W, relu, relu_of=relu_of, wbits_out=wbits_out, relu_wbits=relu_wbits, s2d=s2d)
            e1.record()
            em = 4 if s2d else (3 if wbits_out is not None else (2 if relu_wbits is not None else (1 if relu_of is not None else 0)))
            name = f"conv_wino_kernel<{cout // 16}, {cin // 8}, {em}, 8>"
            self.rec.append((name, e0, e1, 2.0 * B * H * W * cin * cout * 9, 4.0 * B * H * W * (cin + cout * (2 if em == 1 else (1 + 1 / 32 if em >= 2 else 1))),
                             (B, H, W, cin, cout, 3, False)))
        self.ops.conv2d_winograd_raw = timed_wino

        def timed_split(src, u, dst0, dst0_s2d, dst1, cin, B, H, W):
            # the [16, 32]-channel data gradient in one launch (three output blocks per wave, four waves per workgroup); EM 5 row-major, 6 = the 16 channels space-to-depth
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.orig_split(src, u, dst0, dst0_s2d, dst1, cin, B, H, W)
            e1.record()
            self.rec.append((f"conv_wino_kernel<3, {cin // 8}, {6 if dst0_s2d else 5}, 4>", e0, e1, 2.0 * B * H * W * cin * 48 * 9, 4.0 * B * H * W * (cin + 48),
                             (B, H, W, cin, 48, 3, False)))
        self.ops.conv2d_winograd_split_raw = timed_split

        def timed_predbce(src, u, bias, pred_wp, pred_bias, pred_cout, pos, tmpl, logits, loss, dx, ws, B, H, W, expected_grad):
            # the last decoder convolution with the predictor, the criterion and the predictor's data gradient in its epilogue: the convolution's FLOPs + the two
            # 1 x 1 products; bytes = the input, the logits and dX (the convolution's own output is never written)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.orig_predbce(src, u, bias, pred_wp, pred_bias, pred_cout, pos, tmpl, logits, loss, dx, ws, B, H, W, expected_grad)
            e1.record()
            self.rec.append((f"conv_wino_kernel<2, 4, {7 if pred_cout <= 16 else 8}, 8>", e0, e1, 2.0 * B * H * W * 32 * (32 * 9 + 2 * pred_cout), 4.0 * B * H * W * (32 + pred_cout + 32),
                             (B, H, W, 32, 32, 3, False)))
        self.ops.conv2d_winograd_pred_bce_raw = timed_predbce

        def timed_cat(srcs, u, bias, dst, B, H, W, relu, addend=None, pool=None, wbits_out=None, pool_code=None):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.orig_cat(srcs, u, bias, dst, B, H, W, relu, addend=addend, pool=pool, wbits_out=wbits_out, pool_code=pool_code)
            e1.record()
            cin = sum(s_[1] for s_ in srcs)
            epi = 2 if addend is not None else (3 if pool is not None else 0)      # (epilogue: 0 plain, 2 additive term, 3 pooled copy; 4 / 5: plain / additive term + 1-bit mask written; 6: pooled copy + code bytes)
            if epi == 3 and pool_code is not None:
                epi = 6
            elif wbits_out is not None and epi != 3:
                epi = 4 if epi == 0 else 5
            name = f"conv_wino_cat_kernel<2, {epi}>"
            self.rec.append((name, e0, e1, 2.0 * B * H * W * cin * 32 * 9, 4.0 * B * H * W * (cin + 32 * (2 if addend is not None else (1.25 if pool is not None else 1))),
                             (B, H, W, cin, 32, 3, False)))
        self.ops.conv2d_winograd_cat_raw = timed_cat

        def timed_16(srcs, u, bias, dst, cout, B, H, W, relu, relu_of=None, addend=None, pool=None):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.orig_16(srcs, u, bias, dst, cout, B, H, W, relu, relu_of=relu_of, addend=addend, pool=pool)
            e1.record()
            cin = sum(s_[1] for s_ in srcs)
            epi = 1 if relu_of is not None else (2 if addend is not None else (3 if pool is not None else 0))
            name = f"conv_wino16_kernel<{epi}>"      # (epilogue: 0 plain, 1 through a ReLU backward, 2 additive term, 3 pooled copy)
            self.rec.append((name, e0, e1, 2.0 * B * H * W * cin * cout * 9, 4.0 * B * H * W * (cin + cout * (2 if epi in (1, 2) else (1.25 if epi == 3 else 1))),
                             (B, H, W, cin, cout, 3, False)))
        self.ops.conv2d_winograd16_raw = timed_16

        def timed_up(src, u, bias, dst, cin, cout, B, H, W, relu=False):
            # the up-convolution with its bilinear x2 inside: the convolution's FLOPs at the up-sampled size; bytes = the LOW-resolution input + the output
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.orig_up(src, u, bias, dst, cin, cout, B, H, W, relu)
            e1.record()
            kind = self.ops._lib().ynet_upsample2x_conv2d_winograd_supported(B, H, W, cin, cout, 3)
            self.rec.append(("conv_wino16_up_kernel" if kind == 2 else f"conv_wino_up_kernel<{cin // 8}>", e0, e1, 2.0 * B * H * W * cin * cout * 9, 4.0 * B * H * W * (cin / 4.0 + cout),
                             (B, H, W, cin, cout, 3, False)))
        self.ops.upsample2x_conv2d_raw = timed_up

        def timed(srcs, mask, wp, bias, dsts, B, H, W, K, relu, relu_of=None, pooled=None, bits_out=None, relu_bits=None, wino=None, **more):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            took = self.orig(srcs, mask, wp, bias, dsts, B, H, W, K, relu, relu_of=relu_of, pooled=pooled, bits_out=bits_out, relu_bits=relu_bits,
                             wino=wino, **more)
            e1.record()
            cin = sum(s[1] for s in srcs)
            dl = list(dsts)
            while len(dl) > 1 and dl[-1][0] is None:      # unwanted trailing outputs are not computed
                dl.pop()
            cout = sum(d[1] for d in dl)
            flops = 2.0 * B * H * W * cin * cout * K * K
            byts = 4.0 * B * H * W * (sum(s[1] for s in srcs if s[2] != 0) * (2 if mask else 1)
                                      + sum(d[1] for d in dl if d[0]) * (2 if relu_of else (1.25 if pooled else (1 + 1 / 32 if (bits_out or relu_bits) else 1))))
            plan = self.ops._lib().ynet_conv2d_plan(B, H, W, cout, K)
            rows, tiles, m16, dma = plan & 255, (plan >> 8) & 255, (plan >> 16) & 1, (plan >> 17) & 1
            cc = plan >> 21                       # input channels per staged chunk (the kernel's CC template argument)
            if took is not None and str(took).startswith("winograd"):
                return took                       # (timed launch by launch in timed_wino / timed_cat, under rocprof's kernel names)
            if dma:
                if cc > 8:                        # launch_dma_small: the deep chunks are for plain launches with enough input channels
                    if relu_of or pooled or bits_out or relu_bits:
                        cc = 8
                    while cc > 8 and cin <= cc // 2:
                        cc //= 2
                x4 = (plan >> 18) & 1
                fold = 1 << ((plan >> 19) & 3)
                kind = "emask_" if relu_of else ("pool_" if pooled else ("bits_" if bits_out else ("emaskb_" if relu_bits else "")))
                name = (f"conv_dma_{kind}kernel<{tiles}, {rows}, {cc}, {'true' if mask else 'false'}, "
                        f"{'true' if x4 else 'false'}, {fold}>")
            else:
                name = (f"conv_mfma_kernel<{K}, {tiles}, {rows}, {cc}, {'true' if mask else 'false'}, "
                        f"{'true' if m16 else 'false'}>")
            self.rec.append((name, e0, e1, flops, byts, (B, H, W, cin, cout, K, bool(mask))))
            return took
        self.ops.conv2d_raw = timed
        return self

    def __exit__(self, *a):
        self.ops.conv_auto = self.orig_auto
        self.ops.conv2d_raw = self.orig
        self.ops.conv2d_winograd_raw = self.orig_wino
        self.ops.conv2d_winograd_cat_raw = self.orig_cat
        self.ops.conv2d_winograd16_raw = self.orig_16
        self.ops.upsample2x_conv2d_raw = self.orig_up
        self.ops.conv2d_winograd_split_raw = self.orig_split
        self.ops.conv2d_winograd_pred_bce_raw = self.orig_predbce

    def layers(self, steps):
        """The launches of one step in call order: [kernel, (B, H, W, cin, cout, K, masked), median microseconds, direct-form TFLOP/s,
        direct-form GFLOP (2 * K^2 * Cin * Cout per pixel; a conv_wino launch EXECUTES 16 / 36 of it), algorithmic bytes]."""
        torch.cuda.synchronize()
        per = len(self.rec) // steps
        out = []
        for i in range(per):
            us = sorted(self.rec[i + s * per][1].elapsed_time(self.rec[i + s * per][2]) * 1e3 for s in range(steps))[steps // 2]
            name, _, _, fl, by, shape = self.rec[i]
            out.append([name, list(shape), round(us, 2), round(fl / us / 1e6, 2), round(fl / 1e9, 5), int(by)])
        return out

    def summary(self, by_shape=False):
        """Totals per kernel name, or per LAUNCH SHAPE (kernel name, (B, H, W, cin, cout, K, masked)): one instantiation serves layers
        of very different sizes, so a roofline fraction belongs to a shape, not to a name."""
        torch.cuda.synchronize()
        agg = {}
        for name, e0, e1, fl, by, shape in self.rec:
            d = agg.setdefault((name, tuple(shape)) if by_shape else name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
        return agg


class FlopCounter:
    """Counts the convolution FLOPs (2 x MACs, as SURVEY.md 8d counts them) the product really LAUNCHES in one step: the
    algorithmic figure of STEP_WORK assumes the reference's graph, while the product executes less where it shares work
    (Y-Net-Mod's scene branch once per batch; the shared skip terms of the evaluation sweep)."""

    def __init__(self, ops):
        self.ops, self.flops = ops, 0.0
        self.saved = {}

    def __enter__(self):
        ops, me = self.ops, self
        names = ("conv2d_raw", "conv2d_wgrad_raw", "lora_conv2d_wgrad_raw", "conv2d_shared_term", "pred_bce", "pred_softargmax", "upsample2x_conv2d_raw")
        self.saved = {n: getattr(ops, n) for n in names}

        def conv2d_raw(srcs, mask, wp, bias, dsts, B, H, W, K, relu, **kw):
            dl = list(dsts)
            while len(dl) > 1 and dl[-1][0] is None:
                dl.pop()
            per = 2.0 * B * H * W * sum(s_[1] for s_ in srcs) * K * K
            took = me.saved["conv2d_raw"](srcs, mask, wp, bias, dsts, B, H, W, K, relu, **kw)
            if took is not None and str(took).startswith("winograd"):      # 16 multiplies per 2 x 2 block and channel pair instead of 36; unwanted destinations are not computed
                me.flops += per * sum(d[1] for d in dl if d[0] is not None) * (16.0 / 36.0)
            else:
                me.flops += per * sum(d[1] for d in dl)
            return took

        def conv2d_wgrad_raw(srcs, dy, mask, weight, want_b, *a, **kw):
            cout, cin, k, _ = weight.shape
            B, _, H, W = dy.shape
            me.flops += 2.0 * B * H * W * cin * cout * k * k
            return me.saved["conv2d_wgrad_raw"](srcs, dy, mask, weight, want_b, *a, **kw)

        def lora_conv2d_wgrad_raw(srcs, dy, mask, weight, lora_a, lora_b, scale):
            cout, cin, k, _ = weight.shape
            B, _, H, W = dy.shape
            r = lora_a.shape[0] // k
            me.flops += 2.0 * B * H * W * (54 * cin + 18 * cout) * r      # projected planes instead of the full filter gradient
            return me.saved["lora_conv2d_wgrad_raw"](srcs, dy, mask, weight, lora_a, lora_b, scale)

        def conv2d_shared_term(x, x_times, rest, weight, bias, relu, cache, term, c0, c1):
            cout, cin, k, _ = weight.shape
            B, _, H, W = rest[0].shape
            n0 = ops.wino_stats["launches"]
            y = me.saved["conv2d_shared_term"](x, x_times, rest, weight, bias, relu, cache, term, c0, c1)
            me.flops += 2.0 * B * H * W * (cin - (c1 - c0)) * cout * k * k * (16.0 / 36.0 if ops.wino_stats["launches"] > n0 else 1.0)
            return y

        def pred_bce(x, weight, bias, target, expected_grad, cache):
            B, cin, H, W = x.shape
            me.flops += 2.0 * B * H * W * cin * weight.shape[0] * (2 if x.requires_grad else 1)      # predictor + its dgrad
            return me.saved["pred_bce"](x, weight, bias, target, expected_grad, cache)

        def pred_softargmax(x, weight, bias):
            B, cin, H, W = x.shape
            me.flops += 2.0 * B * H * W * cin * weight.shape[0]
            return me.saved["pred_softargmax"](x, weight, bias)

        def upsample2x_conv2d_raw(src, u, bias, dst, cin, cout, B, H, W, relu=False):
            me.flops += 2.0 * B * H * W * cin * cout * 9 * (16.0 / 36.0)      # (a Winograd launch)
            return me.saved["upsample2x_conv2d_raw"](src, u, bias, dst, cin, cout, B, H, W, relu)

        for n, f in (("conv2d_raw", conv2d_raw), ("conv2d_wgrad_raw", conv2d_wgrad_raw), ("lora_conv2d_wgrad_raw", lora_conv2d_wgrad_raw),
                     ("conv2d_shared_term", conv2d_shared_term), ("upsample2x_conv2d_raw", upsample2x_conv2d_raw),
                     ("pred_bce", pred_bce), ("pred_softargmax", pred_softargmax)):
            setattr(ops, n, f)
        return self

    def __exit__(self, *a):
        for n, f in self.saved.items():
            setattr(self.ops, n, f)


def readout_roofline(ops, ynet_mod, cfg, B, H, W, dev):
    """The HBM-bound kernel of the evaluation sweep, timed in isolation: every trajectory sample's read-out = predictor 1x1 +
    soft-argmax in one pass over the decoder's last activation [B, 32, H, W] (the logits are never written); where that
    launch does not apply, the soft-argmax over [B, pred, H, W] planes."""
    cin = int(cfg.dec[-1])
    n_img = B * max(1, min(20, 256 // B))      # images per decoder pass of the sweep (evaluate(): max_effective_batch 256)
    x = torch.relu(torch.randn(n_img, cin, H, W, device=dev))
    wt, bs_ = torch.randn(cfg.pred_len, cin, 1, 1, device=dev) * 0.2, torch.zeros(cfg.pred_len, device=dev)
    fused = ops.pred_softargmax_supported(x, wt) and ynet_mod.FUSED_READOUT
    if not fused:
        x = torch.randn(n_img, cfg.pred_len, H, W, device=dev)
    fn = (lambda: ops.pred_softargmax(x, wt, bs_)) if fused else (lambda: ops.softargmax2d(x))
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    gbs = x.numel() * 4 / us / 1e3
    traffic, src = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic_c5.json")
    if fused and os.path.exists(pmc):
        with open(pmc) as f:
            tab = json.load(f)
        ent = next((v for k, v in tab.items() if k.startswith(f"pred_softargmax_kernel<{cin},")), None)
        if ent and n_img == 256:       # (the committed counters are per launch of the 256-image pass of C5)
            # 16-byte-per-lane streaming reads are tallied at half their bytes on gfx950 (MI355X_MICROARCH.md, HBM section)
            traffic = ent["hbm_bytes_per_launch_fetch_x2"]
            src = ("profiles/pmc_traffic_c5.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --config C5` with the sample "
                   "groups on one stream, tools/profile_c5.sh; NOT measured in this run)")
    return {"bound": "hbm", "kernel": f"pred_softargmax_kernel<{cin}> (+ combine)" if fused else "softargmax_kernel",
            "achieved": gbs, "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": src, "avg_launch_us": us,
            "algorithmic_mb_per_launch": x.numel() * 4 / 1e6, "images_per_launch": n_img,
            "note": "20 back-to-back launches between one HIP-event pair; algorithmic bytes = the input tensor "
                    "read once (outputs are B x pred x 2 floats)"}


def cpu_baseline(O, cfg, H, W, batch, seconds_budget=20.0, big_batch=None):
    """CPU oracle (port of the reference's ATen-op path) on the host cores: bounded sample.  `cfg`: the oracle's Cfg."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # torch's intra-op pool degrades badly when oversubscribed (256 threads on the GPU box's host ran
    # 30x slower than 32): pick the fastest of a few thread counts on a quick conv probe.
    import torch.nn.functional as F
    probe_x, probe_w = torch.randn(4, 32, 128, 128), torch.randn(32, 32, 3, 3)
    best, cores = None, 1
    for t in sorted({c for c in (8, 16, 32, 64, avail) if c <= avail}):
        torch.set_num_threads(t)
        F.conv2d(probe_x, probe_w, padding=1)
        t0 = time.perf_counter()
        for _ in range(3):
            F.conv2d(probe_x, probe_w, padding=1)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, t
    torch.set_num_threads(cores)
    sd = O.make_state_dict(cfg, seed=0, lora_b_std=0.05)
    scene = O.synthetic_scene(cfg, H, W, 0)
    replay = {"state_dict": {k: v.clone() for k, v in sd.items()}, "scene": scene, "trajectories": []}      # what parity_check re-runs on the HIP path
    S = cfg.template_size
    in_t, gt_t = O.dist_template(S), O.gaussian_template(S, cfg.kernlen, cfg.nsig)
    names = O.trainable_names(cfg, sd)
    ms = {n: torch.zeros_like(sd[n]) for n in names}
    vs = {n: torch.zeros_like(sd[n]) for n in names}
    times = []
    t_start = time.perf_counter()
    step = 0
    first = []
    while True:
        traj = O.synthetic_trajectories(cfg, batch, H, W, 100 + step)
        t0 = time.perf_counter()
        r = O.train_step(sd, cfg, scene, traj, in_t, gt_t, names)
        if len(first) < 3:       # kept for parity_check: the HIP path repeats exactly these steps (eager, capture, replay)
            replay["trajectories"].append(traj)
            first.append({"batch": batch, "loss": float(r["loss"]), "ade": float(r["ade"].mean()), "fde": float(r["fde"].mean())})
        for n in names:
            sd[n], ms[n], vs[n] = O.adam_update(sd[n], r["grads"][n], ms[n], vs[n], step + 1, 1e-3)
        dt = time.perf_counter() - t0
        step += 1
        if step > 1:                    # first step = warm-up
            times.append(dt)
        if step >= 3 and (time.perf_counter() - t_start > seconds_budget or len(times) >= 12):      # ~10-20 s of CPU work
            break
    med = float(np.median(times))
    out = {"value": batch / med, "unit": "trajectories/s", "cores": cores, "kind": "port",
           "sample": f"{len(times)} steps of batch {batch} after 1 warm-up (oracle/ynet_oracle.py train_step + Adam, "
                     f"same config and raster size), median step {med * 1e3:.0f} ms; torch.set_num_threads({cores}) = the fastest of "
                     f"{{8, 16, 32, 64, {avail}}} on a conv probe ({avail} host cores available)"}
    if big_batch and big_batch != batch:
        # SURVEY 8(d) names B = 4 and B = 32 for the CPU leg: two steps at the benchmarked batch (first = warm-up)
        tb = []
        for i in range(2):
            traj = O.synthetic_trajectories(cfg, big_batch, H, W, 200 + i)
            t0 = time.perf_counter()
            r = O.train_step(sd, cfg, scene, traj, in_t, gt_t, names)
            for n in names:
                sd[n], ms[n], vs[n] = O.adam_update(sd[n], r["grads"][n], ms[n], vs[n], step + 1 + i, 1e-3)
            tb.append(time.perf_counter() - t0)
        out["at_benchmarked_batch"] = {"batch": big_batch, "value": big_batch / tb[-1], "unit": "trajectories/s",
                                       "sample": f"2nd of 2 steps of batch {big_batch}, {tb[-1] * 1e3:.0f} ms, same {cores} threads"}
    return out, first, replay


def parity_check(first, gpu_steps, launched):
    """The HIP path's first THREE steps against the CPU oracle's first three steps on the SAME inputs (state dict seed 0,
    trajectories seeds 100, 101, 102, batch = --cpu-batch, Adam lr 1e-3 between them): loss to 2e-5 relative, ADE / FDE to
    1e-4 (the north-star tolerance).  train_epoch runs the three steps of one shape as [eager, capture + replay, replay]:
    the third one is a pure replay of the captured hipGraph -- the thing the timed region launches."""
    steps, ok = [], True
    for want, (ade, fde, loss), how in zip(first, gpu_steps, launched):
        rel = abs(loss - want["loss"]) / abs(want["loss"])
        d_ade, d_fde = abs(ade - want["ade"]), abs(fde - want["fde"])
        good = bool(rel <= 2e-5 and d_ade <= 1e-4 and d_fde <= 1e-4)
        ok = ok and good
        steps.append({"launch": how, "loss_hip": loss, "loss_oracle": want["loss"], "loss_rel_err": rel,
                      "ade_hip": ade, "ade_oracle": want["ade"], "ade_abs_err": d_ade,
                      "fde_hip": fde, "fde_oracle": want["fde"], "fde_abs_err": d_fde, "ok": good})
    last = steps[-1]
    return {"batch": first[0]["batch"], "steps": steps, "step_launch_checked": last["launch"],
            "loss_rel_err": max(s_["loss_rel_err"] for s_ in steps), "ade_abs_err": max(s_["ade_abs_err"] for s_ in steps),
            "fde_abs_err": max(s_["fde_abs_err"] for s_ in steps),
            "tolerance": {"loss_rel": 2e-5, "ade_fde_abs": 1e-4}, "ok": ok}


def step_roofline(value_per_gpu, gf, mb, executed_gflop_per_traj):
    """The whole step against both roofs (per GPU); the dominant kernel's figure is in "roofline"."""
    return {
        "algorithmic_gflop_per_trajectory": gf, "tflops_per_gpu": value_per_gpu * gf / 1e3,
        "frac_of_fp32_peak": value_per_gpu * gf / 1e3 / PEAK_FP32_MFMA_TFLOPS,
        # what the launches of one step really compute (counted at the op layer in one eager step; without tile padding):
        # below the algorithmic figure where the product shares work (C4 scene branch, C5 shared skip terms); only THIS
        # fraction is hardware utilisation
        "executed_gflop_per_trajectory": executed_gflop_per_traj,
        "executed_tflops_per_gpu": value_per_gpu * executed_gflop_per_traj / 1e3,
        "executed_frac_of_fp32_peak": value_per_gpu * executed_gflop_per_traj / 1e3 / PEAK_FP32_MFMA_TFLOPS,
        "algorithmic_mb_per_trajectory": mb,
        "hbm_gbs_per_gpu": None if mb is None else value_per_gpu * mb / 1e3,
        "frac_of_hbm_peak": None if mb is None else value_per_gpu * mb / 1e3 / PEAK_HBM_GBS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-precapture", action="store_true",
                    help="capture the step's hipGraph inside the warm-up steps instead of before them (the first timed steps then follow a ~15 ms pause and run slower)")
    ap.add_argument("--batch", type=int, default=None,
                    help="trajectories per GPU per step (default: BASELINE.json's per-GPU batch: 32; C4 16; C5 128)")
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-repeats", action="store_true", help="time the contract's region only (no two repeat regions)")
    ap.add_argument("--no-c5", action="store_true", help="skip the short C5 leg of the default run")
    ap.add_argument("--no-legs", action="store_true", help="skip the short C1 / C4 legs of the default run")
    ap.add_argument("--conv-layers", default=None, metavar="FILE",
                    help="also write the instrumented step's convolution launches one by one (kernel, shape, microseconds) to FILE")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 5 s sustained region after the contract's timed region")
    ap.add_argument("--sustained-seconds", type=float, default=5.5)
    ap.add_argument("--layers", action="store_true", help="print one line per conv launch of the instrumented step (stderr)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU over RCCL), as CHILD processes
        # and before this process has made any HIP call, then exit with their code.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        import tempfile
        logs = tempfile.mkdtemp(prefix="ynet_bench_ranks_")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), "--log-dir", logs, "--tee", "3",
               os.path.abspath(__file__)] + sys.argv[1:]
        rc = subprocess.call(cmd)
        if rc != 0:      # a rank failed: non-zero exit of the launcher, with the tail of every rank's stderr (failing one first)
            tails = []
            for f in sorted(glob.glob(os.path.join(logs, "**", "stderr.log"), recursive=True)):
                with open(f, errors="replace") as fh:
                    lines = fh.read().splitlines()
                if lines:
                    bad = any("Traceback" in ln or "Error" in ln for ln in lines)
                    tails.append((not bad, f, lines[-25:]))
            for _, f, lines in sorted(tails):
                print(f"---- tail of {os.path.relpath(f, logs)}", file=sys.stderr)
                print("\n".join(lines), file=sys.stderr)
            print(f"bench.py: a rank failed (torch.distributed.run exit code {rc})", file=sys.stderr)
        raise SystemExit(rc if rc else 0)

    D = pkg("dist")
    rank, local, world = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    if os.environ.get("YNET_BENCH_SINGLE_DEVICE") == "1":      # development: every rank on cuda:0 (with YNET_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if os.environ.get("YNET_BENCH_FAIL_RANK") == str(rank) and world > 1:      # (tests: the launcher must report a dying rank)
        raise SystemExit(f"rank {rank} fails on purpose (YNET_BENCH_FAIL_RANK)")
    ynet, trainer, te, ops = pkg("models.ynet"), pkg("models.trainer"), pkg("utils.train_epoch"), pkg("ops")
    cfg, H, W, workload = make_cfg(args.config)
    if args.batch is None:
        args.batch = {"C4": 16, "C5": 128}.get(args.config, 32)
    B, N = args.batch, world
    # model, templates, scene: built by the PACKAGE (the oracle is imported further down, inside the cpu_baseline / parity_check legs only)
    model = build_model(cfg, dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    # YNET_DP_FORCE=1 with --gpus 1: a process group of ONE rank (RCCL) and every collective of the N-rank step issued for real --
    # the split-graph step with the eager all-reduce between its two hipGraphs (or the one-shot kernel inside one graph) on a 1-GPU box
    forced_dp = world == 1 and os.environ.get("YNET_DP_FORCE") == "1"
    if forced_dp and not dist.is_initialized():
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            free_port = sock.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port))
        dist.init_process_group(backend=os.environ.get("YNET_DIST_BACKEND") or "nccl", rank=0, world_size=1)
    dp = D.DataParallel(model.parameters()) if (world > 1 or forced_dp) else None
    crit = trainer.HipBCEWithLogitsLoss()
    in_t, gt_t = templates(cfg, dev)
    images = {"scene0": synthetic_scene(cfg, H, W, 0)[0].to(dev)}

    ev = pkg("utils.evaluate")

    def prep(n_steps, seed):
        """The inputs of `n_steps` steps (host coordinates, as the reference's loader hands them over) -- generated BEFORE a timed region."""
        return loader_for(synthetic_trajectories(cfg, B * N * n_steps, H, W, seed))

    def go(loader, graph=None):
        if args.config == "C5":
            a, f, _, _ = ev.evaluate(model, loader, images, dev, "sdd", None, in_t, list(cfg.waypoints), "test",
                                     20, 1, cfg.obs_len, B * N, cfg.resize_factor, cfg.temperature, dp=dp)
            return a, f, 0.0
        return te.train_epoch(model, loader, images, opt, crit, cfg.loss_scale, dev, "sdd", None, gt_t, in_t,
                              list(cfg.waypoints), 0, cfg.obs_len, cfg.pred_len, B * N, 10000, cfg.resize_factor,
                              cfg.network, False, dp=dp, graph=graph)

    def run(n_steps, seed, graph=None):
        return go(prep(n_steps, seed), graph)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, seed, loader=None):
        """One timed region: inputs generated first, then EXACTLY n_steps steps between two fences; MAX over ranks."""
        if loader is None:
            loader = prep(n_steps, seed)
        fence()
        t0 = time.perf_counter()
        res = go(loader)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, res

    # Per-kernel roofline of the dominant kernel: three instrumented EAGER steps on one stream, before the timed region
    # (eager launches are the only place where a HIP-event pair brackets one kernel: the timed region replays a captured
    # step with concurrent branches; measured after it the same launches read ~5 % longer on a chip the denser captured
    # work has warmed up).  All ranks run them (collectives inside).
    # FLOPs the product really launches per step (one counted step; all ranks run it: collectives inside)
    with FlopCounter(ops) as fc:
        run(1, 4, graph=False)
    executed_gflop_per_traj = fc.flops / B / 1e9      # (this rank's launches over this rank's B trajectories)
    ct, n_inst = None, 3
    if not args.no_roofline and args.config != "C5":
        ops.overlap_decoders = False      # kernels are timed in isolation: the two decoder streams run back to back
        run(1, 3, graph=False)            # (first launches: code objects, function attributes, allocator growth)
        if rank == 0:
            with ConvTimer(ops) as ct:
                run(n_inst, 3, graph=False)
        else:
            run(n_inst, 3, graph=False)
        ops.overlap_decoders = True
        fence()

    # (the timed region's inputs are generated BEFORE the warm-up steps, so that the K timed steps follow the W warm-up steps directly: generating
    #  them in between left the GPU idle for ~20 ms, and the first three steps after such a pause run 3-7 % slower -- tools/step_periods.py)
    first_loader = prep(args.steps, 2)
    if args.config != "C5" and not args.no_precapture:
        # set-up, not a step of the benchmark: the step's hipGraph is captured here (an eager step, then the capturing one), and the garbage the capture
        # leaves behind is collected, so that the W warm-up steps below are W replays and nothing of the capture is left to happen between them and the timed steps
        import gc
        run(2, 7)
        gc.collect()
        fence()
    if args.warmup > 0:
        run(args.warmup, 1)
    elapsed, (ade, fde, loss) = timed(args.steps, 2, first_loader)
    value = B * N * args.steps / elapsed
    # The contract's timed region is the one above (EXACTLY --steps steps); two more identical regions give the spread
    regions = [elapsed / args.steps * 1e3]
    for rep in range(0 if args.no_repeats else 2):
        regions.append(timed(args.steps, 5 + rep)[0] / args.steps * 1e3)
    # >= 5 s of the same replayed steps after the contract region: long enough for the driver's SMI sampler to see the GPU busy,
    # and a check of the clocks under sustained load (the contract region is 0.2-0.5 s)
    sustained = None
    if not args.no_sustained and args.config != "C5":
        n_sus = max(args.steps, int(np.ceil(args.sustained_seconds * 1e3 / regions[0])))
        dt_s, _ = timed(n_sus, 9)
        sustained = {"steps": n_sus, "seconds": dt_s, "ms_per_step": dt_s / n_sus * 1e3, "value": B * N * n_sus / dt_s,
                     "unit": "trajectories/s", "note": "one more timed region of the same replayed steps, sized for >= "
                     f"{args.sustained_seconds:g} s; `value` / `ms_per_step` of the line stay the contract region's"}

    out = {
        "metric": ("trajectories/sec eval sweep K=20 (Y-Net, SDD longterm)" if args.config == "C5"
                   else "trajectories/sec fwd+bwd (Y-Net+LoRA, SDD shortterm)"), "value": value, "unit": "trajectories/s",
        "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "batch_per_gpu": B, "global_batch": B * N, "raster": f"{H}x{W}",
                   "obs_len": cfg.obs_len, "pred_len": cfg.pred_len, "train_net": cfg.train_net,
                   "parallelism": f"dp{N}", "trainable_floats": sum(p.numel() for p in model.parameters() if p.requires_grad)},
        "final_loss": loss,
        "timed_regions": {"ms_per_step": [round(r_, 4) for r_ in regions], "median_ms_per_step": float(np.median(regions)),
                          "min_ms_per_step": min(regions), "max_ms_per_step": max(regions),
                          "median_value": B * N / (float(np.median(regions)) * 1e-3),
                          "note": "region 0 is the contract's timed region (value / ms_per_step); the others repeat it; the inputs of "
                                  "every region are generated before its clock starts"},
        "step_launch": (pkg("utils.evaluate").last_sweep_launch() if args.config == "C5" else
                        ("hipGraph replay" if pkg("utils.step_graph").enabled(None, dev) else "eager")),
    }
    if sustained is not None:
        out["sustained"] = sustained
    if out["step_launch"] != "eager" and args.config != "C5":      # what the step cache really holds: a failed capture means the timed steps ran eagerly
        sg = pkg("utils.step_graph")
        entries = [e for c in sg._caches.get(model, {}).values() for e in c.entries.values()]
        out["step_graphs"] = {"captured": sum(1 for e in entries if e.ready), "failed": sum(1 for e in entries if e.failed),
                              "graphs_per_step": max([len(e.graphs) for e in entries if e.ready] or [0]),
                              "collective_in_graph": any(getattr(e, "collective_in_graph", False) for e in entries if e.ready)}
        if not any(e.ready for e in entries):
            out["step_launch"] = "eager (no step was captured)"
    # proof of the process group the step ran on: size and backend as torch.distributed reports them, and every rank's device
    mine = {"rank": rank, "device": str(dev), "name": torch.cuda.get_device_name(dev),
            "uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", "")), "pid": os.getpid()}
    if dp is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        out["world"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": ranks,
                        "allreduce_floats_per_step": int(dp.flat.numel())}
        # one process per GPU: with at least as many devices as ranks every rank must sit on a device of its own
        uuids = [r_["uuid"] for r_ in ranks]
        if torch.cuda.device_count() >= world and os.environ.get("YNET_BENCH_SINGLE_DEVICE") != "1":
            assert len(set(uuids)) == world, f"ranks share a device: {[(r_['rank'], r_['device'], r_['uuid']) for r_ in ranks]}"
        out["world"]["distinct_devices"] = len(set(uuids))
    else:
        out["world"] = {"world_size": 1, "backend": None, "ranks": [mine], "allreduce_floats_per_step": 0}
    gf, mb = STEP_WORK[args.config]
    out["step_roofline"] = step_roofline(value / N, gf, mb, executed_gflop_per_traj)

    if args.no_roofline:
        pass
    elif args.config == "C5":
        if rank == 0:
            out["roofline"] = readout_roofline(ops, pkg("models.ynet"), cfg, B, H, W, dev)
    elif rank == 0:
        agg = ct.summary()
        if args.conv_layers:
            with open(args.conv_layers, "w") as f:
                json.dump(ct.layers(n_inst), f)
        for v in agg.values():               # per step
            for k in ("launches", "ms", "flops", "bytes"):
                v[k] = v[k] / n_inst
            v["launches"] = int(round(v["launches"]))
        if args.layers:
            for name, e0, e1, fl, by, shape in ct.rec:
                ms = e0.elapsed_time(e1)
                print(f"LAYER B,H,W,cin,cout,K,mask={shape} {ms * 1e3:8.1f} us {fl / ms / 1e9:7.2f} TF/s", file=sys.stderr)
        # The dominant LAUNCH SHAPE (kernel instantiation + (B, H, W, cin, cout)), not a name that mixes 27 launches of different sizes.
        # `achieved` / `frac` count the FLOPs the matrix pipes EXECUTE (a Winograd F(2x2, 3x3) launch: 16 / 36 of the direct form's
        # 2 * 9 * Cin * Cout per pixel), so frac <= 1 by construction; the direct-form figure (SURVEY 8d's per-unit work, what the
        # reference computes) is kept as `direct_equiv_*`.
        shapes = ct.summary(by_shape=True)
        for v in shapes.values():
            for k in ("launches", "ms", "flops", "bytes"):
                v[k] = v[k] / n_inst

        def executed_factor(kernel):
            return 16.0 / 36.0 if kernel.startswith("conv_wino") else 1.0

        def shape_entry(key, v):
            kernel, shp = key
            us = v["ms"] * 1e3 / v["launches"]
            direct_tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
            ex_tf = direct_tf * executed_factor(kernel)
            gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            return {"kernel": kernel, "shape": dict(zip(("B", "H", "W", "cin", "cout", "K", "input_masked"), shp)),
                    "launches_per_step": int(round(v["launches"])), "avg_launch_us": round(us, 2), "ms_per_step": round(v["ms"], 4),
                    "executed_gflop_per_launch": round(v["flops"] * executed_factor(kernel) / v["launches"] / 1e9, 4),
                    "achieved": round(ex_tf, 2), "frac": round(ex_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                    "direct_equiv_gflop_per_launch": round(v["flops"] / v["launches"] / 1e9, 4),
                    "direct_equiv_achieved": round(direct_tf, 2), "direct_equiv_frac": round(direct_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                    "algorithmic_mb_per_launch": round(v["bytes"] / v["launches"] / 1e6, 3), "algorithmic_hbm_gbs": round(gbs, 1),
                    "hbm_frac": round(gbs / PEAK_HBM_GBS, 4)}

        ranked = sorted(shapes.items(), key=lambda kv: -kv[1]["ms"])
        top = [shape_entry(k_, v_) for k_, v_ in ranked[:3]]
        (name, dom_shape), d = ranked[0]
        dom = top[0]
        # what a HIP-event pair reads around a ~1 us kernel: the marker / dispatch latency that every per-launch figure here
        # contains and rocprofv3's kernel durations do not (reported, NOT subtracted)
        floor = []
        one = torch.zeros(1, device=dev)
        for _ in range(30):
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record()
            one.fill_(0.0)
            f1.record()
            floor.append((f0, f1))
        torch.cuda.synchronize()
        floor_us = sorted(a.elapsed_time(b) * 1e3 for a, b in floor)[len(floor) // 2]
        # HBM traffic of this launch shape from the counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, joined with
        # the launch order of one eager serial step by tools/conv_shapes.py; committed under profiles/)
        traffic, traffic_src = None, None
        for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_conv_shapes_C2.json")), reverse=True):
            with open(fn) as f:
                for e in json.load(f).get("shapes", []):
                    if e.get("kernel") == name and [e["shape"].get(k_) for k_ in ("B", "H", "W", "cin", "cout", "K")] == list(dom_shape[:6]) \
                            and e.get("pmc_hbm_bytes_per_launch") is not None:
                        traffic, traffic_src = e["pmc_hbm_bytes_per_launch"], os.path.relpath(fn, ROOT)
            if traffic is not None:
                break
        out["roofline"] = {"bound": "mfma", "kernel": name, "shape": dom["shape"], "achieved": dom["achieved"], "peak": PEAK_FP32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": dom["frac"], "traffic": traffic,
                           "flops_counted": "EXECUTED by the matrix pipes" + (": Winograd F(2x2, 3x3) on the fp32 matrix cores (csrc/conv_wino.hip), 16 multiplies per "
                                            "2x2 output block and channel pair instead of the direct form's 36" if name.startswith("conv_wino") else ""),
                           "direct_equiv_achieved": dom["direct_equiv_achieved"], "direct_equiv_frac": dom["direct_equiv_frac"],
                           "direct_equiv_note": "SURVEY 8d's algorithmic work (2 * 9 * Cin * Cout per pixel, what the reference computes) over the same time: "
                                                "above 1.0 of the peak is the algorithmic saving of the Winograd form, not a roofline fraction",
                           "traffic_source": None if traffic is None else
                           f"{traffic_src} (FETCH_SIZE, doubled for the 16-byte LDS-DMA kernels, + WRITE_SIZE of this launch shape, rocprofv3 --pmc passes of this command on an earlier run "
                           "of the same build; NOT measured in this run)",
                           "launches_per_step": dom["launches_per_step"], "avg_launch_us": dom["avg_launch_us"],
                           "event_pair_floor_us": floor_us,
                           "executed_gflop_per_launch": dom["executed_gflop_per_launch"],
                           "algorithmic_mb_per_launch": dom["algorithmic_mb_per_launch"], "algorithmic_hbm_gbs": dom["algorithmic_hbm_gbs"],
                           "hbm_frac": dom["hbm_frac"],
                           "top_shapes": top,
                           "profile": (traffic_src or "profiles/r06_conv_shapes_C2.json") + ": every launch shape of one step with rocprofv3's kernel-only duration "
                                      "(--kernel-trace of this command with YNET_STEP_GRAPH=0 YNET_SERIAL_DECODERS=1), its executed FLOPs and counter "
                                      "traffic; the *_bench_C2_serial_kernel_stats.csv of the same round is the --stats table of the same trace",
                           "note": "per-launch HIP-event timing of 3 instrumented EAGER steps on one stream, run before the warm-up of the timed "
                                   "region; a HIP-event pair also reads the marker / dispatch latency around the kernel (`event_pair_floor_us` "
                                   "around a 1-element fill), so `avg_launch_us` sits that much above rocprofv3's kernel-only duration and "
                                   "`frac` below the fraction computed from the profile.  The timed region itself replays the captured step, "
                                   "whose decoder branches run CONCURRENTLY: per-kernel durations inside it are not a kernel-quality measure; "
                                   "`value` and `step_roofline` are."}
        out["conv_kernels"] = {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                   "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in agg.items()}
        for k, v in out["conv_kernels"].items():
            if k.startswith("conv_wino"):      # `tflops` counts the convolution's (direct-form) FLOPs; the kernel executes 16 / 36 of them
                v["tflops_executed"] = round(v["tflops"] / 2.25, 2)
                v["form"] = "Winograd F(2x2, 3x3), csrc/conv_wino.hip: tflops = the direct form's 2 * 9 * Cin * Cout per pixel over the time"
        total_conv_ms = sum(v["ms"] for v in agg.values())
        out["conv_share_of_step"] = total_conv_ms / out["ms_per_step"]
    if dp is not None:
        # ---- the collective on its own and the split of a replayed step around it (graph A = forward / backward, eager
        # all-reduce of the flat gradient buffer, graph B = optimizer / read-out), HIP events on the step's stream
        sgm = pkg("utils.step_graph")
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        probe = dp.flat.clone()
        keep = dp.flat
        dp.flat = probe                      # (time the transport on a scratch copy: the gradients stay as they are)
        for _ in range(5):
            dp.allreduce()
        fence()
        ev0.record()
        for _ in range(50):
            dp.allreduce()
        ev1.record()
        torch.cuda.synchronize()
        dp.flat = keep
        out["world"]["allreduce_us_per_step"] = ev0.elapsed_time(ev1) * 1e3 / 50
        out["world"]["allreduce_transport"] = "oneshot (ynet_allreduce_sum, HIP IPC)" if dp._comm is not None else f"torch.distributed ({dist.get_backend()})"
        out["world"]["transport_note"] = dp.transport_note
        split = None
        for c in sgm._caches.get(model, {}).values():
            for e in c.entries.values():
                if e.ready and getattr(e, "split", False):
                    split = e.profile_split(10)
        out["world"]["replayed_step_split_ms"] = split      # {"graph_a", "allreduce", "graph_b"} per step, or None (eager / single graph)
        dist.barrier()
    if rank == 0 and N == 1 and args.config == "C2" and not args.no_c5:
        # ---- BASELINE.json's HBM-bound roofline point (configs[4]: K = 20 goal-decoder sweep, B = 128) in the default run:
        # two warm-up batches (the first ones grow the caching allocator's pools: 150 instead of 121 ms per batch) + four timed
        # batches of the evaluation sweep, and its read-out kernel timed in isolation
        cfg5, H5, W5, workload5 = make_cfg("C5")
        m5 = build_model(cfg5, dev)
        B5 = 128
        in5, _ = templates(cfg5, dev)
        img5 = {"scene0": synthetic_scene(cfg5, H5, W5, 0)[0].to(dev)}

        def sweep(loader):
            return ev.evaluate(m5, loader, img5, dev, "sdd", None, in5, list(cfg5.waypoints), "test", 20, 1,
                               cfg5.obs_len, B5, cfg5.resize_factor, cfg5.temperature)
        sweep(loader_for(synthetic_trajectories(cfg5, B5 * 2, H5, W5, 11)))
        l5 = loader_for(synthetic_trajectories(cfg5, B5 * 4, H5, W5, 12))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a5, f5, _, _ = sweep(l5)
        torch.cuda.synchronize()
        dt5 = time.perf_counter() - t0
        out["c5"] = {"workload": workload5, "batch": B5, "batches_timed": 4, "value": 4 * B5 / dt5, "unit": "trajectories/s",
                     "ms_per_batch": dt5 / 4 * 1e3, "ade": float(a5), "fde": float(f5),
                     "sweep_launch": pkg("utils.evaluate").last_sweep_launch(),
                     "roofline": readout_roofline(ops, ynet, cfg5, B5, H5, W5, dev)}
        del m5
    if rank == 0 and N == 1 and args.config == "C2" and not args.no_legs:
        # ---- the other single-GPU training configs of BASELINE.json, driver-timed: C1 (every weight trains: the filter-gradient
        # kernels and ynet_adam_step over 1.64 M parameters) and C4 (Y-Net-Mod, 512x512, per-GPU batch 16): 3 steps of warm-up
        # (eager, capture + replay, replay), then 5 replayed steps between two synchronisations
        for leg, Bl in (("C1", 32), ("C4", 16)):
            cl, Hl, Wl, wl = make_cfg(leg)
            ml = build_model(cl, dev)
            optl = torch.optim.Adam(ml.parameters(), lr=1e-3)
            inl, gtl = templates(cl, dev)
            imgl = {"scene0": synthetic_scene(cl, Hl, Wl, 0)[0].to(dev)}

            def leg_run(loader, graph=None):
                return te.train_epoch(ml, loader, imgl, optl, crit, cl.loss_scale, dev, "sdd", None, gtl, inl, list(cl.waypoints), 0,
                                      cl.obs_len, cl.pred_len, Bl, 10000, cl.resize_factor, cl.network, False, graph=graph)
            with FlopCounter(ops) as fcl:
                leg_run(loader_for(synthetic_trajectories(cl, Bl, Hl, Wl, 20)), graph=False)
            leg_run(loader_for(synthetic_trajectories(cl, Bl * 3, Hl, Wl, 21)))
            ll = loader_for(synthetic_trajectories(cl, Bl * 5, Hl, Wl, 22))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            leg_run(ll)
            torch.cuda.synchronize()
            dtl = time.perf_counter() - t0
            sgl = pkg("utils.step_graph")
            captured = sum(1 for c in sgl._caches.get(ml, {}).values() for e in c.entries.values() if e.ready)
            gfl, mbl = STEP_WORK[leg]
            out[leg.lower()] = {"workload": wl, "batch": Bl, "steps_timed": 5, "value": 5 * Bl / dtl, "unit": "trajectories/s",
                                "ms_per_step": dtl / 5 * 1e3, "step_launch": "hipGraph replay" if captured else "eager",
                                "step_roofline": step_roofline(5 * Bl / dtl, gfl, mbl, fcl.flops / Bl / 1e9)}
            del ml, optl
    if rank == 0 and N == 1 and not args.no_cpu_baseline and args.config != "C5":
        # ---- the ONLY place the oracle is imported: the CPU baseline and the parity check (oracle = checker)
        from oracle import ynet_oracle as O
        out["cpu_baseline"], first, replay = cpu_baseline(O, oracle_cfg(O, cfg), H, W, args.cpu_batch, big_batch=B)
        # the same three steps on the HIP path: fresh model from the oracle's state dict, its scene, its trajectories; one batch
        # per train_epoch call, so the calls run [eager, capture + replay, replay] and each returns ITS step's loss / ADE / FDE
        m2 = build_model(cfg, dev, state_dict=replay["state_dict"])
        opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
        images2 = {"scene0": replay["scene"][0].to(dev)}
        sg = pkg("utils.step_graph")
        gpu_steps, launched = [], []
        for i, traj in enumerate(replay["trajectories"]):
            before = sum(1 for c in sg._caches.get(m2, {}).values() for e in c.entries.values() if e.ready)
            gpu_steps.append(te.train_epoch(m2, loader_for(traj), images2, opt2, crit, cfg.loss_scale, dev, "sdd", None, gt_t, in_t,
                                            list(cfg.waypoints), i, cfg.obs_len, cfg.pred_len, args.cpu_batch, 10000,
                                            cfg.resize_factor, cfg.network, False))
            after = sum(1 for c in sg._caches.get(m2, {}).values() for e in c.entries.values() if e.ready)
            launched.append("replay" if before else ("capture + replay" if after else "eager"))
        out["parity_check"] = parity_check(first, gpu_steps, launched)
    if rank == 0:
        # the other BASELINE configurations measured by this run, once more under ONE key (value / ms only: the full objects are `c1`, `c4`, `c5`)
        legs = {k: {kk: out[k][kk] for kk in ("workload", "batch", "value", "unit", "ms_per_step", "ms_per_batch") if kk in out[k]} for k in ("c1", "c4", "c5") if k in out}
        if legs:
            out["legs"] = legs
        out["conv_dispatch"] = "ynet_conv2d_auto (csrc/conv_auto.cpp)" if getattr(ops, "conv_auto", False) else "ops._conv2d_raw_py (YNET_CONV_AUTO=0)"
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
