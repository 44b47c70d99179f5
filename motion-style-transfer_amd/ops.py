"""Operators of the MI355X Y-Net path: thin autograd wrappers over the C ABI (include/ynet_hip.h).

Every function here enqueues hand-written gfx950 kernels from csrc/ on the current torch HIP
stream.  There is no CPU path and no ATen fallback: host tensors raise.  torch is used for device
memory (torch.empty), streams and autograd bookkeeping only.
"""
import ctypes
from typing import List, Sequence

import numpy as np
import torch

from . import _lib as L

MAX_SRC = 4
_VP = ctypes.c_void_p


def _lib():
    return L.load()


def _stream():
    return _VP(torch.cuda.current_stream().cuda_stream)


def _need_gpu(t: torch.Tensor, what: str, dtypes=(torch.float32,)):
    if not torch.is_tensor(t):
        raise TypeError(f"{what}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the MI355X path runs on HIP devices only "
                           f"(no CPU fallback exists by design)")
    if dtypes is not None and t.dtype not in dtypes:
        raise TypeError(f"{what}: {' / '.join(str(d).replace('torch.', '') for d in dtypes)} expected, got {t.dtype}")
    # kernels are enqueued on the CURRENT device's stream: a tensor of another GPU would be addressed from the wrong
    # device's queue.  (YNetTrainer and dist.init_from_env select the device; one process drives one GPU.)
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"{what}: tensor lives on {t.device} but the current HIP device is cuda:{torch.cuda.current_device()}; "
                           f"call torch.cuda.set_device({t.device.index}) (or use `with torch.cuda.device(...)`) first")


# ------------------------------------------------------------------------------------------------
# lazy channel concatenation (fused into the consumer conv; replaces torch.cat at
# models/ynet.py:387,466,574, utils/train_epoch.py:103-104, utils/evaluate.py:259-260)
# ------------------------------------------------------------------------------------------------
class LazyCat:
    """A channel concatenation that is never materialised: conv kernels read its parts directly.
    Any other torch function receives the materialised tensor (``__torch_function__``)."""

    def __init__(self, parts: Sequence[torch.Tensor]):
        flat: List[torch.Tensor] = []
        for p in parts:
            flat.extend(p.parts if isinstance(p, LazyCat) else [p])
        b, _, h, w = flat[0].shape     # parts are tensors or BatchRepeat views
        for p in flat:
            if p.dim() != 4 or p.shape[0] != b or p.shape[2:] != (h, w):
                raise ValueError(f"lazy_cat: incompatible shapes {[tuple(q.shape) for q in flat]}")
        self.parts = flat

    @property
    def shape(self):
        p = self.parts[0]
        return torch.Size((p.shape[0], sum(q.shape[1] for q in self.parts), p.shape[2], p.shape[3]))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 4

    @property
    def device(self):
        return self.parts[0].device

    @property
    def dtype(self):
        return self.parts[0].dtype

    def materialize(self) -> torch.Tensor:
        return torch.cat([(p.tensor.repeat(p.times, 1, 1, 1) if isinstance(p, BatchRepeat) else
                           (p.expand() if isinstance(p, BatchExpand) else p)).contiguous() for p in self.parts], dim=1)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.cat:
            seq = args[0] if args else kwargs["tensors"]
            dim = args[1] if len(args) > 1 else kwargs.get("dim", kwargs.get("axis", 0))
            if dim == 1:
                return LazyCat(list(seq))

        def mat(o):
            if isinstance(o, LazyCat):
                return o.materialize()
            if isinstance(o, (list, tuple)):
                return type(o)(mat(i) for i in o)
            return o
        return func(*mat(args), **{k: mat(v) for k, v in kwargs.items()})


class BatchRepeat:
    """``tensor.repeat(times, 1, 1, 1)`` that is never materialised: a conv source whose images repeat along the
    batch (kernel reads image b % B).  Used by utils/evaluate.py to share the encoder features of a trajectory
    among its K goal samples.  Inference only (no gradient flows into it)."""

    def __init__(self, tensor: torch.Tensor, times: int):
        if tensor.dim() != 4 or times < 1:
            raise ValueError("BatchRepeat: expected a BxCxHxW tensor and times >= 1")
        self.tensor, self.times = tensor.detach(), int(times)

    @property
    def shape(self):
        t = self.tensor
        return torch.Size((t.shape[0] * self.times, t.shape[1], t.shape[2], t.shape[3]))

    def dim(self):
        return 4

    @property
    def device(self):
        return self.tensor.device

    @property
    def dtype(self):
        return self.tensor.dtype


class _BatchExpandFn(torch.autograd.Function):
    """x [1,C,H,W] -> its stride-0 expansion to B images; backward: the sum of the B per-image gradients by ynet_batch_sum."""

    @staticmethod
    def forward(ctx, x, batch):
        ctx.batch = int(batch)
        return x.expand(ctx.batch, -1, -1, -1)

    @staticmethod
    def backward(ctx, g):
        B = ctx.batch
        n = g[0].numel()
        if not (g.is_cuda and g.dtype == torch.float32 and n % 4 == 0):
            return g.sum(dim=0, keepdim=True), None
        g = g.contiguous()
        out = torch.empty((1,) + tuple(g.shape[1:]), device=g.device, dtype=torch.float32)
        lib = _lib()
        L.check(lib.ynet_batch_sum(g.data_ptr(), out.data_ptr(), B, n, n, _stream()), lib)
        return out, None


class BatchExpand:
    """``tensor.expand(B, -1, -1, -1)`` of a one-image tensor, expanded anew by EVERY convolution that reads it (ops.conv2d): each
    consumer then has its own expand node in the autograd graph, whose backward sums that consumer's per-image gradient over the batch
    -- the sums of two consumers are added as ONE-image tensors.  With a single expanded tensor shared by the two decoders autograd adds
    their full per-image gradients first (a B x C x H x W add: 0.43 ms at the top level of C4) and sums over the batch afterwards."""

    def __init__(self, tensor: torch.Tensor, batch: int):
        if tensor.dim() != 4 or tensor.shape[0] != 1 or batch < 1:
            raise ValueError("BatchExpand: expected a 1xCxHxW tensor and a batch size >= 1")
        self.tensor, self.batch = tensor, int(batch)

    @property
    def shape(self):
        t = self.tensor
        return torch.Size((self.batch, t.shape[1], t.shape[2], t.shape[3]))

    def dim(self):
        return 4

    @property
    def device(self):
        return self.tensor.device

    @property
    def dtype(self):
        return self.tensor.dtype

    def expand(self):
        return _BatchExpandFn.apply(self.tensor, self.batch)

    def record_stream(self, stream):
        self.tensor.record_stream(stream)


def batch_repeat(x, times: int):
    """Repeat a tensor / LazyCat along the batch without materialising it (conv sources only)."""
    if times == 1:
        return x
    if isinstance(x, LazyCat):
        return LazyCat([BatchRepeat(p, times) for p in x.parts])
    return BatchRepeat(x, times)


def lazy_cat(parts: Sequence[torch.Tensor]):
    return parts[0] if len(parts) == 1 and not isinstance(parts[0], LazyCat) else LazyCat(parts)


def _parts(x) -> List[torch.Tensor]:
    return list(x.parts) if isinstance(x, LazyCat) else [x]


def _plane_desc(t: torch.Tensor, what: str):
    """-> (tensor kept alive, C, batch stride) with planes contiguous and channel stride H*W."""
    _need_gpu(t, what)
    if t.dim() != 4:
        raise ValueError(f"{what}: expected BxCxHxW, got {tuple(t.shape)}")
    b, c, h, w = t.shape
    st = t.stride()
    ok = (w == 1 or st[3] == 1) and (h == 1 or st[2] == w) and (c == 1 or st[1] == h * w)
    if not ok:
        t = t.contiguous()
        st = t.stride()
    return t, c, (st[0] if b > 1 else c * h * w)


def _arrays(descs):
    n = len(descs)
    ptrs = (_VP * n)(*[d[0] for d in descs])
    cs = (ctypes.c_int * n)(*[d[1] for d in descs])
    bss = (ctypes.c_longlong * n)(*[d[2] for d in descs])
    return ctypes.cast(ptrs, L.PP), cs, bss


def _bmods(descs):
    """Optional 4th field of a source descriptor: batch modulus (0 = none)."""
    if not any(len(d) > 3 and d[3] for d in descs):
        return None
    return (ctypes.c_int * len(descs))(*[(d[3] if len(d) > 3 else 0) for d in descs])


# ------------------------------------------------------------------------------------------------
# convolution (+ fused concat / bias / ReLU / LoRA) with hand-written backward
# ------------------------------------------------------------------------------------------------
def pack_weight(w: torch.Tensor, mode: int) -> torch.Tensor:
    _need_gpu(w, "pack_weight")
    cout, cin, k, _ = w.shape
    lib = _lib()
    out = torch.empty(lib.ynet_packed_weight_floats(cout, cin, k, mode), device=w.device, dtype=torch.float32)
    L.check(lib.ynet_pack_weight(w.data_ptr(), out.data_ptr(), cout, cin, k, mode, _stream()), lib)
    return out


def lora_compose(w, lora_a, lora_b, scale: float) -> torch.Tensor:
    for t, n in ((w, "weight"), (lora_a, "lora_A"), (lora_b, "lora_B")):
        _need_gpu(t, "lora_compose " + n)
    cout, cin, k, _ = w.shape
    r = lora_a.shape[0] // k
    if tuple(lora_a.shape) != (r * k, cin * k) or tuple(lora_b.shape) != (cout * k, r * k):
        raise ValueError(f"lora_compose: lora_A {tuple(lora_a.shape)} / lora_B {tuple(lora_b.shape)} do not fit "
                         f"weight {tuple(w.shape)}")
    out = torch.empty_like(w, memory_format=torch.contiguous_format)
    lib = _lib()
    L.check(lib.ynet_lora_compose(w.contiguous().data_ptr(), lora_a.contiguous().data_ptr(),
                                  lora_b.contiguous().data_ptr(), scale, out.data_ptr(), cout, cin, k, r, _stream()), lib)
    return out


def lora_compose_pack(w, lora_a, lora_b, scale: float, bufs=None):
    """Both packed filters (forward, dgrad) of W + s * (lora_B @ lora_A).view(W.shape) in one launch; ``bufs`` = the
    pair returned by an earlier call for the same layer (their zero padding is kept)."""
    for t, n in ((w, "weight"), (lora_a, "lora_A"), (lora_b, "lora_B")):
        _need_gpu(t, "lora_compose_pack " + n)
    cout, cin, k, _ = w.shape
    r = lora_a.shape[0] // k
    if tuple(lora_a.shape) != (r * k, cin * k) or tuple(lora_b.shape) != (cout * k, r * k):
        raise ValueError(f"lora_compose: lora_A {tuple(lora_a.shape)} / lora_B {tuple(lora_b.shape)} do not fit "
                         f"weight {tuple(w.shape)}")
    lib = _lib()
    if bufs is None:
        bufs = tuple(torch.zeros(lib.ynet_packed_weight_floats(cout, cin, k, mode), device=w.device, dtype=torch.float32)
                     for mode in (0, 1))
    L.check(lib.ynet_lora_compose_pack(w.contiguous().data_ptr(), lora_a.contiguous().data_ptr(),
                                       lora_b.contiguous().data_ptr(), scale, bufs[0].data_ptr(), bufs[1].data_ptr(),
                                       cout, cin, k, r, _stream()), lib)
    return bufs


def lora_grad(dw, lora_a, lora_b, scale: float):
    cout, cin, k, _ = dw.shape
    r = lora_a.shape[0] // k
    d_a, d_b = torch.empty_like(lora_a), torch.empty_like(lora_b)
    lib = _lib()
    L.check(lib.ynet_lora_grad(dw.data_ptr(), lora_a.contiguous().data_ptr(), lora_b.contiguous().data_ptr(), scale,
                               d_a.data_ptr(), d_b.data_ptr(), cout, cin, k, r, _stream()), lib)
    return d_a, d_b


import os as _os
import weakref

_conv_ws = {}
_side_streams = {}
# goal / trajectory decoders on two streams (utils/train_epoch.py); YNET_SERIAL_DECODERS=1 or setting this
# to False runs them back to back on the current stream (used when timing kernels in isolation)
overlap_decoders = _os.environ.get("YNET_SERIAL_DECODERS", "0") != "1"


def side_streams(device, which=None):
    """Persistent side streams per device: (goal decoder, trajectory decoder) -- or stream number `which` (2: read-out)."""
    dev = torch.device(device)
    if not overlap_decoders:
        cur = torch.cuda.current_stream(dev)
        return (cur, cur) if which is None else cur
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _side_streams:
        _side_streams[key] = tuple(torch.cuda.Stream(device=dev) for _ in range(3))
    return _side_streams[key][:2] if which is None else _side_streams[key][which]


# Adapter gradients beside the data-gradient chain (mosa_*: only lora_A / lora_B train).  The encoder's backward is a strict chain
# [pool backward -> dgrad -> dgrad -> ...] of small-map launches that leave most of the chip idle, and nothing on that chain waits for
# a layer's adapter gradient: inside fold_skip_gradients() it is computed on a fourth stream (a branch of the captured step) and
# written straight to lora_A.grad / lora_B.grad; the context's exit joins the branch.  (Opt-in with the context like the skip fold;
# outside it -- torch.autograd.grad, hooks, accumulation into an existing .grad -- the gradients flow through autograd as before.)
wgrad_branch = False
_wgrad_branch_allowed = _os.environ.get("YNET_WGRAD_BRANCH", "1") != "0"
_wgrad_streams = {}
_wgrad_pending = {}


def _wgrad_stream(device):
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _wgrad_streams:
        _wgrad_streams[key] = torch.cuda.Stream(device=dev)
    return key, _wgrad_streams[key]


_wgrad_adds = {}       # branch key -> [(existing .grad, the gradient to add into it)]: added in ONE launch when the branch is joined


def join_wgrad_branch():
    """The current stream of every device with adapter gradients in flight waits for them.  Gradients that go INTO an existing .grad (the data-parallel flat buffer's
    views, an accumulation) are added here, on the branch, in one multi-tensor launch -- not as two little launches behind every layer's filter gradient (18 of them on
    the tail of an N-rank step)."""
    for key, side in list(_wgrad_pending.items()):
        adds = _wgrad_adds.pop(key, None)
        if adds:
            with torch.cuda.stream(side), torch.no_grad():
                torch._foreach_add_([a for a, _ in adds], [b for _, b in adds])
        torch.cuda.current_stream(torch.device(key[0], key[1])).wait_stream(side)
    _wgrad_pending.clear()
    _wgrad_adds.clear()


_wino_allowed = _os.environ.get("YNET_WINOGRAD", "1") != "0"     # YNET_WINOGRAD=0: every convolution takes the implicit-GEMM kernels
_wino_eval = _os.environ.get("YNET_WINOGRAD_EVAL", "1") != "0"  # YNET_WINOGRAD_EVAL=0: only under autograd (training steps) -- see conv2d's note
wino_stats = {"launches": 0}
# development switches of tests/wino_fp64.py (which launch family carries a deviation of evaluate()'s sweep): under no_grad only
_wino_eval_min_hw = 0        # smallest H * W of a map that takes a Winograd launch
_wino_cat_eval = True        # the concatenated-source / shared-term launches
_wino_plain_eval = True      # the one-source launches


def _wino_made(entry):
    """A Winograd filter entry just made on the current stream: remembers that stream and an event behind the transform launch."""
    if not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
        return entry + (None, None)
    ev = torch.cuda.Event()
    ev.record()
    return entry + (torch.cuda.current_stream().cuda_stream, ev)


def _wino_ready(entry):
    """Filters are transformed lazily at a layer's first Winograd launch -- possibly on one of evaluate()'s two sweep streams, with the
    other stream's cache hit a few microseconds behind and no dependency on the transform: a hit from another stream waits for the
    maker's event until that has completed (the first call of a process only; never inside a capture, which follows an eager pass)."""
    stream, ev = entry[-2], entry[-1]
    if ev is not None and not torch.cuda.is_current_stream_capturing():
        cur = torch.cuda.current_stream()
        if cur.cuda_stream != stream and not ev.query():
            cur.wait_event(ev)
    return entry


def winograd_filter(wp: torch.Tensor, cin: int, cout: int, col0: int = 0, cols_total: int = None) -> torch.Tensor:
    """The Winograd-domain form (G g G^T, MFMA fragment order) of the output channels [col0, col0 + cout) of a packed filter of
    pack_weight with cols_total output channels (ynet_winograd_filter)."""
    lib = _lib()
    u = torch.empty(lib.ynet_winograd_filter_floats(cin, cout), device=wp.device, dtype=torch.float32)
    L.check(lib.ynet_winograd_filter(wp.data_ptr(), u.data_ptr(), cin, cout, col0, cout if cols_total is None else cols_total, _stream()), lib)
    return u


def conv2d_winograd_raw(src, u, bias, dst, cin, cout, B, H, W, relu, relu_of=None, wbits_out=None, relu_wbits=None, s2d=False):
    """src / dst: (ptr, batch_stride in floats); u: winograd_filter(...) of the layer's packed filter; relu_of: (ptr, batch_stride) of the
    post-ReLU activation whose backward is applied to dst (ynet_conv2d_winograd_dgrad_relu: a data gradient, no bias / ReLU).  wbits_out: an
    int32 tensor of ynet_winograd_relu_bits_words(B, H, W) words that receives the 1-bit mask of the (ReLU, 32-channel) output; relu_wbits:
    such a mask, applied to a data gradient in place of relu_of's activation fetch."""
    lib = _lib()
    if s2d:      # the plain data gradient stored space-to-depth (ynet_conv2d_winograd_s2d: the gradient of an up-convolution's output)
        if bias is not None or relu or relu_of is not None or wbits_out is not None or relu_wbits is not None:
            raise ValueError("conv2d_winograd_raw: s2d is for a plain data gradient")
        L.check(lib.ynet_conv2d_winograd_s2d(src[0], src[1], u.data_ptr(), dst[0], dst[1], cin, cout, B, H, W, _stream()), lib)
        return
    if wbits_out is not None:
        if cout != 32 or not relu or relu_of is not None or relu_wbits is not None:
            raise ValueError("conv2d_winograd_raw: wbits_out is for a forward ReLU launch with 32 outputs")
        L.check(lib.ynet_conv2d_winograd_relu_bits(src[0], src[1], u.data_ptr(), bias.data_ptr() if bias is not None else None, dst[0], dst[1], cin, B, H, W,
                                                   wbits_out.data_ptr(), _stream()), lib)
        return
    if relu_wbits is not None:
        if cout != 32 or bias is not None or relu or relu_of is not None:
            raise ValueError("conv2d_winograd_raw: relu_wbits is for a data gradient with 32 outputs")
        L.check(lib.ynet_conv2d_winograd_dgrad_relu_bits(src[0], src[1], u.data_ptr(), dst[0], dst[1], relu_wbits.data_ptr(), cin, B, H, W, _stream()), lib)
        return
    if relu_of is not None:
        if bias is not None or relu:
            raise ValueError("conv2d_winograd_raw: relu_of is for a data gradient")
        L.check(lib.ynet_conv2d_winograd_dgrad_relu(src[0], src[1], u.data_ptr(), dst[0], dst[1], relu_of[0], relu_of[1], cin, cout, B, H, W, _stream()), lib)
        return
    L.check(lib.ynet_conv2d_winograd(src[0], src[1], u.data_ptr(), bias.data_ptr() if bias is not None else None, dst[0], dst[1],
                                     cin, cout, B, H, W, 1 if relu else 0, _stream()), lib)


def conv2d_winograd_pred_bce_raw(src, u, bias, pred_wp, pred_bias, pred_cout, pos, tmpl, logits, loss, dx, ws, B, H, W, expected_grad):
    """[conv3x3 32 -> 32 + ReLU -> 1 x 1 predictor -> BCE-with-logits -> predictor's data gradient] in one launch (ynet_conv2d_winograd_pred_bce_blob): src = (ptr, channels,
    batch_stride) of the convolution's input, u its Winograd-domain filter; pos / tmpl the blob form of the target (see _blob_target)."""
    lib = _lib()
    L.check(lib.ynet_conv2d_winograd_pred_bce_blob(src[0], src[2], u.data_ptr(), bias.data_ptr() if bias is not None else None, pred_wp.data_ptr(),
                                                   pred_bias.data_ptr() if pred_bias is not None else None, pred_cout, pos.data_ptr(), tmpl.blob.data_ptr(),
                                                   tmpl.blob.shape[0], tmpl.size, logits.data_ptr(), loss.data_ptr(), dx.data_ptr(), 32 * H * W, ws.data_ptr(), B, H, W,
                                                   expected_grad, _stream()), lib)


def conv2d_winograd_split_raw(src, u, dst0, dst0_s2d, dst1, cin, B, H, W):
    """The plain data gradient with the destinations [16 channels, 32 channels] in one launch (ynet_conv2d_winograd_split): src / dst0 / dst1 = (ptr, batch_stride),
    u = winograd_filter(wp, cin, 48, col0, cols_total); dst0 space-to-depth when dst0_s2d."""
    lib = _lib()
    L.check(lib.ynet_conv2d_winograd_split(src[0], src[1], u.data_ptr(), dst0[0], dst0[1], 1 if dst0_s2d else 0, dst1[0], dst1[1], cin, B, H, W, _stream()), lib)


def conv2d_winograd_cat_raw(srcs, u, bias, dst, B, H, W, relu, addend=None, pool=None, wbits_out=None, pool_code=None):
    """srcs: [(ptr, channels, batch_stride)] (at most three, 56 padded channels); dst: (ptr, batch_stride), 32 channels; u: the filter in
    ynet_winograd_filter_cat's layout for these sources; addend: (ptr, image_stride, modulus) of a term added in front of the ReLU;
    pool: (ptr, batch_stride) of the 2 x 2 max-pooled copy of the output, written by the same launch; wbits_out: receives the 1-bit mask of
    the (ReLU) output (see conv2d_winograd_raw)."""
    lib = _lib()
    sp, sc, sb = _arrays(srcs)
    b = bias.data_ptr() if bias is not None else None
    if wbits_out is not None:
        if pool is not None or not relu:
            raise ValueError("conv2d_winograd_cat_raw: wbits_out is for a ReLU launch without the pooled copy")
        L.check(lib.ynet_conv2d_winograd_cat_relu_bits(sp, sc, sb, len(srcs), u.data_ptr(), b, dst[0], dst[1], B, H, W, addend[0] if addend else None,
                                                       addend[1] if addend else 0, addend[2] if addend else 0, wbits_out.data_ptr(), _stream()), lib)
        return
    if pool is not None and pool_code is not None:      # (+ one byte per pooled block for the pool's backward: ynet_maxpool2_bwd_add_code)
        if not relu:
            raise ValueError("conv2d_winograd_cat_raw: pool_code is for a ReLU launch")
        L.check(lib.ynet_conv2d_winograd_cat_pool_code(sp, sc, sb, len(srcs), u.data_ptr(), b, dst[0], dst[1], pool[0], pool[1], pool_code.data_ptr(), B, H, W,
                                                       _stream()), lib)
        return
    if pool is not None:
        L.check(lib.ynet_conv2d_winograd_cat_pool(sp, sc, sb, len(srcs), u.data_ptr(), b, dst[0], dst[1], pool[0], pool[1], 32, B, H, W,
                                                  1 if relu else 0, _stream()), lib)
    elif addend is None:
        L.check(lib.ynet_conv2d_winograd_cat(sp, sc, sb, len(srcs), u.data_ptr(), b, dst[0], dst[1], 32, B, H, W, 1 if relu else 0, _stream()), lib)
    else:
        L.check(lib.ynet_conv2d_winograd_cat_add(sp, sc, sb, len(srcs), u.data_ptr(), b, dst[0], dst[1], 32, B, H, W, 1 if relu else 0,
                                                 addend[0], addend[1], addend[2], _stream()), lib)


_pool_code_allowed = _os.environ.get("YNET_POOL_CODE", "1") != "0"      # YNET_POOL_CODE=0: the max-pool's backward re-reads the full-resolution activation (no arg-max / ReLU byte per block)
_wino_relu_bits_allowed = _os.environ.get("YNET_WINOGRAD_RELU_BITS", "1") != "0"      # YNET_WINOGRAD_RELU_BITS=0: the Winograd data gradients fetch the float activation (no 1-bit mask in their tiling)
_wino16_allowed = _os.environ.get("YNET_WINOGRAD16", "1") != "0"     # YNET_WINOGRAD16=0: no conv_wino16_kernel launches (round 5's slice form)
# YNET_WINOGRAD16_SLICE16=1: 16-output-channel launches (32 -> 16 at 256^2) on the slice form too -- measured SLOWER there than
# conv_wino_kernel<1, 4> (141 against 134 us at B 32: the launch streams 402 MB, and six staged rows per unit do not pay for one slice)
_wino16_for_16 = _os.environ.get("YNET_WINOGRAD16_SLICE16", "0") != "0"
_split48_allowed = _os.environ.get("YNET_WINOGRAD_SPLIT48", "1") != "0"      # a [16, 32]-channel data gradient as ONE launch (ynet_conv2d_winograd_split)


def conv2d_winograd16_raw(srcs, u, bias, dst, cout, B, H, W, relu, relu_of=None, addend=None, pool=None):
    """The slice form of the Winograd convolution (ynet_conv2d_winograd16: 16 output channels per workgroup, two row pairs per wave):
    srcs [(ptr, channels, batch_stride)] (at most three, 84 padded channels); dst (ptr, batch_stride) of cout in {16, 32, 64, 128} channels;
    u: the filter in ynet_winograd16_filter's layout; at most one of relu_of (ptr, batch_stride) -- a data gradient written through
    that activation's ReLU backward --, addend (ptr, image_stride, modulus) and pool (ptr, batch_stride)."""
    lib = _lib()
    sp, sc, sb = _arrays(srcs)
    L.check(lib.ynet_conv2d_winograd16(sp, sc, sb, len(srcs), u.data_ptr(), bias.data_ptr() if bias is not None else None, dst[0], dst[1], cout, B, H, W,
                                       1 if relu else 0, relu_of[0] if relu_of else None, relu_of[1] if relu_of else 0,
                                       addend[0] if addend else None, addend[1] if addend else 0, addend[2] if addend else 0,
                                       pool[0] if pool else None, pool[1] if pool else 0, _stream()), lib)


def _wino16_supported(srcs_c, cout, B, H, W, K=3):
    if not (_wino16_allowed and _wino_allowed and K == 3 and 1 <= len(srcs_c) <= 3):
        return False
    return bool(_lib().ynet_conv2d_winograd16_supported(B, H, W, (ctypes.c_int * len(srcs_c))(*srcs_c), len(srcs_c), cout, K))


def _wino16_filter(wino, wp, row0, cs, cout, col0, ctot):
    """The slice-major Winograd filter (ynet_winograd16_filter) of output channels [col0, col0 + cout) of the packed filter wp for sources of
    cs channels each, from the filter's input-channel row row0 on; kept in the layer's cache."""
    lib = _lib()
    cache, what = wino
    key = "wino16_%s_%d_%d_%d" % (what, row0, col0, cout)
    ent = cache.get(key)
    if ent is None or ent[0] is not wp or ent[2] != cs:
        ca = (ctypes.c_int * len(cs))(*cs)
        u = torch.empty(lib.ynet_winograd16_filter_floats(ca, len(cs), cout), device=wp.device, dtype=torch.float32)
        cols_pad = -(-ctot // 64) * 64
        L.check(lib.ynet_winograd16_filter(wp.data_ptr() + 4 * row0 * 9 * cols_pad, u.data_ptr(), ca, len(cs), cout, col0, ctot, _stream()), lib)
        ent = cache[key] = _wino_made((wp, u, cs))
    _wino_ready(ent)
    return ent


def _wino16(wino, wp, row0, srcs, bias, dst, cout, col0, ctot, B, H, W, relu, relu_of=None, addend=None, pool=None):
    """One ynet_conv2d_winograd16 launch over output channels [col0, col0 + cout) of the packed filter wp (ctot output channels in all)
    and its input-channel rows from row0 on, the transformed filter kept in the layer's cache."""
    ent = _wino16_filter(wino, wp, row0, tuple(s_[1] for s_ in srcs), cout, col0, ctot)
    conv2d_winograd16_raw(srcs, ent[1], None if bias is None else bias[col0:col0 + cout], dst, cout, B, H, W, relu, relu_of=relu_of, addend=addend, pool=pool)
    wino_stats["launches"] += 1
    wino_stats["launches16"] = wino_stats.get("launches16", 0) + 1


def _pad4(c):
    return (c + 3) & ~3


# YNET_CONV_AUTO=0: the launches of a convolution are composed HERE (_conv2d_raw_py, the round-5 dispatcher, kept as the instrumentable
# twin: bench.py's ConvTimer brackets every launch of a call with its own HIP-event pair through it) instead of by ynet_conv2d_auto in the
# library.  Both choose the same kernels for the same operands (tests/test_gpu_kernels.py::test_conv2d_auto_takes_the_launches_of_the_python_dispatcher).
conv_auto = _os.environ.get("YNET_CONV_AUTO", "1") != "0"
_AUTO_TAG_CAT = {10: ("winograd_cat:2,3", "winograd_cat:2,6|code"), 40: ("winograd_cat:2,2", "winograd_cat:2,5|wbits"), 41: ("winograd_cat:2,0", "winograd_cat:2,4|wbits"),
                 30: ("winograd_cat:2,2", "winograd_cat:2,5|wbits")}


def _auto_tag(tk):
    """The tag the Python dispatcher returns for the same launches (callers and bench.py read it), from a YnetConvTaken."""
    if tk.family == 0:
        return None
    if tk.family == 1:
        return "winograd:" + "+".join("%d,%d,%d" % tuple(tk.tmpl[i]) for i in range(tk.nlaunch)) + ("|wbits" if tk.wrote_wbits else "")
    if tk.family == 2:
        return _AUTO_TAG_CAT[tk.variant][1 if (tk.wrote_wbits or tk.wrote_pool_code) else 0]
    if tk.family == 3:
        return "winograd16:" + "+".join("%d" % tk.tmpl[i][0] for i in range(tk.nlaunch))
    return "winograd_up:%d" % tk.family


def conv2d_auto_raw(srcs, mask, wp, bias, dsts, B, H, W, K, relu, relu_of=None, pooled=None, bits_out=None, relu_bits=None, wino=None, wbits_out=None,
                    relu_wbits=None, pool_code=None, addend=None, upsample2x=False, wp_version=1, dst_s2d=None):
    """ONE call of ynet_conv2d_auto (include/ynet_hip.h): the library chooses the kernel family, splits wide layers and keeps the transformed
    filters in a cache that lives in the layer's filter cache `wino[0]` under "auto_<direction>".  Operands as conv2d_raw; addend: (ptr,
    image_stride, modulus); upsample2x: the source is the low-resolution map (H, W are the up-sampled size).  Returns (tag, YnetConvTaken)."""
    lib = _lib()
    d = L.ConvAuto()
    d.nsrc, d.ndst = len(srcs), len(dsts)
    for i, s_ in enumerate(srcs):
        d.src[i], d.src_c[i], d.src_bs[i] = s_[0], s_[1], s_[2]
        d.src_bmod[i] = s_[3] if len(s_) > 3 else 0
    for i, t_ in enumerate(dsts):
        d.dst[i], d.dst_c[i], d.dst_bs[i] = t_[0], t_[1], t_[2]
        if dst_s2d is not None and dst_s2d[i]:
            d.dst_s2d[i] = 1
    if mask:
        d.mask, d.mask_bs = mask[0], mask[1]
    d.wp, d.bias = wp.data_ptr(), (bias.data_ptr() if bias is not None else None)
    d.B, d.H, d.W, d.K, d.relu, d.upsample2x = B, H, W, K, 1 if relu else 0, 1 if upsample2x else 0
    if relu_of is not None:
        d.relu_of, d.relu_of_bs = relu_of[0], relu_of[1]
    if pooled is not None:
        d.pooled, d.pooled_bs = pooled[0], pooled[1]
        if pool_code is not None:
            d.pool_code = pool_code.data_ptr()
    if addend is not None:
        d.addend, d.addend_bs, d.addend_bmod = addend[0], addend[1], addend[2]
    d.bits_out, d.relu_bits = bits_out, relu_bits
    if wbits_out is not None:
        d.wbits_out = wbits_out.data_ptr()
    if relu_wbits is not None:
        d.relu_wbits = relu_wbits.data_ptr()
    flags = 0
    if wino is None or not _wino_allowed:
        flags |= L.AUTO_NO_WINOGRAD
    if not _wino16_allowed:
        flags |= L.AUTO_NO_WINOGRAD16
    if _wino16_for_16:
        flags |= L.AUTO_WINOGRAD16_FOR_16
    if not _pool_code_allowed:
        flags |= L.AUTO_NO_POOL_CODE
    if not _split48_allowed:
        flags |= L.AUTO_NO_SPLIT48
    d.flags = flags
    ent = None
    if wino is not None and not (flags & L.AUTO_NO_WINOGRAD):
        cache, what = wino
        need = lib.ynet_conv2d_auto_cache_floats(ctypes.byref(d))
        if need < 0:
            L.check(1, lib)
        if need > 0:
            key = "auto_" + what
            ent = cache.get(key)
            if ent is None or ent[0] is not wp or ent[1].numel() < need:
                ent = cache[key] = [wp, torch.empty(need, device=wp.device, dtype=torch.float32), (ctypes.c_ulonglong * 2)(0, 0), None, None]
            elif ent[4] is not None and not torch.cuda.is_current_stream_capturing():
                # (made on another stream a few microseconds ago -- evaluate()'s two sweep streams: wait for the maker's event, as _wino_ready)
                cur = torch.cuda.current_stream()
                if cur.cuda_stream != ent[3] and not ent[4].query():
                    cur.wait_event(ent[4])
            d.cache, d.cache_floats, d.cache_tag, d.wp_version = ent[1].data_ptr(), ent[1].numel(), ent[2], int(wp_version)
    if B * H * W <= 65536:                                       # small maps only (see ynet_conv2d_workspace_floats)
        nws = lib.ynet_conv2d_auto_workspace_floats(ctypes.byref(d))
        if nws > 0:
            key = (wp.device, torch.cuda.current_stream().cuda_stream)   # grow-only scratch per stream (stream-ordered reuse)
            ws = _conv_ws.get(key)
            if ws is None or ws.numel() < nws:
                ws = _conv_ws[key] = torch.empty(nws, device=wp.device, dtype=torch.float32)
            d.workspace, d.workspace_floats = ws.data_ptr(), nws
    tk = L.ConvTaken()
    L.check(lib.ynet_conv2d_auto(ctypes.byref(d), ctypes.byref(tk), _stream()), lib)
    if tk.transformed and ent is not None and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        ev = torch.cuda.Event()
        ev.record()
        ent[3], ent[4] = torch.cuda.current_stream().cuda_stream, ev
    if tk.family:
        wino_stats["launches"] += tk.nlaunch
        if tk.family in (3, 5):
            wino_stats["launches16"] = wino_stats.get("launches16", 0) + tk.nlaunch
    return _auto_tag(tk), tk


_dgrad_family_memo = {}


def dgrad_relu_family(B, H, W, dy_c, dx_c, K=3):
    """The kernel family (YnetConvTaken.family: 0 implicit GEMM, 1 conv_wino_kernel, 3 conv_wino16_kernel) that will write the data gradient
    [B, dx_c, H, W] of a dy_c-channel gradient THROUGH the ReLU backward of the layer below -- asked of the dispatcher itself
    (ynet_conv2d_auto_plan with the operands of that launch: one aligned source, one aligned destination, relu_of), so that the forward pass of
    a conv -> ReLU -> conv chain and the backward pass agree with the launch that is finally taken."""
    key = (B, H, W, dy_c, dx_c, K, _wino_allowed, _wino16_allowed, _wino16_for_16)
    fam = _dgrad_family_memo.get(key)
    if fam is None:
        d, tk = L.ConvAuto(), L.ConvTaken()
        d.nsrc = d.ndst = 1
        d.src[0], d.src_c[0], d.src_bs[0] = 256, dy_c, dy_c * H * W          # (alignment stand-ins: the callers check the real tensors)
        d.dst[0], d.dst_c[0], d.dst_bs[0] = 256, dx_c, dx_c * H * W
        d.relu_of, d.relu_of_bs = 256, dx_c * H * W
        d.wp = 256
        d.B, d.H, d.W, d.K = B, H, W, K
        d.flags = (0 if _wino_allowed else L.AUTO_NO_WINOGRAD) | (0 if _wino16_allowed else L.AUTO_NO_WINOGRAD16) | (L.AUTO_WINOGRAD16_FOR_16 if _wino16_for_16 else 0)
        lib = _lib()
        L.check(lib.ynet_conv2d_auto_plan(ctypes.byref(d), ctypes.byref(tk)), lib)
        fam = _dgrad_family_memo[key] = int(tk.family)
    return fam


def conv2d_raw(srcs, mask, wp, bias, dsts, B, H, W, K, relu, relu_of=None, pooled=None, bits_out=None, relu_bits=None, wino=None, wbits_out=None,
               relu_wbits=None, pool_code=None, dst_s2d=None, info=None):
    """dst_s2d (a list of flags, one per destination) / info (a dict that receives "wrote_s2d", the bit mask of the destinations written space-to-depth):
    the gradient of an up-convolution's output, see upconv_s2d_tables; honoured where ONE plain Winograd launch writes the whole destination.
    The convolution / data-gradient launch set of one layer: srcs / dsts lists of (ptr, channels, batch_stride[, batch modulus]); the operands
    are those of _conv2d_raw_py below (the round-5 dispatcher, whose docstring describes them and the returned tag).  Since round 6 the
    composition happens in the library (ynet_conv2d_auto, csrc/conv_auto.cpp); this function only applies evaluate()'s development gates."""
    if not conv_auto:
        return _conv2d_raw_py(srcs, mask, wp, bias, dsts, B, H, W, K, relu, relu_of=relu_of, pooled=pooled, bits_out=bits_out, relu_bits=relu_bits, wino=wino,
                              wbits_out=wbits_out, relu_wbits=relu_wbits, pool_code=pool_code, dst_s2d=dst_s2d, info=info)
    if wino is not None and not torch.is_grad_enabled() and (
            H * W < _wino_eval_min_hw or not (_wino_plain_eval if len(srcs) == 1 and srcs[0][1] in (16, 32) else _wino_cat_eval)):
        wino = None
    tag, tk = conv2d_auto_raw(srcs, mask, wp, bias, dsts, B, H, W, K, relu, relu_of=relu_of, pooled=pooled, bits_out=bits_out, relu_bits=relu_bits, wino=wino,
                              wbits_out=wbits_out, relu_wbits=relu_wbits, pool_code=pool_code, dst_s2d=dst_s2d)
    if info is not None:
        info["wrote_s2d"] = int(tk.wrote_s2d)
    return tag


def _conv2d_raw_py(srcs, mask, wp, bias, dsts, B, H, W, K, relu, relu_of=None, pooled=None, bits_out=None, relu_bits=None, wino=None, wbits_out=None,
                   relu_wbits=None, pool_code=None, dst_s2d=None, info=None):
    """srcs / dsts: lists of (ptr, channels, batch_stride); mask: (ptr, batch_stride) or None.  relu_of: (ptr, batch_stride) of the
    post-ReLU activation whose backward is applied to the single destination (ynet_conv2d_dgrad_relu), or None.  pooled: (ptr,
    batch_stride) of a second output, the 2 x 2 max-pooled copy of the single destination (ynet_conv2d_pool), or None.
    bits_out: address of the 1-bit activation mask a forward ReLU convolution writes next to its output (ynet_conv2d_relu_bits);
    relu_bits: address of such a mask, applied to the single destination of a data gradient (ynet_conv2d_dgrad_relu_bits).
    wino: (the layer's filter cache, "fwd" | "dgrad") -- a plain launch (one source, one destination, no mask, no epilogue variant)
    of a shape ynet_conv2d_winograd_supported admits takes the Winograd F(2x2, 3x3) kernel, its transformed filter kept in that
    cache next to the packed one; returns "winograd:<NCB>,<NCH>,<EM>[+...]" (one group per launch: the template arguments of
    conv_wino_kernel) or "winograd_cat:2,<epilogue>" then, None otherwise.
    wbits_out / relu_wbits: the Winograd-native 1-bit ReLU mask (int32 tensors of ynet_winograd_relu_bits_words words) -- wbits_out is
    WRITTEN only if the launch taken is a 32-output Winograd one (the returned tag then ends in "|wbits": the caller keeps the tensor
    only in that case); relu_wbits gates a 32-channel data gradient in place of relu_of's activation fetch.  pool_code (with pooled): a uint8
    tensor [B, 32, H/2, W/2] that receives arg-max + ReLU bits of every pooled block when the launch taken is the Winograd one for
    concatenated sources (tag "winograd_cat:2,6|code": the caller keeps the tensor only then)."""
    lib = _lib()
    sp, sc, sb = _arrays(srcs)
    if info is not None:
        info["wrote_s2d"] = 0
    if wino is not None and not torch.is_grad_enabled() and (
            H * W < _wino_eval_min_hw or not (_wino_plain_eval if len(srcs) == 1 and srcs[0][1] in (16, 32) else _wino_cat_eval)):
        wino = None
    if bits_out is not None:
        if len(dsts) != 1 or mask is not None or relu_of is not None or pooled is not None or not relu or _bmods(srcs) is not None:
            raise ValueError("conv2d_raw: bits_out is for a forward ReLU convolution with one destination")
        L.check(lib.ynet_conv2d_relu_bits(sp, sc, sb, len(srcs), wp.data_ptr(), bias.data_ptr() if bias is not None else None,
                                          dsts[0][0], dsts[0][1], dsts[0][2], bits_out, B, H, W, K, _stream()), lib)
        return
    if relu_bits is not None:
        if len(srcs) != 1 or len(dsts) != 1 or bias is not None or relu or relu_of is not None:
            raise ValueError("conv2d_raw: relu_bits is for a data gradient with one source and one destination")
        L.check(lib.ynet_conv2d_dgrad_relu_bits(srcs[0][0], srcs[0][1], srcs[0][2], mask[0] if mask else None, mask[1] if mask else 0,
                                                wp.data_ptr(), dsts[0][0], dsts[0][1], dsts[0][2], relu_bits, B, H, W, K, _stream()), lib)
        return
    if pooled is not None:
        if len(dsts) != 1 or mask is not None or relu_of is not None or _bmods(srcs) is not None:
            raise ValueError("conv2d_raw: pooled is for a forward convolution with one destination")
        if (wino is not None and _wino_allowed and K == 3 and dsts[0][1] == 32 and dsts[0][0] % 8 == 0 and dsts[0][2] % 2 == 0
                and all(len(s_) == 3 and s_[0] % 16 == 0 and s_[2] % 4 == 0 for s_ in srcs)):
            cs = (ctypes.c_int * len(srcs))(*[s_[1] for s_ in srcs])
            if lib.ynet_conv2d_winograd_cat_supported(B, H, W, cs, len(srcs), 32, K):
                # the encoder's conv + ReLU in front of a max-pool: the Winograd launch writes the pooled copy itself (a lane holds the block)
                cache, what = wino
                key = "wino_cat_" + what
                ent = cache.get(key)
                if ent is None or ent[0] is not wp or ent[2] != tuple(cs):
                    u = torch.empty(lib.ynet_winograd_filter_cat_floats(cs, len(srcs), 32), device=wp.device, dtype=torch.float32)
                    L.check(lib.ynet_winograd_filter_cat(wp.data_ptr(), u.data_ptr(), cs, len(srcs), 32, 0, 32, _stream()), lib)
                    ent = cache[key] = _wino_made((wp, u, tuple(cs)))
                _wino_ready(ent)
                code = pool_code if (relu and _pool_code_allowed) else None
                conv2d_winograd_cat_raw(srcs, ent[1], bias, (dsts[0][0], dsts[0][2]), B, H, W, relu, pool=pooled, pool_code=code)
                wino_stats["launches"] += 1
                return "winograd_cat:2,6|code" if code is not None else "winograd_cat:2,3"
        if (wino is not None and K == 3 and dsts[0][0] % 8 == 0 and dsts[0][2] % 2 == 0 and all(len(s_) == 3 and s_[0] % 16 == 0 and s_[2] % 4 == 0 for s_ in srcs)
                and _wino16_supported([s_[1] for s_ in srcs], dsts[0][1], B, H, W)):
            # (64 output channels: the encoder's last 64^2 layer in front of its max-pool)
            _wino16(wino, wp, 0, srcs, bias, (dsts[0][0], dsts[0][2]), dsts[0][1], 0, dsts[0][1], B, H, W, relu, pool=pooled)
            return "winograd16:3"
        L.check(lib.ynet_conv2d_pool(sp, sc, sb, len(srcs), wp.data_ptr(), bias.data_ptr() if bias is not None else None,
                                     dsts[0][0], dsts[0][1], dsts[0][2], pooled[0], pooled[1], B, H, W, K, 1 if relu else 0, _stream()), lib)
        return
    if (wino is not None and _wino_allowed and K == 3 and mask is None and len(srcs) == 1 and len(srcs[0]) == 3
            and srcs[0][0] % 16 == 0 and srcs[0][2] % 4 == 0 and (relu_of is None or (relu_of[0] % 8 == 0 and relu_of[1] % 2 == 0))):
        # destination channels in pieces of 32 / 16 (a 48- or 64-channel data gradient is two launches over slices of the filter; the
        # input is read once per piece -- from L2 --, pieces nobody wants are not computed)
        cin, ctot, HW = srcs[0][1], sum(d[1] for d in dsts), H * W
        pieces, piece_dst, c0 = [], [], 0      # piece_dst: (index of the piece's destination, "the piece is that whole destination")
        for di, (ptr, c, bs) in enumerate(dsts):
            if ptr is None:          # (nobody wants these channels -- e.g. the way-point map's gradient --: not computed)
                c0 += c
                continue
            o = 0
            while c - o >= 16 and (c - o) % 16 == 0:
                n = 32 if c - o >= 32 else 16
                pieces.append((ptr + 4 * o * HW, n, bs, c0 + o))
                piece_dst.append((di, n == c))
                o += n
            if o != c:
                pieces = None
                break
            c0 += c
        if (_wino16_for_16 and pieces and len(pieces) == 1 and pieces[0][1] == 16 and pieces[0][0] % 8 == 0 and pieces[0][2] % 2 == 0
                and _wino16_supported([cin], 16, B, H, W)):
            # 16 output channels (the 32 -> 16 up-convolution at 256^2): the slice form with two row pairs per wave
            ptr, n, bs, col0 = pieces[0]
            _wino16(wino, wp, 0, srcs, bias, (ptr, bs), 16, col0, ctot, B, H, W, relu, relu_of=relu_of)
            return "winograd16:%d" % (1 if relu_of is not None else 0)
        wide = any(d_[0] is not None and d_[1] >= 64 for d_ in dsts)      # (a 64-channel destination: one launch of the slice form, below)
        if (_split48_allowed and pieces and not wide and len(pieces) == 2 and relu_of is None and bias is None and not relu and pieces[0][1] == 16 and piece_dst[0][1]
                and pieces[1][1] == 32 and piece_dst[1][1] and pieces[1][3] == pieces[0][3] + 16 and all(p_[0] % 8 == 0 and p_[2] % 2 == 0 for p_ in pieces)
                and all(lib.ynet_conv2d_winograd_supported(B, H, W, cin, p_[1], K) for p_ in pieces) and lib.ynet_conv2d_winograd_split_supported(B, H, W, cin)):
            # [16, 32] channels of a plain data gradient (cat(up-sampled 16, skip 32[, way-point map]) at the decoders' last level): one launch, dy read once
            cache, what = wino
            col0 = pieces[0][3]
            key = "wino_%s_%d_48" % (what, col0)
            ent = cache.get(key)
            if ent is None or ent[0] is not wp:
                ent = cache[key] = _wino_made((wp, winograd_filter(wp, cin, 48, col0, ctot)))
            _wino_ready(ent)
            s2d = bool(dst_s2d and dst_s2d[piece_dst[0][0]])
            conv2d_winograd_split_raw((srcs[0][0], srcs[0][2]), ent[1], (pieces[0][0], pieces[0][2]), s2d, (pieces[1][0], pieces[1][2]), cin, B, H, W)
            if s2d and info is not None:
                info["wrote_s2d"] |= 1 << piece_dst[0][0]
            wino_stats["launches"] += 1
            return "winograd:3,%d,%d" % (cin // 8, 6 if s2d else 5)
        if (pieces and not wide and len(pieces) <= (2 if relu_of is None else 1) and all(p_[0] % 8 == 0 and p_[2] % 2 == 0 for p_ in pieces)
                and all(lib.ynet_conv2d_winograd_supported(B, H, W, cin, p_[1], K) for p_ in pieces)):
            cache, what = wino
            one32 = len(pieces) == 1 and pieces[0][1] == 32
            wb_out = wbits_out if (one32 and relu and relu_of is None) else None      # the forward launch of a conv -> ReLU -> conv chain writes the mask word
            wb_in = relu_wbits if (one32 and relu_of is not None and not relu and bias is None) else None      # ... and the chain's data gradient applies it
            em = 3 if wb_out is not None else (2 if wb_in is not None else (1 if relu_of is not None else 0))
            # (the gradient of an up-convolution's output, stored space-to-depth: csrc/conv_auto.cpp's rule -- a plain launch that writes the whole destination)
            s2ds = [bool(dst_s2d and dst_s2d[di] and whole and em == 0 and bias is None and not relu) for di, whole in piece_dst]
            tag = ("winograd:" + "+".join("%d,%d,%d" % (p_[1] // 16, cin // 8, 4 if f_ else em) for p_, f_ in zip(pieces, s2ds))
                   + ("|wbits" if wb_out is not None else ""))
            for (ptr, n, bs, col0), (di, whole), s2d in zip(pieces, piece_dst, s2ds):
                key = "wino_%s_%d_%d" % (what, col0, n)
                ent = cache.get(key)
                if ent is None or ent[0] is not wp:
                    ent = cache[key] = _wino_made((wp, winograd_filter(wp, cin, n, col0, ctot)))
                _wino_ready(ent)
                conv2d_winograd_raw((srcs[0][0], srcs[0][2]), ent[1], None if bias is None else bias[col0:col0 + n], (ptr, bs), cin, n, B, H, W, relu,
                                    relu_of=None if wb_in is not None else relu_of, wbits_out=wb_out, relu_wbits=wb_in, **({"s2d": True} if s2d else {}))
                if s2d and info is not None:
                    info["wrote_s2d"] |= 1 << di
                wino_stats["launches"] += 1
            return tag
        # the slice form for what the kernels above do not serve: 64 input channels, destinations of 64 channels (one launch per
        # destination over its slice of the filter; a destination nobody wants is not computed)
        wanted, c0 = [], 0
        for ptr, c, bs in dsts:
            if ptr is not None:
                wanted.append((ptr, c, bs, c0))
            c0 += c
        if (wanted and (relu_of is None or len(wanted) == 1) and all(w_[0] % 8 == 0 and w_[2] % 2 == 0 and _wino16_supported([cin], w_[1], B, H, W) for w_ in wanted)):
            for ptr, c, bs, col0 in wanted:
                _wino16(wino, wp, 0, srcs, bias, (ptr, bs), c, col0, ctot, B, H, W, relu, relu_of=relu_of)
            return "winograd16:" + "+".join("%d" % (1 if relu_of is not None else 0) for _ in wanted)
        if (pieces and wide and len(pieces) <= (2 if relu_of is None else 1) and all(p_[0] % 8 == 0 and p_[2] % 2 == 0 for p_ in pieces)
                and all(lib.ynet_conv2d_winograd_supported(B, H, W, cin, p_[1], K) for p_ in pieces)):      # (YNET_WINOGRAD16=0)
            cache, what = wino
            tag = "winograd:" + "+".join("%d,%d,%d" % (p_[1] // 16, cin // 8, 1 if relu_of is not None else 0) for p_ in pieces)
            for ptr, n, bs, col0 in pieces:
                key = "wino_%s_%d_%d" % (what, col0, n)
                ent = cache.get(key)
                if ent is None or ent[0] is not wp:
                    ent = cache[key] = _wino_made((wp, winograd_filter(wp, cin, n, col0, ctot)))
                _wino_ready(ent)
                conv2d_winograd_raw((srcs[0][0], srcs[0][2]), ent[1], None if bias is None else bias[col0:col0 + n], (ptr, bs), cin, n, B, H, W, relu,
                                    relu_of=relu_of)
                wino_stats["launches"] += 1
            return tag
    if (wino is not None and _wino_allowed and K == 3 and mask is None and relu_of is None and all(len(s_) == 3 for s_ in srcs)
            and (len(srcs) > 1 or srcs[0][1] not in (16, 32)) and all(s_[0] % 16 == 0 and s_[2] % 4 == 0 for s_ in srcs)):
        # the decoders' first convolutions: cat(up-sampled features, skip features[, way-point map]) -> 32 (ynet_conv2d_winograd_cat)
        want = [d for d in dsts if d[0] is not None]
        if len(want) == 1 and len(dsts) == 1 and want[0][1] == 32 and want[0][0] % 8 == 0 and want[0][2] % 2 == 0:
            cs = (ctypes.c_int * len(srcs))(*[s_[1] for s_ in srcs])
            first = srcs[0]
            rest_c = [first[1] - 32] + [s_[1] for s_ in srcs[1:]] if first[1] >= 32 else None
            rest_c = [c for c in rest_c if c > 0] if rest_c is not None else None
            if (not lib.ynet_conv2d_winograd_cat_supported(B, H, W, cs, len(srcs), 32, K) and rest_c and len(rest_c) <= 3
                    and lib.ynet_conv2d_winograd_supported(B, H, W, 32, 32, K)
                    and lib.ynet_conv2d_winograd_cat_supported(B, H, W, (ctypes.c_int * len(rest_c))(*rest_c), len(rest_c), 32, K)):
                # 57 .. 88 input channels (the 64 / 65 -> 32 layers at 128^2): two launches -- the first 32 channels into the destination,
                # then the rest with the destination as the additive term in front of bias and ReLU (read and written by the same lane)
                cache, what = wino
                key = "wino_split_" + what
                ent = cache.get(key)
                rc = (ctypes.c_int * len(rest_c))(*rest_c)
                if ent is None or ent[0] is not wp or ent[3] != tuple(rest_c):
                    cols_pad = -(-32 // 64) * 64
                    u0 = winograd_filter(wp, 32, 32, 0, 32)
                    u1 = torch.empty(lib.ynet_winograd_filter_cat_floats(rc, len(rest_c), 32), device=wp.device, dtype=torch.float32)
                    L.check(lib.ynet_winograd_filter_cat(wp.data_ptr() + 4 * 32 * 9 * cols_pad, u1.data_ptr(), rc, len(rest_c), 32, 0, 32, _stream()), lib)
                    ent = cache[key] = _wino_made((wp, u0, u1, tuple(rest_c)))
                _wino_ready(ent)
                HW = H * W
                conv2d_winograd_raw((first[0], first[2]), ent[1], None, (want[0][0], want[0][2]), 32, 32, B, H, W, False)
                rsrcs = ([(first[0] + 4 * 32 * HW, first[1] - 32, first[2])] if first[1] > 32 else []) + list(srcs[1:])
                wb_out = wbits_out if relu else None
                conv2d_winograd_cat_raw(rsrcs, ent[2], bias, (want[0][0], want[0][2]), B, H, W, relu, addend=(want[0][0], want[0][2], 0), wbits_out=wb_out)
                wino_stats["launches"] += 2
                return "winograd_cat:2,5|wbits" if wb_out is not None else "winograd_cat:2,2"
            if lib.ynet_conv2d_winograd_cat_supported(B, H, W, cs, len(srcs), 32, K):
                cache, what = wino
                key = "wino_cat_" + what
                ent = cache.get(key)
                if ent is None or ent[0] is not wp or ent[2] != tuple(cs):
                    u = torch.empty(lib.ynet_winograd_filter_cat_floats(cs, len(srcs), 32), device=wp.device, dtype=torch.float32)
                    L.check(lib.ynet_winograd_filter_cat(wp.data_ptr(), u.data_ptr(), cs, len(srcs), 32, 0, 32, _stream()), lib)
                    ent = cache[key] = _wino_made((wp, u, tuple(cs)))
                _wino_ready(ent)
                wb_out = wbits_out if relu else None
                conv2d_winograd_cat_raw(srcs, ent[1], bias, (want[0][0], want[0][2]), B, H, W, relu, wbits_out=wb_out)
                wino_stats["launches"] += 1
                return "winograd_cat:2,4|wbits" if wb_out is not None else "winograd_cat:2,0"
        if len(want) == 1 and len(dsts) == 1 and want[0][0] % 8 == 0 and want[0][2] % 2 == 0:
            # the slice form: 64 output channels (the decoders' first convolutions at 64^2: cat(up-sampled 32, skip 64[, way-point map]) -> 64)
            cout_w, cs_all = want[0][1], [s_[1] for s_ in srcs]
            if _wino16_supported(cs_all, cout_w, B, H, W):
                _wino16(wino, wp, 0, srcs, bias, (want[0][0], want[0][2]), cout_w, 0, cout_w, B, H, W, relu)
                return "winograd16:0"
            # more than 84 padded input channels: the leading sources into the destination, then the rest with the destination as the
            # additive term in front of bias and ReLU (read and written by the same lane: in place)
            for cut in range(1, len(srcs)):
                if _wino16_supported(cs_all[:cut], cout_w, B, H, W) and _wino16_supported(cs_all[cut:], cout_w, B, H, W):
                    _wino16(wino, wp, 0, srcs[:cut], None, (want[0][0], want[0][2]), cout_w, 0, cout_w, B, H, W, False)
                    _wino16(wino, wp, sum(cs_all[:cut]), srcs[cut:], bias, (want[0][0], want[0][2]), cout_w, 0, cout_w, B, H, W, relu,
                            addend=(want[0][0], want[0][2], 0))
                    return "winograd16:0+2"
    dp, dc, db = _arrays(dsts)
    nws, ws = 0, None
    if B * H * W <= 65536:                                       # small maps only (see ynet_conv2d_workspace_floats)
        nws = lib.ynet_conv2d_workspace_floats(B, H, W, sum(d[1] for d in dsts))
        if nws:
            key = (wp.device, torch.cuda.current_stream().cuda_stream)   # grow-only scratch per stream (stream-ordered reuse)
            ws = _conv_ws.get(key)
            if ws is None or ws.numel() < nws:
                ws = _conv_ws[key] = torch.empty(nws, device=wp.device, dtype=torch.float32)
    if relu_of is not None:
        if len(srcs) != 1 or len(dsts) != 1 or bias is not None or relu:
            raise ValueError("conv2d_raw: relu_of is for a data gradient with one source and one destination")
        L.check(lib.ynet_conv2d_dgrad_relu(srcs[0][0], srcs[0][1], srcs[0][2], mask[0] if mask else None, mask[1] if mask else 0,
                                           wp.data_ptr(), dsts[0][0], dsts[0][1], dsts[0][2], relu_of[0], relu_of[1], B, H, W, K,
                                           ws.data_ptr() if ws is not None else None, nws, _stream()), lib)
        return
    L.check(lib.ynet_conv2d(sp, sc, sb, _bmods(srcs), len(srcs), mask[0] if mask else None, mask[1] if mask else 0,
                            wp.data_ptr(), bias.data_ptr() if bias is not None else None,
                            dp, dc, db, len(dsts), B, H, W, K, 1 if relu else 0,
                            ws.data_ptr() if ws is not None else None, nws, _stream()), lib)


def conv2d_add_supported(B: int, H: int, W: int, cout: int, K: int) -> bool:
    return bool(_lib().ynet_conv2d_add_supported(int(B), int(H), int(W), int(cout), int(K)))


def conv2d_shared_term(x, x_times: int, rest, weight, bias, relu: bool, cache: dict, term: torch.Tensor, c0: int, c1: int):
    """relu(conv(cat(rest_a, repeat(x), rest_b), W) + b) for an input whose channels [c0, c1) repeat along the batch
    (`x` [Bs,c1-c0,H,W] shared by `x_times` batch items each; only its term is needed here, `x` may be None):
    `term` = conv(x, W[:, c0:c1]) was computed once
    (shared_conv_term); only the other channels go through the convolution here (ynet_conv2d_add).  Inference only.
    `rest`: the non-repeating parts in channel order (tensors [B,*,H,W], B = Bs * x_times)."""
    parts = [p for p in rest]
    for t in parts + [term]:
        _need_gpu(t, "conv2d_shared_term")
    cout, cin, k, _ = weight.shape
    B, _, H, W = parts[0].shape
    rest_filter(weight, c0, c1, cache)
    descs = []
    for p_ in parts:
        t, c, bs = _plane_desc(p_.detach(), "conv2d_shared_term input")
        descs.append((t.data_ptr(), c, bs))
    if sum(d[1] for d in descs) != cin - (c1 - c0):
        raise ValueError("conv2d_shared_term: channel counts do not add up")
    y = torch.empty((B, cout, H, W), device=weight.device, dtype=torch.float32)
    lib = _lib()
    sp, sc, sb = _arrays(descs)
    ent = cache.get("wino_rest")
    if (ent is not None and ent[0] is cache["rest_wp"] and ent[2] == tuple(d[1] for d in descs) and _wino_allowed and _wino_eval and k == 3
            and _wino_cat_eval and H * W >= _wino_eval_min_hw and all(d[0] % 16 == 0 and d[2] % 4 == 0 for d in descs) and term.data_ptr() % 8 == 0
            and lib.ynet_conv2d_winograd_cat_supported(B, H, W, (ctypes.c_int * len(descs))(*ent[2]), len(descs), cout, k)):
        # the Winograd form of the same launch (its filter was transformed by rest_filter_winograd, before the sweep's streams fork)
        conv2d_winograd_cat_raw(descs, ent[1], bias.detach() if bias is not None else None, (y.data_ptr(), cout * H * W), B, H, W, relu,
                                addend=(term.data_ptr(), cout * H * W, term.shape[0]))
        wino_stats["launches"] += 1
        return y
    ent16 = cache.get("wino16_rest")
    if (ent16 is not None and ent16[0] is cache["rest_wp"] and ent16[2] == tuple(d[1] for d in descs) and _wino_eval and k == 3
            and _wino_cat_eval and H * W >= _wino_eval_min_hw and all(d[0] % 16 == 0 and d[2] % 4 == 0 for d in descs) and term.data_ptr() % 8 == 0
            and _wino16_supported([d[1] for d in descs], cout, B, H, W, k)):
        conv2d_winograd16_raw(descs, ent16[1], bias.detach() if bias is not None else None, (y.data_ptr(), cout * H * W), cout, B, H, W, relu,
                              addend=(term.data_ptr(), cout * H * W, term.shape[0]))
        wino_stats["launches"] += 1
        wino_stats["launches16"] = wino_stats.get("launches16", 0) + 1
        return y
    L.check(lib.ynet_conv2d_add(sp, sc, sb, None, len(descs), cache["rest_wp"].data_ptr(),
                                bias.detach().data_ptr() if bias is not None else None, y.data_ptr(), cout, cout * H * W,
                                B, H, W, k, 1 if relu else 0, term.data_ptr(), cout * H * W, term.shape[0], _stream()), lib)
    return y


def rest_filter(weight, c0: int, c1: int, cache: dict) -> torch.Tensor:
    """The packed filter over the input channels OUTSIDE [c0, c1) (conv2d_shared_term's convolution), cached per layer.
    utils/evaluate.py runs its K-sample passes on two streams: _SharedSkipTerms packs it on the main stream BEFORE the fork
    (a lazy pack on one lane would be read by the other lane's cache hit with no dependency on the pack kernel)."""
    wkey = ("rest", c0, c1, weight.data_ptr(), weight._version)
    if cache.get("rest_key") != wkey:
        with torch.no_grad():
            w_rest = torch.cat([weight[:, :c0], weight[:, c1:]], dim=1).contiguous()
            cache["rest_key"], cache["rest_wp"] = wkey, pack_weight(w_rest, 0)
    return cache["rest_wp"]


def rest_filter_winograd(weight, c0: int, c1: int, cache: dict, src_c, B: int, H: int, W: int):
    """The Winograd-domain form of rest_filter(...) for sources of src_c channels each (conv2d_shared_term's launches), where
    ynet_conv2d_winograd_cat serves them; like rest_filter it is made on the caller's stream before the sweep's streams fork."""
    wp = rest_filter(weight, c0, c1, cache)
    cout, k = weight.shape[0], weight.shape[2]
    src_c = tuple(int(c) for c in src_c if c > 0)
    lib = _lib()
    cs = (ctypes.c_int * len(src_c))(*src_c)
    if not (_wino_allowed and _wino_eval and k == 3 and src_c and lib.ynet_conv2d_winograd_cat_supported(B, H, W, cs, len(src_c), cout, k)):
        cache.pop("wino_rest", None)
        # the slice form (64 output channels: the 64^2 / 32^2 levels of evaluate()'s folded batches)
        if _wino_eval and src_c and _wino16_supported(list(src_c), cout, B, H, W, k):
            ent = cache.get("wino16_rest")
            if ent is None or ent[0] is not wp or ent[2] != src_c:
                with torch.no_grad():
                    u = torch.empty(lib.ynet_winograd16_filter_floats(cs, len(src_c), cout), device=wp.device, dtype=torch.float32)
                    L.check(lib.ynet_winograd16_filter(wp.data_ptr(), u.data_ptr(), cs, len(src_c), cout, 0, cout, _stream()), lib)
                ent = cache["wino16_rest"] = (wp, u, src_c)
            return ent[1]
        cache.pop("wino16_rest", None)
        return None
    cache.pop("wino16_rest", None)
    ent = cache.get("wino_rest")
    if ent is None or ent[0] is not wp or ent[2] != src_c:
        with torch.no_grad():
            u = torch.empty(lib.ynet_winograd_filter_cat_floats(cs, len(src_c), cout), device=wp.device, dtype=torch.float32)
            L.check(lib.ynet_winograd_filter_cat(wp.data_ptr(), u.data_ptr(), cs, len(src_c), cout, 0, cout, _stream()), lib)
        ent = cache["wino_rest"] = (wp, u, src_c)
    return ent[1]


def shared_conv_term(x: torch.Tensor, weight, c0: int, c1: int, cache: dict) -> torch.Tensor:
    """conv(x, W[:, c0:c1]) without bias / ReLU: the batch-shared part of a convolution (see conv2d_shared_term)."""
    wkey = ("shared", c0, c1, weight.data_ptr(), weight._version)
    if cache.get("shared_key") != wkey:
        with torch.no_grad():
            cache["shared_key"], cache["shared_w"], cache["shared_pack"] = wkey, weight[:, c0:c1].contiguous(), {}
    with torch.no_grad():
        return conv2d(x if isinstance(x, LazyCat) else x.detach(), cache["shared_w"], None, False, cache["shared_pack"])


def _weight_key(weight, lora_a, lora_b):
    key = (weight.data_ptr(), weight._version)
    if lora_a is not None:
        key += (lora_a.data_ptr(), lora_a._version, lora_b.data_ptr(), lora_b._version)
    return key


def _cached(cache: dict, weight, lora_a, lora_b, scale, what: str):
    """Per-layer cache of the two packed filters ('fwd' / 'dgrad'), invalidated when a parameter changes.  A LoRA
    layer composes W + BA*s and writes both layouts in one launch, into buffers it keeps across steps."""
    key = _weight_key(weight, lora_a, lora_b)
    if cache.get("key") != key:
        bufs, shape = cache.get("lora_bufs"), cache.get("lora_shape")
        cache.clear()
        cache["key"] = key
        if bufs is not None:      # the layer's two packed buffers live across steps (their zero padding is written once)
            cache["lora_bufs"], cache["lora_shape"] = bufs, shape
    if what not in cache:
        with torch.no_grad():
            w = weight.detach()
            if lora_a is not None:
                bufs = cache.get("lora_bufs")
                if bufs is not None and (bufs[0].device != w.device or cache.get("lora_shape") != tuple(w.shape)):
                    bufs = None
                # (the conv launches of this step that still read the buffers are ordered before this launch on the
                # stream; backward of an earlier forward with the old weights is rejected by the version check)
                cache["lora_bufs"] = bufs = lora_compose_pack(w, lora_a.detach(), lora_b.detach(), scale, bufs)
                cache["lora_shape"] = tuple(w.shape)
                cache["fwd"], cache["dgrad"] = bufs
            else:
                cache[what] = pack_weight(w if w.is_contiguous() else w.contiguous(), 0 if what == "fwd" else 1)
    return cache[what]


MULTI_PACK_MAX = 48      # YNET_LORA_MULTI_MAX (csrc/lora.hip)


def refresh_filters(model):
    """Write both packed filter layouts of every conv of `model` whose parameters changed since its last packing --
    W + BA*s for an adapted (LoRA) conv, W itself for a plain one -- in ONE launch per 48 layers
    (ynet_lora_compose_pack_multi) instead of one or two ~5 us launches in front of each convolution: what every conv
    would do itself at its first use in a step.  Frozen, already packed layers are skipped, so a mosa_* step composes
    its 9 adapted convs here and a train_net = train / all step re-packs all 46."""
    todo = []
    for m in model.modules():
        cache = getattr(m, "_packed", None)
        if cache is None or not isinstance(getattr(m, "weight", None), torch.Tensor) or m.weight.dim() != 4:
            continue
        has_lora = bool(getattr(m, "r", 0)) and hasattr(m, "lora_A")
        la, lb = (m.lora_A, m.lora_B) if has_lora else (None, None)
        if cache.get("key") != _weight_key(m.weight, la, lb) or "fwd" not in cache or "dgrad" not in cache:
            todo.append((m, la, lb))
    if not todo:
        return
    lib = _lib()
    with torch.no_grad():
        for i0 in range(0, len(todo), MULTI_PACK_MAX):
            part = todo[i0:i0 + MULTI_PACK_MAX]
            n = len(part)
            entries = []
            for m, la, lb in part:
                w = m.weight.detach()
                _need_gpu(w, "refresh_filters weight")
                cout, cin, k, _ = w.shape
                r = 0
                if la is not None:
                    for t, what in ((la, "lora_A"), (lb, "lora_B")):
                        _need_gpu(t, "refresh_filters " + what)
                    r = la.shape[0] // k
                cache = m._packed
                bufs = cache.get("lora_bufs")
                if bufs is not None and (bufs[0].device != w.device or cache.get("lora_shape") != tuple(w.shape)):
                    bufs = None
                if bufs is None:
                    bufs = tuple(torch.zeros(lib.ynet_packed_weight_floats(cout, cin, k, mode), device=w.device, dtype=torch.float32)
                                 for mode in (0, 1))
                entries.append((m, w.contiguous(), la.detach().contiguous() if la is not None else None,
                                lb.detach().contiguous() if lb is not None else None, bufs, cout, cin, k, r,
                                float(m.scaling) if la is not None else 0.0, la, lb))
            vp = lambda xs: ctypes.cast((_VP * n)(*xs), L.PP)        # noqa: E731
            ia = lambda xs: (ctypes.c_int * n)(*xs)                  # noqa: E731
            L.check(lib.ynet_lora_compose_pack_multi(
                n, vp([e[1].data_ptr() for e in entries]), vp([e[2].data_ptr() if e[2] is not None else None for e in entries]),
                vp([e[3].data_ptr() if e[3] is not None else None for e in entries]), (ctypes.c_float * n)(*[e[9] for e in entries]),
                vp([e[4][0].data_ptr() for e in entries]), vp([e[4][1].data_ptr() for e in entries]),
                ia([e[5] for e in entries]), ia([e[6] for e in entries]), ia([e[7] for e in entries]),
                ia([e[8] for e in entries]), _stream()), lib)
            for e in entries:
                m, w, bufs = e[0], e[1], e[4]
                cache = m._packed
                cache.clear()
                cache["key"] = _weight_key(m.weight, e[10], e[11])
                cache["lora_bufs"], cache["lora_shape"] = bufs, tuple(w.shape)
                cache["fwd"], cache["dgrad"] = bufs


refresh_lora_filters = refresh_filters      # (round-2 name)


# ------------------------------------------------------------------------------------------------
# Skip-connection gradients folded into the max-pool backward
# ------------------------------------------------------------------------------------------------
# An encoder feature map feeds the next stage's max-pool AND the two decoders (torch.cat skips).  Left to the
# autograd engine its gradient is summed by two full-size elementwise adds (6 passes over the largest tensors of
# the step).  Instead: the pool's forward registers its input; a conv backward that produces a gradient for a
# registered tensor hands it over here (and returns None to autograd); the pool's backward adds the handed-over
# gradients while it writes its own (ynet_maxpool2_bwd_add).
# This is only correct when the backward pass runs the pool's node after the decoders' -- true for a full
# ``loss.backward()`` (the pool's gradient depends on them) but NOT for pruned graphs such as
# ``torch.autograd.grad(loss, features[i])`` or ``backward(inputs=...)``, where the pool's backward never runs and
# a handed-over gradient would be dropped.  The fold is therefore OPT-IN: ``with ops.fold_skip_gradients():`` around
# forward + backward (utils/train_epoch.py does that); everywhere else autograd sums the gradients itself.
class _SkipEntry:
    __slots__ = ("ref", "shape", "stash", "consumed")


_skip_registry = {}
skip_fold = False                                                   # switched on by fold_skip_gradients() only
_skip_fold_allowed = _os.environ.get("YNET_SKIP_FOLD", "1") != "0"      # YNET_SKIP_FOLD=0: never fold (A/B runs)


# ------------------------------------------------------------------------------------------------
# ReLU backward applied where a gradient is PRODUCED
# ------------------------------------------------------------------------------------------------
# conv + ReLU is one launch here, and the conv's backward zeroes its incoming gradient where the activation y was <= 0 by
# reading y next to dy (the masked dgrad / wgrad variants: 7-12 % slower, a third tile for the wgrad's DMA stream).  The
# producers of such a gradient can apply the mask instead, each where y costs least:
#   * the max-pool backward (y is the pool's input, in registers anyway) and the fused predictor + criterion kernel (y is its x);
#   * the bilinear up-sampling backward (y is read at the LOW resolution, a quarter of the gradient it reduces);
#   * the data gradient of the next conv of a conv -> ReLU -> conv chain (y is that conv's input; its tile is fetched under the
#     last MFMA chunk of each output tile: ynet_conv2d_dgrad_relu);
# the conv then runs its unmasked kernels.  Protocol, opt-in like the skip fold (same context manager):
#   * _Conv2dFn.forward registers the address of every post-ReLU output it produces        (_relu_outputs)
#   * a producer whose input is such a tensor masks its dx and registers (dx address -> y address, dx version)  (_premasked)
#   * _Conv2dFn.backward drops its own mask when its dy is registered for ITS y and has not been written since
#     (an in-place accumulation by the autograd engine bumps the version; a sum into a new tensor has a new address).
# Masking twice is harmless (idempotent), so every doubt resolves to "mask again".
_relu_outputs = {}
_premasked = {}
premask = False                                                     # switched on by fold_skip_gradients() only
premask_stats = {"unmasked_backwards": 0}                           # conv backwards that ran without their own mask (tests)
_premask_allowed = _os.environ.get("YNET_PREMASK", "1") != "0"          # YNET_PREMASK=0: every conv backward masks itself
_relu_bits_allowed = _os.environ.get("YNET_RELU_BITS", "1") != "0"      # YNET_RELU_BITS=0: output-side masks read the float activation


def _is_relu_output(t: torch.Tensor) -> bool:
    e = _relu_outputs.get(t.data_ptr())
    return e is not None and e[0]() is not None and e[1] == tuple(t.shape)


class fold_skip_gradients:
    """Context manager: inside it (forward AND the full backward of the same graph) the gradients of the encoder
    feature maps that feed a max-pool are added inside the pool's backward kernel instead of by autograd, and the
    ReLU backward of a conv whose output gradient comes from a max-pool / up-sampling backward, from the fused predictor +
    criterion or from the data gradient of a single-input conv is applied by that producer (see `_premasked`)."""

    def __enter__(self):
        global skip_fold, premask, wgrad_branch
        self._prev = (skip_fold, premask, wgrad_branch)
        skip_fold = _skip_fold_allowed
        premask = _premask_allowed
        wgrad_branch = _wgrad_branch_allowed and overlap_decoders
        _relu_outputs.clear()
        _premasked.clear()
        _s2d_wanted.clear()
        _s2d_grads.clear()
        _s2d_produced.clear()
        _deferred.clear()
        _unmaterialized.clear()
        return self

    def __exit__(self, *exc):
        global skip_fold, premask, wgrad_branch
        skip_fold, premask, wgrad_branch = self._prev
        join_wgrad_branch()
        _relu_outputs.clear()
        _premasked.clear()
        _s2d_wanted.clear()
        _s2d_grads.clear()
        _s2d_produced.clear()
        _deferred.clear()
        _unmaterialized.clear()
        _blob_targets.clear()      # (holds the positions and the template of every Gaussian target of the step: nothing of a finished step stays alive)
        # gradients handed over to a pool whose backward never ran (an exception, a pruned graph) must not linger
        for k in [k for k, e in _skip_registry.items() if e.consumed or e.ref() is None or e.stash]:
            del _skip_registry[k]
        return False


def _skip_register(x: torch.Tensor):
    # entries of finished steps (tensor gone, or the pool's backward ran) go, together with any gradient handed over
    # to a pool whose backward never ran (pruned graphs)
    for k in [k for k, e in _skip_registry.items() if e.ref() is None or e.consumed]:
        del _skip_registry[k]
    e = _SkipEntry()
    e.ref, e.shape, e.stash, e.consumed = weakref.ref(x), tuple(x.shape), [], False
    _skip_registry[x.data_ptr()] = e


def _skip_entry(t: torch.Tensor):
    e = _skip_registry.get(t.data_ptr())
    if e is None:
        return None
    if e.ref() is None or e.shape != tuple(t.shape):
        del _skip_registry[t.data_ptr()]
        return None
    return e


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, weight, bias, lora_a, lora_b, *srcs):
        relu, scale, cache = meta["relu"], meta.get("scale", 1.0), meta["cache"]
        _need_gpu(weight, "conv2d weight")
        cout, cin, k, _ = weight.shape
        descs, keep = [], []
        reps = meta.get("repeat") or [1] * len(srcs)
        for i, s in enumerate(srcs):
            t, c, bs = _plane_desc(s, f"conv2d input {i}")
            keep.append(t)
            if reps[i] > 1:
                if ctx.needs_input_grad[5 + i]:
                    raise NotImplementedError("conv2d: a batch-repeated input cannot receive a gradient")
                descs.append((t.data_ptr(), c, c * t.shape[2] * t.shape[3] if t.shape[0] == 1 else t.stride(0), t.shape[0]))
            else:
                descs.append((t.data_ptr(), c, bs))
        if len(descs) > MAX_SRC:
            raise ValueError(f"conv2d: at most {MAX_SRC} concatenated inputs (got {len(descs)})")
        if sum(d[1] for d in descs) != cin:
            raise ValueError(f"conv2d: inputs carry {sum(d[1] for d in descs)} channels, weight expects {cin}")
        B, _, H, W = keep[0].shape
        B = max(t.shape[0] * r for t, r in zip(keep, reps))
        for t, r in zip(keep, reps):
            if t.shape[0] * r != B and not (t.shape[0] > 1 and t.stride(0) == 0):
                raise ValueError(f"conv2d: inputs disagree on the batch size ({[tuple(q.shape) for q in keep]}, repeat {reps})")
        wp = _cached(cache, weight, lora_a, lora_b, scale, "fwd")
        y = torch.empty((B, cout, H, W), device=weight.device, dtype=torch.float32)
        b = bias.detach() if bias is not None else None
        pooled = None
        if (meta.get("pool") and _pool_epilogue_allowed and not meta.get("repeat") and H % 2 == 0 and W % 2 == 0
                and all(d[0] % 16 == 0 and d[2] % 4 == 0 for d in descs) and _lib().ynet_conv2d_pool_supported(B, H, W, cout, k)):
            # the next module is MaxPool2d(2, 2): its output comes out of this launch's epilogue (see _MaxPool2Fn.forward)
            pooled = torch.empty((B, cout, H // 2, W // 2), device=weight.device, dtype=torch.float32)
        bits = None
        # (will the consumer's data gradient -- meta["bits"] = its output channels -- be a Winograd launch?  The dispatcher that will take it answers)
        consumer_wino = (isinstance(meta.get("bits"), int) and not isinstance(meta.get("bits"), bool) and bool(meta.get("wino")) and _wino_allowed and k == 3
                         and dgrad_relu_family(B, H, W, int(meta["bits"]), cout, k) != 0)
        if (meta.get("bits") and not consumer_wino and relu and premask and _relu_bits_allowed and pooled is None and not meta.get("repeat")
                and all(d[0] % 16 == 0 and d[2] % 4 == 0 for d in descs)):
            # the next conv of a conv -> ReLU -> conv chain will write its data gradient THROUGH this ReLU's backward: leave it the
            # 1-bit form of the mask (1/32 of the bytes of y, in the register layout of the tiles both launches share)
            n_words = _lib().ynet_conv2d_relu_bits_words(B, H, W, cout, k)
            if n_words > 0:
                bits = torch.empty(n_words, device=weight.device, dtype=torch.int32)
        # ... and where that data gradient will be the 32-channel Winograd launch, this (Winograd) launch leaves it the mask in THAT tiling's
        # register layout: one word per lane and unit (ynet_conv2d_winograd_*_relu_bits; kept only if the launch taken did write it)
        wbits = None
        if (consumer_wino and cout == 32 and relu and premask and _wino_relu_bits_allowed and pooled is None and not meta.get("repeat")
                and int(meta["bits"]) in (16, 32) and bool(_lib().ynet_conv2d_winograd_supported(B, H, W, int(meta["bits"]), 32, k))):
            n_words = _lib().ynet_winograd_relu_bits_words(B, H, W)
            if n_words > 0:
                wbits = torch.empty(n_words, device=weight.device, dtype=torch.int32)
        # (a pooled 32-channel ReLU output under autograd: the Winograd launch also leaves the pool's backward one byte per block -- kept if that launch was taken)
        pcode = None
        if pooled is not None and relu and cout == 32 and _pool_code_allowed and any(ctx.needs_input_grad) and meta.get("wino"):
            pcode = torch.empty((B, cout, H // 2, W // 2), device=weight.device, dtype=torch.uint8)
        # (the last decoder convolution, asked to wait for the fused predictor + criterion: see `_deferred`)
        defer = bool(meta.get("defer") and _conv_pred_bce_allowed and premask and relu and pooled is None and bits is None and wbits is None and len(descs) == 1
                     and len(descs[0]) == 3 and (cin, cout, k) == (32, 32, 3) and lora_a is None and not any(ctx.needs_input_grad[1:5]) and ctx.needs_input_grad[5]
                     and meta.get("wino") and _wino_allowed and descs[0][0] % 16 == 0 and descs[0][2] % 4 == 0 and keep[0].shape[0] == B
                     and _lib().ynet_conv2d_winograd_pred_bce_supported(B, H, W, 32, 32, 1, 1))
        if defer:
            def run(descs=descs, wp=wp, b=b, y=y, dims=(B, H, W), cache=cache):
                conv2d_raw(descs, None, wp, b, [(y.data_ptr(), 32, 32 * dims[1] * dims[2])], dims[0], dims[1], dims[2], 3, True, wino=(cache, "fwd"))
            _deferred[y.data_ptr()] = {"ref": weakref.ref(y), "shape": tuple(y.shape), "run": run, "src": descs[0], "wp": wp, "bias": b, "cache": cache,
                                       "dims": (B, H, W), "keep": keep[0]}
            took = None
        else:
            took = conv2d_raw(descs, None, wp, b, [(y.data_ptr(), cout, cout * H * W)], B, H, W, k, relu,
                              pooled=None if pooled is None else (pooled.data_ptr(), cout * (H // 2) * (W // 2)),
                              bits_out=None if bits is None else bits.data_ptr(), wino=(cache, "fwd") if meta.get("wino") else None, wbits_out=wbits,
                              pool_code=pcode)
        if not (isinstance(took, str) and took.endswith("|wbits")):
            wbits = None
        if not (isinstance(took, str) and took.endswith("|code")):
            pcode = None
        if pooled is not None:
            for k_ in [k_ for k_, e_ in _pooled_outputs.items() if e_[0]() is None]:      # (a pool that never followed)
                del _pooled_outputs[k_]
            _pooled_outputs[y.data_ptr()] = (weakref.ref(y), tuple(y.shape), pooled, pcode)
        ctx.meta = meta
        ctx.n_src = len(srcs)
        ctx.has_bias = bias is not None
        ctx.has_lora = lora_a is not None
        ctx.save_for_backward(weight, lora_a, lora_b, y if relu else None, *keep)
        ctx.w_key = _weight_key(weight, lora_a, lora_b)
        if relu and premask:
            _relu_outputs[y.data_ptr()] = (weakref.ref(y), tuple(y.shape), bits, k, wbits)
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, lora_a, lora_b, y, *srcs = ctx.saved_tensors
        meta = ctx.meta
        relu, scale, cache = meta["relu"], meta.get("scale", 1.0), meta["cache"]
        cout, cin, k, _ = weight.shape
        dy = dy.contiguous()
        B, _, H, W = dy.shape
        mask = (y.data_ptr(), cout * H * W) if relu else None
        if relu and _premasked:
            # the producer of dy already zeroed it where y <= 0 (see `_premasked`): unmasked dgrad / wgrad kernels
            if _premasked.pop(dy.data_ptr(), None) == (y.data_ptr(), dy._version, tuple(dy.shape)):
                mask = None
                premask_stats["unmasked_backwards"] += 1
        if relu and mask is not None and _unmaterialized.get(y.data_ptr()) == tuple(y.shape):
            raise RuntimeError("conv2d backward: this convolution ran inside the fused predictor + criterion launch (its output was never written), but the gradient that "
                               "arrived does not carry its ReLU backward -- the output has another consumer, or a hook replaced its gradient: set YNET_CONV_PRED_BCE=0")
        need = ctx.needs_input_grad
        need_src = list(need[5:5 + ctx.n_src])
        d_srcs = [None] * ctx.n_src
        branch = None
        if (wgrad_branch and ctx.has_lora and dy.is_cuda and not need[1] and not (ctx.has_bias and need[2]) and (need[3] or need[4])
                and lora_a.is_leaf and lora_b.is_leaf and not lora_a._backward_hooks and not lora_b._backward_hooks
                and not getattr(lora_a, "_post_accumulate_grad_hooks", None) and not getattr(lora_b, "_post_accumulate_grad_hooks", None)):
            # fork BEFORE the data gradient is queued: the adapter gradients (below) run beside it (see `wgrad_branch`)
            cur = torch.cuda.current_stream(dy.device)
            bkey, branch = _wgrad_stream(dy.device)
            branch.wait_stream(cur)
        if any(need_src):
            if ctx.w_key != _weight_key(weight, lora_a, lora_b):
                raise RuntimeError("conv2d backward: a parameter was modified in place between forward and backward")
            wp_d = _cached(cache, weight, lora_a, lora_b, scale, "dgrad")
            dsts, want_s2d = [], []
            for i, s in enumerate(srcs):
                c = s.shape[1]
                if need_src[i]:
                    # (a batch-broadcast input -- stride 0 -- gets the full per-image gradient; the expand's own
                    # backward sums it over the batch, as in the reference's semantic_img.expand(...))
                    d_srcs[i] = torch.empty((B, c, H, W), device=dy.device, dtype=torch.float32)
                    dsts.append((d_srcs[i].data_ptr(), c, c * H * W))
                    w_ = _s2d_wanted.get(s.data_ptr()) if (premask and _upconv_s2d_allowed) else None
                    want_s2d.append(bool(w_ is not None and w_[0]() is not None and w_[1] == tuple(s.shape) and c == 16 and H % 2 == 0 and W % 2 == 0))
                else:
                    dsts.append((None, c, 0))
                    want_s2d.append(False)
            s2d_info = {} if any(want_s2d) else None
            # the single input is itself a post-ReLU conv output that no pool folds: apply THAT layer's ReLU backward to the
            # gradient produced here (see `_premasked`)
            s0 = srcs[0]
            emask = None
            if (premask and ctx.n_src == 1 and need_src[0] and _is_relu_output(s0) and s0.is_contiguous() and s0.shape[0] == B
                    and s0.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0 and not (skip_fold and _skip_entry(s0) is not None)
                    and _lib().ynet_conv2d_dgrad_relu_supported(B, H, W, int(s0.shape[1]), int(k))):
                emask = (s0.data_ptr(), s0.shape[1] * H * W)
            ebits = None
            if emask is not None:
                e0 = _relu_outputs.get(s0.data_ptr())
                if (e0 is not None and e0[2] is not None and e0[3] == k and d_srcs[0].data_ptr() % 16 == 0
                        and _lib().ynet_conv2d_relu_bits_words(B, H, W, int(s0.shape[1]), int(k)) == e0[2].numel()):
                    ebits = e0[2]          # (written by s0's own forward launch, for exactly this tiling)
            # (where the Winograd generation serves the launch it applies the float mask itself -- faster than the implicit GEMM with
            #  the 1-bit mask, whose layout belongs to that kernel's tiles)
            wino_em = (emask is not None and mask is None and bool(meta.get("wino")) and _wino_allowed and k == 3 and dy.data_ptr() % 16 == 0
                       and d_srcs[0].data_ptr() % 8 == 0 and s0.data_ptr() % 8 == 0 and dgrad_relu_family(B, H, W, cout, int(s0.shape[1]), int(k)) != 0)
            ewbits = None
            if wino_em and _wino_relu_bits_allowed and int(s0.shape[1]) == 32:
                e0 = _relu_outputs.get(s0.data_ptr())
                if (e0 is not None and len(e0) > 4 and e0[4] is not None and e0[4].numel() == _lib().ynet_winograd_relu_bits_words(B, H, W)
                        and _lib().ynet_conv2d_winograd_supported(B, H, W, cout, 32, int(k))):
                    ewbits = e0[4]         # (written by s0's own Winograd forward launch: the 1-bit mask in this launch's tiling)
            if ebits is not None and not wino_em:
                premask_stats["bit_masks"] = premask_stats.get("bit_masks", 0) + 1
                conv2d_raw([(dy.data_ptr(), cout, cout * H * W)], mask, wp_d, None, dsts, B, H, W, k, False, relu_bits=ebits.data_ptr())
            else:
                if ewbits is not None:
                    premask_stats["wino_bit_masks"] = premask_stats.get("wino_bit_masks", 0) + 1
                conv2d_raw([(dy.data_ptr(), cout, cout * H * W)], mask, wp_d, None, dsts, B, H, W, k, False, relu_of=emask,
                           wino=(cache, "dgrad") if meta.get("wino") else None, relu_wbits=ewbits, dst_s2d=want_s2d if s2d_info is not None else None, info=s2d_info)
                if s2d_info:
                    for i, g_ in enumerate(d_srcs):      # (written space-to-depth: only the up-convolution's backward may read these tensors)
                        if g_ is not None and (s2d_info.get("wrote_s2d", 0) >> i) & 1:
                            _s2d_grads[g_.data_ptr()] = (g_._version, tuple(g_.shape))
                            # (one hand-over per up-convolution output: a second consumer of that tensor gets a row-major gradient, and the sum autograd
                            #  forms of the two is then refused by _UpConvFn.backward instead of being read in the wrong layout)
                            _s2d_wanted.pop(srcs[i].data_ptr(), None)
                            _s2d_produced.add(srcs[i].data_ptr())
            if emask is not None:
                _premasked[d_srcs[0].data_ptr()] = (s0.data_ptr(), d_srcs[0]._version, tuple(d_srcs[0].shape))
            if skip_fold:
                for i, s in enumerate(srcs):
                    if d_srcs[i] is None:
                        continue
                    e = _skip_entry(s)
                    if e is not None and not e.consumed and len(e.stash) < 2:
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream(dy.device))
                        e.stash.append((d_srcs[i], ev))
                        d_srcs[i] = None        # the pool's backward adds it in (see _MaxPool2Fn.backward)
        d_w = d_b = d_a = d_bm = None
        want_w = need[1] or (ctx.has_lora and (need[3] or need[4]))
        want_b = ctx.has_bias and need[2]
        if branch is not None:
            for t in (dy, y, weight, lora_a, lora_b, *srcs):
                if t is not None:
                    t.record_stream(branch)
            with torch.cuda.stream(branch):
                if lora_conv2d_wgrad_supported(srcs, dy, weight, lora_a, preferred=True):
                    d_a, d_bm = lora_conv2d_wgrad_raw(srcs, dy, mask, weight, lora_a.detach(), lora_b.detach(), scale)
                else:
                    dw, _ = conv2d_wgrad_raw(srcs, dy, mask, weight, False)
                    d_a, d_bm = lora_grad(dw, lora_a.detach(), lora_b.detach(), scale)
                # (an existing .grad -- the data-parallel flat buffer's view, an accumulation -- is added to in place, on the branch:
                # this backward is the only writer of these two parameters' gradients)
                for p_, g_, want in ((lora_a, d_a, need[3]), (lora_b, d_bm, need[4])):
                    if want and p_.grad is not None:
                        _wgrad_adds.setdefault(bkey, []).append((p_.grad, g_.view_as(p_.grad)))      # (added when the branch is joined: join_wgrad_branch)
            for p_, g_, want in ((lora_a, d_a, need[3]), (lora_b, d_bm, need[4])):
                g_.record_stream(cur)         # read by the optimizer (and a data-parallel stage) on the step's stream, after the join
                if want and p_.grad is None:
                    p_.grad = g_.view_as(p_)
            d_a = d_bm = None
            _wgrad_pending[bkey] = branch
        elif want_w or want_b:
            if ctx.has_lora and not need[1] and not want_b and lora_conv2d_wgrad_supported(srcs, dy, weight, lora_a, preferred=True):
                # only the adapter trains (mosa_*): dA / dB straight from projected planes, no dW (ynet_lora_conv2d_wgrad)
                d_a, d_bm = lora_conv2d_wgrad_raw(srcs, dy, mask, weight, lora_a.detach(), lora_b.detach(), scale)
            else:
                dw, d_b = conv2d_wgrad_raw(srcs, dy, mask, weight, want_b)
                if ctx.has_lora and (need[3] or need[4]):
                    d_a, d_bm = lora_grad(dw, lora_a.detach(), lora_b.detach(), scale)
                if need[1]:
                    d_w = dw
        return (None, d_w, d_b, d_a if need[3] else None, d_bm if need[4] else None, *d_srcs)


def conv2d_wgrad_raw(srcs, dy, mask, weight, want_b):
    """(dW, db or None) of conv(cat(srcs), W) for the output gradient dy [B,cout,H,W] (masked where mask's plane <= 0)."""
    lib = _lib()
    cout, cin, k, _ = weight.shape
    B, _, H, W = dy.shape
    descs = [(s.data_ptr(), s.shape[1], (s.stride(0) if s.shape[0] > 1 else s.shape[1] * H * W)) for s in srcs]
    sp, sc, sb = _arrays(descs)
    dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
    d_b = torch.empty(cout, device=dy.device, dtype=torch.float32) if want_b else None
    ws = torch.empty(lib.ynet_conv2d_wgrad_workspace_floats(B, H, W, cout, cin, k), device=dy.device, dtype=torch.float32)
    L.check(lib.ynet_conv2d_wgrad(sp, sc, sb, len(descs), dy.data_ptr(), cout * H * W,
                                  mask[0] if mask else None, mask[1] if mask else 0,
                                  dw.data_ptr(), d_b.data_ptr() if want_b else None, ws.data_ptr(),
                                  B, H, W, cout, k, _stream()), lib)
    return dw, d_b


_lora_wg_ws = {}


def lora_conv2d_wgrad_supported(srcs, dy, weight, lora_a, preferred: bool = False) -> bool:
    """Can ynet_lora_conv2d_wgrad serve this layer -- and, with `preferred`, is it the faster path for it (measured)?"""
    cout, cin, k, _ = weight.shape
    r = lora_a.shape[0] // k
    W = dy.shape[3]
    query = _lib().ynet_lora_conv2d_wgrad_preferred if preferred else _lib().ynet_lora_conv2d_wgrad_supported
    if not query(int(cin), int(cout), int(k), int(r), int(W)):
        return False
    return all(s.data_ptr() % 16 == 0 and (s.shape[0] == 1 or s.stride(0) % 4 == 0) for s in srcs) and dy.data_ptr() % 16 == 0


def lora_conv2d_wgrad_raw(srcs, dy, mask, weight, lora_a, lora_b, scale):
    """(d lora_A, d lora_B) of the adapted conv(cat(srcs), W + s * (B @ A).view(W.shape)) for the output gradient dy, without
    the filter gradient in between (ynet_lora_conv2d_wgrad; models/ynet.py:141-144)."""
    lib = _lib()
    cout, cin, k, _ = weight.shape
    r = lora_a.shape[0] // k
    B, _, H, W = dy.shape
    descs = [(s.data_ptr(), s.shape[1], (s.stride(0) if s.shape[0] > 1 else s.shape[1] * H * W)) for s in srcs]
    sp, sc, sb = _arrays(descs)
    la, lb = lora_a.contiguous(), lora_b.contiguous()
    d_a, d_b = torch.empty_like(la), torch.empty_like(lb)
    n_ws = lib.ynet_lora_conv2d_wgrad_workspace_floats(cin, cout)
    key = (dy.device, torch.cuda.current_stream().cuda_stream)      # grow-only scratch per stream (stream-ordered reuse)
    ws = _lora_wg_ws.get(key)
    if ws is None or ws.numel() < n_ws:
        ws = _lora_wg_ws[key] = torch.empty(n_ws, device=dy.device, dtype=torch.float32)
    L.check(lib.ynet_lora_conv2d_wgrad(sp, sc, sb, len(descs), dy.data_ptr(), cout * H * W,
                                       mask[0] if mask else None, mask[1] if mask else 0,
                                       la.data_ptr(), lb.data_ptr(), float(scale), d_a.data_ptr(), d_b.data_ptr(), ws.data_ptr(),
                                       B, H, W, cout, k, r, _stream()), lib)
    return d_a, d_b


def conv2d(x, weight, bias, relu: bool, cache: dict, lora_a=None, lora_b=None, scale: float = 1.0, pool: bool = False,
           bits=False, defer: bool = False):
    """[ReLU](conv(cat(x), W_eff) + bias); x is a tensor or a LazyCat.  pool: the caller applies max_pool2 to the result next --
    where the kernel can, the pooled copy is written by this launch and that max_pool2 call finds it (no kernel).  bits: the
    caller feeds the (post-ReLU) result to another convolution next -- inside fold_skip_gradients() this launch then also writes
    the 1-bit form of its ReLU mask, which that convolution's data gradient applies to what it writes (ynet_conv2d_relu_bits); an int:
    that next convolution's output channels -- where its data gradient will be a Winograd launch (which reads the float activation) no
    bits are written."""
    parts = [p.expand() if isinstance(p, BatchExpand) else p for p in _parts(x)]      # (a fresh expand node per consumer)
    # wino: the plain 16 / 32-channel large-map launches take the Winograd generation (conv2d_raw).  Its results differ from the
    # implicit GEMM's by fp32 rounding that is uncorrelated with the reference's own (the implicit GEMM sums in nearly the reference's
    # order).  Measured against the CPU oracle (tests/wino_margin.py): a training step's loss / ADE / FDE / per-trajectory read-outs
    # and gradients deviate exactly as much with it as without; in evaluate()'s K-sample sweep every trajectory's ADE / FDE does too
    # (max 6e-5 / 3e-5 either way), while single coordinates of single goal samples -- 0.3 % of them, where the decoded heat-map is
    # diffuse -- move by up to 5e-3 px instead of 4e-5.  YNET_WINOGRAD_EVAL=0 keeps evaluate() on the implicit GEMM.
    meta = {"relu": bool(relu), "scale": float(scale), "cache": cache, "pool": bool(pool), "bits": bits if torch.is_grad_enabled() else False,
            "wino": torch.is_grad_enabled() or _wino_eval, "defer": bool(defer) and torch.is_grad_enabled()}
    if any(isinstance(p, BatchRepeat) for p in parts):
        if torch.is_grad_enabled() and any(t.requires_grad for t in (weight, bias, lora_a, lora_b) if t is not None):
            raise NotImplementedError("conv2d: batch-repeated inputs are for inference (torch.no_grad) only")
        meta["repeat"] = [p.times if isinstance(p, BatchRepeat) else 1 for p in parts]
        parts = [p.tensor if isinstance(p, BatchRepeat) else p for p in parts]
    return _Conv2dFn.apply(meta, weight, bias, lora_a, lora_b, *parts)


# ------------------------------------------------------------------------------------------------
# pooling / resampling
# ------------------------------------------------------------------------------------------------
# conv outputs whose 2 x 2 max-pooled copy was written by the conv's own epilogue (ops.conv2d(..., pool=True)):
# address of y -> (weak reference to y, its shape, the pooled tensor); consumed by the max_pool2 call that follows
_pooled_outputs = {}
pool_code_stats = {"launches": 0}
_pool_epilogue_allowed = _os.environ.get("YNET_POOL_EPILOGUE", "1") != "0"


class _MaxPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_gpu(x, "max_pool2d")
        x = x.contiguous()
        B, C, H, W = x.shape
        e = _pooled_outputs.pop(x.data_ptr(), None) if _pooled_outputs else None
        ctx.code = None
        if e is not None and e[0]() is x and e[1] == tuple(x.shape) and x._version == 0:
            y = e[2]                 # written by the producing conv's epilogue: no pass over x here
            ctx.code = e[3] if len(e) > 3 else None      # (... and, from a Winograd launch, arg-max + ReLU bits per block for backward)
        else:
            y = torch.empty((B, C, H // 2, W // 2), device=x.device, dtype=torch.float32)
            lib = _lib()
            L.check(lib.ynet_maxpool2_fwd(x.data_ptr(), y.data_ptr(), B * C, H, W, _stream()), lib)
        ctx.save_for_backward(x)
        ctx.folds = bool(skip_fold and ctx.needs_input_grad[0] and H % 2 == 0 and W % 2 == 0)
        if ctx.folds:
            _skip_register(x)
        # x is a post-ReLU conv output: this pool's backward applies that ReLU's backward to the gradient it produces
        ctx.premask = bool(premask and ctx.needs_input_grad[0] and H % 2 == 0 and W % 2 == 0 and x.data_ptr() % 8 == 0
                           and _is_relu_output(x))
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        lib = _lib()
        adds = []
        if ctx.folds:
            e = _skip_entry(x)
            if e is not None:
                adds, e.stash, e.consumed = e.stash, [], True
        pm = bool(ctx.premask and premask)
        if adds or pm:
            cur = torch.cuda.current_stream(dy.device)
            for t, ev in adds:          # produced on the decoders' streams
                cur.wait_event(ev)
                t.record_stream(cur)
            a0 = adds[0][0] if adds else None
            a1 = adds[1][0] if len(adds) > 1 else None
            code, ctx.code = ctx.code, None
            if code is not None and _pool_code_allowed:      # x (saved: autograd has checked that nobody wrote into it) is not read again
                pool_code_stats["launches"] += 1
                L.check(lib.ynet_maxpool2_bwd_add_code(code.data_ptr(), dy.contiguous().data_ptr(), a0.data_ptr() if a0 is not None else None,
                                                       a1.data_ptr() if a1 is not None else None, dx.data_ptr(), B * C, H, W,
                                                       1 if pm else 0, _stream()), lib)
            else:
                L.check(lib.ynet_maxpool2_bwd_add(x.data_ptr(), dy.contiguous().data_ptr(), a0.data_ptr() if a0 is not None else None,
                                                  a1.data_ptr() if a1 is not None else None, dx.data_ptr(), B * C, H, W,
                                                  1 if pm else 0, _stream()), lib)
            if pm:
                _premasked[dx.data_ptr()] = (x.data_ptr(), dx._version, tuple(dx.shape))
        else:
            L.check(lib.ynet_maxpool2_bwd(x.data_ptr(), dy.contiguous().data_ptr(), dx.data_ptr(), B * C, H, W, _stream()), lib)
        return dx


def max_pool2(x):
    return _MaxPool2Fn.apply(x)


class _Upsample2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_gpu(x, "upsample2x")
        x = x.contiguous()
        B, C, H, W = x.shape
        y = torch.empty((B, C, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
        lib = _lib()
        L.check(lib.ynet_upsample2x_fwd(x.data_ptr(), y.data_ptr(), B * C, H, W, _stream()), lib)
        ctx.shape = (B, C, H, W)
        # x is a post-ReLU conv output: this backward applies that ReLU's backward to the gradient it produces (`_premasked`)
        ctx.premask = bool(premask and ctx.needs_input_grad[0] and _is_relu_output(x))
        ctx.save_for_backward(x if ctx.premask else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        (x,) = ctx.saved_tensors
        dx = torch.empty((B, C, H, W), device=dy.device, dtype=torch.float32)
        lib = _lib()
        if ctx.premask and premask and x is not None:
            L.check(lib.ynet_upsample2x_bwd_relu(dy.contiguous().data_ptr(), dx.data_ptr(), x.data_ptr(), B * C, H, W, _stream()), lib)
            _premasked[dx.data_ptr()] = (x.data_ptr(), dx._version, tuple(dx.shape))
        else:
            L.check(lib.ynet_upsample2x_bwd(dy.contiguous().data_ptr(), dx.data_ptr(), B * C, H, W, _stream()), lib)
        return dx


def upsample2x(x):
    return _Upsample2xFn.apply(x)


_upconv_allowed = _os.environ.get("YNET_WINOGRAD_UP", "1") != "0"      # YNET_WINOGRAD_UP=0: the bilinear x2 stays a pass of its own in front of the up-convolution
upconv_stats = {"fused": 0}


def upsample2x_conv2d_raw(src, u, bias, dst, cin, cout, B, H, W, relu=False):
    """dst = [relu](conv3x3(bilinear x2 of src) + bias) in one launch (ynet_upsample2x_conv2d_winograd): src (ptr, batch_stride) of cin planes
    of (H / 2) x (W / 2), dst (ptr, batch_stride) of cout planes of H x W, u = winograd_filter(packed filter, cin, cout)."""
    lib = _lib()
    L.check(lib.ynet_upsample2x_conv2d_winograd(src[0], src[1], u.data_ptr(), bias.data_ptr() if bias is not None else None, dst[0], dst[1], cin, cout, B, H, W,
                                                1 if relu else 0, _stream()), lib)


# ---- the up-convolution's backward without the up-sampled gradient (round 6; include/ynet_hip.h: ynet_upconv_dgrad_ring) ----
# Up^T . conv^T is a 3 x 3 convolution at the LOW resolution over the space-to-depth output gradient.  Protocol (inside fold_skip_gradients() only, like the other
# producer / consumer hand-overs): _UpConvFn.forward registers its output; the conv backward that produces that tensor's gradient asks the dispatcher to write it
# space-to-depth (one plain 16-channel Winograd launch: the decoders' last level) and registers the gradient; _UpConvFn.backward, handed a registered gradient, runs the
# effective-filter data gradient at the low resolution + the ring correction instead of [data gradient at the up-sampled size -> bilinear backward].
_upconv_s2d_allowed = _os.environ.get("YNET_UPCONV_S2D", "1") != "0"
_s2d_wanted = {}       # output of an up-convolution: data_ptr -> (weakref, shape)
_s2d_grads = {}        # a gradient written space-to-depth: data_ptr -> (version, shape)
# ---- the last decoder convolution inside the fused predictor + criterion (round 6; ynet_conv2d_winograd_pred_bce_blob) -------------------------------------------
# `conv2d(..., defer=True)` (a frozen 32 -> 32 conv + ReLU under fold_skip_gradients()) builds its autograd node but launches nothing: its output tensor is registered
# here, and the pred_bce call that consumes it runs [convolution -> predictor -> criterion -> predictor's data gradient] as ONE launch -- the 32 activation planes are
# never written.  Anything else that wants the tensor calls materialize_deferred() first (pred_bce itself does when the fused launch does not apply).  The convolution's
# backward receives the fused launch's dx, which carries that convolution's ReLU backward already (`_premasked`); a backward that would need the activation itself
# (its mask was not applied by the producer) is refused loudly.
_conv_pred_bce_allowed = _os.environ.get("YNET_CONV_PRED_BCE", "1") != "0"
_deferred = {}         # data_ptr of y -> dict(ref, shape, run) : convolutions not yet launched
_unmaterialized = {}   # data_ptr of y -> shape : outputs whose convolution ran inside the fused launch (the memory holds nothing)
conv_pred_bce_stats = {"fused": 0, "materialized": 0}


def materialize_deferred(t):
    """Launch the convolution behind `t` if it was deferred (see above); a no-op for every other tensor."""
    if not _deferred or not torch.is_tensor(t):
        return t
    e = _deferred.get(t.data_ptr())
    if e is not None and e["ref"]() is not None and e["shape"] == tuple(t.shape):
        del _deferred[t.data_ptr()]
        e["run"]()
        conv_pred_bce_stats["materialized"] += 1
    return t


_s2d_produced = set()  # outputs of up-convolutions (data_ptr) for which a gradient WAS written space-to-depth: their backward must be handed exactly that tensor
upconv_stats_s2d = {"backwards": 0}
_S2D_M = ((0.75, 0.25, 0.0), (0.25, 0.75, 0.75), (0.0, 0.0, 0.25)), ((0.25, 0.0, 0.0), (0.75, 0.75, 0.25), (0.0, 0.25, 0.75))
_S2D_DM = ((-0.25, 0.25, 0.0), (0.25, 0.0, 0.0)), ((0.0, 0.0, 0.25), (0.0, 0.25, -0.25))


def upconv_s2d_tables(weight, cache):
    """(packed effective filter for the low-resolution data gradient, ring tables [16][4 cout][cin]) of an up-convolution's filter [cout][cin][3][3], cached per weight
    version in the layer's cache.  Keff[(py, px, co)][ci][a][b] = sum M_py[a][ty] K[co][ci][ty][tx] M_px[b][tx]; the tables as include/ynet_hip.h lists them."""
    key = ("s2d", weight.data_ptr(), weight._version)
    ent = cache.get("s2d_tables")
    if ent is None or ent[0] != key:
        with torch.no_grad():
            K = weight.detach().double()
            cout, cin = K.shape[0], K.shape[1]
            M = torch.tensor(_S2D_M, dtype=torch.float64, device=K.device)           # [p][a][t]
            dM = torch.tensor(_S2D_DM, dtype=torch.float64, device=K.device)         # [side][p][t]
            keff = torch.einsum("pau,oiuv,qbv->pqoiab", M, K, M).reshape(4 * cout, cin, 3, 3).float().contiguous()
            tabs = []
            for s_ in range(2):      # rows: [b][c'][ci]
                tabs.append(torch.einsum("pu,oiuv,qbv->bpqoi", dM[s_], K, M).reshape(3, 4 * cout, cin))
            for s_ in range(2):      # columns: [a][c'][ci]
                tabs.append(torch.einsum("pau,oiuv,qv->apqoi", M, K, dM[s_]).reshape(3, 4 * cout, cin))
            corners = [torch.einsum("pu,oiuv,qv->pqoi", dM[sv], K, dM[sh]).reshape(1, 4 * cout, cin) for sv in range(2) for sh in range(2)]
            tables = torch.cat(tabs + corners, 0).float().contiguous()
            ent = cache["s2d_tables"] = (key, pack_weight(keff, 1), tables, {})
    return ent[1], ent[2], ent[3]


class _UpConvFn(torch.autograd.Function):
    """y = conv3x3(upsample2x(x), W) + b without the up-sampled tensor (models/ynet.py:463-464 as one launch).  The filter is frozen (no
    filter gradient needs the up-sampled input); backward = the convolution's data gradient at the up-sampled size, then the bilinear
    backward -- through the ReLU backward of the layer that produced x, exactly as _Upsample2xFn does."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache):
        _need_gpu(x, "upsample2x_conv2d")
        x = x.contiguous()
        B, cin, Hl, Wl = x.shape
        cout = weight.shape[0]
        H, W = 2 * Hl, 2 * Wl
        wp = _cached(cache, weight, None, None, 1.0, "fwd")
        if _lib().ynet_upsample2x_conv2d_winograd_supported(B, H, W, cin, cout, 3) == 2:      # the slice form: its own filter layout
            ent = _wino16_filter((cache, "fwd"), wp, 0, (cin,), cout, 0, cout)
        else:
            key = "wino_fwd_0_%d" % cout      # (the entry ops.conv2d_raw keeps for the unfused launch of the same layer)
            ent = cache.get(key)
            if ent is None or ent[0] is not wp:
                ent = cache[key] = _wino_made((wp, winograd_filter(wp, cin, cout, 0, cout)))
            _wino_ready(ent)
        y = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
        upsample2x_conv2d_raw((x.data_ptr(), cin * Hl * Wl), ent[1], bias.detach() if bias is not None else None, (y.data_ptr(), cout * H * W), cin, cout, B, H, W)
        wino_stats["launches"] += 1
        upconv_stats["fused"] += 1
        ctx.cache, ctx.shape, ctx.cout = cache, (B, cin, Hl, Wl), cout
        ctx.w_key = _weight_key(weight, None, None)
        ctx.premask = bool(premask and ctx.needs_input_grad[0] and _is_relu_output(x))
        ctx.save_for_backward(weight, x if ctx.premask else None)
        if (_upconv_s2d_allowed and premask and ctx.needs_input_grad[0] and cout == 16 and x.data_ptr() % 16 == 0
                and _wino16_supported([4 * cout], cin, B, Hl, Wl)):
            # (inside fold_skip_gradients(): the gradient of y may arrive space-to-depth -- see the protocol above)
            _s2d_wanted[y.data_ptr()] = (weakref.ref(y), tuple(y.shape))
        ctx.y_ptr = y.data_ptr()
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, x = ctx.saved_tensors
        B, cin, Hl, Wl = ctx.shape
        H, W, cout = 2 * Hl, 2 * Wl, ctx.cout
        if ctx.w_key != _weight_key(weight, None, None):
            raise RuntimeError("upsample2x_conv2d backward: the filter was modified in place between forward and backward")
        dy = dy.contiguous()
        reg = _s2d_grads.pop(dy.data_ptr(), None)
        handed = reg is not None and reg == (dy._version, tuple(dy.shape))
        if ctx.y_ptr in _s2d_produced:
            _s2d_produced.discard(ctx.y_ptr)
            if not handed:
                raise RuntimeError("upsample2x_conv2d backward: a gradient of this up-convolution's output was written space-to-depth for it, but the gradient that "
                                   "arrived is another tensor (the output has a second consumer, or a hook replaced its gradient): set YNET_UPCONV_S2D=0 for such a graph")
        if handed:
            # dy's memory holds the gradient space-to-depth, [B, 4 cout, Hl, Wl]: the data gradient of the effective filter at the low resolution (through the ReLU
            # backward of x where that is wanted), then what the bilinear clamp and the up-sampled image's zero padding add on the outermost ring
            wp_eff, tables, wcache = upconv_s2d_tables(weight, ctx.cache)
            dx = torch.empty((B, cin, Hl, Wl), device=dy.device, dtype=torch.float32)
            masked = bool(ctx.premask and premask and x is not None)
            conv2d_raw([(dy.data_ptr(), 4 * cout, 4 * cout * Hl * Wl)], None, wp_eff, None, [(dx.data_ptr(), cin, cin * Hl * Wl)], B, Hl, Wl, 3, False,
                       relu_of=(x.data_ptr(), cin * Hl * Wl) if masked else None, wino=(wcache, "dgrad"))
            lib = _lib()
            L.check(lib.ynet_upconv_dgrad_ring(dy.data_ptr(), 4 * cout * Hl * Wl, tables.data_ptr(), x.data_ptr() if masked else None, cin * Hl * Wl, dx.data_ptr(),
                                               cin * Hl * Wl, B, 4 * cout, cin, Hl, Wl, _stream()), lib)
            if masked:
                _premasked[dx.data_ptr()] = (x.data_ptr(), dx._version, tuple(dx.shape))
            upconv_stats_s2d["backwards"] += 1
            return dx, None, None, None
        wp_d = _cached(ctx.cache, weight, None, None, 1.0, "dgrad")
        d_up = torch.empty((B, cin, H, W), device=dy.device, dtype=torch.float32)
        conv2d_raw([(dy.data_ptr(), cout, cout * H * W)], None, wp_d, None, [(d_up.data_ptr(), cin, cin * H * W)], B, H, W, 3, False, wino=(ctx.cache, "dgrad"))
        dx = torch.empty((B, cin, Hl, Wl), device=dy.device, dtype=torch.float32)
        lib = _lib()
        if ctx.premask and premask and x is not None:
            L.check(lib.ynet_upsample2x_bwd_relu(d_up.data_ptr(), dx.data_ptr(), x.data_ptr(), B * cin, Hl, Wl, _stream()), lib)
            _premasked[dx.data_ptr()] = (x.data_ptr(), dx._version, tuple(dx.shape))
        else:
            L.check(lib.ynet_upsample2x_bwd(d_up.data_ptr(), dx.data_ptr(), B * cin, Hl, Wl, _stream()), lib)
        return dx, None, None, None


def upsample2x_conv2d(x, conv):
    """conv(upsample2x(x)) for a decoder's up-convolution `conv` (a plain HipConv2d, 3 x 3, no ReLU): ONE launch where the shape is served, no
    filter gradient is wanted (inference, or a frozen filter: the filter gradient would need the up-sampled input) and nobody hooks the
    module; the two modules otherwise."""
    w, b = conv.weight, conv.bias
    if (_upconv_allowed and _wino_allowed and torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32
            and ((not torch.is_grad_enabled() and _wino_eval) or (torch.is_grad_enabled() and not w.requires_grad and (b is None or not b.requires_grad)))
            and not conv._forward_hooks and not conv._forward_pre_hooks and not conv._backward_hooks
            and tuple(w.shape[1:]) == (x.shape[1], 3, 3) and x.data_ptr() % 16 == 0
            and _lib().ynet_upsample2x_conv2d_winograd_supported(int(x.shape[0]), 2 * int(x.shape[2]), 2 * int(x.shape[3]), int(x.shape[1]), int(w.shape[0]), 3)):
        return _UpConvFn.apply(x, w, b, conv._packed)
    return conv(upsample2x(x))


def avgpool_pyramid(x: torch.Tensor, n_levels: int) -> List[torch.Tensor]:
    """[x, AvgPool2d(2)(x), ..., AvgPool2d(2**(n_levels-1))(x)] in one pass (no gradient)."""
    _need_gpu(x, "avgpool_pyramid")
    x = x.detach().contiguous()
    B, C, H, W = x.shape
    outs = [torch.empty((B, C, H >> i, W >> i), device=x.device, dtype=torch.float32) for i in range(1, n_levels)]
    if outs:
        lib = _lib()
        ptrs = (_VP * len(outs))(*[o.data_ptr() for o in outs])
        L.check(lib.ynet_avgpool_pyramid(x.data_ptr(), ctypes.cast(ptrs, L.PP), len(outs), B * C, H, W, _stream()), lib)
    return [x] + outs


# ------------------------------------------------------------------------------------------------
# The serial adapters' element-wise tail (models/ynet.py:24-26,64-66,117-131): BatchNorm2d, residual add, ReLU (round 6: off ATen)
# ------------------------------------------------------------------------------------------------
class _BatchNorm2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, train, factor, eps):
        _need_gpu(x, "batch_norm2d input")
        x = x.contiguous()
        B, C, H, W = x.shape
        lib = _lib()
        y = torch.empty_like(x)
        mean, invstd = torch.empty(C, device=x.device, dtype=torch.float32), torch.empty(C, device=x.device, dtype=torch.float32)
        ws = torch.empty(lib.ynet_batchnorm_workspace_doubles(C), device=x.device, dtype=torch.float64)
        ptr = lambda t: None if t is None else t.data_ptr()      # noqa: E731
        L.check(lib.ynet_batchnorm2d_fwd(x.data_ptr(), y.data_ptr(), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), mean.data_ptr(), invstd.data_ptr(),
                                         ws.data_ptr(), B, C, H * W, 1 if train else 0, float(factor), float(eps), _stream()), lib)
        ctx.train = bool(train)
        ctx.save_for_backward(x, gamma, mean if train else running_mean.detach().clone(), invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, invstd = ctx.saved_tensors
        B, C, H, W = x.shape
        lib = _lib()
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty(C, device=x.device, dtype=torch.float32) if gamma is not None else None
        db = torch.empty(C, device=x.device, dtype=torch.float32) if gamma is not None else None
        ws = torch.empty(lib.ynet_batchnorm_workspace_doubles(C), device=x.device, dtype=torch.float64)
        L.check(lib.ynet_batchnorm2d_bwd(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr() if gamma is not None else None, dx.data_ptr(),
                                         dg.data_ptr() if dg is not None else None, db.data_ptr() if db is not None else None, ws.data_ptr(), B, C, H * W,
                                         1 if ctx.train else 0, _stream()), lib)
        return dx, dg, db, None, None, None, None, None


def batch_norm2d(x, gamma, beta, running_mean, running_var, train: bool, factor: float, eps: float):
    """F.batch_norm of a [B, C, H, W] device tensor (ynet_batchnorm2d_fwd / _bwd): batch statistics and the running-statistics update when `train`, the running
    statistics otherwise; `factor` = the exponential average factor nn.BatchNorm2d computes (momentum, or 1 / num_batches_tracked)."""
    return _BatchNorm2dFn.apply(x, gamma, beta, running_mean, running_var, train, factor, eps)


class _AddReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, relu):
        _need_gpu(a, "add_relu")
        _need_gpu(b, "add_relu")
        if a.shape != b.shape:
            raise ValueError(f"add_relu: shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        lib = _lib()
        L.check(lib.ynet_add_relu(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), 1 if relu else 0, _stream()), lib)
        ctx.relu = bool(relu)
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.relu:
            return dy, dy, None
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        lib = _lib()
        L.check(lib.ynet_relu_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), _stream()), lib)
        return dx, dx, None


def add_relu(a, b, relu: bool = False):
    """[relu](a + b) in one launch: the adapters' residual add and the ReLU behind the sum (models/ynet.py:66,117-131)."""
    return _AddReluFn.apply(a, b, relu)


# ------------------------------------------------------------------------------------------------
# BCE-with-logits (mean)
# ------------------------------------------------------------------------------------------------
class _BCEFn(torch.autograd.Function):
    """With a gradient wanted the forward pass writes dx = (sigmoid(x) - t) * expected_grad / n next to the loss, and
    backward only rescales it on the device when the upstream gradient is not `expected_grad` (train_epoch passes
    loss_scale): the logits and targets are read once per step."""

    @staticmethod
    def forward(ctx, x, t, expected_grad):
        _need_gpu(x, "bce_with_logits input")
        _need_gpu(t, "bce_with_logits target")
        if x.shape != t.shape:
            raise ValueError(f"Target size ({tuple(t.shape)}) must be the same as input size ({tuple(x.shape)})")
        x, t = x.contiguous(), t.contiguous()
        lib = _lib()
        loss = torch.empty((), device=x.device, dtype=torch.float32)
        ws = torch.empty(lib.ynet_bce_workspace_bytes() // 8, device=x.device, dtype=torch.float64)
        ctx.dx = None
        if ctx.needs_input_grad[0]:
            expected_grad = float(expected_grad)
            if not (expected_grad != 0.0 and abs(expected_grad) < float("inf")):
                expected_grad = 1.0
            ctx.dx, ctx.expected = torch.empty_like(x), expected_grad
            L.check(lib.ynet_bce_logits_fwd_grad(x.data_ptr(), t.data_ptr(), x.numel(), expected_grad, loss.data_ptr(),
                                                 ctx.dx.data_ptr(), ws.data_ptr(), _stream()), lib)
        else:
            L.check(lib.ynet_bce_logits_fwd(x.data_ptr(), t.data_ptr(), x.numel(), loss.data_ptr(), ws.data_ptr(), _stream()), lib)
        ctx.save_for_backward(x, t)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        lib = _lib()
        g = g.contiguous().float()
        dx, ctx.dx = ctx.dx, None
        if dx is not None:                      # first backward through this node: the forward pass left the gradient
            L.check(lib.ynet_bce_grad_rescale(dx.data_ptr(), g.data_ptr(), ctx.expected, x.numel(), _stream()), lib)
        else:                                   # retain_graph: recompute from the saved logits
            dx = torch.empty_like(x)
            L.check(lib.ynet_bce_logits_bwd(x.data_ptr(), t.data_ptr(), g.data_ptr(), dx.data_ptr(), x.numel(), _stream()), lib)
        return dx, None, None


# ------------------------------------------------------------------------------------------------
# 1x1 predictor + BCE-with-logits (+ the predictor's dgrad) in one pass
# ------------------------------------------------------------------------------------------------
_pred_bce_ws = {}


class _PredBCEFn(torch.autograd.Function):
    """(logits, loss) = (conv1x1(x, W) + b, mean BCE(logits, target)).  The forward kernel also leaves the gradients
    for the upstream gradient `expected_grad` of the loss: dx for the decoder (the predictor's dgrad) and, when the
    predictor trains, dlogits for its wgrad.  The logits are returned for the read-out only (not differentiable)."""

    @staticmethod
    def forward(ctx, x, weight, bias, target, expected_grad, cache):
        _need_gpu(x, "pred_bce input")
        _need_gpu(target, "pred_bce target")
        dfr = _deferred.get(x.data_ptr()) if _deferred else None      # x = the output of a convolution that has not been launched yet (see `_deferred`)
        if dfr is not None and not (dfr["ref"]() is not None and dfr["shape"] == tuple(x.shape) and x.is_contiguous()):
            dfr = None
        x, target = x.contiguous(), target.contiguous()
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        if tuple(target.shape) != (B, cout, H, W):
            raise ValueError(f"Target size ({tuple(target.shape)}) must be the same as input size ({(B, cout, H, W)})")
        lib = _lib()
        wp = _cached(cache, weight, None, None, 1.0, "fwd")
        if dfr is not None:
            need_w_ = ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2])
            blob_t_ = _blob_target(target, B, cout, H, W)
            if (ctx.needs_input_grad[0] and not need_w_ and blob_t_ is not None and premask and cin == 32 and dfr["dims"] == (B, H, W)
                    and lib.ynet_conv2d_winograd_pred_bce_supported(B, H, W, 32, 32, cout, int(blob_t_[1].blob.shape[0]))):
                # ONE launch: the convolution, its ReLU, the predictor, the criterion and the predictor's data gradient (through that ReLU's backward)
                del _deferred[x.data_ptr()]
                pos, tmpl = blob_t_
                pos.record_stream(torch.cuda.current_stream(x.device))
                y = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
                loss = torch.empty((), device=x.device, dtype=torch.float32)
                key = (x.device, torch.cuda.current_stream().cuda_stream)
                ws = _pred_bce_ws.get(key)
                if ws is None:
                    ws = _pred_bce_ws[key] = torch.zeros(lib.ynet_pred_bce_workspace_bytes() // 8 + 1, device=x.device, dtype=torch.float64)
                expected_grad = float(expected_grad)
                if not (expected_grad != 0.0 and abs(expected_grad) < float("inf")):
                    expected_grad = 1.0
                ccache = dfr["cache"]
                ent = ccache.get("wino_fwd_0_32")       # (the entry ops.conv2d_raw keeps for the unfused launch of the same layer)
                if ent is None or ent[0] is not dfr["wp"]:
                    ent = ccache["wino_fwd_0_32"] = _wino_made((dfr["wp"], winograd_filter(dfr["wp"], 32, 32, 0, 32)))
                _wino_ready(ent)
                ctx.dx = torch.empty_like(x)
                ctx.dy = None
                conv2d_winograd_pred_bce_raw(dfr["src"], ent[1], dfr["bias"], wp, bias.detach() if bias is not None else None, cout, pos, tmpl, y, loss, ctx.dx, ws,
                                             B, H, W, expected_grad)
                bce_blob_stats["launches"] += 1
                conv_pred_bce_stats["fused"] += 1
                _unmaterialized[x.data_ptr()] = tuple(x.shape)
                ctx.premask_y = x.data_ptr()
                ctx.blob_keep = (pos, tmpl, dfr["keep"])
                ctx.expected = expected_grad
                ctx.has_bias = bias is not None
                ctx.save_for_backward(x, weight)
                ctx.mark_non_differentiable(y)
                ctx.set_materialize_grads(False)
                return y, loss
            materialize_deferred(x)
        y = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
        loss = torch.empty((), device=x.device, dtype=torch.float32)
        key = (x.device, torch.cuda.current_stream().cuda_stream)
        ws = _pred_bce_ws.get(key)
        if ws is None:
            ws = _pred_bce_ws[key] = torch.zeros(lib.ynet_pred_bce_workspace_bytes() // 8 + 1, device=x.device, dtype=torch.float64)
        expected_grad = float(expected_grad)
        if not (expected_grad != 0.0 and abs(expected_grad) < float("inf")):
            expected_grad = 1.0
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2])
        ctx.dx = torch.empty_like(x) if need_x else None
        ctx.dy = torch.empty_like(y) if need_w else None
        # x is the post-ReLU output of decoder[4][2]: the kernel writes dx with that ReLU's backward applied
        ctx.premask_y = x.data_ptr() if (need_x and premask and cin <= 32 and _is_relu_output(x)) else None
        blob_t = _blob_target(target, B, cout, H, W)
        if blob_t is not None:      # the target is a Gaussian blob per plane: computed from the positions, its planes are not read
            pos, tmpl = blob_t
            pos.record_stream(torch.cuda.current_stream(x.device))
            bce_blob_stats["launches"] += 1
            L.check(lib.ynet_pred_bce_blob(x.data_ptr(), cin * H * W, wp.data_ptr(), bias.detach().data_ptr() if bias is not None else None,
                                           pos.data_ptr(), tmpl.blob.data_ptr(), tmpl.blob.shape[0], tmpl.size, H, W, y.data_ptr(), loss.data_ptr(),
                                           ctx.dx.data_ptr() if need_x else None, ctx.dy.data_ptr() if need_w else None,
                                           ws.data_ptr(), B, cin, cout, expected_grad, 1 if ctx.premask_y is not None else 0, _stream()), lib)
            ctx.blob_keep = (pos, tmpl)
        else:
            L.check(lib.ynet_pred_bce(x.data_ptr(), cin * H * W, wp.data_ptr(), bias.detach().data_ptr() if bias is not None else None,
                                      target.data_ptr(), y.data_ptr(), loss.data_ptr(),
                                      ctx.dx.data_ptr() if need_x else None, ctx.dy.data_ptr() if need_w else None,
                                      ws.data_ptr(), B, cin, cout, H * W, expected_grad, 1 if ctx.premask_y is not None else 0,
                                      _stream()), lib)
        ctx.expected = expected_grad
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight)
        ctx.mark_non_differentiable(y)
        # (the logits are an output without a gradient: left to its default, autograd materialises a ZERO tensor of their size
        # for backward -- a 100 MB fill per decoder, 2 x 32 us on the step's critical path in the trace)
        ctx.set_materialize_grads(False)
        return y, loss

    @staticmethod
    def backward(ctx, _gy, g):
        x, weight = ctx.saved_tensors
        lib = _lib()
        if g is None:      # (the loss did not take part in the differentiated graph)
            return None, None, None, None, None, None
        g = g.contiguous().float()
        dx, dy, ctx.dx, ctx.dy = ctx.dx, ctx.dy, None, None
        if (ctx.needs_input_grad[0] and dx is None) or ((ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and dy is None):
            raise RuntimeError("pred_bce: backward through the fused predictor + criterion runs once (no retain_graph)")
        d_w = d_b = None
        if dx is not None:
            L.check(lib.ynet_bce_grad_rescale(dx.data_ptr(), g.data_ptr(), ctx.expected, dx.numel(), _stream()), lib)
            if ctx.premask_y is not None and premask:
                _premasked[dx.data_ptr()] = (ctx.premask_y, dx._version, tuple(dx.shape))
        if dy is not None:
            L.check(lib.ynet_bce_grad_rescale(dy.data_ptr(), g.data_ptr(), ctx.expected, dy.numel(), _stream()), lib)
            d_w, d_b = conv2d_wgrad_raw([x], dy, None, weight, ctx.has_bias and ctx.needs_input_grad[2])
            if not ctx.needs_input_grad[1]:
                d_w = None
        return dx, d_w, d_b, None, None, None


def pred_bce(x, weight, bias, target, expected_grad: float, cache: dict):
    """(predictor(x), BCEWithLogitsLoss()(predictor(x), target)) in one pass over x (see ynet_pred_bce)."""
    return _PredBCEFn.apply(x, weight, bias, target, expected_grad, cache)


def pred_bce_supported(x, weight) -> bool:
    return (torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and (x.shape[2] * x.shape[3]) % 4 == 0
            and weight.shape[0] <= 32 and tuple(weight.shape[2:]) == (1, 1))


def bce_with_logits(x, t, expected_grad: float = 1.0):
    """nn.BCEWithLogitsLoss() (mean).  `expected_grad`: the gradient the caller expects to arrive at the loss (a hint:
    any other value is handled by one more pass over dx in backward)."""
    return _BCEFn.apply(x, t, expected_grad)


# ------------------------------------------------------------------------------------------------
# read-out and heat-map construction (no gradient on this path)
# ------------------------------------------------------------------------------------------------
def softargmax2d(x: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] -> [B,C,2] (x, y) in pixels (utils/softargmax.py:55-81)."""
    if not torch.is_tensor(x):
        raise TypeError(f"Input input type is not a torch.Tensor. Got {type(x)}")
    if x.dim() != 4:
        raise ValueError(f"Invalid input shape, we expect BxCxHxW. Got: {x.shape}")
    t, c, bs = _plane_desc(x.detach(), "softargmax")
    B, _, H, W = t.shape
    if W % 4 == 0 and (bs % 4 or (H * W) % 4 or t.data_ptr() % 16):
        t = t.contiguous()
        bs = c * H * W
    out = torch.empty((B, c, 2), device=t.device, dtype=torch.float32)
    lib = _lib()
    L.check(lib.ynet_softargmax2d(t.data_ptr(), out.data_ptr(), B, c, bs, H, W, _stream()), lib)
    return out


def train_readout(pred_traj_map: torch.Tensor, pred_goal_map: torch.Tensor, gt_future: torch.Tensor, resize_factor: float):
    """utils/train_epoch.py:118-126 in two launches (ynet_train_readout): soft-argmax of every trajectory heat-map and of the
    goal decoder's last one, then per-trajectory ADE / FDE.  -> (pred_traj [B,P,2], pred_goal [B,1,2], ade [B], fde [B])."""
    for t, n in ((pred_traj_map, "trajectory maps"), (pred_goal_map, "goal maps"), (gt_future, "ground truth")):
        _need_gpu(t, "train_readout " + n)
    tm, gm = pred_traj_map.detach().contiguous(), pred_goal_map.detach().contiguous()
    B, P, H, W = tm.shape
    if tuple(gm.shape[0:1] + gm.shape[2:]) != (B, H, W) or tuple(gt_future.shape) != (B, P, 2):
        raise ValueError(f"train_readout: shapes {tuple(tm.shape)}, {tuple(gm.shape)}, {tuple(gt_future.shape)} do not fit")
    gt = gt_future.detach().contiguous()
    dev = tm.device
    pred_traj = torch.empty((B, P, 2), device=dev, dtype=torch.float32)
    pred_goal = torch.empty((B, 1, 2), device=dev, dtype=torch.float32)
    ade, fde = torch.empty(B, device=dev, dtype=torch.float32), torch.empty(B, device=dev, dtype=torch.float32)
    lib = _lib()
    L.check(lib.ynet_train_readout(tm.data_ptr(), P * H * W, gm.data_ptr(), gm.shape[1] * H * W, gm.shape[1] - 1, gt.data_ptr(),
                                   pred_traj.data_ptr(), pred_goal.data_ptr(), ade.data_ptr(), fde.data_ptr(), B, P, H, W,
                                   float(resize_factor), _stream()), lib)
    return pred_traj, pred_goal, ade, fde


class LazyPredictor:
    """The heat-map logits ``conv1x1(x, weight, bias)`` that nobody needs materialised: what a decoder returns with
    ``readout=True`` and what ``SoftArgmax2D`` turns into coordinates in one launch (pred_softargmax)."""

    def __init__(self, x, weight, bias):
        self.x, self.weight, self.bias = x, weight, bias

    @property
    def shape(self):
        return (self.x.shape[0], self.weight.shape[0], self.x.shape[2], self.x.shape[3])


def pred_softargmax_supported(x, weight) -> bool:
    """Can ``softargmax2d(conv1x1(x, weight))`` run as one launch (ynet_pred_softargmax)?"""
    if not (torch.is_tensor(x) and x.dim() == 4 and x.is_cuda and weight.dim() == 4 and weight.shape[2] == 1 and weight.shape[3] == 1):
        return False
    _, cin, H, W = x.shape
    return cin == weight.shape[1] and bool(_lib().ynet_pred_softargmax_supported(int(cin), int(weight.shape[0]), int(H), int(W)))


def pred_softargmax(x: torch.Tensor, weight: torch.Tensor, bias) -> torch.Tensor:
    """SoftArgmax2D(conv1x1(x, weight, bias)): [B,cin,H,W] -> [B,cout,2] (x, y) in pixels without materialising the
    [B,cout,H,W] logits (models/ynet.py:469 + utils/softargmax.py:55-81 as chained at utils/evaluate.py:259-262).
    Inference only."""
    for t, n in ((x, "input"), (weight, "weight")):
        _need_gpu(t, "pred_softargmax " + n)
    t, cin, bs = _plane_desc(x.detach(), "pred_softargmax")
    B, _, H, W = t.shape
    if bs % 4 or t.data_ptr() % 16:
        t = t.contiguous()
        bs = cin * H * W
    cout = weight.shape[0]
    lib = _lib()
    w = weight.detach().reshape(cout, cin).contiguous()
    b = bias.detach().contiguous() if bias is not None else None
    out = torch.empty((B, cout, 2), device=t.device, dtype=torch.float32)
    ws = torch.empty(lib.ynet_pred_softargmax_workspace_floats(B, H, W), device=t.device, dtype=torch.float32)
    L.check(lib.ynet_pred_softargmax(t.data_ptr(), bs, w.data_ptr(), b.data_ptr() if b is not None else None, out.data_ptr(),
                                     ws.data_ptr(), B, cin, cout, H, W, _stream()), lib)
    return out


def sigmoid(x: torch.Tensor) -> torch.Tensor:
    """Elementwise sigmoid of any tensor (models/ynet.py:585-586) on the same kernel."""
    _need_gpu(x, "sigmoid")
    flat = x.detach().contiguous().view(1, 1, 1, -1)
    return sigmoid_temp(flat, [0], 1.0).view(x.shape)


def sigmoid_temp(x: torch.Tensor, channels: Sequence[int], temperature: float) -> torch.Tensor:
    """sigmoid(x[:, channels] / temperature) (utils/evaluate.py:128-131) in one pass."""
    _need_gpu(x, "sigmoid_temp")
    x = x.detach().contiguous()
    B, C, H, W = x.shape
    sel = [int(c) % C for c in channels]
    if len(sel) > 8:          # the kernel takes up to 8 selected channels per launch
        return torch.cat([sigmoid_temp(x, sel[i:i + 8], temperature) for i in range(0, len(sel), 8)], dim=1)
    y = torch.empty((B, len(sel), H, W), device=x.device, dtype=torch.float32)
    lib = _lib()
    arr = (ctypes.c_int * len(sel))(*sel)
    L.check(lib.ynet_sigmoid_temp(x.data_ptr(), y.data_ptr(), B, C, H * W, arr, len(sel), float(temperature), _stream()), lib)
    return y


_patch_status = {}


def check_patch_windows(template_shape, xy, H: int, W: int):
    """Host-side check that every H x W window around the (x, y) coordinates ``xy`` lies inside the template
    (the reference would build a ragged patch and fail in torch.stack, utils/image_utils.py:40-63)."""
    SH, SW = int(template_shape[0]), int(template_shape[1])
    host = np.asarray(xy.detach().cpu() if torch.is_tensor(xy) else xy, dtype=np.float32).reshape(-1, 2)
    rx, ry = np.round(host[:, 0]).astype(np.int64), np.round(host[:, 1]).astype(np.int64)
    ox, oy = SW // 2 - rx, SH // 2 - ry
    if ((ox < 0) | (oy < 0) | (ox + W > SW) | (oy + H > SH)).any():
        raise ValueError(f"get_patch: a {H}x{W} window around one of the coordinates leaves the {SH}x{SW} template")
    return host


class AnalyticTemplate:
    """A heat-map template that is never materialised: ``gather_patches`` computes its windows in
    the kernel (ynet_heatmap_analytic), bit-identical to slicing the S x S array of utils/image_utils.py:15-37.
    kind 'dist': create_dist_mat(size) (normalised distance); kind 'gaussian': create_gaussian_heatmap_template(size,
    kernlen, nsig, normalize) -- only its kernlen x kernlen blob is kept (fp32, on the device)."""

    def __init__(self, kind: str, size: int, device, blob=None, normalize: bool = True):
        if kind not in ("dist", "gaussian"):
            raise ValueError("AnalyticTemplate: kind must be 'dist' or 'gaussian'")
        self.kind, self.size, self.device = kind, int(size), torch.device(device)
        self.shape = torch.Size((self.size, self.size))
        self.dtype = torch.float32
        self.normalize = bool(normalize)
        half = self.size // 2
        self.dmax = float(np.sqrt(np.float64(2 * half * half))) if normalize else 0.0
        if kind == "dist" and not normalize:
            raise ValueError("AnalyticTemplate: the un-normalised distance map is not on the Y-Net path")
        self.blob = None
        if kind == "gaussian":
            b = np.asarray(blob, dtype=np.float64)
            self.blob = torch.from_numpy(b.astype(np.float32)).contiguous().to(self.device)      # same cast as torch.Tensor(template)
        self.is_cuda = self.device.type == "cuda"

    def data_ptr(self):
        return id(self)

    def dim(self):
        return 2

    def to(self, device):
        device = torch.device(device)
        if device == self.device:
            return self
        out = AnalyticTemplate.__new__(AnalyticTemplate)
        out.__dict__.update(self.__dict__)
        out.device, out.is_cuda = device, device.type == "cuda"
        out.blob = None if self.blob is None else self.blob.to(device)
        return out

    def materialize(self) -> torch.Tensor:
        """The S x S tensor the reference would hold (tests / callers that index the template directly)."""
        S = self.size
        if self.kind == "dist":
            off = np.arange(S, dtype=np.int64) - S // 2
            d = np.sqrt((off[:, None] ** 2 + off[None, :] ** 2).astype(np.float64))
            return torch.Tensor(d / d.max() * 2).to(self.device)
        t = torch.zeros(S, S)
        m = self.blob.shape[0]
        lo = S // 2 - m // 2
        t[lo:lo + m, lo:lo + m] = self.blob.cpu()
        return t.to(self.device)


def gather_patches(template, xy, H: int, W: int) -> torch.Tensor:
    """[N,2] (x,y) coordinates -> [N,H,W] windows of `template` (utils/image_utils.py:40-63)."""
    if isinstance(template, AnalyticTemplate):
        return _analytic_patches(template, xy, H, W)
    _need_gpu(template, "get_patch template")
    if template.dim() != 2:
        raise ValueError("get_patch: template must be 2-D")
    template = template.contiguous()
    SH, SW = template.shape
    dev = template.device
    if torch.is_tensor(xy) and xy.is_cuda:
        coords = xy.detach().reshape(-1, 2).float().contiguous()
    else:
        host = np.ascontiguousarray(check_patch_windows((SH, SW), xy, H, W))
        coords = torch.from_numpy(host).to(dev, non_blocking=True)
    n = coords.shape[0]
    out = torch.empty((n, H, W), device=dev, dtype=torch.float32)
    st = _patch_status.get(dev)
    if st is None:
        st = _patch_status[dev] = torch.zeros(1, device=dev, dtype=torch.int32)
    lib = _lib()
    L.check(lib.ynet_gather_patch(template.data_ptr(), SH, SW, coords.data_ptr(), out.data_ptr(), n, H, W,
                                  st.data_ptr(), _stream()), lib)
    return out


def _analytic_patches(t: AnalyticTemplate, xy, H: int, W: int) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("get_patch: the MI355X path runs on HIP devices only (no CPU fallback exists by design)")
    dev = t.device
    if dev.index is not None and dev.index != torch.cuda.current_device():
        raise RuntimeError(f"get_patch: template lives on {dev} but the current HIP device is cuda:{torch.cuda.current_device()}")
    if torch.is_tensor(xy) and xy.is_cuda:
        coords = xy.detach().reshape(-1, 2).float().contiguous()
    else:
        coords = torch.from_numpy(np.ascontiguousarray(check_patch_windows(t.shape, xy, H, W))).to(dev, non_blocking=True)
    n = coords.shape[0]
    out = torch.empty((n, H, W), device=dev, dtype=torch.float32)
    st = _status_flag(_patch_status, dev)
    lib = _lib()
    L.check(lib.ynet_heatmap_analytic(coords.data_ptr(), out.data_ptr(), n, H, W, t.size, 0 if t.kind == "dist" else 1, t.dmax,
                                      t.blob.data_ptr() if t.blob is not None else None,
                                      t.blob.shape[0] if t.blob is not None else 0, st.data_ptr(), _stream()), lib)
    if t.kind == "gaussian" and _bce_blob_allowed:
        # the fused predictor + criterion computes such a target from the positions instead of reading its planes (ynet_pred_bce_blob):
        # remembered by storage, size and version -- a tensor somebody wrote into since is read as the tensor it is
        for k_ in [k_ for k_, e_ in _blob_targets.items() if e_[0]() is None]:
            del _blob_targets[k_]
        _blob_targets[out.data_ptr()] = (weakref.ref(out), out._version, n, H, W, coords, t, coords._version)
    return out


_bce_blob_allowed = _os.environ.get("YNET_BCE_BLOB_TARGET", "1") != "0"      # (development switch)
_blob_targets = {}
bce_blob_stats = {"launches": 0}


def _blob_target(target, B, cout, H, W):
    """(positions [B * cout, 2], AnalyticTemplate) if `target` [B, cout, H, W] is, untouched, what gather_patches wrote for a Gaussian template."""
    e = _blob_targets.get(target.data_ptr()) if _bce_blob_allowed else None
    if (e is None or e[0]() is None or not target.is_contiguous() or target._version != e[1] or (e[2], e[3], e[4]) != (B * cout, H, W) or W % 4 != 0
            or e[5]._version != e[7]):
        return None
    base = e[0]()
    if base.untyped_storage().data_ptr() != target.untyped_storage().data_ptr() or target.numel() != base.numel():
        return None
    return e[5], e[6]


def pad_planes(x: torch.Tensor, division_factor: int = 32) -> torch.Tensor:
    """pad (utils/image_utils.py:95-107) of a [..., H, W] device tensor: zero border at the bottom / right up to a multiple of
    `division_factor` (cv2.copyMakeBorder with BORDER_CONSTANT)."""
    _need_gpu(x, "pad")
    H, W = x.shape[-2:]
    Hp, Wp = -(-H // division_factor) * division_factor, -(-W // division_factor) * division_factor
    if (Hp, Wp) == (H, W):
        return x
    xc = x.contiguous()
    y = torch.empty(tuple(x.shape[:-2]) + (Hp, Wp), device=x.device, dtype=torch.float32)
    lib = _lib()
    L.check(lib.ynet_pad2d(xc.data_ptr(), y.data_ptr(), max(1, xc.numel() // (H * W)), H, W, Hp, Wp, _stream()), lib)
    return y


def seg_onehot_pad(labels: torch.Tensor, classes: int = 6, division_factor: int = 32) -> torch.Tensor:
    """pad + preprocess_image_for_segmentation(seg_mask=True) (utils/image_utils.py:74-81, 95-107): an integer label map
    [H, W] on the device -> one-hot fp32 planes [classes, Hp, Wp]; the padded border is class 0 (it is padded before the
    encoding, as in the reference)."""
    if not torch.is_tensor(labels) or not labels.is_cuda:
        raise RuntimeError("seg_onehot_pad: the MI355X path runs on HIP devices only (no CPU fallback exists by design)")
    if labels.dim() != 2 or labels.is_floating_point() and bool((labels != labels.round()).any()):
        raise ValueError("seg_onehot_pad: expected a 2-D map of integer class labels")
    lab = labels.to(torch.int32).contiguous()
    H, W = lab.shape
    Hp, Wp = -(-H // division_factor) * division_factor, -(-W // division_factor) * division_factor
    y = torch.empty((int(classes), Hp, Wp), device=lab.device, dtype=torch.float32)
    lib = _lib()
    L.check(lib.ynet_seg_onehot_pad(lab.data_ptr(), y.data_ptr(), H, W, Hp, Wp, int(classes), _stream()), lib)
    return y


def cv_round(v: float) -> int:
    """OpenCV's cvRound: nearest integer, halves to even (what cv2.resize applies to src * factor for the output size)."""
    return int(np.rint(np.float64(v)))


def resize_nearest(labels: torch.Tensor, factor: float) -> torch.Tensor:
    """resize(images, factor, seg_mask=True) (utils/image_utils.py:83-87 = cv2.resize(im, (0, 0), fx=factor, fy=factor,
    interpolation=cv2.INTER_NEAREST)) of ONE integer label map [H, W] on the device, restating OpenCV's published rule (see
    ynet_resize_nearest).  PARITY UNPINNED: cv2 is absent from the image.  -> [round(H * f), round(W * f)], same dtype."""
    if not torch.is_tensor(labels) or not labels.is_cuda:
        raise RuntimeError("resize_nearest: the MI355X path runs on HIP devices only (no CPU fallback exists by design)")
    if labels.dim() != 2 or labels.is_floating_point() and bool((labels != labels.round()).any()):
        raise ValueError("resize_nearest: expected a 2-D map of integer class labels")
    if not factor > 0:
        raise ValueError("resize_nearest: the factor must be positive")
    lab = labels.to(torch.int32).contiguous()
    H, W = lab.shape
    Ho, Wo = cv_round(H * float(factor)), cv_round(W * float(factor))
    if Ho < 1 or Wo < 1:
        raise ValueError(f"resize_nearest: {H}x{W} * {factor} leaves an empty map")
    out = torch.empty((Ho, Wo), device=lab.device, dtype=torch.int32)
    lib = _lib()
    L.check(lib.ynet_resize_nearest(lab.data_ptr(), out.data_ptr(), H, W, Ho, Wo, float(factor), float(factor), _stream()), lib)
    return out.to(labels.dtype)


def rot90_flip(image: torch.Tensor, k: int = 0, flip: bool = False) -> torch.Tensor:
    """k times cv2.rotate(image, ROTATE_90_COUNTERCLOCKWISE), then (flip) cv2.flip(image, 1) -- utils/data_utils.py:133-134, 162 -- of a
    device tensor [H, W] or [..., H, W] with 4-byte elements (int32 label maps, fp32 planes): np.rot90(image, k) / np.fliplr over the last two
    dimensions, bit-exact (an index permutation, ynet_rot90_flip)."""
    _need_gpu(image, "rot90_flip", None)
    if image.dim() < 2 or image.element_size() != 4:
        raise ValueError("rot90_flip: expected [..., H, W] with 4-byte elements (int32 / float32)")
    k = int(k) % 4
    src = image.contiguous()
    H, W = src.shape[-2:]
    out = torch.empty(tuple(src.shape[:-2]) + ((W, H) if k & 1 else (H, W)), device=src.device, dtype=src.dtype)
    if src.numel() == 0:
        return out
    lib = _lib()
    L.check(lib.ynet_rot90_flip(src.data_ptr(), out.data_ptr(), src.numel() // (H * W), H, W, k, 1 if flip else 0, _stream()), lib)
    return out


def rot_coords(xy: torch.Tensor, center, matrix, offset) -> torch.Tensor:
    """(xy - center) @ matrix + offset on a float64 device tensor [n, 2], in place (the coordinate side of rot() / fliplr(),
    utils/data_utils.py:127-131,140-141,158-161,169-170; ynet_rot_coords)."""
    _need_gpu(xy, "rot_coords", (torch.float64,))
    if xy.dtype != torch.float64 or xy.dim() != 2 or xy.shape[1] != 2 or not xy.is_contiguous():
        raise ValueError("rot_coords: expected a contiguous float64 tensor [n, 2]")
    if xy.shape[0]:
        lib = _lib()
        L.check(lib.ynet_rot_coords(xy.data_ptr(), xy.shape[0], float(center[0]), float(center[1]), float(matrix[0][0]), float(matrix[0][1]),
                                    float(matrix[1][0]), float(matrix[1][1]), float(offset[0]), float(offset[1]), _stream()), lib)
    return xy


def kmeans2d(points: torch.Tensor, init_idx: torch.Tensor, tol: float = 1e-3, iter_limit: int = 1000):
    """Lloyd's k-means of P independent point sets on the device (one workgroup per set; TTST).
    points [P,N,2] fp32 with integer-valued coordinates, init_idx [P,K] int32 -> (centers [P,K,2], status [P] int32:
    bit 0 = a cluster went empty (centres undefined, redo on the host path), bits 8.. = iterations)."""
    _need_gpu(points, "kmeans2d points")
    P, N, _ = points.shape
    K = init_idx.shape[1]
    pts = points.contiguous().float()
    idx = init_idx.to(device=points.device, dtype=torch.int32).contiguous()
    centers = torch.empty((P, K, 2), device=points.device, dtype=torch.float32)
    status = torch.empty(P, device=points.device, dtype=torch.int32)
    lib = _lib()
    L.check(lib.ynet_kmeans2d(pts.data_ptr(), idx.data_ptr(), centers.data_ptr(), status.data_ptr(), P, N, K,
                              float(tol), int(iter_limit), _stream()), lib)
    return centers, status


_sample_status = {}


def _status_flag(store, dev):
    st = store.get(dev)
    if st is None:
        st = store[dev] = torch.zeros(1, device=dev, dtype=torch.int32)
    return st


def multinomial(prob: torch.Tensor, num_samples: int, replacement: bool = False, rel_threshold=None, seed=0) -> torch.Tensor:
    """torch.multinomial(prob [rows, n], num_samples, replacement) on the device with the documented Philox4x32-10
    generator of ynet_multinomial (include/ynet_hip.h): the draws are a pure function of (prob, seed), reproduced on
    the CPU by oracle/ynet_oracle.py:device_multinomial.  -> int64 [rows, num_samples]."""
    _need_gpu(prob, "multinomial")
    if prob.dim() != 2:
        raise ValueError("multinomial: expected a [rows, n] matrix")
    rows, n = prob.shape
    if prob.stride(1) != 1:
        prob = prob.contiguous()
    out = torch.empty((rows, num_samples), device=prob.device, dtype=torch.int64)
    st = _status_flag(_sample_status, prob.device)
    lib = _lib()
    if torch.is_tensor(seed):
        # the seed as a device input (one int64 element, read by the kernel): what a captured evaluation sweep passes
        if not (seed.is_cuda and seed.dtype == torch.int64 and seed.numel() == 1):
            raise ValueError("multinomial: a tensor seed must be one int64 element on the device")
        L.check(lib.ynet_multinomial_devseed(prob.data_ptr(), rows, prob.stride(0) if rows > 1 else n, n, int(num_samples),
                                             1 if replacement else 0, float(rel_threshold or 0.0), seed.data_ptr(),
                                             out.data_ptr(), st.data_ptr(), _stream()), lib)
        return out
    L.check(lib.ynet_multinomial(prob.data_ptr(), rows, prob.stride(0) if rows > 1 else n, n, int(num_samples),
                                 1 if replacement else 0, float(rel_threshold or 0.0), int(seed) & (2 ** 64 - 1),
                                 out.data_ptr(), st.data_ptr(), _stream()), lib)
    return out


def cws_prior(sig: torch.Tensor, mean_xy: torch.Tensor, dist_xy: torch.Tensor, sigma_factor: float, ratio: float, rot: bool,
              want_map: bool = False, want_xy: bool = True):
    """Conditioned-waypoint-sampling prior (utils/evaluate.py:9-34, 198-211) for rows = mean_xy.shape[0] (a multiple of the
    B persons of sig [B, H, W]; row r uses person r % B): (normalised map [rows,H,W] or None, expectation [rows,2] or None)."""
    _need_gpu(sig, "cws_prior sigmoid map")
    B, H, W = sig.shape
    if sig.stride(2) != 1 or sig.stride(1) != W:
        sig = sig.contiguous()
    mean_xy, dist_xy = mean_xy.detach().float().contiguous(), dist_xy.detach().float().contiguous()
    rows = mean_xy.shape[0]
    if rows % B or tuple(dist_xy.shape) != (rows, 2) or tuple(mean_xy.shape) != (rows, 2):
        raise ValueError(f"cws_prior: {rows} rows for {B} persons")
    out_map = torch.empty((rows, H, W), device=sig.device, dtype=torch.float32) if want_map else None
    out_xy = torch.empty((rows, 2), device=sig.device, dtype=torch.float32) if want_xy else None
    lib = _lib()
    L.check(lib.ynet_cws_prior(sig.data_ptr(), sig.stride(0) if B > 1 else H * W, B, mean_xy.data_ptr(), dist_xy.data_ptr(),
                               rows, H, W, float(sigma_factor), float(ratio), 1 if rot else 0,
                               out_map.data_ptr() if want_map else None, out_xy.data_ptr() if want_xy else None, _stream()), lib)
    return out_map, out_xy


def check_patch_status():
    """Raise if a device-side coordinate ever left the template (checked at sync points)."""
    for dev, st in _patch_status.items():
        if int(st.item()) != 0:
            st.zero_()
            raise ValueError("get_patch: a window left the template (device-side coordinates)")
    for dev, st in _sample_status.items():
        if int(st.item()) != 0:
            st.zero_()
            raise RuntimeError("invalid multinomial distribution (a row with too few positive entries)")
