"""Data-parallel sharding of a scene batch over the GPUs of one node (new: the reference is
single-process).

One process per GPU (torchrun); backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests.
Every batch of trajectories is split into contiguous row shards; the semantic map, templates and
weights are replicated.  BCE is a mean over B*pred*H*W (models/trainer.py:206), so a rank scales its
local loss by B_local / B_global and ONE all-reduce(SUM) of the flat trainable-gradient buffer per
step (32 KB for mosa_1 ... 6.6 MB full model; the loss rides in its last slot) reproduces the single-GPU gradient; every rank then
applies the identical Adam update.  The flat buffer is persistent and ``p.grad`` are views into it,
so the collective runs in place on the compute stream with no packing copies.
"""
import os
import weakref
from typing import List, Tuple

import torch
import torch.distributed as dist


def _close_comm(lib, comm):
    try:
        lib.ynet_comm_destroy(comm)
    except Exception:      # noqa: BLE001 -- interpreter teardown
        pass


class DataParallel:
    def __init__(self, params, group=None, collective=None, force=None):
        """``force`` (default: YNET_DP_FORCE=1): issue every collective even in a world of ONE rank -- all-reduce, scalar
        sum, seed broadcast all go through torch.distributed / the one-shot kernel and the captured step takes the same
        shape as on N ranks (two graphs around an eager RCCL all-reduce, or the one-shot kernel recorded inside one graph).
        The sums of one rank are the inputs, so results equal ``dp=None`` bit for bit; it is how the RCCL stream hand-off is
        exercised on a box with a single GPU (tests/test_gpu_dp.py, `bench.py --gpus 1` under YNET_DP_FORCE=1).
        ``collective``: "rccl" (torch.distributed all_reduce: RCCL over xGMI, or gloo in tests; the default) or "oneshot"
        (ynet_allreduce_sum: every rank reads its peers' buffers through HIP IPC in one hop, rank-ordered sums; one node,
        <= 16 ranks; default when YNET_ALLREDUCE=oneshot).  torch.distributed stays the control plane either way."""
        if not dist.is_initialized():
            raise RuntimeError("DataParallel needs an initialised torch.distributed process group")
        self.group = group
        self.collective = collective or os.environ.get("YNET_ALLREDUCE", "rccl")
        self._comm = None
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if force is None:
            force = os.environ.get("YNET_DP_FORCE", "0") == "1"
        self.active = self.world > 1 or bool(force)       # do the collectives run at all?
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        # one extra slot carries the shard-weighted loss, so a step needs exactly ONE collective
        self.n_grad = n
        self.flat = torch.zeros(n + 1, device=dev, dtype=torch.float32)
        self._views = []
        off = 0
        for p in self.params:
            self._views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.bind()
        self.transport_note = None          # why the requested transport was not used (bench.py prints it)
        if self.collective == "oneshot" and self.active:
            self._connect_oneshot()
            self._self_test_oneshot()
        if self._comm is not None:
            self._finalizer = weakref.finalize(self, _close_comm, self._lib, self._comm)      # IPC mappings are released at teardown

    def _agree(self, ok: int) -> bool:
        """Every rank takes the same decision (MIN over ranks), through the control plane."""
        verdict = torch.tensor([int(ok)], dtype=torch.int32, device=self.flat.device)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=self.group)
        return int(verdict.item()) == 1

    def _fall_back(self, what: str, mine_ok: bool, why: str):
        import warnings
        self.transport_note = f"one-shot all-reduce {what} failed" + (f" on this rank: {why}" if not mine_ok else " on a peer") + \
            "; using torch.distributed"
        warnings.warn(self.transport_note)
        self.close()
        self.collective = "rccl"

    def _connect_oneshot(self):
        """Mailbox allocation, handle exchange and IPC mapping; a failure on ANY rank (no fine-grained memory, a handle that
        cannot be opened) makes ALL ranks fall back to torch.distributed together -- nobody is left waiting at a barrier."""
        import ctypes
        from . import _lib as L
        if not self.flat.is_cuda:
            raise RuntimeError("the one-shot all-reduce exchanges device buffers: parameters must live on a HIP device")
        lib = L.load()
        self._lib = lib
        comm = ctypes.c_void_p()
        nb = lib.ynet_comm_handle_bytes()
        mine = ctypes.create_string_buffer(nb)
        ok, why = 1, ""
        try:
            L.check(lib.ynet_comm_create(self.rank, self.world, self.flat.numel(), ctypes.byref(comm)), lib)
            self._comm = comm
            L.check(lib.ynet_comm_export(comm, mine), lib)
        except Exception as e:      # noqa: BLE001
            ok, why = 0, f"{type(e).__name__}: {e}"
        handles = [None] * self.world
        dist.all_gather_object(handles, mine.raw if ok else b"", group=self.group)       # control plane: torch.distributed
        if ok and all(len(h) == nb for h in handles):
            try:
                L.check(lib.ynet_comm_connect(comm, b"".join(handles)), lib)
            except Exception as e:      # noqa: BLE001
                ok, why = 0, f"{type(e).__name__}: {e}"
        elif ok:
            ok, why = 0, "a peer exported no handle"
        if not self._agree(ok):
            self._fall_back("set-up", bool(ok), why)

    def _self_test_oneshot(self):
        """Start-up self-test of the one-shot transport: all-reduce a buffer of rank ids (expected: world * (world - 1) / 2
        in every element, on every rank).  Any failure -- a time-out, wrong sums, an IPC mapping that does not observe the
        peers -- falls back to torch.distributed (RCCL) on ALL ranks together and records the reason."""
        if self._comm is None:
            return
        ok, why = 1, ""
        try:
            probe = torch.full_like(self.flat, float(self.rank))
            from . import _lib as L
            L.check(self._lib.ynet_allreduce_sum(self._comm, probe.data_ptr(), probe.numel(),
                                                 torch.cuda.current_stream(probe.device).cuda_stream), self._lib)
            want = self.world * (self.world - 1) / 2.0
            if not bool((probe == want).all()):       # (synchronises; NaN after a time-out compares unequal)
                ok, why = 0, f"self-test sums differ from {want} (or a peer timed out)"
            elif self._lib.ynet_comm_status(self._comm) != 0:
                ok, why = 0, "a wait for a peer timed out"
            elif os.environ.get("YNET_ONESHOT_FORCE_FAIL") == str(self.rank):      # (tests: the collective fall-back path)
                ok, why = 0, "forced failure (YNET_ONESHOT_FORCE_FAIL)"
        except Exception as e:      # noqa: BLE001
            ok, why = 0, f"{type(e).__name__}: {e}"
        if not self._agree(ok):
            self._fall_back("self-test", bool(ok), why)

    def check(self):
        """Raise if the one-shot transport ever timed out waiting for a peer (its kernel poisons the gradients and the loss
        with NaN in that case; this names the cause).  Called at the synchronisation points of an epoch."""
        if self._comm is not None and self._lib.ynet_comm_status(self._comm) != 0:
            raise RuntimeError(f"rank {self.rank}: the one-shot all-reduce timed out waiting for a peer (~20 s): the gradients of "
                               f"that step were not reduced (buffer poisoned with NaN); a rank crashed or stalled")

    def close(self):
        fin = getattr(self, "_finalizer", None)
        if fin is not None:
            fin.detach()
            self._finalizer = None
        if self._comm is not None:
            self._lib.ynet_comm_destroy(self._comm)
            self._comm = None

    # -- sharding ---------------------------------------------------------------------------
    def shard(self, n: int) -> Tuple[int, int]:
        """Rows [lo, hi) of an n-row batch owned by this rank (ragged tails allowed, may be empty)."""
        base, extra = divmod(n, self.world)
        lo = self.rank * base + min(self.rank, extra)
        return lo, lo + base + (1 if self.rank < extra else 0)

    # -- gradients --------------------------------------------------------------------------
    def bind(self):
        for p, v in zip(self.params, self._views):
            p.grad = v

    def zero_grad(self):
        self.flat.zero_()
        self.bind()

    def stage(self, loss: torch.Tensor = None):
        """Before the collective: every p.grad is its view of the flat buffer, the shard-weighted loss sits in the last slot."""
        for p, v in zip(self.params, self._views):      # a rank with an empty shard never ran backward
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
        if loss is not None:
            self.flat[-1:].copy_(loss.detach().reshape(1))

    def allreduce(self):
        """The ONE collective of a step: SUM all-reduce of the flat buffer, in place, on the current stream."""
        if self.active:
            if self._comm is not None:
                from . import _lib as L
                L.check(self._lib.ynet_allreduce_sum(self._comm, self.flat.data_ptr(), self.flat.numel(),
                                                     torch.cuda.current_stream(self.flat.device).cuda_stream), self._lib)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    def capturable_collective(self) -> bool:
        """Can allreduce() be recorded into a hipGraph?  The one-shot kernel can (its call counter lives on the device);
        torch.distributed collectives stay between the two graphs of a captured step (utils/step_graph.py)."""
        return self._comm is not None

    def loss_value(self) -> torch.Tensor:
        return self.flat[-1].clone()

    def allreduce_grads(self, loss: torch.Tensor = None) -> torch.Tensor:
        """stage + allreduce; returns the global loss.  (The captured step runs the three parts separately: a
        torch.distributed collective stays between its two hipGraphs, the one-shot kernel is recorded into its single graph,
        utils/step_graph.py.)"""
        self.stage(loss)
        self.allreduce()
        return self.loss_value()

    def sum_scalar(self, t: torch.Tensor) -> torch.Tensor:
        self.check()          # (the epoch's synchronisation point)
        if self.active:
            t = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def gather_rows(self, t: torch.Tensor, sizes: List[int]) -> torch.Tensor:
        """Concatenate per-rank row blocks of (possibly different) length sizes[r] on every rank."""
        if not self.active:
            return t
        m = max(sizes)
        pad = torch.zeros((m,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        pad[:t.shape[0]] = t
        out = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(out, pad, group=self.group)
        return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)

    def shared_seed(self) -> int:
        """A random 63-bit seed drawn on rank 0 and broadcast: every rank seeds its scene-shuffling generator with it,
        so all ranks walk the scenes in the same order (train_epoch shards trajectory[i:i+bs] of the SAME scene)."""
        dev = self.flat.device
        t = torch.zeros(1, dtype=torch.int64, device=dev)
        if self.rank == 0:
            t[0] = int(torch.randint(0, 2 ** 62, (1,)).item())
        if self.active:
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        return int(t.item())

    def shard_sizes(self, n: int) -> List[int]:
        base, extra = divmod(n, self.world)
        return [base + (1 if r < extra else 0) for r in range(self.world)]


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """torchrun-style bring-up: returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # YNET_DIST_BACKEND=gloo: development runs of the multi-process path on a box with fewer GPUs than ranks
            backend = os.environ.get("YNET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        # every backend: the HIP kernels launch on the current device's streams.  Only when this rank HAS a device of its own:
        # CPU-only workers (gloo tests) and boxes with fewer GPUs than ranks (YNET_DIST_BACKEND=gloo development runs, every
        # rank on device 0) must not touch HIP here / must not get "invalid device ordinal".
        n_dev = torch.cuda.device_count()
        if n_dev > 0 and os.environ.get("YNET_BENCH_SINGLE_DEVICE") != "1":
            if local < n_dev:
                torch.cuda.set_device(local)
            elif backend == "nccl":
                raise RuntimeError(f"LOCAL_RANK {local} but only {n_dev} HIP devices are visible: RCCL needs one GPU per rank "
                                   f"(YNET_DIST_BACKEND=gloo YNET_BENCH_SINGLE_DEVICE=1 runs every rank on cuda:0)")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
