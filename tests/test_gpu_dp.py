"""The REAL multi-rank product path on the GPU: two fresh processes (torch.distributed.run) run
utils/train_epoch.train_epoch(dp=...) on the HIP kernels; gradients, loss, metrics and post-Adam weights must equal the
single-process run of the same epochs.  On a 1-GPU box both ranks share cuda:0 and the collective runs over gloo; the
arithmetic (shards, B_local/B_global loss weights, expected_grad of the one-pass BCE, empty shards, one all-reduce per
step, epoch-end (sum, sum, count) reduction) is exactly what RCCL ranks execute.  SURVEY 8(e), utils/train_epoch.py:44."""
import os
import socket
import sys

import pytest
import torch

from conftest import ROOT, launch
from dp_worker import case_inputs, run_epochs

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# (rows, batch size): even shards; ragged shards (3 -> 2 + 1) and a ragged last batch; batch of 1 -> rank 1's shard is EMPTY.
# collective "oneshot": ynet_allreduce_sum (HIP-IPC mailboxes, one hop, rank-ordered sums) instead of torch.distributed.
# Four ranks: shards of one trajectory each, and (batch 3) one EMPTY rank in every step -- the rank-ordered sums of the
# one-shot all-reduce over four mailboxes; eight ranks (the node size of the north star) with one trajectory each.
@pytest.mark.parametrize("n_rows,batch_size,collective,world", [(8, 4, "rccl", 2), (8, 3, "rccl", 2), (3, 1, "rccl", 2), (8, 3, "oneshot", 2),
                                                                (3, 1, "oneshot", 2), (8, 4, "oneshot", 4), (8, 3, "rccl", 4), (8, 8, "oneshot", 8)])
def test_two_rank_train_epoch_equals_single_process(dev, tmp_path, n_rows, batch_size, collective, world):
    out = str(tmp_path / "dp.pt")
    n_gpu = torch.cuda.device_count()
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "YNET_ALLREDUCE": collective}
    if n_gpu < world:
        env.update(YNET_DIST_BACKEND="gloo", YNET_BENCH_SINGLE_DEVICE="1")
    rc, tail = launch([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                       "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                       os.path.join(ROOT, "tests", "dp_worker.py"), out, str(n_rows), str(batch_size)], env=env, timeout=600)
    assert rc == 0, tail
    got = torch.load(out, weights_only=False)
    assert got["world"]["world_size"] == world and len({r["pid"] for r in got["world"]["ranks"]}) == world
    assert got["world"]["backend"] == ("nccl" if n_gpu >= world else "gloo")
    assert got["world"]["collective"] == collective and got["world"]["transport_note"] is None      # (the one-shot start-up self-test passed)

    cfg, sd, scene, traj = case_inputs(n_rows)
    want = run_epochs(cfg, sd, scene, traj, batch_size, dev, lambda m: None)
    for (a1, f1, l1), (a0, f0, l0) in zip(got["results"], want["results"]):
        assert abs(l1 - l0) <= 2e-5 * abs(l0), (l1, l0)
        assert abs(a1 - a0) <= 1e-4 and abs(f1 - f0) <= 1e-4, ((a1, f1), (a0, f0))
    for n, w in want["grads"].items():      # gradients of the LAST step (after identical earlier updates)
        g = got["grads"][n]
        assert float((g - w).abs().max()) <= 2e-4 * float(w.abs().max()) + 1e-7, n
    # evaluate(dp=) after the training epochs: the gathered per-trajectory errors equal the single-process sweep
    import numpy as np
    np.testing.assert_allclose(got["eval"][2], want["eval"][2], rtol=0, atol=1e-4)
    np.testing.assert_allclose(got["eval"][3], want["eval"][3], rtol=0, atol=1e-4)
    assert abs(got["eval"][0] - want["eval"][0]) <= 1e-4 and abs(got["eval"][1] - want["eval"][1]) <= 1e-4
    steps = 2 * ((n_rows + batch_size - 1) // batch_size)
    for n, w in want["weights"].items():    # every Adam step moves a weight by <= lr: the two runs stay within a few % of that
        big = want["grads"][n].abs() > 1e-3 * want["grads"][n].abs().max()      # (Adam turns a rounding-level gradient into +-lr)
        d = float((got["weights"][n] - w)[big].abs().max()) if bool(big.any()) else 0.0
        assert d <= 0.05 * 1e-3 * steps, (n, d)


def test_oneshot_self_test_failure_falls_back_to_torch_distributed_on_all_ranks(dev, tmp_path):
    """the start-up self-test of the one-shot transport (all-reduce of the rank ids); a failure on ONE
    rank makes EVERY rank fall back to torch.distributed together, the reason is recorded, and the epochs still equal the
    single-process run."""
    out = str(tmp_path / "dp.pt")
    n_gpu = torch.cuda.device_count()
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "YNET_ALLREDUCE": "oneshot", "YNET_ONESHOT_FORCE_FAIL": "1"}
    if n_gpu < 2:
        env.update(YNET_DIST_BACKEND="gloo", YNET_BENCH_SINGLE_DEVICE="1")
    rc, tail = launch([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                       "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                       os.path.join(ROOT, "tests", "dp_worker.py"), out, "8", "3"], env=env, timeout=600)
    assert rc == 0, tail
    got = torch.load(out, weights_only=False)
    assert got["world"]["requested"] == "oneshot" and got["world"]["collective"] == "rccl"
    assert "self-test failed on a peer" in got["world"]["transport_note"]      # rank 0's view of rank 1's failure
    cfg, sd, scene, traj = case_inputs(8)
    want = run_epochs(cfg, sd, scene, traj, 3, dev, lambda m: None)
    for (a1, f1, l1), (a0, f0, l0) in zip(got["results"], want["results"]):
        assert abs(l1 - l0) <= 2e-5 * abs(l0) and abs(a1 - a0) <= 1e-4 and abs(f1 - f0) <= 1e-4


def test_bench_launcher_reports_a_failing_rank(dev):
    """`python bench.py --gpus 2` starts its ranks itself; a rank that dies must make the launcher exit non-zero and show that
    rank's tail (YNET_BENCH_FAIL_RANK is the test hook)."""
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "YNET_BENCH_FAIL_RANK": "1"}
    if torch.cuda.device_count() < 2:
        env.update(YNET_DIST_BACKEND="gloo", YNET_BENCH_SINGLE_DEVICE="1")
    rc, tail = launch([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2",
                       "--no-cpu-baseline", "--no-roofline", "--no-repeats"], env=env, timeout=600)
    assert rc != 0
    assert "rank 1 fails on purpose" in tail and "a rank failed" in tail, tail[-1500:]


@pytest.mark.parametrize("collective", ["rccl", "oneshot"])
def test_bench_multi_rank_line_reports_the_collective_and_the_step_graphs(dev, collective):
    """`python bench.py --gpus 2` on the development path (one device, gloo control plane) prints the fields the
    driver's SCALE run will carry -- the process group, the transport, `allreduce_us_per_step` -- and how the replayed step is
    launched: with a torch.distributed collective two graphs around the eager all-reduce (`replayed_step_split_ms` = graph A /
    collective / graph B), with the one-shot HIP-IPC all-reduce ONE graph per step with the collective recorded inside it
    (device-resident call counter, csrc/comm.hip)."""
    import json
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "YNET_ALLREDUCE": collective}
    n_gpu = torch.cuda.device_count()
    if n_gpu < 2:
        env.update(YNET_DIST_BACKEND="gloo", YNET_BENCH_SINGLE_DEVICE="1")
    rc, tail = launch([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "3", "--batch", "4",
                       "--no-cpu-baseline", "--no-roofline", "--no-repeats", "--no-sustained"], env=env, timeout=900)
    assert rc == 0, tail[-3000:]
    lines = [ln[ln.index('{"metric"'):] for ln in tail.splitlines() if '{"metric"' in ln]
    assert lines, tail[-2000:]
    d = json.loads(lines[-1])
    w = d["world"]
    assert d["n_gpus"] == 2 and w["world_size"] == 2 and len({r["pid"] for r in w["ranks"]}) == 2
    assert w["backend"] == ("nccl" if n_gpu >= 2 else "gloo") and w["distinct_devices"] == (2 if n_gpu >= 2 else 1)
    assert w["allreduce_floats_per_step"] == d["config"]["trainable_floats"] + 1 and w["allreduce_us_per_step"] > 0
    assert d["config"]["global_batch"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    sgr = d["step_graphs"]
    assert d["step_launch"] == "hipGraph replay" and sgr["captured"] >= 1 and sgr["failed"] == 0
    if collective == "oneshot":
        assert w["transport_note"] is None and w["allreduce_transport"].startswith("oneshot")
        assert sgr["graphs_per_step"] == 1 and sgr["collective_in_graph"] is True and w["replayed_step_split_ms"] is None
    else:
        assert w["allreduce_transport"].startswith("torch.distributed")
        assert sgr["graphs_per_step"] == 2 and sgr["collective_in_graph"] is False
        split = w["replayed_step_split_ms"]
        assert set(split) >= {"graph_a", "allreduce", "graph_b"} and split["graph_a"] > split["graph_b"] > 0


@pytest.mark.parametrize("collective", ["rccl", "oneshot"])
def test_forced_single_rank_rccl_step_is_bit_equal_to_the_plain_step(dev, tmp_path, collective):
    """VERDICT r5, missing 1: no RCCL collective had ever executed in this code base -- world 1 skipped them and the multi-rank
    GPU tests fall back to the host-synchronous gloo on a 1-GPU box.  tests/dp_force_worker.py creates a ONE-rank process group
    on backend nccl (= RCCL) and dist.DataParallel(force=True) issues every collective of the N-rank step: at C2 B = 32 the
    steps [eager, capture + replay, replay] run as graph A -> eager RCCL all-reduce (asynchronous, on torch's NCCL stream) ->
    graph B -- the default multi-GPU transport -- and must equal the dp=None run bit for bit (loss, ADE / FDE, the last step's
    gradients, the weights after three Adam updates); under YNET_ALLREDUCE=oneshot the kernel is recorded inside ONE graph."""
    import json
    out = str(tmp_path / "forced.json")
    rc, tail = launch([sys.executable, os.path.join(ROOT, "tests", "dp_force_worker.py"), out, "32"],
                      env={"HSA_ENABLE_IPC_MODE_LEGACY": "0", "YNET_ALLREDUCE": collective, "MASTER_ADDR": "127.0.0.1",
                           "MASTER_PORT": str(_free_port())}, timeout=900)
    assert rc == 0, tail[-3000:]
    with open(out) as f:
        v = json.load(f)
    assert v["backend"] == "nccl" and v["world_size"] == 1
    assert v["collective"] == collective and v["transport_note"] is None and v["seed_ok"]
    assert v["launched"] == [["eager", "replay", "replay"]] * 2 and v["failed"] == [0, 0]
    if collective == "rccl":
        assert v["graphs_per_step"] == [1, 2] and v["collective_in_graph"] is False
        s = v["split_ms"]
        assert s["graph_a"] > s["graph_b"] > 0 and s["allreduce"] > 0
    else:
        assert v["graphs_per_step"] == [1, 1] and v["collective_in_graph"] is True
    assert v["results_equal"], v["results"]
    assert v["n_tensors"] == 18 and v["weights_moved"]
    assert v["grads_differ"] == [] and v["weights_differ"] == []
