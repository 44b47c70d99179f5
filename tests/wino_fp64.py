"""Adjudication of the Winograd launches of evaluate() against an fp64 run of the oracle (development aid -- the
checks themselves are tests/test_gpu_headline.py).  The C5 sweep (B = 128, K goal samples, forced way-points) is computed four ways:
the oracle in fp32 (= the reference's arithmetic), the oracle in fp64 (the exact value of the same function of the same fp32 weights
and inputs), the HIP path with the Winograd launches and the HIP path on the implicit GEMM only.  Printed: how far each of the three
fp32 results is from fp64 per coordinate and per sample ADE [K,B], the worst coordinates with all four values, and -- with `bisect` --
the same distance for every launch family switched off in turn.
    gpurun --timeout 1500 -- 'python tests/wino_fp64.py 20 > gpurun_out/wino_fp64.log 2>&1'
Reference: utils/evaluate.py:248-291, utils/softargmax.py:55-81."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_headline as T       # noqa: E402
from conftest import build_model, pkg      # noqa: E402
from oracle import ynet_oracle as O        # noqa: E402


def oracle64(sd, cfg, scene, traj, in_t, K, wps):
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    return O.eval_batch(sd64, cfg, scene.double(), traj, in_t.double(), n_goal=K, n_traj=1, waypoint_samples=wps.double())


def stats(name, got, ref64, ref32_err=None):
    d = (got.double() - ref64).abs()
    line = f"{name:28s} vs fp64: max {float(d.max()):.3e} px, mean {float(d.mean()):.3e}, beyond 1e-4: {int((d > 1e-4).sum())} of {d.numel()}"
    if ref32_err is not None:
        bound = torch.clamp(3.0 * ref32_err, min=1e-4)
        line += f"; beyond max(1e-4, 3 |oracle32 - fp64|): {int((d > bound).sum())}"
    print(line, flush=True)
    return d


def run_hip(model, ev, dev, cfg, scene, traj, in_t, K, B, wps):
    caught = []
    h = model.softargmax_.register_forward_hook(lambda m, i, o: caught.append(o.detach().cpu()))
    ade, fde, df, _ = ev.evaluate(
        model, T.loader_for(traj), {"scene0": scene[0]}, dev, "sdd", None, in_t.to(dev), list(cfg.waypoints), "test", K, 1,
        cfg.obs_len, B, cfg.resize_factor, cfg.temperature, forced_samples={0: wps})
    h.remove()
    return torch.cat(caught).view(K, B, cfg.pred_len, 2), df


def main(K=20, B=128, bisect=False):
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    dev = torch.device("cuda:0")
    ops = pkg("ops")
    ev = pkg("utils.evaluate")
    cfg = O.sdd_long(train_net="train")
    H = W = 256
    sd = O.make_state_dict(cfg, seed=0)
    scene, traj = O.synthetic_scene(cfg, H, W, 0), O.synthetic_trajectories(cfg, B, H, W, 22)
    in_t = O.dist_template(cfg.template_size)
    gen = torch.Generator().manual_seed(5)
    w32 = O.eval_batch(sd, cfg, scene, traj, in_t, n_goal=K, n_traj=1, generator=gen)
    wps = w32["waypoint_samples"]
    w64 = oracle64(sd, cfg, scene, traj, in_t, K, wps)
    t64 = w64["trajs"]
    fut = traj[:, cfg.obs_len:]
    e32 = stats("oracle fp32", w32["trajs"], t64)
    out = {"K": K, "B": B, "oracle32_vs_fp64_max": float(e32.max())}

    def ade_k(trajs):
        return O.displacement_error(fut.double(), trajs.double(), cfg.resize_factor).mean(dim=2)

    a64 = ade_k(t64)
    print(f"per-sample ADE [K,B]: oracle32 vs fp64 max {float((ade_k(w32['trajs']) - a64).abs().max()):.3e}", flush=True)
    model = build_model(cfg, sd, dev)
    variants = [("hip winograd", dict(_wino_eval=True)), ("hip implicit gemm", dict(_wino_eval=False))]
    res = {}
    for name, sw in variants:
        old = {k: getattr(ops, k) for k in sw}
        for k, v in sw.items():
            setattr(ops, k, v)
        n0 = ops.wino_stats["launches"]
        got, df = run_hip(model, ev, dev, cfg, scene, traj, in_t, K, B, wps)
        for k, v in old.items():
            setattr(ops, k, v)
        d = stats(f"{name} ({ops.wino_stats['launches'] - n0} wino launches)", got, t64, e32)
        da = (ade_k(got) - a64).abs()
        d32 = (got - w32["trajs"]).abs()
        print(f"    vs oracle32: max {float(d32.max()):.3e} px, beyond 1e-4: {int((d32 > 1e-4).sum())}; per-sample ADE vs fp64 max {float(da.max()):.3e}, "
              f"vs oracle32 max {float((ade_k(got) - ade_k(w32['trajs'])).abs().max()):.3e}; best-of-K ADE vs oracle32 max "
              f"{np.abs(df['ade'].to_numpy() - w32['ade'].numpy()).max():.3e}", flush=True)
        res[name] = (got, d)
        out[name] = {"vs_fp64_max": float(d.max()), "vs_fp64_beyond_1e-4": int((d > 1e-4).sum()), "vs_oracle32_max": float(d32.max()),
                     "per_sample_ade_vs_fp64_max": float(da.max())}
    # the worst coordinates of the Winograd run, with all four values and how diffuse the plane is (spread of the oracle's own fp32 error)
    got_w, d_w = res["hip winograd"]
    got_d, _ = res["hip implicit gemm"]
    flat = d_w.flatten()
    top = torch.topk(flat, 12).indices
    print("worst coordinates of the Winograd run: (k, b, t, xy)  fp64 | oracle32 | hip direct | hip winograd", flush=True)
    for i in top.tolist():
        k, b, t, c = np.unravel_index(i, d_w.shape)
        print(f"  ({k:2d},{b:3d},{t:2d},{c})  {float(t64[k, b, t, c]):.6f} | {float(w32['trajs'][k, b, t, c]):.6f} | {float(got_d[k, b, t, c]):.6f} | "
              f"{float(got_w[k, b, t, c]):.6f}", flush=True)
    if bisect:
        # which launch family carries the deviation: each switched off in turn (python-side switches of ops.py)
        for name, sw in [("no cat / shared-term launches", dict(_wino_cat_eval=False)), ("no one-source launches", dict(_wino_plain_eval=False)),
                         ("only >= 128^2 maps", dict(_wino_eval_min_hw=128 * 128)), ("only 256^2 maps", dict(_wino_eval_min_hw=256 * 256))]:
            if not all(hasattr(ops, k) for k in sw):
                print(f"{name}: switch not present in ops.py", flush=True)
                continue
            old = {k: getattr(ops, k) for k in sw}
            for k, v in sw.items():
                setattr(ops, k, v)
            n0 = ops.wino_stats["launches"]
            got, df = run_hip(model, ev, dev, cfg, scene, traj, in_t, K, B, wps)
            for k, v in old.items():
                setattr(ops, k, v)
            d = stats(f"{name} ({ops.wino_stats['launches'] - n0})", got, t64, e32)
            out[name] = {"vs_fp64_max": float(d.max()), "vs_fp64_beyond_1e-4": int((d > 1e-4).sum())}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, bisect="bisect" in sys.argv)
