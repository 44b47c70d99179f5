#!/usr/bin/env python3
"""Per-LAUNCH-SHAPE roofline table of one C2 step: joins the launch order bench.py records for one eager serial
step (`bench.py --conv-layers FILE`: kernel instantiation, (B, H, W, cin, cout, K), HIP-event microseconds, direct-form GFLOP,
algorithmic bytes) with rocprofv3's kernel-only durations of the same eager serial run (--kernel-trace, YNET_STEP_GRAPH=0
YNET_SERIAL_DECODERS=1) and, when present, with the FETCH_SIZE / WRITE_SIZE counter passes -- all three list the convolution
launches of a step in the same order, which the kernel names verify row by row.

    python tools/conv_shapes.py LAYERS.json TRACE_DIR OUT.json [--fetch DIR --write DIR]

Per shape: launches per step, rocprof average / min / max microseconds, executed GFLOP (a conv_wino launch executes 16 / 36 of the
direct form's 2 * 9 * Cin * Cout per pixel), executed TFLOP/s and its fraction of the 157.3 TFLOP/s fp32 matrix peak, algorithmic
MB and GB/s (fraction of 8 TB/s), counter traffic 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section)."""
import glob
import json
import os
import re
import sqlite3
import sys

PEAK_TF, PEAK_GBS = 157.3, 8000.0
CONV = ("conv_dma_", "conv_mfma_kernel", "conv_wino")


def short(name):
    name = re.sub(r"^void ", "", name.strip())
    return re.sub(r"\((ConvArgs|WgradArgs|WinoArgs|WinoCatArgs|Wino16Args|WinoUpArgs|Wino16UpArgs)\)$", "", name)


def kernel_rows(d):
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    con = sqlite3.connect(dbs[0])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")] or [c[0] for c in con.execute("select * from kernels limit 1").description]
    s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = sorted(((short(n), int(a), int(b)) for n, a, b in con.execute(f'select name, "{s}", "{e}" from kernels')), key=lambda r: r[1])
    return [(n, b - a) for n, a, b in rows if n.startswith(CONV)]


def counter_rows(d, counter):
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if not dbs:
        return None
    con = sqlite3.connect(dbs[0])
    cols = [c[0] for c in con.execute("select * from pmc_events limit 1").description]
    order = "dispatch_id" if "dispatch_id" in cols else ("start" if "start" in cols else cols[0])
    rows = [(short(n), float(v)) for n, v in con.execute(f'select name, counter_value from pmc_events where counter_name = ? order by "{order}"', (counter,))]
    return [(n, v) for n, v in rows if n.startswith(CONV)]


def align(seq_names, layer_names, what):
    """seq_names: the convolution launches of the whole run in order; layer_names: one step's.  Every eager step launches the same
    sequence, so the run is a whole number of repetitions -- checked name by name."""
    P = len(layer_names)
    if len(seq_names) % P != 0:
        raise SystemExit(f"{what}: {len(seq_names)} convolution launches are not a multiple of the step's {P}")
    for i, n in enumerate(seq_names):
        if n != layer_names[i % P]:
            raise SystemExit(f"{what}: launch {i} is {n}, the step's launch {i % P} is {layer_names[i % P]}")
    return len(seq_names) // P


def main():
    layers = json.load(open(sys.argv[1]))
    trace, out = sys.argv[2], sys.argv[3]
    names = [l[0] for l in layers]
    P = len(layers)
    rows = kernel_rows(trace)
    steps = align([n for n, _ in rows], names, "kernel trace")
    per_launch = [[rows[s * P + i][1] / 1e3 for s in range(steps)] for i in range(P)]
    pmc = {}
    for key, flag in (("fetch", "--fetch"), ("write", "--write")):
        if flag in sys.argv:
            cr = counter_rows(sys.argv[sys.argv.index(flag) + 1], "FETCH_SIZE" if key == "fetch" else "WRITE_SIZE")
            if cr:
                n = align([c[0] for c in cr], names, key + " counters")
                pmc[key] = [sum(cr[s * P + i][1] for s in range(n)) / n * 1024.0 for i in range(P)]      # the counters are in KB
    shapes = {}
    for i, l in enumerate(layers):
        name, shp, ev_us, _tf, gf, by = l[:6]
        e = shapes.setdefault((name, tuple(shp)), dict(kernel=name, shape=dict(zip(("B", "H", "W", "cin", "cout", "K", "input_masked"), shp)),
                                                       launches_per_step=0, us=[], event_us=[], direct_gflop=gf, bytes=by, fetch=[], write=[]))
        e["launches_per_step"] += 1
        e["us"] += per_launch[i]
        e["event_us"].append(ev_us)
        if "fetch" in pmc:
            e["fetch"].append(pmc["fetch"][i])
        if "write" in pmc:
            e["write"].append(pmc["write"][i])
    res = []
    for e in shapes.values():
        us = sum(e["us"]) / len(e["us"])
        ex = e["direct_gflop"] * (16.0 / 36.0 if e["kernel"].startswith("conv_wino") else 1.0)
        tf = ex / us * 1e3          # GFLOP per microsecond = 1e15 FLOP/s
        gbs = e["bytes"] / us / 1e3
        r = dict(kernel=e["kernel"], shape=e["shape"], launches_per_step=e["launches_per_step"], rocprof_avg_us=round(us, 2),
                 rocprof_min_us=round(min(e["us"]), 2), rocprof_max_us=round(max(e["us"]), 2), ms_per_step=round(us * e["launches_per_step"] / 1e3, 4),
                 hip_event_us=round(sum(e["event_us"]) / len(e["event_us"]), 2), direct_gflop_per_launch=e["direct_gflop"],
                 executed_gflop_per_launch=round(ex, 5), executed_tflops=round(tf, 2), frac_of_fp32_mfma_peak=round(tf / PEAK_TF, 4),
                 direct_equiv_tflops=round(e["direct_gflop"] / us * 1e3, 2), algorithmic_mb_per_launch=round(e["bytes"] / 1e6, 3),
                 algorithmic_gbs=round(gbs, 1), frac_of_hbm_peak=round(gbs / PEAK_GBS, 4), pmc_hbm_bytes_per_launch=None)
        if e["fetch"] and e["write"]:
            f, w = sum(e["fetch"]) / len(e["fetch"]), sum(e["write"]) / len(e["write"])
            # 16-byte-per-lane streaming reads are tallied at half their bytes on gfx950 (MI355X_MICROARCH.md, HBM section): the LDS-DMA kernels
            # (conv_dma* / conv_wino*) fetch that way; the register-staged conv_mfma_kernel does not, its FETCH_SIZE counts as it is
            x2 = e["kernel"].startswith(("conv_dma", "conv_wino"))
            fb = (2 * f if x2 else f) + w
            r.update(pmc_fetch_bytes_raw=round(f), pmc_write_bytes=round(w), fetch_x2_applied=x2, pmc_hbm_bytes_per_launch=round(fb),
                     pmc_over_algorithmic=round(fb / e["bytes"], 3))
        res.append(r)
    res.sort(key=lambda r: -r["ms_per_step"])
    doc = {"what": "every convolution launch shape of one C2 step (B 32, 256^2): rocprofv3 kernel-only durations of the eager serial run joined "
                   "with bench.py's launch order; frac_of_fp32_mfma_peak = executed_gflop_per_launch / rocprof_avg_us / 157.3 TFLOP/s",
           "steps_in_trace": steps, "launches_per_step": P, "conv_ms_per_step": round(sum(r["ms_per_step"] for r in res), 4),
           "peaks": {"fp32_mfma_tflops": PEAK_TF, "hbm_gbs": PEAK_GBS}, "shapes": res}
    json.dump(doc, open(out, "w"), indent=1)
    for r in res[:12]:
        print(f"{r['kernel']:48s} {str(list(r['shape'].values())[:5]):28s} x{r['launches_per_step']:2d} {r['rocprof_avg_us']:8.1f} us  exec {r['executed_tflops']:6.1f} TF = "
              f"{r['frac_of_fp32_mfma_peak']:.3f}  hbm {r['frac_of_hbm_peak']:.3f}  pmc/alg {r.get('pmc_over_algorithmic')}")


if __name__ == "__main__":
    main()
