#!/usr/bin/env python3
"""Who runs beside whom in a captured step (development aid): from tools/trace_summary.py --rows output ([name, start ns, duration ns] of
the last N kernels of a rocprofv3 kernel trace) the time every kernel class spends alone / beside another class, and the wall time
no MFMA-bound kernel covers.
    python tools/overlap.py gpurun_out/prof_<tag>/<tag>_rows.json [steps]"""
import json
import sys


def cls(n):
    if "conv_wino" in n:
        return "wino"
    if "conv_dma" in n or "conv_mfma" in n:
        return "direct"
    if "wgrad" in n or "small_gemm" in n or "reduce_partials" in n or "lora_" in n:
        return "wgrad"
    return "glue"


def main():
    rows = json.load(open(sys.argv[1]))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ev = []
    for n, a, d in rows:
        ev.append((a, 1, cls(n)))
        ev.append((a + d, -1, cls(n)))
    ev.sort()
    live = {"wino": 0, "direct": 0, "wgrad": 0, "glue": 0}
    t_prev = ev[0][0]
    acc = {}
    for t, s, c in ev:
        if t > t_prev:
            key = "+".join(k for k in ("wino", "direct", "wgrad", "glue") if live[k]) or "idle"
            acc[key] = acc.get(key, 0) + (t - t_prev)
        live[c] += s
        t_prev = t
    wall = ev[-1][0] - ev[0][0]
    print(f"wall {wall / 1e6 / steps:.3f} ms per step ({steps} steps)")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"  {k:28s} {v / 1e6 / steps:7.3f} ms  {100.0 * v / wall:5.1f} %")
    busy = {}
    for n, a, d in rows:
        busy[cls(n)] = busy.get(cls(n), 0) + d
    print("busy (sum of durations) per step:", {k: round(v / 1e6 / steps, 3) for k, v in busy.items()})


if __name__ == "__main__":
    main()
