#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/profile_c2.sh) into profiles/<tag>_bench_C2_kernel_stats.csv and profiles/pmc_traffic.json.

    python tools/pmc_aggregate.py r01
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE for kernels that stream 16 bytes per lane (the conv / wgrad
LDS-DMA kernels and the vectorised glue), as MI355X_MICROARCH.md (HBM section) prescribes for gfx950; the raw sum
is kept next to it.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name.strip())
    return re.sub(r"\((ConvArgs|WgradArgs|WinoArgs|WinoCatArgs|Wino16Args|WinoUpArgs|Wino16UpArgs)\)$", "", name)


def db_of(d):
    f = glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)
    return f[0] if f else None


def counter_means(d, counter):
    acc = {}
    db = db_of(d)
    if db:      # rocprofv3's default rocpd (SQLite) output: view pmc_events = one row per (dispatch, counter)
        import sqlite3
        con = sqlite3.connect(db)
        for name, value in con.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)):
            s = acc.setdefault(short(name), [0.0, 0])
            s[0] += float(value)
            s[1] += 1
        return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = short(row["Kernel_Name"])
                s = acc.setdefault(k, [0.0, 0])
                s[0] += float(row["Counter_Value"])
                s[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    to = sys.argv[sys.argv.index("--to") + 1] if "--to" in sys.argv else None      # on the box: only the traffic summary, into that directory
    stats = [] if to else glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    dst = os.path.join(ROOT, "profiles", f"{tag}_bench_C2_kernel_stats.csv")
    if stats:
        shutil.copy(stats[0], dst)
        print("wrote", dst)
    elif not to and db_of(os.path.join(src, "trace")):
        import sqlite3
        import statistics
        con = sqlite3.connect(db_of(os.path.join(src, "trace")))
        per = {}
        for name, dur in con.execute("select name, duration from kernels"):
            per.setdefault(name, []).append(int(dur))
        total = sum(sum(v) for v in per.values())
        with open(dst, "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 4), min(v), max(v),
                            round(statistics.pstdev(v), 3)])
        print("wrote", dst, "(from the rocpd database)")
    fetch = counter_means(os.path.join(src, "fetch"), "FETCH_SIZE")
    write = counter_means(os.path.join(src, "write"), "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        fb = fetch.get(k, (0.0, 0))[0] * 1024.0        # the counters are reported in KB
        wb = write.get(k, (0.0, 0))[0] * 1024.0
        out[k] = {"fetch_bytes_raw": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb,
                  "hbm_bytes_per_launch_fetch_x2": 2.0 * fb + wb,
                  "launches_sampled": max(fetch.get(k, (0, 0))[1], write.get(k, (0, 0))[1]),
                  "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB*1024; x2 FETCH = gfx950 correction for 16-B/lane streams"}
    if out:
        dst = os.path.join(to or os.path.join(ROOT, "profiles"), "pmc_traffic.json")
        with open(dst, "w") as f:
            json.dump(out, f, indent=1)
        print("wrote", dst, len(out), "kernels")


if __name__ == "__main__":
    main()
