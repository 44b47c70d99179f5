#!/usr/bin/env python3
"""On-box summary of a rocprofv3 --kernel-trace run (rocpd SQLite database or kernel_trace CSV):
  <out>_kernel_stats.csv   per-kernel calls / total / average / min / max (the --stats table)
  <out>_timeline.json      busy time (sum of kernel durations; > wall when graph branches overlap) and covered time (union of the
                           kernel intervals) against the wall time from the first kernel start to the last kernel end, over
                           the whole run and over its last `--tail-frac` (the timed steps), plus the histogram of the gaps
                           (time with no kernel running)
usage: trace_summary.py <dir with the rocprofv3 output> <out prefix> [--tail-frac 0.5]"""
import csv
import glob
import json
import os
import sqlite3
import statistics
import sys


def rows_from(d):
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if dbs:
        con = sqlite3.connect(dbs[0])
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")] or [c[0] for c in con.execute("select * from kernels limit 1").description]
        s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
        return [(n, int(a), int(b)) for n, a, b in con.execute(f'select name, "{s}", "{e}" from kernels')]
    out = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                out.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return out


def main():
    d, out = sys.argv[1], sys.argv[2]
    frac = float(sys.argv[sys.argv.index("--tail-frac") + 1]) if "--tail-frac" in sys.argv else 0.5
    rows = sorted(rows_from(d), key=lambda r: r[1])
    if not rows:
        print("no kernel rows found under", d)
        return
    per = {}
    for n, a, b in rows:
        per.setdefault(n, []).append(b - a)
    total = sum(sum(v) for v in per.values())
    with open(out + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([n, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 4), min(v), max(v),
                        round(statistics.pstdev(v), 3)])

    def span(rs):
        busy = sum(b - a for _, a, b in rs)
        wall = max(b for _, _, b in rs) - rs[0][1]
        # gaps of the UNION of the kernel intervals (kernels of parallel graph branches overlap: a gap is time with NO kernel running)
        gaps, covered, reach = [], 0, rs[0][1]
        for _, a, b in rs:
            if a > reach:
                gaps.append(a - reach)
                covered += 0
            covered += max(0, b - max(a, reach))
            reach = max(reach, b)
        hist = {"<1us": 0, "1-2us": 0, "2-5us": 0, "5-10us": 0, "10-50us": 0, ">50us": 0}
        for g in gaps:
            k = "<1us" if g < 1e3 else "1-2us" if g < 2e3 else "2-5us" if g < 5e3 else "5-10us" if g < 1e4 else "10-50us" if g < 5e4 else ">50us"
            hist[k] += 1
        return {"kernels": len(rs), "busy_ms": busy / 1e6, "wall_ms": wall / 1e6, "busy_over_wall": busy / wall,
                "covered_ms": covered / 1e6, "covered_over_wall": covered / wall,
                "gap_ms_total": sum(gaps) / 1e6, "gap_us_median": statistics.median(gaps) / 1e3 if gaps else 0, "gap_hist": hist}
    if "--dump" in sys.argv:        # timeline of the last `--dump N` kernels: start offset, duration, concurrency
        k = int(sys.argv[sys.argv.index("--dump") + 1])
        last = rows[-k:]
        t0 = last[0][1]
        with open(out + "_dump.txt", "w") as f:
            for i, (n, a, b) in enumerate(last):
                conc = sum(1 for (_, a2, b2) in last[max(0, i - 8):i + 8] if a2 < b and b2 > a) - 1
                f.write(f"{(a - t0) / 1e3:10.1f} us  +{(b - a) / 1e3:8.1f} us  x{conc}  {n[:90]}\n")
    if "--rows" in sys.argv:        # the last `--rows N` kernels as JSON rows [name, start ns (from the first row), duration ns]: overlap analysis off the box
        k = int(sys.argv[sys.argv.index("--rows") + 1])
        last = rows[-k:]
        t0 = last[0][1]
        json.dump([[n[:80], a - t0, b - a] for n, a, b in last], open(out + "_rows.json", "w"))
    tail = rows[int(len(rows) * (1 - frac)):]
    json.dump({"all": span(rows), f"last_{frac}": span(tail)}, open(out + "_timeline.json", "w"), indent=1)
    print(open(out + "_timeline.json").read())


if __name__ == "__main__":
    main()
