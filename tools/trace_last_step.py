#!/usr/bin/env python3
"""The kernels of the LAST step of a rocprofv3 --kernel-trace run in start order: start offset, duration, queue, name -- to see what sits on the critical path between the
big launches.  usage: trace_last_step.py <dir> [marker kernel substring that opens a step, default heatmap_analytic] [which step from the end, default 1]"""
import glob
import os
import sqlite3
import sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "heatmap_analytic"
db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table', 'view')")]
kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
cols = [c[1] for c in con.execute(f"pragma table_info({kt[0]})")] or [c[0] for c in con.execute(f"select * from {kt[0]} limit 1").description]
s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = sorted(con.execute(f"select name, {s}, {e}" + (f", {q}" if q else ", 0") + f" from {kt[0]}").fetchall(), key=lambda r: r[1])
starts = [i for i, r in enumerate(rows) if marker in r[0]]
# a step opens with a burst of marker kernels: the bursts' first kernels
bursts = [i for i in starts if not any(j in starts for j in (i - 1, i - 2, i - 3))]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # 1: the last step (with the epoch's closing reductions behind it), 2: the one before it, ...
first = bursts[-back]
step = rows[first:(bursts[-back + 1] + 3 if back > 1 else len(rows))]
t0 = step[0][1]
for name, a, b, qu in step:
    print(f"{(a - t0) / 1e3:9.1f} us  {(b - a) / 1e3:8.1f} us  q{qu}  {name[:110]}")
