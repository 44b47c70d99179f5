#!/usr/bin/env python3
"""Which queue / stream did each part of a split step run on?  (graph A -> eager RCCL all-reduce -> graph B.)

Reads a rocprofv3 --kernel-trace output directory (rocpd SQLite database or kernel_trace CSV), takes the LAST step of the run
(steps are separated at `zero`/fill kernels of the flat gradient buffer: the first kernel of graph A is found by the longest
idle gap pattern, so simply: the last 1/8 of the kernels) and prints
  * the columns the database offers for a kernel dispatch (queue id, stream id where the tool records one),
  * per queue / stream: the number of kernels and the first / last kernel names in the window,
  * every RCCL kernel (names containing nccl / rccl) with its queue and the kernels that end just before / start just after it,
  * the largest gaps between consecutive kernels (where the host-side collective call sits when RCCL launches nothing: an
    in-place all-reduce among ONE rank is a no-op inside RCCL, only torch's event hand-off around it executes).
usage: trace_streams.py <dir>"""
import csv
import glob
import os
import sqlite3
import sys


def load(d):
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if dbs:
        con = sqlite3.connect(dbs[0])
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        if not cols:
            cols = [c[0] for c in con.execute("select * from kernels limit 1").description]
        print("columns of `kernels`:", cols)
        rows = [dict(zip(cols, r)) for r in con.execute("select * from kernels")]
        s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
        for r in rows:
            r["_s"], r["_e"], r["_n"] = int(r[s]), int(r[e]), r["name"]
            r["_q"] = tuple(r.get(k) for k in ("queue_id", "stream_id", "queue", "stream", "tid") if k in r)
        return rows, [k for k in ("queue_id", "stream_id", "queue", "stream", "tid") if k in cols]
    rows = []
    keys = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            keys = [k for k in ("Queue_Id", "Stream_Id", "Thread_Id") if k in rd.fieldnames]
            print("columns of kernel_trace.csv:", rd.fieldnames)
            for r in rd:
                r["_s"], r["_e"], r["_n"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]
                r["_q"] = tuple(r[k] for k in keys)
                rows.append(r)
    return rows, keys


def short(n):
    return n.split("(")[0][:70]


def main():
    rows, keys = load(sys.argv[1])
    rows.sort(key=lambda r: r["_s"])
    if not rows:
        print("no kernels")
        return
    print(f"{len(rows)} kernels; dispatch keys {keys}")
    win = rows[-max(50, len(rows) // 8):]
    per = {}
    for r in win:
        per.setdefault(r["_q"], []).append(r)
    print(f"\nlast {len(win)} kernels, by {keys}:")
    for q, rs in sorted(per.items(), key=lambda kv: -len(kv[1])):
        print(f"  {q}: {len(rs):5d} kernels   first {short(rs[0]['_n'])}   last {short(rs[-1]['_n'])}")
    nccl = [i for i, r in enumerate(rows) if "nccl" in r["_n"].lower() or "rccl" in r["_n"].lower()]
    print(f"\nRCCL kernels in the whole trace: {len(nccl)}")
    for i in nccl[-6:]:
        r = rows[i]
        before = max((x for x in rows[:i] if x["_e"] <= r["_s"]), key=lambda x: x["_e"], default=None)
        after = min((x for x in rows[i + 1:] if x["_s"] >= r["_e"]), key=lambda x: x["_s"], default=None)
        print(f"  {short(r['_n'])} on {r['_q']}: {(r['_e'] - r['_s']) / 1e3:.1f} us")
        if before:
            print(f"      after  {short(before['_n'])} on {before['_q']} (ended {(r['_s'] - before['_e']) / 1e3:.1f} us earlier)")
        if after:
            print(f"      before {short(after['_n'])} on {after['_q']} (starts {(after['_s'] - r['_e']) / 1e3:.1f} us later)")
    gaps = []
    reach = win[0]["_e"]
    prev = win[0]
    for r in win[1:]:
        if r["_s"] > reach:
            gaps.append((r["_s"] - reach, prev, r))
        if r["_e"] > reach:
            reach, prev = r["_e"], r
    print("\nlargest idle gaps in the window (us: kernel that ended -> kernel that started):")
    for g, a, b in sorted(gaps, key=lambda t: -t[0])[:8]:
        print(f"  {g / 1e3:8.1f}   {short(a['_n'])} {a['_q']}  ->  {short(b['_n'])} {b['_q']}")


if __name__ == "__main__":
    main()
