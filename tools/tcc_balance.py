#!/usr/bin/env python3
"""Channel balance of the L2's memory-side requests from a `rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ` run (one row per TCC channel
instance and dispatch in the rocpd database):  tcc_balance.py <dir with the rocprofv3 output> <out.json> <kernel name substring> ...
-> per kernel (its LAST dispatch) and counter: channels, min / max / mean requests, max / mean (1.0 = perfectly balanced)."""
import glob
import json
import os
import sqlite3
import sys

d, out, keys = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
    con = sqlite3.connect(db)
    rows = {}
    for name, disp, counter, value in con.execute("select name, dispatch_id, counter_name, counter_value from pmc_events"):
        for k in keys:
            if k in name:
                rows.setdefault((k, counter), {}).setdefault(disp, []).append(float(value))
    con.close()
    for (k, counter), by in rows.items():
        vals = by[max(by)]
        if len(vals) >= 2:
            mean = sum(vals) / len(vals)
            res.setdefault(k, {})[counter] = {"channels": len(vals), "min": min(vals), "max": max(vals), "mean": mean,
                                              "max_over_mean": max(vals) / mean if mean else None, "dispatches_seen": len(by)}
json.dump({"note": "rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ: one value per TCC channel instance of the kernel's last dispatch; "
                   "max / mean = channel imbalance (1.0 = balanced)", "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
