#!/bin/bash
# Kernel trace + stats of another bench configuration (C1, C3, C4, C5): gpurun -- 'bash tools/profile_cfg.sh C5 3 1'
set -u
CFG=${1:-C5}; STEPS=${2:-3}; WARM=${3:-1}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o run -- python3 "$R/bench.py" --config "$CFG" --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline --no-roofline > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
python3 - "$OUT" <<'PY'
import glob, sqlite3, sys, os
out = sys.argv[1]
db = glob.glob(os.path.join(out, "trace", "**", "*results.db"), recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(os.path.join(out, "kernel_stats.txt"), "w") as f:
    for r in rows:
        f.write(f"{r[2]/tot*100:6.2f}% {r[1]:6d} calls avg {r[3]/1e3:9.1f} us  min {r[4]/1e3:8.1f} max {r[5]/1e3:8.1f}  {r[0][:110]}\n")
    f.write(f"total kernel time {tot/1e6:.2f} ms\n")
os.remove(db)
PY
tail -1 "$OUT/trace.log" | cut -c1-300
