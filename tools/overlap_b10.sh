#!/bin/bash
# Who runs beside whom in the CAPTURED step at the reference scripts' batch 10 (evidence of how far the two decoders' small-map
# launches already overlap): a rocprofv3 kernel trace of `bench.py --batch 10` (graph replay), its last 8 steps through tools/overlap.py.
#   gpurun --timeout 900 -- 'bash tools/overlap_b10.sh r06'
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_b10_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_b10g
for b in 10 32; do
  rm -rf /tmp/tr_g$b
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_g$b -o t -- python3 $R/bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-c5 --no-legs --no-sustained --no-repeats > "$OUT/trace_g$b.log" 2>&1
  echo "trace b$b rc=$?"
  # kernels per step: count from the serial table is not needed -- take the last 8 steps' worth by the number of launches of a replayed step
  N=$(python3 - "$OUT/trace_g$b.log" <<'PY'
import sys
print(0)
PY
)
  python3 "$R/tools/trace_summary.py" /tmp/tr_g$b "$OUT/${TAG}_b${b}_graph" --tail-frac 0.3 --rows 4000 > "$OUT/timeline_g$b.txt"
  python3 - "$OUT/${TAG}_b${b}_graph_rows.json" "$OUT/${TAG}_bench_C2_b${b}_overlap.txt" "$R" <<'PY'
import json, subprocess, sys
rows = json.load(open(sys.argv[1]))
# one replayed step starts with the coordinates' copy kernel / the zero fill of the flat gradients: find the period by the most frequent first kernel name spacing
names = [r[0] for r in rows]
first = None
for cand in ("zero", "fill", "Fill", "copyBuffer"):
    idx = [i for i, n in enumerate(names) if cand in n]
    if len(idx) >= 4:
        first = idx
        break
adam = [i for i, n in enumerate(names) if "adam_update_kernel" in n]
steps = len(adam) - 1
rows = rows[adam[0] + 1:adam[-1] + 1]          # whole steps: from behind one optimizer launch to the last one
json.dump(rows, open(sys.argv[1] + ".steps.json", "w"))
out = subprocess.run([sys.executable, sys.argv[3] + "/tools/overlap.py", sys.argv[1] + ".steps.json", str(steps)], capture_output=True, text=True).stdout
open(sys.argv[2], "w").write(out)
print(out)
PY
done
