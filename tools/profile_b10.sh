#!/bin/bash
# Serial (eager, one stream) kernel tables of the C2 step at batch 10 and batch 32 from the same box, for a per-kernel
# comparison of how the step scales down:  gpurun --timeout 900 -- 'bash tools/profile_b10.sh r04'
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_b10_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
rm -rf /tmp/tr_*
export YNET_STEP_GRAPH=0 YNET_SERIAL_DECODERS=1
for b in 10 32; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_b$b -o t -- $B --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-c5 --no-legs --no-sustained --no-repeats > "$OUT/trace_b$b.log" 2>&1
  echo "trace b$b rc=$?"; python3 "$R/tools/trace_summary.py" /tmp/tr_b$b "$OUT/${TAG}_b${b}_serial" --tail-frac 0.6 > "$OUT/timeline_b$b.txt"
done
ls -la "$OUT"
