#!/bin/bash
# Kernel trace of a bench.py configuration (run on the GPU box through gpurun):
#   gpurun -- 'bash tools/profile_step.sh <tag> [bench.py arguments]'
# -> gpurun_out/prof_<tag>/<tag>_kernel_stats.csv, <tag>_timeline.json (copy what should be judged into profiles/).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/trace_$TAG -o t -- python3 "$R/bench.py" --no-cpu-baseline --no-roofline "$@" > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
tail -1 "$OUT/trace.log" | cut -c1-400
python3 "$R/tools/trace_summary.py" /tmp/trace_$TAG "$OUT/$TAG" --tail-frac 0.6 --dump ${DUMP:-700} --rows ${ROWS:-700}
