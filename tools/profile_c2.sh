#!/bin/bash
# Round profile of the C2 bench (run on the GPU box through gpurun): kernel trace + stats, then the two PMC
# passes (FETCH_SIZE / WRITE_SIZE need separate passes; never combined with a trace domain).  Results land in
# gpurun_out/prof_<tag>/; tools/pmc_aggregate.py turns them into the files committed under profiles/.
#   gpurun --timeout 1500 -- 'bash tools/profile_c2.sh r01'
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export YNET_SERIAL_DECODERS=1      # isolated per-kernel durations (bench.py's roofline object times them the same way)
timeout 500 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o c2 -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o c2 -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > "$OUT/fetch.log" 2>&1
echo "fetch rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o c2 -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > "$OUT/write.log" 2>&1
echo "write rc=$?"
find "$OUT" -name "*.csv" | head -20
