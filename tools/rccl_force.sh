#!/bin/bash
# the RCCL path executed for real on a 1-GPU box (forced single-rank process group, backend nccl).
#   gpurun --timeout 1500 -- 'bash tools/rccl_force.sh r06'
# 1. the -m gpu test (bit-equality with dp=None, both transports)   2. `bench.py --gpus 1` under YNET_DP_FORCE=1 (split-graph step timed)
# 3. a rocprofv3 kernel trace of the same, from which tools/trace_streams.py lists the streams graph A, the RCCL kernel and graph B ran on
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/rccl_$TAG
mkdir -p "$OUT"
cd "$R"
python3 -m pytest tests/test_gpu_dp.py -q -x -k forced_single_rank > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c5 --no-legs --no-sustained"
$B > "$OUT/${TAG}_bench_C2_plain_line.json" 2> "$OUT/plain.err"; echo "plain rc=$?"
YNET_DP_FORCE=1 $B > "$OUT/${TAG}_bench_C2_forced_rccl_line.json" 2> "$OUT/forced.err"; echo "forced rccl rc=$?"
YNET_DP_FORCE=1 YNET_ALLREDUCE=oneshot $B > "$OUT/${TAG}_bench_C2_forced_oneshot_line.json" 2> "$OUT/forced1.err"; echo "forced oneshot rc=$?"
cd /tmp && export TMPDIR=/tmp
export YNET_DP_FORCE=1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_rccl -o t -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-c5 --no-legs --no-sustained --no-repeats > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
python3 "$R/tools/trace_streams.py" /tmp/tr_rccl > "$OUT/${TAG}_rccl_forced_streams.txt" 2>&1; tail -30 "$OUT/${TAG}_rccl_forced_streams.txt"
for f in plain forced_rccl forced_oneshot; do python3 - "$OUT/${TAG}_bench_C2_${f}_line.json" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], d.get("step_graphs"), {k:v for k,v in d["world"].items() if k!="ranks"})
PY
done
