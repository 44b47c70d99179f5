#!/bin/bash
# serial kernel table of one configuration under an environment switch: gpurun -- 'bash tools/prof_serial.sh YNET_CONV_PRED_BCE "0 1" [pattern ...]'
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALS=$2; shift 2
OUT=$R/gpurun_out/prof_serial
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export YNET_STEP_GRAPH=0 YNET_SERIAL_DECODERS=1
for v in $VALS; do
  rm -rf /tmp/tr_ser_$v
  export $VAR=$v
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_ser_$v -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-c5 --no-legs --no-sustained --no-repeats > $OUT/trace_$v.log 2>&1
  python3 $R/tools/trace_summary.py /tmp/tr_ser_$v $OUT/${VAR}_$v --tail-frac 0.6 > /dev/null
  echo "== $VAR=$v"; python3 - $OUT/${VAR}_${v}_kernel_stats.csv "$@" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))[1:]
pats = sys.argv[2:]
tot = sum(float(r[2]) for r in rows)
steps = 28
print("total kernel ms per step", round(tot / 1e6 / steps, 3))
for r in rows:
    if any(k in r[0] for k in pats):
        print(f"  {r[0][:60]:60s} calls/step {int(r[1]) / steps:5.1f}  avg us {float(r[3]) / 1e3:8.1f}  ms/step {float(r[2]) / 1e6 / steps:6.3f}")
PY
done
