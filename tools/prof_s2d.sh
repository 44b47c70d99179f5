#!/bin/bash
# serial kernel table of a C2 step with the low-resolution up-convolution backward (YNET_UPCONV_S2D=1, default) and without
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_s2d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export YNET_STEP_GRAPH=0 YNET_SERIAL_DECODERS=1
for v in 0 1; do
  rm -rf /tmp/tr_s2d_$v
  YNET_UPCONV_S2D=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_s2d_$v -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-c5 --no-legs --no-sustained --no-repeats > $OUT/trace_$v.log 2>&1
  python3 $R/tools/trace_summary.py /tmp/tr_s2d_$v $OUT/s2d_$v --tail-frac 0.6 > /dev/null
  echo "== YNET_UPCONV_S2D=$v"; python3 - $OUT/s2d_${v}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))[1:]
tot = sum(float(r[2]) for r in rows)
steps = 28
print("total kernel ms per step", round(tot / 1e6 / steps, 3))
for r in rows:
    if any(k in r[0] for k in ("upconv_ring", "upsample2x_bwd", "conv_wino_kernel<1, 4", "conv_wino_kernel<2, 2, 0", "conv_wino16_kernel<1>", "conv_wino16_kernel<0>")):
        print(f"  {r[0][:60]:60s} calls/step {int(r[1]) / steps:5.1f}  avg us {float(r[3]) / 1e3:8.1f}  ms/step {float(r[2]) / 1e6 / steps:6.3f}")
PY
done
