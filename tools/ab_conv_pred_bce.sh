#!/bin/bash
# A/B of the last decoder convolution inside the predictor + criterion launch (YNET_CONV_PRED_BCE): gpurun --timeout 1200 -- 'bash tools/ab_conv_pred_bce.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
for v in 0 1; do
  for c in ${AB_CONFIGS:-C2}; do
    YNET_CONV_PRED_BCE=$v python3 bench.py --config $c --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-c5 --no-legs --sustained-seconds 3 2>/dev/null | python3 -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"metric\"'):
        d=json.loads(ln); print('conv_pred_bce=$v $c', round(d['value'],1), d['timed_regions']['ms_per_step'], round((d.get('sustained') or {}).get('ms_per_step',0),4), d.get('parity_check',{}).get('ok'))"
  done
done
done
