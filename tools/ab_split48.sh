#!/bin/bash
# A/B of the [16, 32]-channel data gradient in one launch (YNET_WINOGRAD_SPLIT48): gpurun --timeout 1200 -- 'bash tools/ab_split48.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
for v in 0 1; do
  for c in ${AB_CONFIGS:-C2}; do
    YNET_WINOGRAD_SPLIT48=$v python3 bench.py --config $c --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-c5 --no-legs --sustained-seconds 3 2>/dev/null | python3 -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"metric\"'):
        d=json.loads(ln); print('split48=$v $c', round(d['value'],1), d['timed_regions']['ms_per_step'], round((d.get('sustained') or {}).get('ms_per_step',0),4))"
  done
done
done
