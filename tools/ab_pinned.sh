#!/bin/bash
# A/B of the pinned coordinate upload (utils/step_graph.py::_upload_coords): gpurun --timeout 900 -- 'bash tools/ab_pinned.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
for v in 0 1; do
  for b in ${AB_BATCHES:-32 10}; do
    YNET_PINNED_COORDS=$v python3 bench.py --batch $b --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-c5 --no-legs --sustained-seconds 3 2>/dev/null | python3 -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{\"metric\"'):
        d=json.loads(ln); print('pinned=$v batch=$b', round(d['value'],1), d['timed_regions']['ms_per_step'], round(d['sustained']['ms_per_step'],4))"
  done
done
done
