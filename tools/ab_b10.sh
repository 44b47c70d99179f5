#!/bin/bash
# A/B of dispatch thresholds on the captured C2 step at batch 10 (and 32):  gpurun -- 'bash tools/ab_b10.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
run() {  # label, batch, env...
  local label=$1 b=$2; shift 2
  env "$@" python bench.py --batch $b --steps 40 --no-cpu-baseline --no-c5 --no-legs --no-sustained --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); print('$label', 'B$b', [round(x,4) for x in d['timed_regions']['ms_per_step']])"
}
for b in 10 32; do
run base $b YNET_X=0
run r2min64 $b YNET_CONV_R2_MIN=64
run r2min32 $b YNET_CONV_R2_MIN=32
run r2min64_dmar1 $b YNET_CONV_R2_MIN=64 YNET_CONV_DMA_R1=1
run r2min64_ks128 $b YNET_CONV_R2_MIN=64 YNET_KSPLIT_ITEMS=128
run r2min64_ks512 $b YNET_CONV_R2_MIN=64 YNET_KSPLIT_ITEMS=512
run r2min64_kt1024 $b YNET_CONV_R2_MIN=64 YNET_KSPLIT_TARGET=1024
run r2min64_noks $b YNET_CONV_R2_MIN=64 YNET_CONV_NO_KSPLIT=1
run r2min64_narrow4096 $b YNET_CONV_R2_MIN=64 YNET_CONV_NARROW=4096
run r2min64_narrow1024 $b YNET_CONV_R2_MIN=64 YNET_CONV_NARROW=1024
run r2min64_r4min512 $b YNET_CONV_R2_MIN=64 YNET_CONV_R4_MIN=512
run r2min64_r4min2048 $b YNET_CONV_R2_MIN=64 YNET_CONV_R4_MIN=2048
done
