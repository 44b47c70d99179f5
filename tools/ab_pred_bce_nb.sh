#!/bin/bash
# pred_bce_kernel: input planes fetched ahead of their products (NB) against the launch time -- gpurun --timeout 1500 -- 'bash tools/ab_pred_bce_nb.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== production build (NB 8)"; python3 tools/glue_bench.py 2>/dev/null | grep "pred_bce cout=12 "
for nb in 4 16 32; do
  rm -rf /tmp/csrc_diag; cp -r $R/motion-style-transfer_amd/csrc /tmp/csrc_diag
  (cd /tmp/csrc_diag && rm -f glue.o libynet_hip.so && sed -i 's#../../include/ynet_hip.h#'$R'/include/ynet_hip.h#' conv_auto.cpp Makefile && make EXTRA="-DYNET_PRED_BCE_NB=$nb -Rpass-analysis=kernel-resource-usage" -j8 > /tmp/diag_build_$nb.log 2>&1) || { tail -5 /tmp/diag_build_$nb.log; continue; }
  grep -A8 "pred_bce_kernelILi12ELi4ELb0" /tmp/diag_build_$nb.log | grep -E "VGPRs:|Spill|Occupancy" | head -4
  echo "== NB=$nb"; YNET_HIP_LIB=/tmp/csrc_diag/libynet_hip.so python3 tools/glue_bench.py 2>/dev/null | grep "pred_bce cout=12 "
done
