#!/bin/bash
# What does the fused predictor epilogue cost?  Development builds of conv_wino.hip with parts of wino_epilogue_pred removed (YNET_PRED_EPI_DIAG bits: 1 no exp / log / rcp,
# 2 no target lookup; results are WRONG in those builds), timed by tools/conv_pred_bce_probe.py on one box.   gpurun --timeout 1500 -- 'bash tools/ab_conv_pred_bce_epi.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== production build"; python3 tools/conv_pred_bce_probe.py 2>/dev/null
for d in 1 2 3; do
  rm -rf /tmp/csrc_diag; cp -r $R/motion-style-transfer_amd/csrc /tmp/csrc_diag
  (cd /tmp/csrc_diag && rm -f conv_wino.o libynet_hip.so && sed -i 's#../../include/ynet_hip.h#'$R'/include/ynet_hip.h#' conv_auto.cpp Makefile && make EXTRA=-DYNET_PRED_EPI_DIAG=$d -j8 > /tmp/diag_build_$d.log 2>&1) || { tail -5 /tmp/diag_build_$d.log; continue; }
  echo "== YNET_PRED_EPI_DIAG=$d"; YNET_HIP_LIB=/tmp/csrc_diag/libynet_hip.so python3 tools/conv_pred_bce_probe.py 2>/dev/null | grep fused
done
