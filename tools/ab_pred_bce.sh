#!/bin/bash
# What bounds pred_bce_kernel?  Development builds of glue.hip with parts of its arithmetic removed (YNET_PRED_BCE_DIAG bits: 1 no
# forward products, 2 no dgrad products, 4 no exp / log / rcp; results are WRONG in those builds), timed by tools/glue_bench.py on one box.
#   gpurun --timeout 1500 -- 'bash tools/ab_pred_bce.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== production build"; python3 tools/glue_bench.py 2>/dev/null | grep "pred_bce cout=12"
for d in 1 2 4 7; do
  rm -rf /tmp/csrc_diag; cp -r $R/motion-style-transfer_amd/csrc /tmp/csrc_diag; mkdir -p /tmp/include; cp $R/include/ynet_hip.h /tmp/include/ 2>/dev/null
  (cd /tmp/csrc_diag && rm -f glue.o libynet_hip.so && sed -i 's#../../include/ynet_hip.h#'$R'/include/ynet_hip.h#' conv_auto.cpp Makefile && make EXTRA=-DYNET_PRED_BCE_DIAG=$d -j8 > /tmp/diag_build_$d.log 2>&1) || { tail -5 /tmp/diag_build_$d.log; continue; }
  echo "== YNET_PRED_BCE_DIAG=$d"; YNET_HIP_LIB=/tmp/csrc_diag/libynet_hip.so python3 tools/glue_bench.py 2>/dev/null | grep "pred_bce cout=12"
done
