#!/bin/bash
# gpurun --timeout 600 -- 'bash tools/run_wino44.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/conv_wino44 $R/tools/conv_wino44_proto.hip || exit 1
for d in 0 2 4 6 8 10 12 14; do DIAG=$d /tmp/conv_wino44 32 256 256 32 32 | grep -E "diag|us per"; done
for d in 0 8 14; do DIAG=$d NW4=1 /tmp/conv_wino44 32 256 256 32 32 | grep -E "diag|us per"; done
