#!/bin/bash
# The GPU parity suites under the switches that change routes or kernels (results must not change): bash tools/switch_matrix.sh [farm]
# (farm: only the legs of the 64-lane snapshot path and its walks -- farm64.hip.h)
FARM="GNX_CLONG=2@GNX_W64=2 GNX_CLONG=2@GNX_W64=2@GNX_W64_FARM=0 GNX_CLONG=2@GNX_W64=2@GNX_W64_FARM_PIPE=0 GNX_CLONG=2@GNX_W64=2@GNX_W64_FARM=3 GNX_MEGA_STRIPS=3@GNX_W64=2 GNX_MEGA_STRIPS=3@GNX_W64=2@GNX_W64_FARM=0 GNX_CLONG=2@GNX_W64=2@GNX_W64_R=6@GNX_W64_RC=4 GNX_CLONG=2@GNX_W64=2@GNX_W64_R=16@GNX_W64_RC=4 GNX_MEGA_STRIPS=3@GNX_W64=2@GNX_W64_R=8@GNX_W64_RC=4"
if [ "$1" = "farm" ]; then
  for swa in $FARM; do
    sw=${swa//@/ }
    r=$(env $sw timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_const_long.py tests/test_host_entry.py tests/test_n1_gpu.py tests/test_n2_gsw.py -m gpu -x -q -k "not ten_million" 2>&1 | grep -E "passed|failed" | tail -1)
    echo "$sw: $r"
  done
  exit 0
fi
for sw in "GNX_FASTPATH=0" "GNX_FASTPATH=2" "GNX_NO_PIPE=1" "GNX_CLONG=2" "GNX_CLONG=0" "GNX_REF_UNPACK=1" "GNX_CL_WALK_NP=4" "GNX_CL_CKC=448" "GNX_CL_PUB=64" "GNX_FP_MAXIT=0" "GNX_FP_SPEC=1" "GNX_WALK_LANE=1" "GNX_TICKET_DELAY=3" "GNX_NO_HFORM=1" "GNX_SCORED_SUB=3" "GNX_CL_WG=0" "GNX_CL_WALK_SPEC=0" "GNX_CL_WALK_SPEC=4" "GNX_FP_SMALL=0" "GNX_WALK_WIDE=0" "GNX_LAT=0" "GNX_LAT=2" "GNX_WIDE=2" "GNX_REBASE=1" "GNX_SCORE_GENERIC=1" "GNX_MEGA_STRIPS=3" "GNX_CLONG=2 GNX_W64=2" "GNX_W64=0" "GNX_MEGA_STRIPS=3 GNX_W64=2" "GNX_CLONG=2 GNX_W64=2 GNX_W64_FARM=0" "GNX_CLONG=2 GNX_W64=2 GNX_W64_FARM_PIPE=0" "GNX_MEGA_STRIPS=3 GNX_W64=2 GNX_W64_FARM=0"; do
  r=$(env $sw timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_const_long.py tests/test_host_entry.py tests/test_n1_gpu.py tests/test_n2_gsw.py -m gpu -x -q -k "not ten_million" 2>&1 | grep -E "passed|failed" | tail -1)
  echo "$sw: $r"
done
