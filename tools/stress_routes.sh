#!/bin/bash
# tools/stress.py under the switches that force the round-5 / round-6 routes (random batches vs oracle).  Usage: bash tools/stress_routes.sh [seconds per leg]
s=${1:-120}
for sw in "GNX_LAT=2" "GNX_WIDE=2" "GNX_MEGA_STRIPS=2" "GNX_REBASE=1 GNX_CLONG=2" "GNX_LAT=0" "GNX_CLONG=2 GNX_W64=2" "GNX_MEGA_STRIPS=2 GNX_W64=2" "GNX_CLONG=2 GNX_W64=2 GNX_W64_FARM=3" "GNX_CLONG=2 GNX_W64=2 GNX_W64_FARM_PIPE=0" "GNX_CLONG=2 GNX_W64=2 GNX_W64_R=6 GNX_W64_RC=4" "GNX_CLONG=2 GNX_W64=2 GNX_W64_R=8 GNX_W64_RC=4" "GNX_CLONG=2 GNX_W64=2 GNX_W64_R=16" "GNX_MEGA_STRIPS=2 GNX_W64=2 GNX_W64_R=6 GNX_W64_RC=4"; do
  r=$(env $sw timeout $((s + 120)) python tools/stress.py $s 91 2>&1 | tail -1)
  echo "$sw: $r"
done
