#!/bin/bash
# round 5, session S: soaks on the round's final library
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python tools/gpu_soak_mixed.py 70 2025092921 900 > $O/r05s_soak_mixed.log 2>&1; tail -1 $O/r05s_soak_mixed.log
timeout 1200 python tools/gpu_soak_lanes.py 40 92921 > $O/r05s_soak_lanes.log 2>&1; tail -1 $O/r05s_soak_lanes.log
timeout 1500 python tools/gpu_soak.py 40 92922 1200 300 > $O/r05s_soak_single.log 2>&1; tail -1 $O/r05s_soak_single.log
