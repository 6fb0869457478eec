#!/bin/bash
# round 5, session I: the whole GPU suite, then soaks on mdec-k3.7 / adpcm-k5.0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/r05i_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05i_pytest.log
tail -14 $O/r05i_pytest.log
timeout 1500 python tools/gpu_soak_mixed.py 60 555 900 > $O/r05i_soak_mixed.log 2>&1; tail -3 $O/r05i_soak_mixed.log
timeout 900 python tools/gpu_soak.py 120 8675 1400 600 > $O/r05i_soak_single_launch.log 2>&1; tail -2 $O/r05i_soak_single_launch.log
timeout 600 python tools/gpu_soak_lanes.py 40 4242 700 > $O/r05i_soak_lanes.log 2>&1; tail -2 $O/r05i_soak_lanes.log
timeout 600 python tools/gpu_adpcm_soak.py 300 > $O/r05i_adpcm_soak.log 2>&1; tail -2 $O/r05i_adpcm_soak.log
