#!/bin/bash
# round 5, session O11: kernel arguments loaded on demand, the frame address from the scalar frame index: bytes, soak, k3.6 against it on one box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_mdec.py -q -x > $O/r05o_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05o_pytest.log; tail -2 $O/r05o_pytest.log
timeout 600 python tools/gpu_soak_mixed.py 12 779 900 > $O/r05o_soak_mixed.log 2>&1; tail -2 $O/r05o_soak_mixed.log
timeout 1500 python tools/gpu_ab_rates.py build_ab/*.so psxavenc_amd/libpsxav_hip.so v3a4 a4 mixed a8 --rounds 2 --json $O/r05o11_ab.json 2>&1 | tee $O/r05o11_ab.log
