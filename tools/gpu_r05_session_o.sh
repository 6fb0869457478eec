#!/bin/bash
# round 5, session O16: exact evaluations overrule the checkpoint's projections: bytes, soak, the committed library against this one on one box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_mdec.py -q -x > $O/r05o_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05o_pytest.log; tail -2 $O/r05o_pytest.log
timeout 600 python tools/gpu_soak_mixed.py 12 783 900 > $O/r05o_soak_mixed.log 2>&1; tail -1 $O/r05o_soak_mixed.log
timeout 300 python tools/gpu_a8_batches.py a8 2>&1 | grep "^batch" | sed 's/.*ends/ends/'
timeout 1500 python tools/gpu_ab_rates.py build_ab/libpsxav_hip_prev.so psxavenc_amd/libpsxav_hip.so a8 mixed a4 v3a4 --rounds 2 --json $O/r05o16_ab.json 2>&1 | tee $O/r05o16_ab.log
