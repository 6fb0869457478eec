#!/bin/bash
# round 5, session O18: the checkpoint's standard error from the spread of the rounds' sums (clusters), margins 0.8 / 1.0 / 1.3
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_mdec.py -q -x > $O/r05o_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05o_pytest.log; tail -2 $O/r05o_pytest.log
timeout 1500 python tools/gpu_ab_rates.py build_ab/libpsxav_hip_prev.so psxavenc_amd/libpsxav_hip.so a8 mixed a4 v3a4 v3a8_32k --rounds 2 2>&1 | sed "s/^/default /"
for m in 700 1300; do
PSXHIP_MDEC_CKMARGIN=$m timeout 600 python tools/gpu_ab_rates.py psxavenc_amd/libpsxav_hip.so a8 mixed v3a4 --rounds 1 2>&1 | sed "s/^/margin=$m /"
done
timeout 600 python tools/gpu_r05_diag.py mixed a8 --json $O/r05o18_diag.json > $O/r05o18_diag.log 2>&1
python - <<PY
import json
d=json.load(open("$O/r05o18_diag.json"))
for kind in ("mixed","a8"):
    w=d[kind]['warm_launch']
    print(kind, w['passes_per_frame'], w['passes_hist_0_1_2_3_4_5plus'], 'ck', w['stopped_at_checkpoint'], w['pass_traces_of_frames_with_3_or_more_passes'])
PY
