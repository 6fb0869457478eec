#!/bin/bash
# round 5, session O19: a stopped pass's verdict steers the next pass (not the unchanged state's model): bytes, soak, A/B on one box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_mdec.py -q -x > $O/r05o_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05o_pytest.log; tail -2 $O/r05o_pytest.log
timeout 600 python tools/gpu_soak_mixed.py 12 784 900 > $O/r05o_soak_mixed.log 2>&1; tail -1 $O/r05o_soak_mixed.log
timeout 1500 python tools/gpu_ab_rates.py build_ab/libpsxav_hip_prev.so psxavenc_amd/libpsxav_hip.so a8 mixed a4 v3a4 v3a8_32k --rounds 2 2>&1
timeout 600 python tools/gpu_r05_diag.py mixed --json $O/r05o19_diag.json > $O/r05o19_diag.log 2>&1
python - <<PY
import json
d=json.load(open("$O/r05o19_diag.json"))
for k in ("warm_launch","cold_launch"):
    w=d["mixed"][k]
    print(k, w['passes_per_frame'], w['passes_hist_0_1_2_3_4_5plus'], 'ck', w['stopped_at_checkpoint'], w['pass_traces_of_frames_with_3_or_more_passes'][:10])
print({kk:vv['frames_per_sec'] for kk,vv in d["mixed"]['rates'].items() if kk!='quant_scale_hist_4000_frames'})
PY
