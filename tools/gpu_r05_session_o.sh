#!/bin/bash
# round 5, session O22: the pilot leaning down when the scale below the bracket's top is over the limit by less than room >> n (n = 5, 4, 3)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python tools/gpu_ab_rates.py build_ab/libpsxav_hip_lean0.so build_ab/libpsxav_hip_lean5.so build_ab/libpsxav_hip_lean4.so build_ab/libpsxav_hip_lean3.so mixed a8 v3a4 --rounds 2 2>&1
for v in 0 4 3; do
PSXAV_HIP_LIB=$PWD/build_ab/libpsxav_hip_lean$v.so timeout 600 python tools/gpu_r05_diag.py mixed --json $O/r05o22_diag$v.json > $O/r05o22_diag$v.log 2>&1
python - <<PY
import json
d=json.load(open("$O/r05o22_diag$v.json"))
for k in ("warm_launch","cold_launch"):
    w=d["mixed"][k]; print("lean$v", k, w['passes_per_frame'], w['passes_hist_0_1_2_3_4_5plus'], w["first_guess_minus_answer_hist_-4..+4"])
PY
done
