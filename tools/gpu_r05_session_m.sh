#!/bin/bash
# round 5, session M: exit-path trims (no pilot-word atomic when no pilot ran): the headline in order and with two lanes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_mdec.py -q -x > $O/r05m_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05m_pytest.log; tail -2 $O/r05m_pytest.log
for l in 1 2; do timeout 300 python bench.py --lanes $l --steps 5 --launches-per-step 1600 --no-secondary --no-cpu-baseline | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('a4 lanes $l', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_stats'])"; done
timeout 600 python tools/gpu_r05_diag.py a4 a8 mixed v3a4 --json $O/r05m_diag.json > $O/r05m_diag.log 2>&1
python - <<PY
import json
d=json.load(open("$O/r05m_diag.json"))
for k,v in d.items():
    if k=='library': continue
    print('==',k, {kk:vv['frames_per_sec'] for kk,vv in v['rates'].items() if kk!='quant_scale_hist_4000_frames'})
PY
