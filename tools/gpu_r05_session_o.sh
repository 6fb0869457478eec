#!/bin/bash
# round 5, session O24: warm-up 64 instead of 32 units for chunks of 1024 (a rank's share of config 5), all three materials, 8 / 4 / 2 ranks
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for w in 8 4 2; do
for ch in "0" "1024,64"; do
  if [ "$ch" = "0" ]; then a=""; else a="--chunk $ch"; fi
  timeout 900 python tools/gpu_r05_predict_8gpu.py --world $w $a 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    i=l.find('{')
    if i<0: continue
    d=json.loads(l[i:]); n=d['n_gpus']
    print('world=$w chunk=$ch', l[:i].strip()[:12], 'step', d['n_gpu_step_ms_predicted'], 'ms ->', d['n_gpu_sectors_per_sec_predicted'], 'rounds', [(r['slowest_ms'], r['verify_passes_max']) for r in n['per_round']])
"
done
done
