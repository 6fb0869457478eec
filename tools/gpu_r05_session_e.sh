#!/bin/bash
# round 5, session E: the device-resident STR path -- bytes, then config 3 with 1 .. 16 streams per call
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_str_device.py tests/test_gpu_dropin.py -q -x > $O/r05e_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05e_pytest.log
tail -8 $O/r05e_pytest.log
for S in 1 2 4 8 16; do timeout 300 python bench.py --config strcd --str-streams $S --steps 40 --warmup 5 --cpu-seconds 3 > $O/r05e_strcd_S$S.json 2> $O/r05e_strcd_S$S.err; tail -2 $O/r05e_strcd_S$S.err; python -c "
import json,sys
d=json.loads(open('$O/r05e_strcd_S$S.json').read().strip().splitlines()[-1])
print('S=$S', d['value'], 'ms', d['ms_per_step'], d['parity'], json.dumps(d['config']['legs'])[:600], d['cpu_baseline'] and d['cpu_baseline']['value'])"; done
timeout 300 python bench.py --config strcd --str-streams 8 --audio-kind 2 --steps 40 --warmup 5 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('white noise S=8', d['value'], d['ms_per_step'], d['parity'])"
