#!/bin/bash
# round 5, session D: model-steered pilot, far verdicts not taken by piloted frames, single-frame tickets by default; ADPCM seed A/B
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_mdec.py tests/test_gpu_adpcm.py -q -x > $O/r05d_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05d_pytest.log
tail -4 $O/r05d_pytest.log
timeout 600 python tools/gpu_r05_diag.py a4 a8 mixed --json $O/r05d_diag_run1.json > $O/r05d_diag_run1.log 2>&1
PSXHIP_MDEC_RUN=2 timeout 600 python tools/gpu_r05_diag.py a4 mixed --json $O/r05d_diag_run2.json > $O/r05d_diag_run2.log 2>&1
for sd in 0 1; do for k in 0 4 5; do echo "== xacd 600 s kind $k seed $sd"; PSXHIP_ADPCM_SEED=$sd timeout 300 python bench.py --config xacd --audio-seconds 600 --audio-kind $k --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['verify_passes_last_step'], d['parity'])"; done; done
PSXHIP_MDEC_STATS=0 timeout 300 python tools/gpu_mdec_probe.py v3_8k v3_32k v2_16k a16 a2 a24_4k 2>&1 | tail -6
