#!/bin/bash
# round 5, session K: two pilot macroblocks per wavefront on big frames (cold / distrusting 640x480), the chunking threshold of host streams
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_mdec.py -q -x > $O/r05k_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05k_pytest.log; tail -3 $O/r05k_pytest.log
for t in 0 2; do PSXHIP_MDEC_TRUST=$t timeout 600 python tools/gpu_r05_diag.py v3a4 v3a8_32k mixed --json $O/r05k_diag_trust$t.json > $O/r05k_diag_trust$t.log 2>&1; done
python - <<PY
import json
for t in (0,2):
    d=json.load(open("$O/r05k_diag_trust%d.json"%t))
    for k,v in d.items():
        if k=='library': continue
        print('== TRUST',t,k, {kk:vv['frames_per_sec'] for kk,vv in v['rates'].items() if kk!='quant_scale_hist_4000_frames'})
        for w in ('warm_launch','cold_launch'):
            x=v[w]; print('   ',w,'frames',x['frames'],'hist',x['passes_hist_0_1_2_3_4_5plus'],'right',x['first_guess_right'],'off1',x['first_guess_off_by_one'],'offmore',x['first_guess_off_by_more'],'pilot%',x['phase_share_pct_ticket_resetdc_pilot_passes_scanmerge_writeout'][2])
PY
make -s -C examples percall_bench
for th in 4096 1024 512 256; do echo "== chunk threshold $th"; PSXHIP_ADPCM_CHUNK_THRESHOLD=$th ./examples/percall_bench 200 50 50 1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['psx_audio_spu_encode_by_samples_per_call'])"; done
for l in 1 2; do timeout 300 python bench.py --config sbs_v3 --total-frames 1250 --lanes $l --steps 8 --warmup 2 --no-secondary --no-cpu-baseline | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sbs_v3_1250 lanes $l', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_stats'])"; done
timeout 300 python bench.py --amp 8 --lanes 1 --steps 5 --launches-per-step 400 --no-secondary --no-cpu-baseline | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('a8 lanes 1', d['value'], d['roofline']['kernel_ms_stats'])"
