#!/bin/bash
# round 5, session H: pilot sample lattice + robust two-point model + far-verdict margin: bytes, then flip-flop / uniform / mixed content
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_mdec.py -q -x > $O/r05h_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05h_pytest.log
tail -4 $O/r05h_pytest.log
timeout 900 python tools/gpu_r05_diag.py v3a4 a8 a4 mixed v3a8_32k v2_16k --json $O/r05h_diag.json > $O/r05h_diag.log 2>&1
PSXHIP_MDEC_TRUST=1 timeout 600 python tools/gpu_r05_diag.py v3a4 mixed --json $O/r05h_diag_trust1.json > $O/r05h_diag_trust1.log 2>&1
python - <<PY
import json
for name in ("r05h_diag", "r05h_diag_trust1"):
    d=json.load(open("$O/%s.json"%name))
    for k,v in d.items():
        if k=='library': continue
        print('==',name,k, {kk:vv['frames_per_sec'] for kk,vv in v['rates'].items() if kk!='quant_scale_hist_4000_frames'})
        for w in ('warm_launch','cold_launch'):
            x=v[w]; print('   ',w,'frames',x['frames'],'passes/start',x['passes_per_frame'],'hist',x['passes_hist_0_1_2_3_4_5plus'],'right',x['first_guess_right'],'off1',x['first_guess_off_by_one'],'offmore',x['first_guess_off_by_more'],'ckpt',x['stopped_at_checkpoint'],'pilot%',x['phase_share_pct_ticket_resetdc_pilot_passes_scanmerge_writeout'][2],'ends',x['group_end_us_min_p10_p50_p90_max'])
            print('        >=3:', x.get('cases_with_3_or_more_passes')[:10])
PY
