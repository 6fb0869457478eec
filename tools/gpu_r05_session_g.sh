#!/bin/bash
# round 5, session G: where did 640x480 v3 at 8 KiB go?  (1.83 M on k3.6, 1.19 M on k3.7): the trust policy on / off on flip-flop content
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
for t in 0 1 2; do PSXHIP_MDEC_TRUST=$t timeout 600 python tools/gpu_r05_diag.py v3a4 a8 --json $O/r05g_diag_trust$t.json > $O/r05g_diag_trust$t.log 2>&1; done
python - <<PY
import json
for t in (0,1,2):
    d=json.load(open("$O/r05g_diag_trust%d.json"%t))
    for k,v in d.items():
        if k=='library': continue
        print('== TRUST',t,k, {kk:vv['frames_per_sec'] for kk,vv in v['rates'].items() if kk!='quant_scale_hist_4000_frames'})
        for w in ('warm_launch','cold_launch'):
            x=v[w]; print('   ',w,'frames',x['frames'],'passes/start',x['passes_per_frame'],'hist',x['passes_hist_0_1_2_3_4_5plus'],'right',x['first_guess_right'],'off1',x['first_guess_off_by_one'],'offmore',x['first_guess_off_by_more'],'ckpt',x['stopped_at_checkpoint'],'phases',x['phase_share_pct_ticket_resetdc_pilot_passes_scanmerge_writeout'],'ends',x['group_end_us_min_p10_p50_p90_max'])
            print('        top', x['top_cases_guess_abort_answer_passes_count'][:8])
PY
