#!/bin/bash
# round 5, session O17: the checkpoint's margin (thousandths of a standard error) again, now that settled scales are not overruled
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for m in 800 1000 1300 1700; do
PSXHIP_MDEC_CKMARGIN=$m timeout 600 python tools/gpu_ab_rates.py psxavenc_amd/libpsxav_hip.so a8 mixed v3a4 v3a8_32k --rounds 1 2>&1 | sed "s/^/margin=$m /"
done
