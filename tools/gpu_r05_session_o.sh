#!/bin/bash
# round 5, session O21: clusters of 1 / 2 / 3 for the quarter's sample, and what they cost in fetched bytes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
for c in 0 3 2 1; do
PSXHIP_MDEC_SAMPLE_CLUSTER=$c timeout 600 python tools/gpu_ab_rates.py psxavenc_amd/libpsxav_hip.so a8 mixed a4 v3a4 v3a8_32k --rounds 1 2>&1 | sed "s/^/cluster=$c /"
done
for c in 0 2 1; do
  for shape in "a4 --frames 1000" "v3 --config sbs_v3 --total-frames 1250"; do
    tag=${shape%% *}; args=${shape#* }
    out=$O/r05o21_${tag}_c$c; rm -rf $out; mkdir -p $out
    PSXHIP_MDEC_SAMPLE_CLUSTER=$c rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o r -- python bench.py --lanes 1 --steps 2 --warmup 1 --launches-per-step 16 --no-cpu-baseline --no-secondary $args > $out/fetch.log 2>&1
    python tools/rocpd_summary.py $(find $out -name "*.db" | sort) 2>/dev/null | grep "mdec_encode.*FETCH_SIZE" | sed "s/^/fetch cluster=$c $tag /"; find $out -name "*.db" -delete
  done
done
