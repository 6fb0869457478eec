#!/bin/bash
# round 6: config 3 at ONE stream (2 chains x 90 000 units, the GPU nearly empty): chunk length x warm-up length of the XA track's
# speculate-and-verify session.  A longer warm-up costs latency once (all chunks warm up side by side), and saves verify passes.
cd "$(dirname "$0")/.."
for kind in 0 5 2; do
for spec in "0:0" "64:64" "64:128" "64:256" "64:512" "128:128" "128:256" "128:512" "256:256" "256:512" "512:512" "32:256" "32:512"; do
  c=${spec%%:*}; w=${spec#*:}
  if [ "$c" = 0 ]; then env="" ; else env="PSXHIP_ADPCM_CHUNK=$c PSXHIP_ADPCM_WARM=$w"; fi
  env $env python bench.py --config strcd --audio-kind $kind --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kind $kind chunk %4s warm %4s  step %.3f ms  %.2f M sectors/s  parity %s  s8 %s' % ('$c','$w', d['ms_per_step'], d['value']/1e6, d['parity']['bit_exact'], d['config'].get('eight_streams_sectors_per_sec')))"
done; done
