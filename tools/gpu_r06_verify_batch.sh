#!/bin/bash
# round 6: what the host round trips of the verify phase cost config 3 at one stream: every batch n passes launched back to back
# (n = 48: the whole phase is one batch, one synchronisation; the passes behind the fixpoint return at once, ~4 us each)
cd "$(dirname "$0")/.."
for n in 0 8 16 32 48 64; do
  if [ "$n" = 0 ]; then env=""; else env="PSXHIP_ADPCM_VERIFY_BATCH=$n"; fi
  for r in 1 2; do
  env $env python bench.py --config strcd --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch %2s  step %.3f ms  %.2f M sectors/s  parity %s' % ('$n', d['ms_per_step'], d['value']/1e6, d['parity']['bit_exact']))"
  done
done
