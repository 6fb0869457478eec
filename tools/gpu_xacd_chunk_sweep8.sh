#!/bin/bash
# an eighth of config 5 (one GPU's share at 8 GPUs: 450 s x 8 channels, 9.7 M units) against chunk length / warm-up
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
for cw in "1024 32" "768 32" "512 32" "512 64" "384 32" "256 32" "238 32" "238 64" "128 16"; do
  set -- $cw
  PSXHIP_ADPCM_CHUNK=$1 PSXHIP_ADPCM_WARM=$2 python bench.py --config xacd --audio-seconds 450 --steps 10 --warmup 2 --no-secondary --no-cpu-baseline > $O/xs8_$1_$2.json 2> $O/xs8_$1_$2.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/xs8_$1_$2.json").read().splitlines() if l.startswith("{")][-1])
    print("chunk %5s warm %4s  %12.0f sectors/s  ms/step %.3f  passes %s  parity %s" % ("$1", "$2", d["value"], d["ms_per_step"], d["config"].get("verify_passes_last_step"), d["parity"].get("bit_exact")))
except Exception as e:
    print("$1 $2 ERR", e, open("$O/xs8_$1_$2.err").read()[-400:])
PY
done
