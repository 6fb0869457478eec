#!/bin/bash
# config 5 (xacd, full length) against chunk length / warm-up of the speculate-and-verify encode
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
for cw in "4096 128" "3797 128" "3797 64" "2048 128" "1899 128" "1899 64" "1899 32" "1266 64" "950 64" "950 32"; do
  set -- $cw
  PSXHIP_ADPCM_CHUNK=$1 PSXHIP_ADPCM_WARM=$2 python bench.py --config xacd --steps 6 --warmup 2 --no-secondary --no-cpu-baseline > $O/xsw_$1_$2.json 2> $O/xsw_$1_$2.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/xsw_$1_$2.json").read().splitlines() if l.startswith("{")][-1])
    print("chunk %5s warm %4s  %12.0f sectors/s  ms/step %.3f  passes %s  parity %s" % ("$1", "$2", d["value"], d["ms_per_step"], d["config"].get("verify_passes_last_step"), d["parity"].get("bit_exact")))
except Exception as e:
    print("$1 $2 ERR", e, open("$O/xsw_$1_$2.err").read()[-400:])
PY
done
