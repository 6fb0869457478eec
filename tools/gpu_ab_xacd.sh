#!/bin/bash
# A/B of library builds on the ADPCM workloads: tools/gpu_ab_xacd.sh name=path ...   (xacd at full length, strcd, SPU chains)
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
for i in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; path=${spec#*=}; lib=""
    [ "$path" != cur ] && lib="$PWD/$path"
    for wl in "xacd:--config xacd --steps 6" "strcd:--config strcd --steps 60"; do
      w=${wl%%:*}; args=${wl#*:}
      PSXAV_HIP_LIB=$lib python bench.py --warmup 2 --no-secondary --no-cpu-baseline $args > $O/abx_${name}_${w}_$i.json 2> $O/abx_${name}_${w}_$i.err
      python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/abx_${name}_${w}_$i.json").read().splitlines() if l.startswith("{")][-1])
    print("%-8s %-6s run $i  %12.0f %s  ms/step %.4f  parity %s" % ("$name", "$w", d["value"], d["unit"], d["ms_per_step"], d["parity"].get("bit_exact")))
except Exception as e:
    print("$name $w $i ERR", e, open("$O/abx_${name}_${w}_$i.err").read()[-600:])
PY
    done
  done
done
