#!/bin/bash
# A/B of library builds on ONE box: tools/gpu_ab.sh <tag> <name=path-to-.so or "cur"> ...   (kernel time of in-order launches, three workloads)
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
tag=$1; shift
for i in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; path=${spec#*=}; lib=""
    [ "$path" != cur ] && lib="$PWD/$path"
    for wl in "a4:--amp 4 --launches-per-step 400" "a8:--amp 8 --launches-per-step 200" "v3:--config sbs_v3 --total-frames 1250 --launches-per-step 40"; do
      w=${wl%%:*}; args=${wl#*:}
      PSXAV_HIP_LIB=$lib python bench.py --steps 4 --warmup 2 --lanes 1 --no-secondary --no-cpu-baseline $args > $O/ab_${tag}_${name}_${w}_$i.json 2> $O/ab_${tag}_${name}_${w}_$i.err
      python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/ab_${tag}_${name}_${w}_$i.json").read().splitlines() if l.startswith("{")][-1])
    print("%-8s %-3s run $i  %10.0f frames/s  kernel_ms %.5f  parity %s" % ("$name", "$w", d["value"], d["roofline"]["kernel_ms"], d["parity"]["bit_exact"]))
except Exception as e:
    print("$name $w $i ERR", e, open("$O/ab_${tag}_${name}_${w}_$i.err").read()[-600:])
PY
    done
  done
done
