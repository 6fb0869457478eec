#!/bin/bash
# experiment: pass order with runs of g consecutive raster rounds (PSXHIP_MDEC_PASS_RUN=g): kernel time + HBM-side fetch per launch
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
for g in 1 2 3 5; do
  export PSXHIP_MDEC_PASS_RUN=$g
  for wl in "a4:--amp 4 --launches-per-step 400" "a8:--amp 8 --launches-per-step 200" "v3:--config sbs_v3 --total-frames 1250 --launches-per-step 40"; do
    w=${wl%%:*}; args=${wl#*:}
    python bench.py --steps 4 --warmup 2 --lanes 1 --no-secondary --no-cpu-baseline $args > $O/pr_${g}_$w.json 2>/dev/null
    out=$O/pr_pmc_${g}_$w; rm -rf $out; mkdir -p $out
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out -o r -- python bench.py --steps 2 --warmup 1 --launches-per-step 16 --lanes 1 --no-cpu-baseline --no-secondary $args > $out/log 2>&1
    f=$(python tools/rocpd_summary.py $(find $out -name '*.db') 2>/dev/null | grep "mdec_encode_frames" | grep FETCH_SIZE | awk '{print $NF}')
    python - <<PY
import json
d = json.loads([l for l in open("$O/pr_${g}_$w.json").read().splitlines() if l.startswith("{")][-1])
alg = d["roofline"]["algorithmic_bytes_per_launch"]
print("run %s %-3s kernel_ms %.5f  fetch %.1f MB (x2 = %.1f MB)  parity %s" % ("$g", "$w", d["roofline"]["kernel_ms"], float("$f" or 0) / 1024, 2 * float("$f" or 0) / 1024, d["parity"]["bit_exact"]))
PY
  done
done
