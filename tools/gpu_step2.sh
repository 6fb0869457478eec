#!/bin/bash
# like gpu_step.sh, with the HBM-side fetch counter (ONE counter per pass: FETCH_SIZE + WRITE_SIZE together exceed the hardware and rocprofv3 hangs in its abort handler; every pass under its own timeout): tools/gpu_step2.sh <tag> name=path ...
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
bash tools/gpu_ab.sh $tag "$@" 2>&1 | grep -v "run 2"
for spec in "$@"; do
  name=${spec%%=*}; path=${spec#*=}; lib=""
  [ "$path" != cur ] && lib="$PWD/$path"
  for wl in "a4:--amp 4" "a8:--amp 8" "v3:--config sbs_v3 --total-frames 1250"; do
    w=${wl%%:*}; args=${wl#*:}
    out=gpurun_out/f2_${tag}_${name}_$w; rm -rf $out; mkdir -p $out
    PSXAV_HIP_LIB=$lib timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out -o r -- python bench.py --steps 2 --warmup 1 --launches-per-step 16 --lanes 1 --no-cpu-baseline --no-secondary $args > $out/log 2>&1
    python tools/rocpd_summary.py $(find $out -name '*.db') 2>/dev/null | grep "mdec_encode_frames" | grep "_SIZE" | awk -v n=$name -v w=$w '{printf "%s %s %s %.1f MB%s\n", n, w, $(NF-2), $NF/1024, ($(NF-2)=="FETCH_SIZE" ? " (x2 on gfx950)" : "")}'
  done
done
