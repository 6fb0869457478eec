#!/bin/bash
# round 5, session P: where k3.7's extra vector instructions are: per launch, per frame or per macroblock (instruction counters, k3.6 against the current build, three shapes)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05p; mkdir -p $O
i=3
for lib in psxavenc_amd/libpsxav_hip.so; do
  i=$((i+1))
  for shape in "a4_300mb --frames 1000" "a4_1200mb --width 640 --height 480 --budget 32768 --frames 250" "a4_70mb --width 160 --height 112 --budget 2048 --frames 4000"; do
    tag=${shape%% *}; args=${shape#* }
    out=$O/${tag}_lib$i; mkdir -p $out
    PSXAV_HIP_LIB=$PWD/$lib rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $out/sq -o r -- python bench.py --lanes 1 --steps 2 --warmup 1 --launches-per-step 16 --no-cpu-baseline --no-secondary $args > $out/sq.log 2>&1
    python tools/rocpd_summary.py --json $out/summary.json $(find $out -name "*.db" | sort) > $out/summary.txt 2>&1; find $out -name "*.db" -delete
    echo "== $tag $lib"; grep -i "INSTS_VALU\|INSTS_SALU\|INSTS_LDS\|launches" $out/summary.txt | head -8
  done
done
