#!/bin/bash
# rocprofv3 passes for config 3 device-resident (strcd, S streams per call): kernel-trace, then FETCH_SIZE / WRITE_SIZE on their own
# (MI355X_MICROARCH.md) and the instruction counters; the step is several kernels on two streams, the summary adds them up per step.
#   usage: tools/gpu_r05_strcd_pmc.sh [streams]
set -u
S=${1:-8}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/prof_strcd_pmc_S$S
rm -rf $out; mkdir -p $out
full="python bench.py --config strcd --str-streams $S --steps 40 --warmup 5 --no-cpu-baseline"
cmd="python bench.py --config strcd --str-streams $S --steps 6 --warmup 2 --no-cpu-baseline --steps-only"
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- $full > $out/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o r -- $cmd > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o r -- $cmd > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $out/sq -o r -- $cmd > $out/sq.log 2>&1
python tools/rocpd_summary.py --json $out/summary.json $(find $out -name "*.db" | sort) > $out/summary.txt 2>&1; find $out -name "*.db" -delete
grep "^{\"metric\"" $out/kt.log | tail -1 > $out/bench_line.json
grep -E "mdec_encode|str_video|adpcm_chunks|xa_assemble|^kernel|^==" $out/summary.txt | cut -c1-160
