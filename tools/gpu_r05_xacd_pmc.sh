#!/bin/bash
# rocprofv3 passes for config 5 (xacd, all 540 000 sectors) on one synthetic material: kernel-trace, then one --pmc pass per counter
# group (FETCH_SIZE / WRITE_SIZE on their own, MI355X_MICROARCH.md).   usage: tools/gpu_r05_xacd_pmc.sh <tag> <audio-kind>
set -u
tag=$1; kind=$2
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/prof_xacd_$tag
rm -rf $out; mkdir -p $out
full="python bench.py --config xacd --audio-kind $kind --steps 5 --warmup 2 --no-cpu-baseline"
cmd="python bench.py --config xacd --audio-kind $kind --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- $full > $out/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o r -- $cmd > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o r -- $cmd > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $out/sq -o r -- $cmd > $out/sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_BRANCH --kernel-trace -d $out/sq2 -o r -- $cmd > $out/sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --kernel-trace -d $out/sq3 -o r -- $cmd > $out/sq3.log 2>&1
python tools/rocpd_summary.py --json $out/summary.json $(find $out -name "*.db" | sort) > $out/summary.txt 2>&1; find $out -name "*.db" -delete
grep "^{\"metric\"" $out/kt.log | tail -1 > $out/bench_line.json
grep -E "adpcm_chunks|xa_assemble|^kernel|^==" $out/summary.txt | cut -c1-160
