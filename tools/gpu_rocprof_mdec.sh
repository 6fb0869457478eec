#!/bin/bash
# rocprofv3 passes for one bench.py workload (run on the GPU box, from the repo root).
#   usage: tools/gpu_rocprof_mdec.sh <tag> [bench.py args...]
# kernel-trace and each PMC group run separately (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE need their own passes).
# Summaries land in gpurun_out/prof_<tag>/summary.txt
set -u
tag=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
# kernel-trace: the bench's own command line (defaults: 20 steps x 400 launches over 4 distinct batches), so that the kernel's
# average duration here and bench.py's HIP-event mean are measurements of the same thing; PMC passes: a shorter run of the
# same workload (counter collection serialises the kernels)
# --lanes 1: every launch waits for the one before, so a kernel's span in the trace IS its duration (bench.py's default overlaps
# consecutive launches of its one context; its roofline block measures the in-order duration live, which is what this must agree with)
full="python bench.py --lanes 1 --launches-per-step 400 --no-cpu-baseline --no-secondary $*"
cmd="python bench.py --lanes 1 --steps 2 --warmup 1 --launches-per-step 16 --no-cpu-baseline --no-secondary $*"
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- $full > $out/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o r -- $cmd > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o r -- $cmd > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $out/sq -o r -- $cmd > $out/sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $out/sq2 -o r -- $cmd > $out/sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --kernel-trace -d $out/sq3 -o r -- $cmd > $out/sq3.log 2>&1
python tools/rocpd_summary.py --json $out/summary.json $(find $out -name "*.db" | sort) > $out/summary.txt 2>&1; find $out -name "*.db" -delete
grep "^{\"metric\"" $out/kt.log | tail -1 > $out/bench_line.json
cat $out/summary.txt
