#!/bin/bash
# PMC counters of the xacd workload's kernels (config 5 at a tenth of its length): instruction mix of the ADPCM kernels
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/pmc_xacd; mkdir -p $out
cmd="python bench.py --config xacd --audio-seconds 360 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $out/sq -o r -- $cmd > $out/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $out/sq3 -o r -- $cmd > $out/sq3.log 2>&1
python tools/rocpd_summary.py $(find $out -name '*.db' | sort) 2>&1 | grep -E "adpcm_chunks|xa_assemble|^kernel" | cut -c1-150
