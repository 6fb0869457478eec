#!/bin/bash
# quick PMC comparison of library builds on one box: tools/gpu_pmc_quick.sh <tag> <bench args...> -- name=path ...
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
for spec in "$@"; do
  name=${spec%%=*}; path=${spec#*=}; lib=""
  [ "$path" != cur ] && lib="$PWD/$path"
  out=gpurun_out/pmcq_${tag}_$name; mkdir -p $out
  cmd="python bench.py --steps 2 --warmup 1 --launches-per-step 16 --lanes 1 --no-cpu-baseline --no-secondary ${args[*]}"
  PSXAV_HIP_LIB=$lib rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $out/sq -o r -- $cmd > $out/sq.log 2>&1
  PSXAV_HIP_LIB=$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace -d $out/sq2 -o r -- $cmd > $out/sq2.log 2>&1
  PSXAV_HIP_LIB=$lib rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU --kernel-trace -d $out/sq3 -o r -- $cmd > $out/sq3.log 2>&1
  echo "=== $name"; python tools/rocpd_summary.py $(find $out -name '*.db' | sort) 2>&1 | grep -E "mdec_encode_frames|^== " | grep -v "^==" | awk '{print $2, $3, $4}' | sort -u
done
