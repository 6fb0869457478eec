#!/bin/bash
# one optimisation step on one box: timing A/B (three workloads, parity checked) + PMC counters on the headline workload
# usage: tools/gpu_step.sh <tag> name=path ...
cd "$(dirname "$0")/.."
tag=$1; shift
bash tools/gpu_ab.sh $tag "$@" 2>&1 | grep -v "run 2"
bash tools/gpu_pmc_quick.sh $tag --amp 4 -- "$@" > /dev/null 2>&1
dirs=""; for spec in "$@"; do dirs="$dirs gpurun_out/pmcq_${tag}_${spec%%=*}"; done
python tools/pmc_compare.py $dirs | grep -E "counter|INSTS_VALU|INSTS_SALU|INSTS_LDS|INSTS_BRANCH|kernel_ns\(sq\)|WAIT_INST_ANY|LDS_BANK"
