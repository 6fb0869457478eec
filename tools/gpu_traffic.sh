#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the MDEC kernel for one bench.py workload (two short PMC passes)
tag=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/traffic_$tag
rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/$c -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/$c.log 2>&1
done
python tools/rocpd_summary.py $(find $out -name '*.db' | sort) | grep -E "mdec_encode.*(FETCH|WRITE)" | cut -c50-120
