#!/bin/bash
# kernel trace of config 5 (xacd, full length)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/prof_xacd; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python bench.py --config xacd --steps 5 --no-cpu-baseline > $out/kt.log 2>&1
python tools/rocpd_summary.py $(find $out -name '*.db' | sort) > $out/summary.txt 2>&1
head -14 $out/summary.txt | cut -c1-150
tail -1 $out/kt.log | cut -c1-400
