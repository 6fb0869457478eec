#!/bin/bash
# round 6: the split kernel (one frame across many workgroups) under the kernel trace, per segment size M
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
make -s -C $R/examples percall_bench
for M in 1 2 4 8 16; do
  echo "== M=$M"; PSXHIP_MDEC_SPLIT_M=$M timeout 120 $R/examples/percall_bench 10 10 400 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['encode_frame_bs_320x240_v2'])"
done
echo "== frame kernel"; PSXHIP_MDEC_SPLIT_MAX=0 timeout 120 $R/examples/percall_bench 10 10 400 | tail -1
rm -rf $O/split_kt; PSXHIP_MDEC_SPLIT_M=2 timeout 300 rocprofv3 --kernel-trace --stats -d $O/split_kt -o kt --output-format csv -- $R/examples/percall_bench 10 10 400 > /dev/null 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$O/split_kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
