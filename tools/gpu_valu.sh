#!/bin/bash
# one rocprofv3 SQ pass of a bench.py workload: wave-instructions per macroblock of the MDEC kernel.
#   usage: tools/gpu_valu.sh <tag> <macroblocks per launch> [bench.py args...]
set -u
tag=$1; mbs=$2; shift; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/valu_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace -d $out -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --launches-per-step 1 "$@" > $out/log 2>&1
python - $out $mbs $tag <<'PY'
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
mbs = float(sys.argv[2])
c = sqlite3.connect(db)
rows = c.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%mdec_encode%' group by counter_name").fetchall()
d = dict(rows)
t = c.execute("select avg(end-start) from kernels where name like '%mdec_encode%'").fetchone()[0]
print("%-10s kernel %.1f us | per MB: VALU %.1f SALU %.1f LDS %.1f VMEM_RD %.2f | VALU busy %.0f%% of %d SIMD-slots | wait_any %.0f%% wait_inst %.0f%% of wave cycles" % (
    sys.argv[3], t / 1e3, d["SQ_INSTS_VALU"] / mbs, d["SQ_INSTS_SALU"] / mbs, d["SQ_INSTS_LDS"] / mbs, d["SQ_INSTS_VMEM_RD"] / mbs,
    100 * d["SQ_ACTIVE_INST_VALU"] / (t * 2.4 / 4 * 1024), t * 2.4 / 4 * 1024,
    100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]))
PY
