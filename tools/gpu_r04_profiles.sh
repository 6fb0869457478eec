#!/bin/bash
# round-4 final session: GPU tests, the driver's bench command, rocprofv3 passes (kernel-trace + PMC) for the frame kernel's
# workloads, a kernel trace of the default (two launch lanes) run, xacd
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -q --durations=6 > $O/r04f_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r04f_pytest.log
tail -12 $O/r04f_pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r04f_bench_default.json 2> $O/r04f_bench_default.err
bash tools/gpu_rocprof_mdec.sh a4 > $O/prof_a4.log 2>&1
bash tools/gpu_rocprof_mdec.sh a8 --amp 8 > $O/prof_a8.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3_1250 --config sbs_v3 --total-frames 1250 --launches-per-step 40 > $O/prof_v3_1250.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3_preset --config sbs_v3 --launches-per-step 5 > $O/prof_v3_preset.log 2>&1
out=$O/prof_a4_lanes2; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python bench.py --launches-per-step 400 --no-cpu-baseline --no-secondary > $out/kt.log 2>&1
python tools/rocpd_summary.py $(find $out -name '*.db' | sort) > $out/summary.txt 2>&1
out=$O/prof_xacd; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python bench.py --config xacd --steps 5 --no-cpu-baseline > $out/kt.log 2>&1
python tools/rocpd_summary.py $(find $out -name '*.db' | sort) > $out/summary.txt 2>&1
for t in a4 a8 v3_1250 v3_preset a4_lanes2 xacd; do echo "=== $t"; head -12 $O/prof_$t/summary.txt; done
python - <<PY
import json
d = json.loads([l for l in open("$O/r04f_bench_default.json").read().splitlines() if l.startswith("{")][-1])
print(d["metric"], d["value"], "ms/step", d["ms_per_step"], "timed", d.get("timed_region_s"), "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "kernel_ms", "traffic")}, "overlapped", d["roofline"].get("overlapped", {}).get("achieved"), "parity", d.get("parity"))
for k, v in (d.get("secondary") or {}).items():
    print(" ", k, json.dumps(v)[:420])
PY
tail -4 $O/r04f_bench_default.err
