#!/bin/bash
# round 5, session F: device STR + world-size-2 tests, the 8-GPU prediction for config 5, per-call break-even, the default bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_str_device.py tests/test_bench_dist.py -q -x > $O/r05f_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05f_pytest.log
tail -6 $O/r05f_pytest.log
timeout 900 python tools/gpu_r05_predict_8gpu.py --json $O/r05f_predict_8gpu_xacd.json > $O/r05f_predict.log 2>&1; tail -4 $O/r05f_predict.log | cut -c1-700
make -s -C examples percall_bench; ./examples/percall_bench 2000 300 300 1 > $O/r05f_percall_sweep.json 2>&1; cat $O/r05f_percall_sweep.json | cut -c1-900
./oracle/cpu_bench spucall oracle/_ref/libpsxav_ref.so > $O/r05f_cpu_spucall.json; cat $O/r05f_cpu_spucall.json
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r05f_bench_default.json 2> $O/r05f_bench_default.err; tail -5 $O/r05f_bench_default.err
python - <<PY
import json
d = json.loads([l for l in open("$O/r05f_bench_default.json").read().splitlines() if l.startswith("{")][-1])
print(d["metric"], d["value"], "ms/step", d["ms_per_step"], "timed", d.get("timed_region_s"), "roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "frac_overlapped", "kernel_ms", "traffic")}, "parity", d.get("parity"))
print(json.dumps(d["config"].get("secondary_summary")))
for k, v in (d.get("secondary") or {}).items():
    print(" ", k, json.dumps(v)[:500])
PY
