#!/bin/bash
# round-4 closing session: full GPU test suite, the driver's bench command, kernel trace of config 5, ADPCM soak
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -q --durations=4 > $O/r04j_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r04j_pytest.log
tail -8 $O/r04j_pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r04j_bench_default.json 2> $O/r04j_bench_default.err
bash tools/gpu_prof_xacd.sh > $O/prof_xacd.log 2>&1; head -8 $O/prof_xacd/summary.txt | cut -c1-150
timeout 600 python tools/gpu_adpcm_soak.py 300 > $O/r04_adpcm_soak_300_final.log 2>&1; tail -2 $O/r04_adpcm_soak_300_final.log
python - <<PY
import json
d = json.loads([l for l in open("$O/r04j_bench_default.json").read().splitlines() if l.startswith("{")][-1])
print(d["metric"], d["value"], "ms/step", d["ms_per_step"], "timed", d.get("timed_region_s"), "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "kernel_ms", "traffic")}, "overlapped", d["roofline"].get("overlapped", {}).get("achieved"), "parity", d.get("parity"))
for k, v in (d.get("secondary") or {}).items():
    print(" ", k, json.dumps(v)[:260])
PY
tail -4 $O/r04j_bench_default.err
