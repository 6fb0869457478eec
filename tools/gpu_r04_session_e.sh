#!/bin/bash
# round-4 closing session on kernel mdec-k3.5: soaks against the oracle (batched and single-launch), the driver's bench command
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 700 python tools/gpu_soak.py 400 20260929 3000 240 > $O/r04_soak_k3.5_single_launch.log 2>&1
tail -3 $O/r04_soak_k3.5_single_launch.log
timeout 400 python tools/gpu_soak.py 96 515151 > $O/r04_soak_k3.5_batched.log 2>&1
tail -3 $O/r04_soak_k3.5_batched.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r04h_bench_default.json 2> $O/r04h_bench_default.err
python - <<PY
import json
d = json.loads([l for l in open("$O/r04h_bench_default.json").read().splitlines() if l.startswith("{")][-1])
print(d["metric"], d["value"], "ms/step", d["ms_per_step"], "timed", d.get("timed_region_s"), "roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "kernel_ms", "traffic")}, "issue", d["roofline"].get("issue"), "overlapped", d["roofline"].get("overlapped", {}).get("achieved"), "parity", d.get("parity"))
for k, v in (d.get("secondary") or {}).items():
    print(" ", k, json.dumps(v)[:300])
PY
tail -4 $O/r04h_bench_default.err
