#!/bin/bash
# round-4 session B: kernel k3.2 (batch table, launch lanes): new tests, then the bench in both launch orders
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
python -m pytest tests/test_gpu_mdec.py tests/test_gpu_dropin.py -m gpu -q --durations=5 -x > $O/r04b_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r04b_pytest.log
tail -12 $O/r04b_pytest.log
python bench.py --steps 5 --warmup 2 --no-config-secondaries --cpu-seconds 3 > $O/r04b_bench_lanes2.json 2> $O/r04b_bench_lanes2.err
python bench.py --steps 5 --warmup 2 --lanes 1 --no-secondary --no-cpu-baseline > $O/r04b_bench_lanes1.json 2> $O/r04b_bench_lanes1.err
python - <<PY
import json
for f in ("lanes2", "lanes1"):
    try:
        d = json.loads([l for l in open("$O/r04b_bench_%s.json" % f).read().splitlines() if l.startswith("{")][-1])
        print(f, d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "in_order", d["roofline"].get("in_order"), "parity", d.get("parity"))
        for k, v in (d.get("secondary") or {}).items():
            print(" ", k, json.dumps(v)[:400])
    except Exception as e:
        print(f, "ERR", e); print(open("$O/r04b_bench_%s.err" % f).read()[-1500:])
PY
