#!/bin/bash
# round-4 session C: the per-call fast paths (one launch per drop-in call): full GPU test suite, per-call timings with and without
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
make -s -C examples
python -m pytest tests -m gpu -q --durations=5 -x > $O/r04c_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r04c_pytest.log
tail -12 $O/r04c_pytest.log
for i in 1 2; do ./examples/percall_bench > $O/r04c_percall_fast_$i.json 2>> $O/r04c_percall.err; cat $O/r04c_percall_fast_$i.json; done
PSXHIP_NO_PERCALL_PATH=1 ./examples/percall_bench > $O/r04c_percall_old.json 2>> $O/r04c_percall.err; cat $O/r04c_percall_old.json
python bench.py --steps 5 --warmup 2 --lanes 1 --no-secondary --no-cpu-baseline > $O/r04c_bench_lanes1.json 2> $O/r04c_bench_lanes1.err
python - <<PY
import json
d = json.loads([l for l in open("$O/r04c_bench_lanes1.json").read().splitlines() if l.startswith("{")][-1])
print("lanes1", d["value"], "kernel_ms", d["roofline"]["kernel_ms"], d["parity"])
PY
