#!/bin/bash
# round 5, session R: the distributed-bench tests again, then the default bench line on the final library with the final counter index
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_bench_dist.py -m gpu -q > $O/r05r_pytest_dist.log 2>&1; tail -2 $O/r05r_pytest_dist.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r05r_bench_default.json 2> $O/r05r_bench_default.err; tail -4 $O/r05r_bench_default.err
python - <<PY
import json
d=json.loads(open("$O/r05r_bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["traffic"], d["roofline"]["frac"])
for k,v in d["secondary"].items():
    if isinstance(v,dict) and "roofline" in v: print(k, v.get("value"), v["roofline"].get("traffic"), v["roofline"].get("traffic_source"))
PY
