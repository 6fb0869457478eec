#!/bin/bash
# round-3 validation session: tests, the four BASELINE presets, a 2-rank share-GPU run, host-path tools
cd "$(dirname "$0")/.."
O=gpurun_out
python -m pytest tests -m gpu -q --durations=8 > $O/r03_pytest_b.log 2>&1; echo "pytest rc=$?" >> $O/r03_pytest_b.log
tail -15 $O/r03_pytest_b.log
python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err
python bench.py --config sbs_v3 > $O/r03_bench_sbs_v3.json 2> $O/r03_bench_sbs_v3.err
python bench.py --config xacd --steps 5 > $O/r03_bench_xacd.json 2> $O/r03_bench_xacd.err
python bench.py --config strcd > $O/r03_bench_strcd.json 2> $O/r03_bench_strcd.err
python bench.py --config sbs_v3 --gpus 2 --dist-backend gloo --share-gpu --steps 5 > $O/r03_bench_sbs_v3_2rank_sharegpu.json 2> $O/r03_bench_sbs_v3_2rank.err
python tools/gpu_multi_host.py > $O/r03_multi_host.log 2>&1
python tools/gpu_percall_audio.py > $O/r03_percall_audio.log 2>&1
for f in default sbs_v3 xacd strcd sbs_v3_2rank_sharegpu; do echo "== $f"; python - <<PY
import json
try:
    d = json.loads(open("$O/r03_bench_$f.json").read().strip().splitlines()[-1])
    print(d["metric"], d["value"], d["unit"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "parity", d.get("parity"))
    print("cpu", json.dumps(d.get("cpu_baseline"))[:600])
    print("secondary", json.dumps(d.get("secondary"))[:600])
except Exception as e:
    print("ERR", e); print(open("$O/r03_bench_$f.err").read()[-1500:] if "$f" != "sbs_v3_2rank_sharegpu" else open("$O/r03_bench_sbs_v3_2rank.err").read()[-1500:])
PY
done
tail -8 $O/r03_multi_host.log; tail -20 $O/r03_percall_audio.log
