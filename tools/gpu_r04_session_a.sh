#!/bin/bash
# round-4 session A: GPU tests, the default bench line (10 s timed region + the other configs as children + the RCCL world-size-1
# self-test), and all four presets through the RCCL leg (--force-dist under torch.distributed.run, kept logs)
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q --durations=8 -x > $O/r04a_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r04a_pytest.log
tail -15 $O/r04a_pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r04a_bench_default.json 2> $O/r04a_bench_default.err
for cfg in sbs_v2 sbs_v3 xacd strcd; do
  extra=""
  [ $cfg = sbs_v2 ] && extra="--launches-per-step 400"
  [ $cfg = sbs_v3 ] && extra="--launches-per-step 20"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-dist \
      --config $cfg --steps 5 --warmup 2 --no-secondary --cpu-seconds 4 $extra > $O/r04a_rccl_world1_$cfg.json 2> $O/r04a_rccl_world1_$cfg.err
  echo "== rccl world-1 $cfg rc=$?"; tail -c 600 $O/r04a_rccl_world1_$cfg.json; echo
done
python - <<PY
import json
d = json.loads([l for l in open("$O/r04a_bench_default.json").read().splitlines() if l.startswith("{")][-1])
print(d["metric"], d["value"], d["unit"], "ms/step", d["ms_per_step"], "timed", d.get("timed_region_s"), "frac", d["roofline"]["frac"], "parity", d.get("parity"))
for k, v in (d.get("secondary") or {}).items():
    print(" ", k, json.dumps(v)[:500])
PY
tail -5 $O/r04a_bench_default.err
