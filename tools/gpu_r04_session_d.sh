#!/bin/bash
# round-4 session D: per-call speculation inside one launch (tests + timings); A/B of the frame kernel k3.1 vs k3.2 on ONE box
cd "$(dirname "$0")/.."
O=gpurun_out
mkdir -p $O
make -s -C examples
python -m pytest tests/test_gpu_adpcm.py tests/test_gpu_dropin.py -m gpu -q --durations=5 -x > $O/r04d_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r04d_pytest.log
tail -12 $O/r04d_pytest.log
for i in 1 2; do ./examples/percall_bench > $O/r04d_percall_$i.json 2>> $O/r04d_percall.err; cat $O/r04d_percall_$i.json; done
for i in 1 2 3; do
  for v in k31 new; do
    lib=""; [ $v = k31 ] && lib="$PWD/build/k31/psxavenc_amd/libpsxav_hip.so"
    PSXAV_HIP_LIB=$lib python bench.py --steps 5 --warmup 2 --lanes 1 --launches-per-step 800 --no-secondary --no-cpu-baseline > $O/r04d_ab_${v}_$i.json 2> $O/r04d_ab_${v}_$i.err
    python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/r04d_ab_${v}_$i.json").read().splitlines() if l.startswith("{")][-1])
    print("$v $i", d["config"]["library"], d["value"], "kernel_ms", d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_stats"])
except Exception as e:
    print("$v $i ERR", e, open("$O/r04d_ab_${v}_$i.err").read()[-800:])
PY
  done
done
