#!/bin/bash
# round 5, session C: mdec-k3.7 with the parallel slot re-arm and the re-pilot on far-off hints: bytes, then runs 1 / 2 / 4
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_mdec.py -q -x > $O/r05c_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05c_pytest.log
tail -4 $O/r05c_pytest.log
for r in 1 2 4; do PSXHIP_MDEC_RUN=$r timeout 600 python tools/gpu_r05_diag.py a4 a8 mixed --json $O/r05c_diag_run$r.json > $O/r05c_diag_run$r.log 2>&1; done
python tools/gpu_r05_runs_ab.py 1000 4; python tools/gpu_r05_runs_ab.py 2048 4; python tools/gpu_r05_runs_ab.py 4000 4
for r in 4 1; do echo "== RUN=$r"; PSXHIP_MDEC_RUN=$r PSXHIP_MDEC_STATS=0 timeout 300 python tools/gpu_mdec_probe.py v3_8k v3_32k v2_16k a16 2>&1 | tail -4; done
