#!/bin/bash
# round 5, session B: kernel mdec-k3.7 (frame tickets as runs, trust policy): bytes first, then the diagnostics with and without runs
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_mdec.py -q -x > $O/r05b_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05b_pytest.log
tail -6 $O/r05b_pytest.log
timeout 600 python tools/gpu_r05_diag.py a4 a8 mixed --json $O/r05b_diag_run4.json > $O/r05b_diag_run4.log 2>&1
PSXHIP_MDEC_RUN=2 timeout 600 python tools/gpu_r05_diag.py a8 mixed --json $O/r05b_diag_run2.json > $O/r05b_diag_run2.log 2>&1
PSXHIP_MDEC_RUN=1 timeout 600 python tools/gpu_r05_diag.py a8 mixed --json $O/r05b_diag_run1.json > $O/r05b_diag_run1.log 2>&1
for r in 4 2 1; do echo "== v3_8k a8 v2_16k RUN=$r"; PSXHIP_MDEC_RUN=$r PSXHIP_MDEC_STATS=0 timeout 300 python tools/gpu_mdec_probe.py v3_8k v3_32k v2_16k a16 2>&1 | tail -4; done
tail -3 $O/r05b_diag_run4.log | cut -c1-600
