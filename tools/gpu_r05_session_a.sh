#!/bin/bash
# round 5, session A: where the tree stands before any kernel change -- lane tests, the off-friendly-point diagnostics, ADPCM
# counters (three materials), the shapes the round-4 diet did not re-measure
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_gpu_mdec.py -q -x -k "lane or host_path or batches" > $O/r05a_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05a_pytest.log
tail -4 $O/r05a_pytest.log
python tools/gpu_r05_diag.py a4 a8 mixed --json $O/r05a_diag.json > $O/r05a_diag.log 2>&1
tail -5 $O/r05a_diag.log | cut -c1-1800
bash tools/gpu_r05_xacd_pmc.sh tonal 0 > $O/r05a_xacd_tonal.log 2>&1
bash tools/gpu_r05_xacd_pmc.sh white 2 > $O/r05a_xacd_white.log 2>&1
bash tools/gpu_r05_xacd_pmc.sh gated 5 > $O/r05a_xacd_gated.log 2>&1
tail -30 $O/r05a_xacd_tonal.log
bash tools/gpu_rocprof_mdec.sh v3_32k --codec 1 --width 640 --height 480 --budget 32768 --amp 8 --frames 1250 --launches-per-step 40 > $O/prof_v3_32k.log 2>&1
bash tools/gpu_rocprof_mdec.sh str_cycle --budget-cycle 16128,18144,18144,18144 --launches-per-step 200 > $O/prof_str_cycle.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3dc_8k --codec 2 --launches-per-step 200 > $O/prof_v3dc_8k.log 2>&1
for t in v3_32k str_cycle v3dc_8k; do echo "=== $t"; head -6 $O/prof_$t/summary.txt | cut -c1-170; cut -c1-300 $O/prof_$t/bench_line.json; done
find gpurun_out -name "*.db" -delete; du -sh gpurun_out
