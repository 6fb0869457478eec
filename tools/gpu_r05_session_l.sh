#!/bin/bash
# round 5, last profiling session on the final library: GPU suite, rocprofv3 (kernel-trace + PMC groups) for every workload of the bench
# line, per-call sweep, the default bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > $O/r05l_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05l_pytest.log; tail -3 $O/r05l_pytest.log
bash tools/gpu_rocprof_mdec.sh a4 > $O/prof_a4.log 2>&1
bash tools/gpu_rocprof_mdec.sh a8 --amp 8 > $O/prof_a8.log 2>&1
bash tools/gpu_rocprof_mdec.sh mixed --content mixed > $O/prof_mixed.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3_1250 --config sbs_v3 --total-frames 1250 --launches-per-step 40 > $O/prof_v3_1250.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3_preset --config sbs_v3 --launches-per-step 5 > $O/prof_v3_preset.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3_32k --codec 1 --width 640 --height 480 --budget 32768 --amp 8 --frames 1250 --launches-per-step 40 > $O/prof_v3_32k.log 2>&1
bash tools/gpu_rocprof_mdec.sh str_cycle --budget-cycle 16128,18144,18144,18144 --launches-per-step 200 > $O/prof_str_cycle.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3dc_8k --codec 2 --launches-per-step 200 > $O/prof_v3dc_8k.log 2>&1
out=$O/prof_a4_lanes2; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python bench.py --launches-per-step 400 --no-cpu-baseline --no-secondary > $out/kt.log 2>&1
python tools/rocpd_summary.py --json $out/summary.json $(find $out -name '*.db' | sort) > $out/summary.txt 2>&1; find $out -name "*.db" -delete
out=$O/prof_strcd_S8; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python bench.py --config strcd --steps 40 --warmup 5 --no-cpu-baseline > $out/kt.log 2>&1
python tools/rocpd_summary.py --json $out/summary.json $(find $out -name '*.db' | sort) > $out/summary.txt 2>&1; find $out -name "*.db" -delete
bash tools/gpu_r05_xacd_pmc.sh tonal 0 > $O/r05l_xacd_tonal.log 2>&1
bash tools/gpu_r05_xacd_pmc.sh white 2 > $O/r05l_xacd_white.log 2>&1
bash tools/gpu_r05_xacd_pmc.sh gated 5 > $O/r05l_xacd_gated.log 2>&1
make -s -C examples percall_bench; ./examples/percall_bench 2000 300 300 1 > $O/r05l_percall_sweep.json 2>&1
./oracle/cpu_bench spucall oracle/_ref/libpsxav_ref.so > $O/r05l_cpu_spucall.json
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r05l_bench_default.json 2> $O/r05l_bench_default.err; tail -4 $O/r05l_bench_default.err
[ -f build_ab/libpsxav_hip_k36.so ] && timeout 1200 python tools/gpu_ab_rates.py build_ab/libpsxav_hip_k36.so psxavenc_amd/libpsxav_hip.so a4 a8 mixed v3a4 --rounds 3 --json $O/r05l_ab_k36_vs_final.json > $O/r05l_ab.log 2>&1
timeout 900 python tools/gpu_r05_diag.py a4 a8 mixed v3a4 --json $O/r05l_diag.json > $O/r05l_diag.log 2>&1
find $O -name "*.db" -delete; du -sh $O
bash tools/gpu_r05_strcd_pmc.sh 8 > $O/r05l_strcd_pmc.log 2>&1
find $O -name "*.db" -delete; du -sh $O
