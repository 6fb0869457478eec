#!/bin/bash
# round 5, final session: mixed-content soak, rocprofv3 (kernel-trace + PMC groups) for every workload of the bench line on mdec-k3.7 /
# adpcm-k5.0, the predictions' missing points, the default bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python tools/gpu_soak_mixed.py 60 555 900 > $O/r05j_soak_mixed.log 2>&1; tail -2 $O/r05j_soak_mixed.log
bash tools/gpu_rocprof_mdec.sh a4 > $O/prof_a4.log 2>&1
bash tools/gpu_rocprof_mdec.sh a8 --amp 8 > $O/prof_a8.log 2>&1
bash tools/gpu_rocprof_mdec.sh mixed --content mixed > $O/prof_mixed.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3_1250 --config sbs_v3 --total-frames 1250 --launches-per-step 40 > $O/prof_v3_1250.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3_preset --config sbs_v3 --launches-per-step 5 > $O/prof_v3_preset.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3_32k --codec 1 --width 640 --height 480 --budget 32768 --amp 8 --frames 1250 --launches-per-step 40 > $O/prof_v3_32k.log 2>&1
bash tools/gpu_rocprof_mdec.sh str_cycle --budget-cycle 16128,18144,18144,18144 --launches-per-step 200 > $O/prof_str_cycle.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3dc_8k --codec 2 --launches-per-step 200 > $O/prof_v3dc_8k.log 2>&1
out=$O/prof_a4_lanes2; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python bench.py --launches-per-step 400 --no-cpu-baseline --no-secondary > $out/kt.log 2>&1
python tools/rocpd_summary.py --json $out/summary.json $(find $out -name '*.db' | sort) > $out/summary.txt 2>&1; find $out -name "*.db" -delete
out=$O/prof_strcd_S8; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python bench.py --config strcd --steps 40 --warmup 5 --no-cpu-baseline > $out/kt.log 2>&1
python tools/rocpd_summary.py --json $out/summary.json $(find $out -name '*.db' | sort) > $out/summary.txt 2>&1; find $out -name "*.db" -delete
bash tools/gpu_r05_xacd_pmc.sh tonal 0 > $O/r05j_xacd_tonal.log 2>&1
bash tools/gpu_r05_xacd_pmc.sh white 2 > $O/r05j_xacd_white.log 2>&1
bash tools/gpu_r05_xacd_pmc.sh gated 5 > $O/r05j_xacd_gated.log 2>&1
for w in 2 4; do timeout 600 python tools/gpu_r05_predict_8gpu.py --world $w --json $O/r05j_predict_${w}gpu_xacd.json > $O/r05j_predict_$w.log 2>&1; done
for tf in 5000 2500; do timeout 300 python bench.py --config sbs_v3 --total-frames $tf --steps 8 --warmup 2 --no-secondary --no-cpu-baseline > $O/r05j_sbs_v3_$tf.json 2>/dev/null; done
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r05j_bench_default.json 2> $O/r05j_bench_default.err; tail -4 $O/r05j_bench_default.err
find $O -name "*.db" -delete; du -sh $O
for t in a4 a8 mixed v3_1250 v3_preset v3_32k str_cycle v3dc_8k a4_lanes2 strcd_S8; do echo "=== $t"; sed -n 2,4p $O/prof_$t/summary.txt | cut -c1-150; done
