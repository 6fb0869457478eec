#!/bin/bash
# closing session on kernel mdec-k3.6 (pass order: raster behind the checkpoint): rocprofv3 passes for the four frame-kernel workloads,
# then session f (GPU suite, the driver's bench command, xacd trace, ADPCM soak), then MDEC soaks
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
timeout 300 bash tools/gpu_rocprof_mdec.sh a4 > $O/prof_a4.log 2>&1
timeout 300 bash tools/gpu_rocprof_mdec.sh a8 --amp 8 > $O/prof_a8.log 2>&1
timeout 400 bash tools/gpu_rocprof_mdec.sh v3_1250 --config sbs_v3 --total-frames 1250 --launches-per-step 40 > $O/prof_v3_1250.log 2>&1
timeout 500 bash tools/gpu_rocprof_mdec.sh v3_preset --config sbs_v3 --launches-per-step 5 > $O/prof_v3_preset.log 2>&1
out=$O/prof_a4_lanes2; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python bench.py --launches-per-step 400 --no-cpu-baseline --no-secondary > $out/kt.log 2>&1
python tools/rocpd_summary.py $(find $out -name '*.db' | sort) > $out/summary.txt 2>&1
timeout 900 bash tools/gpu_r04_session_f.sh
timeout 600 python tools/gpu_soak.py 300 20260930 3000 240 > $O/r04_soak_k3.6_single_launch.log 2>&1; tail -1 $O/r04_soak_k3.6_single_launch.log; grep -c " ok" $O/r04_soak_k3.6_single_launch.log
timeout 300 python tools/gpu_soak.py 72 616161 > $O/r04_soak_k3.6_batched.log 2>&1; tail -1 $O/r04_soak_k3.6_batched.log
timeout 300 python tools/gpu_soak_lanes.py 40 777 600 > $O/r04_soak_lanes_k3.6.log 2>&1; tail -1 $O/r04_soak_lanes_k3.6.log
