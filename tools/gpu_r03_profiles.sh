#!/bin/bash
# round-3 rocprofv3 passes: headline, config 4 preset, the scaler kernel, xacd
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
bash tools/gpu_rocprof_mdec.sh a4 > gpurun_out/prof_a4.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3_preset --config sbs_v3 > gpurun_out/prof_v3_preset.log 2>&1
bash tools/gpu_rocprof_mdec.sh v3_1250 --config sbs_v3 --total-frames 1250 > gpurun_out/prof_v3_1250.log 2>&1
# xacd (600 s) and the scaler: kernel-trace --stats only
out=gpurun_out/prof_xacd; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python bench.py --workload xacd --no-cpu-baseline > $out/kt.log 2>&1
python tools/rocpd_summary.py $(find $out -name '*.db' | sort) > $out/summary.txt 2>&1
out=gpurun_out/prof_scaler; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python tools/gpu_frontend_bench.py > $out/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o r -- python tools/gpu_frontend_bench.py > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o r -- python tools/gpu_frontend_bench.py > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $out/sq -o r -- python tools/gpu_frontend_bench.py > $out/sq.log 2>&1
python tools/rocpd_summary.py $(find $out -name '*.db' | sort) > $out/summary.txt 2>&1
for t in a4 v3_preset v3_1250 xacd scaler; do echo "=== $t"; head -40 gpurun_out/prof_$t/summary.txt; done
