#!/bin/bash
# round-6 closing session: GPU tests, the driver's bench command, rocprofv3 passes (kernel-trace + PMC) for the headline's frame kernel,
# config 5 on the tonal signal (adpcm-k5.1), config 3 at one stream, and the split kernel under the one-frame call pattern
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -q --durations=5 > $O/r06_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r06_pytest.log
tail -9 $O/r06_pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/r06_bench_default.json 2> $O/r06_bench_default.err
wc -c $O/r06_bench_default.json
cp $O/bench_detail_sbs_v2_n1.json $O/r06_bench_default_detail.json
bash tools/gpu_rocprof_mdec.sh a4 > $O/prof_a4.log 2>&1
bash tools/gpu_r05_xacd_pmc.sh tonal 0 > $O/prof_xacd_tonal.log 2>&1
bash tools/gpu_r05_strcd_pmc.sh 1 > $O/prof_strcd_S1.log 2>&1
# the split kernel: the reference's call pattern (examples/percall_bench: 2000 one-frame calls) under the kernel trace, then the counters
make -s -C examples percall_bench
out=$O/prof_split; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- examples/percall_bench 10 10 2000 > $out/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o r -- examples/percall_bench 10 10 200 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o r -- examples/percall_bench 10 10 200 > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $out/sq -o r -- examples/percall_bench 10 10 200 > $out/sq.log 2>&1
python tools/rocpd_summary.py --json $out/summary.json $(find $out -name "*.db" | sort) > $out/summary.txt 2>&1; find $out -name "*.db" -delete
grep -E "mdec_split|^kernel|^==" $out/summary.txt | cut -c1-170
for r in 1 2 3; do PSXHIP_PERCALL_TRACE=1 examples/percall_bench 2000 300 3000 2>&1 | tail -2; done > $O/r06_percall.log 2>&1
tail -2 $O/r06_percall.log | cut -c1-400
for t in a4 xacd_tonal strcd_pmc_S1; do echo "=== $t"; head -14 $O/prof_$t/summary.txt | cut -c1-170; done
