#!/bin/bash
# rocprofv3 passes for the two headline cases of the scaler, one case per run so that the per-dispatch averages mean one geometry:
#   kernel-trace --stats, FETCH_SIZE, WRITE_SIZE, SQ counters.  Summary -> gpurun_out/prof_scaler_cases/summary.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/prof_scaler_cases; rm -rf $out; mkdir -p $out
for c in "rgb 0 640 480 320 240" "yuv 1 640 480 320 240"; do
  set -- $c; tag=$1; shift
  rocprofv3 --kernel-trace --stats -d $out/${tag}_kt -o r -- python tools/gpu_scaler_probe.py $* > $out/${tag}_kt.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/${tag}_fetch -o r -- python tools/gpu_scaler_probe.py $* > $out/${tag}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/${tag}_write -o r -- python tools/gpu_scaler_probe.py $* > $out/${tag}_write.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $out/${tag}_sq -o r -- python tools/gpu_scaler_probe.py $* > $out/${tag}_sq.log 2>&1
  tail -1 $out/${tag}_kt.log
done
( echo "# rocprofv3 passes of tools/gpu_scaler_probe.py, one geometry per run (1000 pictures per dispatch): rgb = RGB24 640x480 -> 320x240, yuv = YUV420P 640x480 -> 320x240"
  python tools/rocpd_summary.py $(find $out -name '*.db' | sort) | grep -E "^==|^kernel|scaler_kernel" ) > $out/summary.txt
cat $out/summary.txt
