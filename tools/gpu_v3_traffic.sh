#!/bin/bash
# config 4's second frame read -- does the DC pre-pass's re-read hit the L2 when (i) half as many frames are in
# flight (16-wavefront shape, one group per CU: PSXHIP_MDEC_LARGE=1) or (ii) the pre-pass walks the frame in reverse of the main pass?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
run() {   # name, lib, env
  name=$1; lib=$2; shift 2
  out=gpurun_out/v3t_$name; mkdir -p $out
  cmd="python bench.py --config sbs_v3 --total-frames 1250 --steps 2 --warmup 1 --launches-per-step 8 --lanes 1 --no-cpu-baseline --no-secondary"
  env "$@" PSXAV_HIP_LIB=$lib python bench.py --config sbs_v3 --total-frames 1250 --steps 4 --warmup 2 --launches-per-step 40 --lanes 1 --no-cpu-baseline --no-secondary > $out/bench.json 2> $out/bench.err
  env "$@" PSXAV_HIP_LIB=$lib rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o r -- $cmd > $out/fetch.log 2>&1
  env "$@" PSXAV_HIP_LIB=$lib rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o r -- $cmd > $out/write.log 2>&1
  env "$@" PSXAV_HIP_LIB=$lib rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $out/tcc -o r -- $cmd > $out/tcc.log 2>&1
  python - <<PY
import json
d = json.loads([l for l in open("$out/bench.json").read().splitlines() if l.startswith("{")][-1])
print("$name", d["value"], "frames/s kernel_ms", d["roofline"]["kernel_ms"], "shape", d["config"]["kernel_shape"], "parity", d["parity"]["bit_exact"])
PY
}
run base "$PWD/build/k33/psxavenc_amd/libpsxav_hip.so" X=1
run large "$PWD/build/k33/psxavenc_amd/libpsxav_hip.so" PSXHIP_MDEC_LARGE=1
run dcrev "$PWD/build/dcrev/psxavenc_amd/libpsxav_hip.so" X=1
python tools/pmc_compare.py gpurun_out/v3t_base gpurun_out/v3t_large gpurun_out/v3t_dcrev
