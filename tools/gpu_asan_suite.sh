#!/bin/bash
# The GPU suite with gcc's AddressSanitizer runtime preloaded: the host-C drop-in layer (host_mdec.c, host_audio.c, host_cdrom.c) is built
# instrumented, and the runtime's interceptors (memcpy / memset / str* / malloc red zones) check every caller in the process -- the C-ABI
# layer's staging copies, the muxer's plan code, ctypes buffers -- against heap red zones.  (clang's ROCm ASan runtime intercepts the HSA
# allocator and does not come up on this box; instrumenting the hipcc-compiled host code is left to a box whose ROCm has the ASan libraries.)
# Build first: the three host C files with gcc -fsanitize=address -O1 -g, linked with the product's other objects into build_ab/libpsxav_hip_asan.so
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
export ASAN_OPTIONS=allocator_may_return_null=1:detect_leaks=0:verify_asan_link_order=0:abort_on_error=0:halt_on_error=0:log_path=$PWD/$O/asan_log
export UBSAN_OPTIONS=print_stacktrace=1:log_path=$PWD/$O/ubsan_log
export LD_LIBRARY_PATH=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),\"lib\"))"):$LD_LIBRARY_PATH
export PSXAV_HIP_LIB=$PWD/build_ab/libpsxav_hip_asan.so
LD_PRELOAD="/usr/lib/x86_64-linux-gnu/libasan.so.6 /usr/lib/x86_64-linux-gnu/libstdc++.so.6" timeout 2400 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_str_device.py tests/test_gpu_adpcm.py tests/test_gpu_mdec.py tests/test_gpu_frontend.py -q -x -p no:cacheprovider > $O/asan_pytest.log 2>&1
echo "rc=$?" >> $O/asan_pytest.log
tail -5 $O/asan_pytest.log
ls $O | grep -c "asan_log\|ubsan_log"; for f in $O/asan_log* $O/ubsan_log*; do [ -f "$f" ] && { echo "== $f"; head -40 "$f"; }; done 2>/dev/null | head -150
