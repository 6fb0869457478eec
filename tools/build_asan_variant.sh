#!/bin/bash
# build_ab/libpsxav_hip_asan.so: the product's objects with the host-C drop-in layer (host_mdec.c, host_audio.c, host_cdrom.c) rebuilt under
# gcc's AddressSanitizer; tools/gpu_asan_suite.sh runs the GPU suite on it with the ASan runtime preloaded.  (Run after `make -C psxavenc_amd/csrc`.)
set -e
cd "$(dirname "$0")/.."
B=/tmp/psxav_asan_build; rm -rf $B; mkdir -p $B/psxavenc_amd build_ab
cp -r psxavenc_amd/csrc $B/psxavenc_amd/; cp -r include $B/
cd $B/psxavenc_amd/csrc
for f in host_mdec host_audio host_cdrom; do gcc -std=c11 -O1 -g -fPIC -Wall -I../../include -D_POSIX_C_SOURCE=201112L -fsanitize=address -fno-omit-frame-pointer -c $f.c -o $f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o libpsxav_hip_asan.so mdec_kernels.o adpcm_kernels.o synth_kernels.o frontend_kernels.o \
    psxhip_api.o psxhip_audio_api.o psxhip_str.o psxhip_spufile.o psxhip_multi.o host_mdec.o host_audio.o host_cdrom.o
cp libpsxav_hip_asan.so "$OLDPWD/build_ab/"
echo "built build_ab/libpsxav_hip_asan.so"
