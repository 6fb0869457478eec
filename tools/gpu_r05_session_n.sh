#!/bin/bash
# round 5, session N: mdec-k3.6 (round 4's closing library, built from its commit) against the current library and experiment builds, on one box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python tools/gpu_ab_rates.py build_ab/*.so psxavenc_amd/libpsxav_hip.so a4 --rounds 2 --json gpurun_out/r05n_ab2.json 2>&1 | tee gpurun_out/r05n_ab2.log
