#!/bin/bash
# round 5, session N: mdec-k3.6 (round 4's closing library, built from its commit) against the current library on one box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python tools/gpu_ab_rates.py build_ab/libpsxav_hip_k36.so psxavenc_amd/libpsxav_hip.so a4 v3a4 --rounds 3 --json gpurun_out/r05n_ab.json 2>&1 | tee gpurun_out/r05n_ab.log
