#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...] : a copy of the current library sources built under build/<name>/ (A/B measurements)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
rm -rf build/$name; mkdir -p build/$name/psxavenc_amd
cp -r include build/$name/
cp -r psxavenc_amd/csrc build/$name/psxavenc_amd/
rm -f build/$name/psxavenc_amd/csrc/*.o
make -s -C build/$name/psxavenc_amd/csrc HIPCC="/opt/rocm/bin/hipcc $*" 2>&1 | grep -E "error|warning" || true
ls -la build/$name/psxavenc_amd/libpsxav_hip.so
