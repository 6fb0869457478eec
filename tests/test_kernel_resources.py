"""Compile-time facts about the device code that the measurements rest on (no GPU needed: hipcc cross-compiles).

A frame kernel that spills vector registers still gives the right bytes -- and 3 MB of scratch traffic per launch (it
happened twice: `mdec-k2.12`, and the first version of the retry queue, both found in PMC passes after the fact).  The
product shapes of the frame kernel and the scaler kernels have to build without scratch."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "psxavenc_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _resource_usage(source):
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                        "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-Rpass-analysis=kernel-resource-usage",
                        "--cuda-device-only", "-c", source, "-o", os.devnull], cwd=CSRC, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return out


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
def test_frame_kernel_product_shapes_build_without_scratch():
    use = _resource_usage("mdec_kernels.hip")
    shapes = {k: v for k, v in use.items() if "mdec_encode_frames_kernel" in k and k.endswith("Lb0EEEvNS_8FrameJobE")}
    assert len(shapes) == 6, sorted(use)                      # 3 codecs x (12 wavefronts, two groups per CU | 16 wavefronts)
    for name, u in shapes.items():
        assert u["ScratchSize"] == "0" and u["VGPRs Spill"] == "0", (name, u)
        waves = 12 if "ELi12E" in name else 16
        # two 12-wavefront groups per CU need 6 wavefronts per SIMD (<= 80 VGPRs), one 16-wavefront group 4 (<= 128)
        assert int(u["VGPRs"]) <= (80 if waves == 12 else 128), (name, u)


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
def test_scaler_kernels_build_without_scratch():
    use = _resource_usage("frontend_kernels.hip")
    shapes = {k: v for k, v in use.items() if "scaler_kernel" in k}
    assert len(shapes) == 2, sorted(use)
    for name, u in shapes.items():
        assert u["ScratchSize"] == "0" and u["VGPRs Spill"] == "0" and int(u["VGPRs"]) <= 128, (name, u)


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
def test_column_coefficient_load_is_left_alone_until_it_is_waited_for():
    """the frame kernel fetches the column pass's sixteen coefficient pairs with an s_load_dwordx16 the compiler does not know is
    in flight (inline asm; the wait stands where the column pass starts, so that the row pass hides the load).  Between the two no
    instruction may read or write those scalar registers -- a copy or a spill there would move registers that are not loaded yet."""
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                        "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "--cuda-device-only", "-S", "mdec_kernels.hip", "-o", "-"],
                       cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_sload", os.path.join(CSRC, "check_sload.py"))      # the check the Makefile gates the build on
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check(r.stdout) >= 12
    # ... and it does refuse a listing in which something touches the registers before the wait
    bad = re.sub(r"(s_load_dwordx16\s+s\[(\d+):\d+\][^\n]*\n)", lambda m: m.group(1) + "\ts_mov_b32 s%s, 0\n" % m.group(2), r.stdout, count=1)
    with pytest.raises(AssertionError):
        mod.check(bad)
