#!/usr/bin/env python3
"""Build-time check of the frame kernel's disassembly (psxavenc_amd/csrc/Makefile runs it on every rebuild of mdec_kernels.hip;
tests/test_kernel_resources.py runs the same function).

The frame kernel fetches the column pass's sixteen coefficient pairs with an s_load_dwordx16 the compiler does not know is in flight
(inline asm; the wait stands where the column pass starts, so that the row pass hides the load).  Between the two no instruction may
read or write those scalar registers -- a copy or a spill there would move registers that are not loaded yet, and the kernel would
compute wrong DCT coefficients without any build or run-time error.  A compiler upgrade or a flag change that breaks
the assumption fails the BUILD here.

usage: hipcc ... --cuda-device-only -S mdec_kernels.hip -o - | python3 check_sload.py"""
import re
import sys


def _sgprs(text):
    """scalar register numbers an instruction line mentions (s12, s[12:15])"""
    regs = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", text):
        regs.add(int(m.group(1)))
    return regs


def check(asm_text, min_loads=12):
    lines = asm_text.splitlines()
    loads = [i for i, ln in enumerate(lines) if "s_load_dwordx16" in ln]
    assert len(loads) >= min_loads, "only %d s_load_dwordx16 found" % len(loads)      # every instantiation of the frame kernel: pilot + pass loop
    for i in loads:
        m = re.search(r"s_load_dwordx16\s+s\[(\d+):(\d+)\]", lines[i])
        assert m, lines[i]
        mine = set(range(int(m.group(1)), int(m.group(2)) + 1))
        for j in range(i + 1, min(i + 400, len(lines))):
            ln = lines[j].split(";")[0].strip()
            if not ln or ln.startswith("."):
                continue
            if ln.startswith("ds_read2_b64") and "s_waitcnt lgkmcnt(0)" in lines[j + 1]:
                break                                         # the column's read + the wait (one asm statement)
            # (s_cbranch_execz only skips a masked region when no lane is active: never in this kernel, whose wavefronts are whole)
            assert ln.startswith("s_cbranch_execz") or not (ln.startswith("s_cbranch") or ln.startswith("s_branch") or ln.startswith("s_endpgm")), (i, j, ln)
            assert not (_sgprs(ln) & mine), "line %d touches s[%d:%d] before the wait: %s" % (j, min(mine), max(mine), ln)
        else:
            raise AssertionError("no wait found after line %d" % i)
    return len(loads)


if __name__ == "__main__":
    n = check(sys.stdin.read())
    print("check_sload: %d column-coefficient loads, none touched before its wait" % n)
