"""The BS VLC / quant / zig-zag facts: generated headers are current, internally consistent, and -- when
the reference tree is present -- identical to what psxavenc/mdec.c lists (mdec.c:39-222)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_tables as G  # noqa: E402


def test_generated_headers_are_current():
    assert open(os.path.join(ROOT, "oracle/bs_vlc_tables.h")).read() == G.gen_oracle()
    assert open(os.path.join(ROOT, "psxavenc_amd/csrc/bs_vlc_lut.h")).read() == G.gen_device()


def test_ac_code_book_is_prefix_free_and_complete():
    t = G.ac_table()
    assert len(t) == 111
    words = [c + s for c in t.values() for s in "01"] + ["10", "000001"]    # + end-of-block, escape prefix
    for i, a in enumerate(words):
        for j, b in enumerate(words):
            if i != j:
                assert not b.startswith(a), (a, b)
    assert max(len(c) for c in t.values()) + 1 < G.ESCAPE_BITS


def test_dc_code_books_prefix_free():
    for zero, pre in ((G.DC_CHROMA_ZERO, G.DC_CHROMA_PREFIX), (G.DC_LUMA_ZERO, G.DC_LUMA_PREFIX)):
        words = [zero] + pre
        for i, a in enumerate(words):
            for j, b in enumerate(words):
                if i != j:
                    assert not b.startswith(a)


def test_zigzag_is_a_permutation_with_known_head():
    z = G.zagzig()
    assert sorted(z) == list(range(64))
    assert z[:10] == [0, 1, 8, 16, 9, 2, 3, 10, 17, 24]


def test_oracle_luts_match_generator(oracle):
    L = oracle.lib()
    t = G.ac_table()
    for run in range(64):
        for level in list(range(-512, -500)) + list(range(-45, 46)) + list(range(500, 511)):
            if level == 0:
                continue
            w = L.orc_mdec_ac_code(run, level)
            bits, val = w >> 24, w & 0xFFFFFF
            key = (run, abs(level))
            if key in t:
                assert bits == len(t[key]) + 1
                assert val == (int(t[key], 2) << 1) | (1 if level < 0 else 0)
            else:
                assert bits == 22 and val == (1 << 16) | (run << 10) | (level & 0x3FF)


# ---------------------------------------------------------------- against the reference text
def _ref_text(reference_root):
    return open(os.path.join(reference_root, "psxavenc/mdec.c")).read()


def test_ac_book_equals_reference(reference_root):
    src = _ref_text(reference_root)
    ents = re.findall(r"\{\s*(\d+),\s*(0x[0-9A-Fa-f]+),\s*AC_PAIR\(\s*(\d+),\s*(\d+)\)\}", src)
    assert len(ents) == 111
    ref = {(int(r), int(l)): format(int(v, 16), "0%db" % int(b)) for b, v, r, l in ents}
    assert ref == G.ac_table()


def test_dc_books_quant_zigzag_equal_reference(reference_root):
    src = _ref_text(reference_root)

    def tree(name):
        body = re.search(name + r"\[\] = \{(.*?)\};", src, re.S).group(1)
        return [(int(b), int(v, 16), int(d)) for b, v, d in re.findall(r"\{(\d+),\s*(0x[0-9A-Fa-f]+),\s*(\d+)\}", body)]

    for name, mine in (("dc_c_huffman_tree", G.DC_CHROMA_PREFIX), ("dc_y_huffman_tree", G.DC_LUMA_PREFIX)):
        t = tree(name)
        assert [d for _, _, d in t] == list(range(8))
        assert [format(v, "0%db" % b) for b, v, _ in t] == mine

    def array(name):
        body = re.search(r"static const uint8_t " + name + r"\[8\*8\] = \{(.*?)\};", src, re.S).group(1)
        return [int(x) for x in re.findall(r"\d+", body)]

    assert array("quant_dec") == G.QUANT
    assert array("dct_zagzig_table") == G.zagzig()
