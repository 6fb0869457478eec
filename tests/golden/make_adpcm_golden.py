#!/usr/bin/env python3
"""Generate tests/golden/adpcm_ref.npz from the REFERENCE's own libpsxav (oracle/_ref/libpsxav_ref.so,
built unchanged from /root/reference/libpsxav/{adpcm,cdrom}.c by oracle/Makefile).

Inputs are regenerated at test time from oracle/synth.c (pure functions of the recorded parameters), the
fixture stores the reference's output bytes (small cases) or their SHA-256 (long cases) plus final states.
Run here (where /root/reference exists):  python tests/golden/make_adpcm_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def stereo_pad(kind, n, seed, pad=4032):
    pcm = np.zeros((n + pad) * 2, np.int16)
    pcm[0:2 * n:2] = O.synth_pcm(seed, 0, 0, n, kind)
    pcm[1:2 * n:2] = O.synth_pcm(seed, 1, 0, n, kind)
    return pcm


def mono_pad(kind, n, seed, pad=4032):
    pcm = np.zeros(n + pad, np.int16)
    pcm[:n] = O.synth_pcm(seed, 0, 0, n, kind)
    return pcm


def main():
    assert O.ref() is not None, "oracle/_ref/libpsxav_ref.so missing: run make -C oracle with /root/reference present"
    out = {}
    cases = []
    # SPU: every signal class, ragged tails (n % 28 != 0), one long case
    for kind in range(6):
        for n in (28, 29, 280, 28 * 100 + 13):
            pcm = O.synth_pcm(11, kind, 0, n, kind)
            data, st = O.ref_spu_encode(pcm)
            key = "spu_k%d_n%d" % (kind, n)
            out[key] = data
            out[key + "_state"] = np.array([st.prev1, st.prev2], np.int32)
            cases.append(key)
    pcm = O.synth_pcm(11, 0, 0, 28 * 20000, 0)
    data, st = O.ref_spu_encode(pcm)
    out["spu_long_sha"] = np.frombuffer(hashlib.sha256(data.tobytes()).digest(), np.uint8)
    out["spu_long_state"] = np.array([st.prev1, st.prev2], np.int32)
    # SPU with pitch 2 (interleaved read)
    pcm2 = stereo_pad(0, 28 * 50, 5, pad=0)
    data, st = O.ref_spu_encode(pcm2, pitch=2, n=28 * 50)
    out["spu_pitch2"] = data
    # SPU simple (trap block / loop flags): the SURVEY 8(c) sine
    i = np.arange(22050)
    sine = np.rint(16384 * np.sin(2 * np.pi * 440 * i / 22050)).astype(np.int16)
    for loop in (-1, 280):
        buf = np.zeros(20000, np.uint8)
        ln = O.ref().psx_audio_spu_encode_simple(O.ptr(sine, O.i16p), sine.size, O.ptr(buf, O.u8p), loop)
        out["spu_simple_sine_loop%d_sha" % loop] = np.frombuffer(hashlib.sha256(buf[:ln].tobytes()).digest(), np.uint8)
        out["spu_simple_sine_loop%d_len" % loop] = np.array([ln], np.int32)
    # XA: format x stereo x bits x freq, several lengths incl. short tails and multi-sector
    for fmt in (0, 1):
        for stereo in (0, 1):
            for bits in (4, 8):
                for freq in (37800, 18900):
                    for kind, n in ((0, 5000), (5, 300), (2, 2016), (3, 100), (1, 4033)):
                        s = O.XaSettings(fmt, stereo, freq, bits, 3, 7)
                        pcm = stereo_pad(kind, n, 21) if stereo else mono_pad(kind, n, 21)
                        data, st = O.ref_xa_encode(s, pcm, n, lba=1234)
                        key = "xa_f%d_s%d_b%d_q%d_k%d_n%d" % (fmt, stereo, bits, freq, kind, n)
                        out[key + "_sha"] = np.frombuffer(hashlib.sha256(data.tobytes()).digest(), np.uint8)
                        out[key + "_len"] = np.array([data.size], np.int32)
                        out[key + "_state"] = np.array([st.left.prev1, st.left.prev2, st.right.prev1, st.right.prev2], np.int32)
    # one full XACD stereo sector kept verbatim
    s = O.XaSettings(1, 1, 37800, 4, 1, 0)
    pcm = stereo_pad(0, 2016, 9)
    data, _ = O.ref_xa_encode(s, pcm, 2016, lba=0)
    out["xa_full_sector"] = data
    np.savez_compressed(os.path.join(HERE, "adpcm_ref.npz"), **out)
    print("wrote adpcm_ref.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
