#!/usr/bin/env python3
"""Generate tests/golden/spufile_ref.npz: SPU / VAG / SPUI / VAGI files produced by driving the REFERENCE's own
psx_audio_spu_encode (oracle/_ref/libpsxav_ref.so, libpsxav/adpcm.c compiled unchanged) through the container framing
of psxavenc/filefmt.c:95-162,212-389 -- the framing loops are restated here call for call (28-sample calls for spu/vag,
one call per channel per chunk for spui/vagi, decoder end-of-input as decoding.c:510-534 defines it).

Inputs are regenerated at test time from the recorded parameters (oracle/synth.c is a pure function); the fixture holds
the file bytes (small cases) or their SHA-256.  Run here (where /root/reference exists):
    python tests/golden/make_spufile_golden.py
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

SPU, VAG, SPUI, VAGI = 2, 3, 4, 5
LOOP_REPEAT, LOOP_START, LOOP_TRAP = 3, 6, 5


def vag_header(fmt, freq, channels, interleave, loop_point, no_dummy, size_per_channel, name):
    h = np.zeros(0x30, np.uint8)
    h[0:3] = list(b"VAG")
    h[3] = ord("i") if fmt == VAGI else ord("p")
    h[4:8] = [0, 0, 0, 0x20]
    if fmt == VAGI:
        h[8:12] = [(interleave >> (8 * k)) & 0xFF for k in range(4)]
    h[0x0C:0x10] = [(size_per_channel >> s) & 0xFF for s in (24, 16, 8, 0)]
    h[0x10:0x14] = [(freq >> s) & 0xFF for s in (24, 16, 8, 0)]
    if fmt == VAGI and loop_point >= 0:
        lsb = (loop_point * freq) // (28 * 1000)
        if not no_dummy:
            lsb += 1
        lp = lsb * 16
        h[0x14:0x18] = [(lp >> s) & 0xFF for s in (24, 16, 8, 0)]
    h[0x1E] = channels
    nm = name.encode()[:16]
    h[0x20:0x20 + len(nm)] = list(nm)
    return h


def ref_spu_file(fmt, pcm, freq=44100, alignment=64, loop_point=-1, enable_loop=False, no_dummy=False, name="out.vag"):
    """encode_file_spu, filefmt.c:212-293 (mono)"""
    R = O.ref()
    st = O.RefChan()
    out = []
    block_count = 0
    if not no_dummy:
        out.append(np.zeros(16, np.uint8))
        block_count += 1
    loop_start_block = -1
    if loop_point >= 0:
        loop_start_block = block_count + (loop_point * freq) // (28 * 1000)
    remaining = pcm.size
    pos = 0
    padded = np.concatenate([pcm, np.zeros(64, np.int16)])
    while remaining > 0:                                  # ensure_av_data: true while the buffer is not empty
        end_of_input = remaining <= 28                    # decoding.c:517-529
        n = min(remaining, 28)
        block = np.zeros(16, np.uint8)
        ln = R.psx_audio_spu_encode(C.byref(st), O.ptr(padded[pos:], O.i16p), n, 1, O.ptr(block, O.u8p))
        assert ln == 16
        if block_count == loop_start_block:
            block[1] |= LOOP_START
        if enable_loop and end_of_input:
            block[1] |= LOOP_REPEAT
        out.append(block)
        pos += n
        remaining -= n
        block_count += 1
    if not enable_loop:
        b = np.zeros(16, np.uint8)
        b[1] = LOOP_TRAP
        out.append(b)
        block_count += 1
    data = np.concatenate(out) if out else np.zeros(0, np.uint8)
    overflow = (block_count * 16) % alignment
    if overflow:
        data = np.concatenate([data, np.zeros(alignment - overflow, np.uint8)])
    if fmt == VAG:
        data = np.concatenate([vag_header(fmt, freq, 1, 0, loop_point, no_dummy, block_count * 16, name), data])
    return data


def ref_spui_file(fmt, pcm, channels, freq=44100, interleave=2048, alignment=2048, loop_point=-1, enable_loop=False,
                  no_dummy=False, name="out.vag"):
    """encode_file_spui, filefmt.c:295-389; pcm interleaved"""
    R = O.ref()
    spc = interleave // 16 * 28
    chunk_size = interleave * channels + alignment - 1
    chunk_size -= chunk_size % alignment
    header_size = 0x30 + alignment - 1
    header_size -= header_size % alignment
    states = (O.RefChan * channels)()
    chunks = []
    n = pcm.size // channels
    remaining, pos, chunk_count = n, 0, 0
    padded = np.concatenate([pcm, np.zeros(64 * channels, np.int16)])
    while remaining > 0:
        end_of_input = remaining <= spc
        samples_length = min(remaining, spc)
        chunk = np.zeros(chunk_size, np.uint8)
        off = 0
        if chunk_count == 0 and not no_dummy:
            off += 16
            samples_length -= 28
        assert samples_length >= 0, "the reference's behaviour for < 28 samples with a dummy block is undefined"
        for ch in range(channels):
            base = off + ch * interleave
            buf = np.zeros(((samples_length + 27) // 28) * 16 + 16, np.uint8)
            ln = R.psx_audio_spu_encode(C.byref(states[ch]), O.ptr(padded[pos * channels + ch:], O.i16p), samples_length, channels,
                                        O.ptr(buf, O.u8p))
            chunk[base:base + ln] = buf[:ln]
            if ln > 0:
                last = base + ln - 16
                if enable_loop or (end_of_input and loop_point >= 0):
                    chunk[last + 1] = LOOP_REPEAT
                elif end_of_input:
                    chunk[last:last + 16] = 0
                    chunk[last + 1] = LOOP_TRAP
        pos += samples_length
        remaining -= samples_length
        chunks.append(chunk)
        chunk_count += 1
    data = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    if fmt == VAGI:
        head = np.zeros(header_size, np.uint8)
        head[:0x30] = vag_header(fmt, freq, channels, interleave, loop_point, no_dummy, chunk_count * interleave, name)
        data = np.concatenate([head, data])
    return data


def interleaved_pcm(seed, kind, n, channels):
    pcm = np.zeros(n * channels, np.int16)
    for c in range(channels):
        pcm[c::channels] = O.synth_pcm(seed, c, 0, n, kind)
    return pcm


# (key, fmt, kwargs, input recipe) -- test_gpu_adpcm.py::test_spu_file_framing_vs_reference_golden replays these
CASES = []
for kind, n in ((0, 22050), (1, 28 * 40), (2, 28 * 40 + 5), (3, 1000), (4, 29)):
    for fmt in (SPU, VAG):
        for opts in ({}, {"loop_point": 10, "enable_loop": True}, {"no_dummy": True, "alignment": 2048}, {"loop_point": 0}):
            CASES.append(("f%d_k%d_n%d_%s" % (fmt, kind, n, "_".join("%s%s" % kv for kv in sorted(opts.items())) or "default"),
                          fmt, dict(opts), dict(seed=31, kind=kind, n=n, channels=1)))
for kind, n, channels in ((0, 3584 * 3 + 100, 2), (1, 3584, 2), (2, 3584 * 2, 1), (3, 5000, 4), (4, 3584 * 2 - 28, 2), (0, 896 * 5 + 7, 2)):
    for fmt in (SPUI, VAGI):
        for opts in ({}, {"loop_point": 50}, {"enable_loop": True, "no_dummy": True}, {"interleave": 512, "alignment": 64}):
            CASES.append(("f%d_k%d_n%d_c%d_%s" % (fmt, kind, n, channels, "_".join("%s%s" % kv for kv in sorted(opts.items())) or "default"),
                          fmt, dict(opts), dict(seed=32, kind=kind, n=n, channels=channels)))


def sine_spu_config():
    """SURVEY 8(d) config 1 `spu`: 22050-sample 440 Hz sine, defaults -> 16 + 788*16 + 16 = 12640 bytes padded to 12672"""
    i = np.arange(22050)
    return np.rint(16384 * np.sin(2 * np.pi * 440 * i / 22050)).astype(np.int16)


def main():
    assert O.ref() is not None, "oracle/_ref/libpsxav_ref.so missing: run make -C oracle with /root/reference present"
    out = {}
    keys = []
    for key, fmt, opts, rec in CASES:
        pcm = interleaved_pcm(rec["seed"], rec["kind"], rec["n"], rec["channels"])
        if fmt in (SPU, VAG):
            data = ref_spu_file(fmt, pcm, **opts)
        else:
            data = ref_spui_file(fmt, pcm, rec["channels"], **opts)
        out[key + "_size"] = np.array([data.size], np.int64)
        out[key + "_sha"] = np.frombuffer(hashlib.sha256(data.tobytes()).digest(), np.uint8)
        if data.size <= 4096:
            out[key + "_bytes"] = data
        keys.append(key)
    sine = ref_spu_file(SPU, sine_spu_config())
    assert sine.size == 12672 and sine[16:20].tolist() == [0x24, 0x00, 0x70, 0x13], (sine.size, sine[16:20])
    out["config_spu_sine"] = sine
    out["keys"] = np.array(keys)
    np.savez_compressed(os.path.join(HERE, "spufile_ref.npz"), **out)
    print("wrote spufile_ref.npz: %d cases" % len(keys))


if __name__ == "__main__":
    main()
