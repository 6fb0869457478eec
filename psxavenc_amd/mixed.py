"""Scene-structured synthetic video for the MDEC path: a frame sequence that is NOT the friendliest point of the space.

The reference's input is decoded video (psxavenc/decoding.c:408-475): scenes of similar frames, cuts between them, flat areas,
hard edges; the rate-control loop (mdec.c:663-723) lands on a different quant scale per scene.  `plan()` is a pure, integer-only
function of (seed, n): runs of 5..30 frames share a noise amplitude 2..40 and a generator seed (frames inside a run drift like the
uniform workload's do), about one frame in twenty is a hand-made special -- flat fields, hard edges at several contrasts, sparse
full-contrast blocks (escape codes) -- chosen so that EVERY frame fits 8192 bytes at 320x240 (the reference asserts otherwise,
mdec.c:723).  Any rank can build any frame range: entry i of the plan says everything about frame i.

`frames_device` builds the frames in HBM (psxhip_synth_frames_device per run + the specials uploaded); `frames_host(oracle_lib, ...)`
builds the same bytes with the CPU twin (oracle/synth.c) for the parity tests.
"""
import numpy as np

N_SPECIAL = 8


def special_frame(kind, w, h):
    """hand-made NV21 frames (chroma flat 128); every one fits 8192 bytes at 320x240 (scales 1, 1, 1, 11, 5, 1, 53, 1)"""
    n = w * h
    yy, xx = np.mgrid[0:h, 0:w]
    kind = kind % N_SPECIAL
    if kind == 0:
        y = np.full((h, w), 128)                                                           # flat mid-grey
    elif kind == 1:
        y = np.full((h, w), 3)                                                             # flat near-black (v3 DC ties)
    elif kind == 2:
        y = 96 + (((yy // 8 + xx // 8) & 1) * 64)                                          # 8x8 checkerboard, moderate contrast: DC only
    elif kind == 3:
        y = 112 + ((xx // 5) & 1) * 32                                                     # vertical bars every 5 px, low contrast
    elif kind == 4:
        y = np.where(yy < h // 4, ((xx // 5) & 1) * 255, 128)                              # a quarter of the picture full-contrast bars (escapes)
    elif kind == 5:
        y = np.where(((xx // 16) % 5 == 0) & ((yy // 16) % 3 == 0), ((xx // 3 + yy // 2) & 1) * 255, 100)   # sparse full-contrast macroblocks
    elif kind == 6:
        y = np.where(((xx * 7 + yy * 13) % 97) < 6, 255, 20)                               # thin bright strokes on black: needs a very coarse scale
    else:
        y = np.where((yy // 16) % 4 == 0, ((xx // 5) & 1) * 255, 128)                      # every fourth macroblock row full-contrast bars
    f = np.full(n * 3 // 2, 128, np.uint8)
    f[:n] = np.clip(y, 0, 255).astype(np.uint8).ravel()
    return f


def plan(seed, n):
    """[(kind, a, b)] per frame: ('synth', amp, run_seed) -- frame i = synth frame (seed run_seed, index i, noise amp) -- or
    ('special', k, 0).  Also returns the list of runs [(first, count, amp, run_seed)] for batched generation."""
    state = [(int(seed) * 2654435761 + 12345) & 0xFFFFFFFF]

    def nxt():
        state[0] = (state[0] * 1664525 + 1013904223) & 0xFFFFFFFF
        return state[0] >> 8

    frames, runs = [], []
    while len(frames) < n:
        length = 5 + nxt() % 26
        amp = 2 + nxt() % 39
        rseed = nxt() & 0xFFFFF
        first = len(frames)
        for _ in range(length):
            if len(frames) >= n:
                break
            if nxt() % 20 == 0:
                frames.append(("special", int(nxt() % N_SPECIAL), 0))
            else:
                frames.append(("synth", int(amp), int(rseed)))
        runs.append((first, len(frames) - first, int(amp), int(rseed)))
    return frames, runs


def frames_host(oracle_lib, w, h, seed, first, n):
    """frames first .. first + n - 1 of the sequence `seed`, built with the CPU twin of the generator"""
    fr, _ = plan(seed, first + n)
    out = np.empty((n, w * h * 3 // 2), np.uint8)
    for i in range(n):
        kind, a, b = fr[first + i]
        out[i] = special_frame(a, w, h) if kind == "special" else oracle_lib.synth_frames(w, h, 1, seed=b, amp=a, first=first + i)[0]
    return out


def frames_device(w, h, seed, first, n, device=0):
    """the same frames in HBM: (n, w*h*3/2) uint8 CUDA tensor"""
    import torch
    from . import synth
    fr, runs = plan(seed, first + n)
    dev = torch.device("cuda", device)
    out = torch.empty((n, w * h * 3 // 2), dtype=torch.uint8, device=dev)
    for (r0, cnt, amp, rseed) in runs:
        a, b = max(r0, first), min(r0 + cnt, first + n)
        if a < b:
            synth.frames_device(w, h, rseed, a, b - a, amp, device=device, out=out[a - first:b - first])
    cache = {}
    for i in range(n):
        kind, k, _ = fr[first + i]
        if kind == "special":
            if k not in cache:
                cache[k] = torch.from_numpy(special_frame(k, w, h)).to(dev)
            out[i].copy_(cache[k])
    return out
