"""ADPCM parity soak (not part of the pytest run): SPU and XA streams of every synthetic signal kind, random lengths / gains /
pitches / formats, through the HIP host entry points (serial and chunked paths), every byte against the oracle
(oracle/adpcm_oracle.c, itself pinned to the reference's own code in tests/test_adpcm_oracle.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from psxavenc_amd import adpcm

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = total_units = 0
t0 = time.time()
for r in range(rounds):
    kind = r % 6
    seed = int(rng.integers(1, 1 << 30))
    gain_shift = int(rng.integers(0, 5))
    if r % 2 == 0:
        # SPU: one stream, mono; long enough for the chunked path every other time
        n = int(rng.integers(28, 28 * (40000 if r % 4 == 0 else 600)))
        pcm = (O.synth_pcm(seed, r, int(rng.integers(0, 1 << 20)), n, kind) >> gain_shift).astype(np.int16)
        want, _ = O.spu_encode(pcm)
        got = adpcm.spu_encode_streams(pcm[None, :], sample_count=n)[0][:want.size]
        ok = np.array_equal(got, want)
        total_units += n // 28
        print("round %2d SPU  kind %d n %8d >>%d %s" % (r, kind, n, gain_shift, "ok" if ok else "MISMATCH"), flush=True)
    else:
        stereo, bits, fmt = bool(rng.integers(0, 2)), int(rng.choice([4, 8])), int(rng.integers(0, 2))
        freq, fno, cno = int(rng.choice([37800, 18900])), int(rng.integers(0, 4)), int(rng.integers(0, 8))
        s = adpcm.XaSettings(fmt, stereo, freq, bits, fno, cno)
        so = O.XaSettings(fmt, int(stereo), freq, bits, fno, cno)
        ch = 2 if stereo else 1
        sectors = int(rng.integers(1, 700 if r % 4 == 1 else 12))
        n = sectors * adpcm.xa_get_samples_per_sector(s) - int(rng.integers(0, 50))
        pcm = np.zeros(n * ch + 8064, np.int16)      # the reference reads zero padding past the end of the data (decoding.c:497-503)
        for c in range(ch):
            pcm[c:n * ch:ch] = (O.synth_pcm(seed, c, 7, n, kind) >> gain_shift).astype(np.int16)
        lba = int(rng.integers(0, 100000))
        want, _ = O.xa_encode(so, pcm, n, lba=lba)
        got = adpcm.xa_encode_streams(s, pcm[None, :], n, lbas=np.array([lba], np.int32))[0]
        ok = got.size == want.size and np.array_equal(got, want)
        total_units += sectors * 18 * (8 if bits == 4 else 4)
        print("round %2d XA   kind %d %s %d-bit fmt %d sectors %4d %s" % (r, kind, "stereo" if stereo else "mono  ", bits, fmt, sectors, "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
print("adpcm soak: %d rounds, %d sound units, %d mismatching rounds, %.0f s" % (rounds, total_units, bad, time.time() - t0))
