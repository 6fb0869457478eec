#!/usr/bin/env python3
"""tools/pin_fdct.py -- the ONE command that moves this repository's MDEC parity from "unpinned at the FDCT" to "pinned".

psxavenc takes its 8x8 forward DCT from FFmpeg (avcodec_dct_alloc / avcodec_dct_init / AVDCT.fdct, psxavenc/mdec.c:524,548,640);
libavcodec is neither in the reference tree nor in the build image, so oracle/mdec_oracle.c restates the routine the reference's release
configuration selects (ff_jpeg_fdct_islow_8) and the GPU kernel is held to that restatement.  Given an FFmpeg installation this script
builds tools/check_fdct_vs_ffmpeg.c against it, runs the real AVDCT.fdct on 200 000 seeded blocks, diffs it against (1) the oracle's
FDCT and (2) -- on an MI355X box with libpsxav_hip.so built -- the device FDCT (psxhip_mdec_fdct_host), and prints ONE line:

    PIN_FDCT PASS libavcodec=<ident> oracle=identical device=identical blocks=200000 sha256=<of AVDCT's output vector>

Run it against the reference's own FFmpeg: version 8.0.1 configured as .github/scripts/build.sh:36-56 does (--disable-mmx among others:
with MMX/SSE2 enabled, x86-64 builds select ff_fdct_sse2 for the default dct_algo, whose low bits differ -- the reference's own
platform dependence).  `--algo islow` forces FF_DCT_INT on any build.

    python tools/pin_fdct.py --ffmpeg-prefix /opt/ffmpeg-8.0.1-nommx          # headers in <prefix>/include, libraries in <prefix>/lib
    python tools/pin_fdct.py                                                # a system FFmpeg (default include / library paths)

Exit status: 0 PASS, 1 FAIL (some coefficient differs), 2 UNAVAILABLE (no FFmpeg to build against: the situation of the build image)."""
import argparse
import hashlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ffmpeg-prefix", default=None, help="installation prefix of the FFmpeg to pin against (default: the compiler's default paths)")
    ap.add_argument("--algo", choices=["auto", "islow"], default="auto", help="auto = AVDCT's default dct_algo, as psxavenc uses it; islow = force FF_DCT_INT")
    ap.add_argument("--blocks", type=int, default=200000)
    ap.add_argument("--no-device", action="store_true", help="oracle only (no MI355X / libpsxav_hip.so needed)")
    ap.add_argument("--cc", default=os.environ.get("CC", "gcc"))
    args = ap.parse_args()
    so = os.path.join(ROOT, "psxavenc_amd", "libpsxav_hip.so")
    device = not args.no_device and os.path.exists(so) and os.path.exists("/dev/kfd")
    with tempfile.TemporaryDirectory() as tmp:
        exe, vec = os.path.join(tmp, "check_fdct"), os.path.join(tmp, "avdct.vec")
        cmd = [args.cc, "-O2", "-I", os.path.join(ROOT, "oracle"), "-I", os.path.join(ROOT, "include")]
        if args.ffmpeg_prefix:
            cmd += ["-I", os.path.join(args.ffmpeg_prefix, "include"), "-L", os.path.join(args.ffmpeg_prefix, "lib"),
                    "-Wl,-rpath," + os.path.join(args.ffmpeg_prefix, "lib")]
        if device:
            cmd += ["-DWITH_DEVICE"]
        cmd += [os.path.join(ROOT, "tools", "check_fdct_vs_ffmpeg.c"), os.path.join(ROOT, "oracle", "mdec_oracle.c"), os.path.join(ROOT, "oracle", "mdec_decode.c")]
        if device:
            cmd += ["-L", os.path.dirname(so), "-lpsxav_hip", "-Wl,-rpath," + os.path.dirname(so)]
        cmd += ["-lavcodec", "-lavutil", "-lm", "-lpthread", "-o", exe]      # (static FFmpeg builds -- the reference's are -- want their dependencies named)
        b = subprocess.run(cmd, capture_output=True, text=True)
        if b.returncode != 0:
            last = (b.stderr.strip().splitlines() or ["?"])[-1]
            print("PIN_FDCT UNAVAILABLE no FFmpeg to build against (%s) -- MDEC parity stays unpinned at the FDCT" % last[:160])
            return 2
        r = subprocess.run([exe, args.algo, str(args.blocks), "0x9E3779B9", vec], capture_output=True, text=True)
        sys.stderr.write(r.stdout)
        ident = oracle = dev = "?"
        for ln in r.stdout.splitlines():
            if ln.startswith("libavcodec "):
                ident = ln[len("libavcodec "):].split(", configuration")[0].replace(" ", "_")
            if "vs oracle restatement" in ln:
                oracle = "identical" if "-> PINNED" in ln else ln.split(":")[1].strip().split(",")[0].replace(" ", "_")
            if "vs device FDCT" in ln:
                dev = "identical" if "-> PINNED" in ln else ln.split(":")[-1].strip().split("->")[0].strip().replace(" ", "_")
        if not device:
            dev = "skipped"
        sha = hashlib.sha256(open(vec, "rb").read()).hexdigest() if os.path.exists(vec) else "?"
        ok = r.returncode == 0 and oracle == "identical" and dev in ("identical", "skipped")
        print("PIN_FDCT %s libavcodec=%s algo=%s oracle=%s device=%s blocks=%d sha256=%s" % ("PASS" if ok else "FAIL", ident, args.algo, oracle, dev, args.blocks, sha))
        return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
