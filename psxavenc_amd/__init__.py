"""psxavenc_amd -- MI355X-native implementation of psxavenc's MDEC BS encode path and SPU/XA ADPCM
search, behind the reference's own function surfaces (see DESIGN.md / INTEGRATION.md).

The product is ``libpsxav_hip.so`` (C ABI, include/*.h).  This package is the thin Python mirror used
by tests and bench.py: device memory and streams come from PyTorch, everything else from the library.
"""
from ._lib import LIB_PATH, PsxHipError, lib  # noqa: F401

BS_CODEC_V2, BS_CODEC_V3, BS_CODEC_V3DC = 0, 1, 2   # bs_codec_t, psxavenc/args.h:61-65
