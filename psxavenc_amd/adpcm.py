"""SPU / XA ADPCM encoder -- Python mirror of include/psxav_audio.h / psxav_hip.h.

Reference surface: ``psx_audio_spu_encode``, ``psx_audio_xa_encode`` (+ ``_simple``, ``_finalize`` and the
size helpers), libpsxav/libpsxav.h:73-101.  State objects carry ``prev1 / prev2`` exactly like
``psx_audio_encoder_channel_state_t`` so calls can be chained (28 samples per call for ``-t spu``, one
sector per call for xa/str, filefmt.c:184,243).  Batched forms encode many independent streams per launch.
"""
import ctypes as C

import numpy as np

from . import _lib

PSX_AUDIO_XA_FORMAT_XA, PSX_AUDIO_XA_FORMAT_XACD = 0, 1           # libpsxav.h:39-42
PSX_AUDIO_XA_FREQ_SINGLE, PSX_AUDIO_XA_FREQ_DOUBLE = 18900, 37800  # libpsxav.h:34-37
PSX_AUDIO_SPU_BLOCK_SIZE, PSX_AUDIO_SPU_SAMPLES_PER_BLOCK = 16, 28
PSX_AUDIO_SPU_LOOP_END, PSX_AUDIO_SPU_LOOP_REPEAT, PSX_AUDIO_SPU_LOOP_START, PSX_AUDIO_SPU_LOOP_TRAP = 1, 3, 6, 5
RECORD_BYTES = 32          # 8-bit codes (and the upper bound)


def record_bytes(bits):
    """bytes between the records of consecutive unit indices (PSXHIP_ADPCM_RECORD_SIZE): 4-bit material keeps 16-byte records in
    the layout of an SPU block"""
    return 16 if bits == 4 else RECORD_BYTES


def _bind():
    L = _lib.lib()
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.psxhip_spu_encode_streams_host.argtypes = [i32, vp, i32, i64, i32, i32, vp, vp, i64]
    L.psxhip_xa_encode_streams_host.argtypes = [i32, i32, i32, i32, i32, i32, i32, vp, i32, i64, i32, vp, vp, vp, i64, i32]
    L.psxhip_adpcm_encode_chains_device.argtypes = [i32, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    L.psxhip_spu_pack_device.argtypes = [i32, vp, i32, vp, vp]
    L.psxhip_xa_assemble_device.argtypes = [i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    return L


class XaSettings:
    """psx_audio_xa_settings_t (libpsxav.h:44-51)"""

    def __init__(self, format=PSX_AUDIO_XA_FORMAT_XA, stereo=True, frequency=37800, bits_per_sample=4, file_number=0,
                 channel_number=0):
        self.format, self.stereo, self.frequency = int(format), bool(stereo), int(frequency)
        self.bits_per_sample, self.file_number, self.channel_number = int(bits_per_sample), int(file_number), int(channel_number)


# ---- size helpers (adpcm.c:235-260): pure arithmetic, kept in Python for the mirror ---------------
def xa_get_buffer_size_per_sector(s):
    return 2336 if s.format == PSX_AUDIO_XA_FORMAT_XA else 2352


def xa_get_samples_per_sector(s):
    return ((112 if s.bits_per_sample == 8 else 224) >> (1 if s.stereo else 0)) * 18


def xa_get_sector_interleave(s):
    v = 2 if s.stereo else 4
    if s.frequency == PSX_AUDIO_XA_FREQ_SINGLE:
        v <<= 1
    if s.bits_per_sample == 4:
        v <<= 1
    return v


def xa_get_buffer_size(s, sample_count):
    sps = xa_get_samples_per_sector(s)
    return ((sample_count + sps - 1) // sps) * xa_get_buffer_size_per_sector(s)


def spu_get_buffer_size(sample_count):
    return ((sample_count + 27) // 28) << 4


# ---- batched host paths -------------------------------------------------------------------------
def spu_encode_streams(samples, pitch=1, states=None, sample_count=None, device=0):
    """samples: int16 (n_streams, >= sample_count*pitch).  Returns (n_streams, 16*ceil(n/28)) uint8;
    `states` (n_streams, 2) int32 [prev1, prev2] is updated in place when given."""
    L = _bind()
    samples = np.ascontiguousarray(samples, dtype=np.int16)
    assert samples.ndim == 2
    n_streams = samples.shape[0]
    n = samples.shape[1] // pitch if sample_count is None else sample_count
    st = np.zeros((n_streams, 2), np.int32) if states is None else states
    assert st.dtype == np.int32 and st.shape == (n_streams, 2) and st.flags.c_contiguous
    out = np.zeros((n_streams, spu_get_buffer_size(n)), np.uint8)
    rc = L.psxhip_spu_encode_streams_host(device, samples.ctypes.data, n_streams, samples.shape[1], pitch, n, st.ctypes.data,
                                          out.ctypes.data, out.shape[1])
    if rc < 0:
        _lib.check(rc)
    return out


def xa_encode_streams(settings, samples, sample_count, lbas=None, states=None, finalize=False, device=0):
    """samples: int16 (n_streams, >= sample_count * channels), interleaved L,R when stereo.
    Returns (n_streams, sectors*sector_size) uint8.  states: (n_streams, 2, 2) int32 [[l1,l2],[r1,r2]]."""
    L = _bind()
    samples = np.ascontiguousarray(samples, dtype=np.int16)
    n_streams = samples.shape[0]
    ch = 2 if settings.stereo else 1
    assert samples.shape[1] >= sample_count * ch
    st = np.zeros((n_streams, 2, 2), np.int32) if states is None else states
    assert st.dtype == np.int32 and st.shape == (n_streams, 2, 2) and st.flags.c_contiguous
    st_dev = np.ascontiguousarray(st[:, :ch, :].reshape(n_streams * ch, 2))
    lb = np.zeros(n_streams, np.int32) if lbas is None else np.ascontiguousarray(lbas, dtype=np.int32)
    upg = 8 if settings.bits_per_sample == 4 else 4
    groups = (sample_count * ch + upg * 28 - 1) // (upg * 28)
    sectors = (groups + 17) // 18
    size = sectors * xa_get_buffer_size_per_sector(settings)
    out = np.zeros((n_streams, max(size, 1)), np.uint8)
    rc = L.psxhip_xa_encode_streams_host(device, settings.format, int(settings.stereo), settings.frequency,
                                         settings.bits_per_sample, settings.file_number, settings.channel_number,
                                         samples.ctypes.data, n_streams, samples.shape[1], sample_count, lb.ctypes.data,
                                         st_dev.ctypes.data, out.ctypes.data, out.shape[1], int(bool(finalize)))
    if rc < 0:
        _lib.check(rc)
    st[:, :ch, :] = st_dev.reshape(n_streams, ch, 2)
    return out[:, :size]


# ---- reference-shaped single-stream calls ---------------------------------------------------------
class ChannelState:
    """psx_audio_encoder_channel_state_t (libpsxav.h:53-57); qerr/mse are dead fields in the reference."""

    def __init__(self):
        self.prev1 = 0
        self.prev2 = 0


class EncoderState:
    """psx_audio_encoder_state_t (libpsxav.h:59-62)"""

    def __init__(self):
        self.left, self.right = ChannelState(), ChannelState()


def psx_audio_spu_encode(state, samples, sample_count, pitch=1, device=0):
    st = np.array([[state.prev1, state.prev2]], np.int32)
    samples = np.asarray(samples, dtype=np.int16).reshape(1, -1)
    out = spu_encode_streams(samples, pitch=pitch, states=st, sample_count=sample_count, device=device)
    state.prev1, state.prev2 = int(st[0, 0]), int(st[0, 1])
    return out[0]


def psx_audio_spu_encode_simple(samples, sample_count, loop_start, device=0):
    """adpcm.c:378-401: zero state, then the trailing LOOP_TRAP block or the loop flags."""
    out = psx_audio_spu_encode(ChannelState(), samples, sample_count, 1, device=device)
    if out.size >= 16:
        if loop_start < 0:
            trap = np.zeros(16, np.uint8)
            trap[1] = PSX_AUDIO_SPU_LOOP_TRAP
            out = np.concatenate([out, trap])
        else:
            out[out.size - 16 + 1] |= PSX_AUDIO_SPU_LOOP_REPEAT
            out[loop_start // 28 * 16 + 1] |= PSX_AUDIO_SPU_LOOP_START
    return out


def psx_audio_xa_encode(settings, state, samples, sample_count, lba, device=0):
    st = np.array([[[state.left.prev1, state.left.prev2], [state.right.prev1, state.right.prev2]]], np.int32)
    samples = np.asarray(samples, dtype=np.int16).reshape(1, -1)
    out = xa_encode_streams(settings, samples, sample_count, lbas=[lba], states=st, device=device)
    state.left.prev1, state.left.prev2 = int(st[0, 0, 0]), int(st[0, 0, 1])
    state.right.prev1, state.right.prev2 = int(st[0, 1, 0]), int(st[0, 1, 1])
    return out[0]


def psx_audio_xa_encode_finalize(settings, output):
    """adpcm.c:334-340: OR the EOF bit into the last sector's submode (both subheader copies)."""
    if output.size >= 2336:
        base = output.size - 2352      # may be -16 for a single .xa sector: only offsets >= 16 are touched
        output[base + 18] |= 0x80
        output[base + 20:base + 24] = output[base + 16:base + 20]
    return output


def psx_audio_xa_encode_simple(settings, samples, sample_count, lba, device=0):
    out = psx_audio_xa_encode(settings, EncoderState(), samples, sample_count, lba, device=device)
    return psx_audio_xa_encode_finalize(settings, out)


# ---- device-resident paths (torch tensors) -----------------------------------------------------------
CHAIN_DTYPE = np.dtype([("sample_offset", "<i8"), ("pitch", "<i4"), ("sample_limit", "<i4"), ("n_units", "<i4"),
                        ("unit_stride", "<i4")])


def make_chains(offsets, pitch, limits, n_units, unit_stride=1):
    """Host-side psxhip_adpcm_chain_t array (include/psxav_hip.h)."""
    offsets = np.asarray(offsets, dtype=np.int64)
    ch = np.zeros(offsets.size, dtype=CHAIN_DTYPE)
    ch["sample_offset"] = offsets
    ch["pitch"] = pitch
    ch["sample_limit"] = limits
    ch["n_units"] = n_units
    ch["unit_stride"] = unit_stride
    return ch


def encode_chains_device(d_samples, chains, unit_base, filter_count, bits, d_states=None, d_units=None, chunk_units=0,
                         warmup_units=16, max_passes=0):
    """Run the ADPCM search for `chains` (host CHAIN_DTYPE array) over int16 CUDA tensor `d_samples`.

    chunk_units == 0: one serial pass per chain (psxhip_adpcm_encode_chains_device, asynchronous).
    chunk_units  > 0: speculate-and-verify along time (psxhip_adpcm_encode_chains_chunked, synchronous).
    Returns (d_units (total_units, record_bytes(bits)) uint8, d_states (n_chains, 2) int32, verify_passes)."""
    import torch
    L = _bind()
    L.psxhip_adpcm_encode_chains_chunked.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    dev = d_samples.device
    chains = np.ascontiguousarray(chains, dtype=CHAIN_DTYPE)
    unit_base = np.ascontiguousarray(unit_base, dtype=np.int32)
    n = chains.size
    total = int((unit_base + (chains["n_units"] - 1) * chains["unit_stride"]).max()) + 1 if n else 0
    if d_states is None:
        d_states = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    if d_units is None:
        d_units = torch.zeros((max(total, 1), record_bytes(bits)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    if chunk_units > 0:
        rc = L.psxhip_adpcm_encode_chains_chunked(dev.index or 0, d_samples.data_ptr(), chains.ctypes.data, unit_base.ctypes.data,
                                                  n, filter_count, bits, d_states.data_ptr(), d_units.data_ptr(), chunk_units,
                                                  warmup_units, max_passes, st)
        if rc < 0:
            _lib.check(rc)
        return d_units, d_states, rc
    d_chains = torch.from_numpy(chains.view(np.uint8).reshape(n, CHAIN_DTYPE.itemsize).copy()).to(dev)
    d_base = torch.from_numpy(unit_base).to(dev)
    _lib.check(L.psxhip_adpcm_encode_chains_device(dev.index or 0, d_samples.data_ptr(), d_chains.data_ptr(), d_base.data_ptr(), n,
                                                   filter_count, bits, d_states.data_ptr(), d_units.data_ptr(), st))
    torch.cuda.current_stream(dev).synchronize()    # d_chains / d_base are temporaries
    return d_units, d_states, 0


def spu_pack_device(d_units, n_blocks):
    import torch
    L = _bind()
    out = torch.empty((n_blocks, 16), dtype=torch.uint8, device=d_units.device)
    st = torch.cuda.current_stream(d_units.device).cuda_stream
    _lib.check(L.psxhip_spu_pack_device(d_units.device.index or 0, d_units.data_ptr(), n_blocks, out.data_ptr(), st))
    return out


def xa_assemble_device(d_units, n_sectors, settings, first_lba=0, d_eof=None):
    import torch
    L = _bind()
    ssz = xa_get_buffer_size_per_sector(settings)
    out = torch.empty((n_sectors, ssz), dtype=torch.uint8, device=d_units.device)
    st = torch.cuda.current_stream(d_units.device).cuda_stream
    _lib.check(L.psxhip_xa_assemble_device(d_units.device.index or 0, d_units.data_ptr(), n_sectors, settings.format,
                                           int(settings.stereo), settings.frequency, settings.bits_per_sample,
                                           settings.file_number, settings.channel_number, first_lba,
                                           d_eof.data_ptr() if d_eof is not None else None, out.data_ptr(), st))
    return out


def pick_chunking(total_units, rows=5, n_cu=None):
    """(chunk_units, warmup_units) for speculate-and-verify: few verify passes vs enough chunks to fill the GPU; large jobs get
    the chunk length at which every wavefront slot (8 per SIMD) holds exactly one wavefront of `rows` chunks (same rule as
    psxhip_audio_api.cpp; tools/gpu_adpcm_sweep.py, tools/gpu_xacd_chunk_sweep.sh).  rows: chains per wavefront, 5 for XA
    (4 filters), 4 for SPU (5 filters)."""
    import os
    if os.environ.get("PSXHIP_ADPCM_CHUNK"):          # experiments only (tools/gpu_xacd_chunk_sweep.sh)
        return int(os.environ["PSXHIP_ADPCM_CHUNK"]), int(os.environ.get("PSXHIP_ADPCM_WARM", "128"))
    if n_cu is None:
        n_cu = 256
        try:
            import torch
            if torch.cuda.is_available():
                n_cu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
        except Exception:
            pass
    per_round = 32 * n_cu * rows
    fill = -(-total_units // per_round)
    if fill >= 1024:
        rounds = -(-fill // 4096)
        return -(-total_units // (per_round * rounds)), 64
    c = total_units // 8192
    p = 64
    while p * 2 <= c and p < 1024:
        p *= 2
    return p, (32 if p >= 1024 else 16)


class AdpcmSession:
    """psxhip_adpcm_session_* (include/psxav_hip.h): a persistent speculate-and-verify encode of a set of chains, whose
    start states can be corrected later (time-sharding across GPUs, psxavenc_amd/parallel.py)."""

    def __init__(self, d_samples, chains, unit_base, filter_count, bits, d_units=None, lead_units=None, chunk_units=128,
                 warmup_units=32):
        import torch
        L = _bind()
        L.psxhip_adpcm_session_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.psxhip_adpcm_session_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.psxhip_adpcm_session_destroy.argtypes = [C.c_void_p]
        L.psxhip_adpcm_session_destroy.restype = None
        self._L = L
        self.chains = np.ascontiguousarray(chains, dtype=CHAIN_DTYPE)
        self.unit_base = np.ascontiguousarray(unit_base, dtype=np.int32)
        self.n_chains = self.chains.size
        dev = d_samples.device
        total = int((self.unit_base + (self.chains["n_units"] - 1) * self.chains["unit_stride"]).max()) + 1 if self.n_chains else 0
        self.d_units = d_units if d_units is not None else torch.zeros((max(total, 1), record_bytes(bits)), dtype=torch.uint8, device=dev)
        self.d_samples = d_samples          # keep alive
        lead = None if lead_units is None else np.ascontiguousarray(lead_units, dtype=np.int32)
        self._h = C.c_void_p()
        _lib.check(L.psxhip_adpcm_session_create(C.byref(self._h), dev.index or 0, d_samples.data_ptr(), self.chains.ctypes.data,
                                                 self.unit_base.ctypes.data, lead.ctypes.data if lead is not None else None,
                                                 self.n_chains, filter_count, bits, self.d_units.data_ptr(), chunk_units,
                                                 warmup_units, torch.cuda.current_stream(dev).cuda_stream))
        self.passes = 0

    def run(self, start_states, known=None, max_passes=0):
        """start_states (n_chains, 2) int32; known (n_chains,) bool/uint8 or None (= all known).
        Returns (final_states (n_chains, 2) int32, changed: bool)."""
        st = np.ascontiguousarray(start_states, dtype=np.int32).reshape(self.n_chains, 2)
        kn = None if known is None else np.ascontiguousarray(known, dtype=np.uint8)
        final = np.zeros((self.n_chains, 2), np.int32)
        changed = C.c_int(0)
        rc = self._L.psxhip_adpcm_session_run(self._h, st.ctypes.data, kn.ctypes.data if kn is not None else None, max_passes,
                                              final.ctypes.data, C.byref(changed))
        if rc < 0:
            _lib.check(rc)
        self.passes += rc
        return final, bool(changed.value)

    def set_timing(self, on=True):
        """HIP events around the speculate launch and the verify passes of the runs that speculate (psxhip_adpcm_session_set_timing)"""
        self._L.psxhip_adpcm_session_set_timing.argtypes = [C.c_void_p, C.c_int]
        _lib.check(self._L.psxhip_adpcm_session_set_timing(self._h, 1 if on else 0))

    def last_timing(self):
        """(speculate_ms, verify_ms) of the last run that speculated with timing on"""
        a, b = C.c_float(0), C.c_float(0)
        self._L.psxhip_adpcm_session_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib.check(self._L.psxhip_adpcm_session_last_timing(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def reset(self):
        """forget the speculative encode: the next run() starts over on the same buffers"""
        self._L.psxhip_adpcm_session_reset.argtypes = [C.c_void_p]
        self._L.psxhip_adpcm_session_reset.restype = None
        self._L.psxhip_adpcm_session_reset(self._h)
        self.passes = 0

    def close(self):
        if self._h:
            self._L.psxhip_adpcm_session_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
