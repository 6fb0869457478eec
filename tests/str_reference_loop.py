"""encode_file_str (psxavenc/filefmt.c:391-520) restated call for call over the CPU oracle, with the decoder it pulls
its input from (psxavenc/decoding.c:510-586) modelled for an input that is all in memory.

Test infrastructure (like oracle_lib): the checker for psxhip_str_encode_host's REFERENCE tail.  Written from the
reference text, not from the product: the sector loop, its exit condition, the frames_needed / end_of_input interplay and
the short-audio hack are the reference's, line by line.  Sector buffers are zero-initialised (the reference muxes into an
uninitialised stack array), frames go through oracle/mdec_oracle.c, audio through the XA encoder handed in
(oracle restatement or the reference's own build).
"""
import ctypes as C
import math

import numpy as np

import oracle_lib as O


class Decoder:
    """decoder_t for in-memory input.  audio_sample_count / video_frame_count are what is left to hand out."""

    def __init__(self, frames, pcm, channels):
        self.frames = frames
        self.vpos = 0
        self.video_frame_count = frames.shape[0]
        mul = max(1, channels)
        pcm = np.asarray(pcm, dtype=np.int16).reshape(-1)
        # "out is always padded out with 4032 "0" samples, this makes calculations elsewhere easier" (decoding.c:521-527)
        self.pcm = np.concatenate([pcm, np.zeros(4032 * mul, np.int16)])
        self.apos = 0
        self.audio_sample_count = pcm.size if channels else 0
        self.end_of_input = False

    # decoding.c:536-560.  The loop polls while a count is <= its threshold; with nothing left to demux the only exit
    # is poll_av_data() returning false, which is where end_of_input is raised (decoding.c:517-530).  With everything
    # left to demux "in the file", counts that are <= the threshold AFTER polling everything are counts that are <= it
    # with all remaining input handed over: the model holds the whole remaining input as the buffer.
    def ensure_av_data(self, needed_audio_samples, needed_video_frames):
        if (needed_audio_samples and self.audio_sample_count <= needed_audio_samples) or \
           (needed_video_frames and self.video_frame_count <= needed_video_frames):
            self.end_of_input = True
            return bool((self.audio_sample_count or not needed_audio_samples) and
                        (self.video_frame_count or not needed_video_frames))
        return True

    def retire_av_data(self, retired_audio_samples, retired_video_frames):       # decoding.c:562-586
        assert retired_audio_samples <= self.audio_sample_count
        assert retired_video_frames <= self.video_frame_count
        self.apos += retired_audio_samples
        self.audio_sample_count -= retired_audio_samples
        self.vpos += retired_video_frames
        self.video_frame_count -= retired_video_frames

    @property
    def audio_samples(self):
        return self.pcm[self.apos:]

    @property
    def video_frames(self):
        return self.frames[min(self.vpos, self.frames.shape[0] - 1)]


def encode_file_str(fmt, codec, w, h, fps_num, fps_den, cd_speed, frames, pcm, channels=2, freq=37800, bits=4,
                    trailing_audio=False, xa_file=1, xa_channel=0, video_id=0x8001, xa_encode=None):
    """Returns (sectors (n, sector_size) uint8, quant_scale_sum, frames_encoded)."""
    xa_encode = xa_encode or O.xa_encode
    ofmt = {6: O.FMT_STR, 7: O.FMT_STRCD, 9: O.FMT_STRV}[fmt]
    xa_settings = O.XaSettings(1 if fmt == 7 else 0, 1 if channels == 2 else 0, freq, bits, xa_file, xa_channel)
    sector_size = O.lib().orc_xa_sector_size(xa_settings)
    decoder = Decoder(frames, pcm, channels)

    if channels:                                           # filefmt.c:399-403
        interleave = O.lib().orc_xa_sector_interleave(xa_settings) * cd_speed
        audio_samples_per_sector = O.lib().orc_xa_samples_per_sector(xa_settings)
        video_sectors_per_block = interleave - 1
    else:                                                  # :415-419
        interleave, audio_samples_per_sector, video_sectors_per_block = 1, 0, 1

    audio_state = None          # :422-423 (zeroed state; the encoder handed in makes its own kind: O.State / the reference's RefState)
    base = (75 * cd_speed) * video_sectors_per_block * fps_den      # :431
    den = interleave * fps_num                                       # :432
    frame_size = float(base) / float(den)                            # :433
    frame_output = np.zeros(2016 * int(math.ceil(frame_size)), np.uint8)     # :438
    enc = O.StrState(0, 0, 0, base, 0, den, 0, 0, frame_output.ctypes.data)  # :439-443
    frames_needed = int(math.ceil(float(video_sectors_per_block) / frame_size))     # :446
    if frames_needed < 2:
        frames_needed = 2

    out = []
    sector_count = 0
    while (not decoder.end_of_input) or enc.frame_data_offset < enc.frame_max_size:      # :450
        decoder.ensure_av_data(audio_samples_per_sector * channels, frames_needed)       # :451
        sector = np.zeros(2352, np.uint8)
        if audio_samples_per_sector == 0:                  # :456-461
            is_video_sector = True
        elif trailing_audio:
            is_video_sector = (sector_count % interleave) < video_sectors_per_block
        else:
            is_video_sector = (sector_count % interleave) > 0

        if is_video_sector:
            # init_sector_buffer_video, filefmt.c:73-91
            if fmt == 7:
                O.lib().orc_cdrom_init_sector(O.ptr(sector, O.u8p), sector_count, 1)
                sector[16:20] = [xa_file, xa_channel & 0x1F, 0x08 | 0x40, 0]
                sector[20:24] = sector[16:20]
            elif fmt == 6:
                sector[0:4] = [xa_file, xa_channel & 0x1F, 0x08 | 0x40, 0]
                sector[4:8] = sector[0:4]
            frames_used = O.lib().orc_mdec_encode_sector_str(C.byref(enc), codec, w, h, ofmt, video_id,
                                                             O.ptr(decoder.video_frames, O.u8p), O.ptr(sector, O.u8p))    # :466-472
            assert 0 <= frames_used <= 1, "more than one frame per sector: the reference strides frames wrongly there (SURVEY App. B4)"
            O.lib().orc_cdrom_calculate_checksums(O.ptr(sector, O.u8p), 1)       # :474
            decoder.retire_av_data(0, frames_used)                                # :475
        else:
            samples_length = decoder.audio_sample_count // channels              # :477
            if samples_length > audio_samples_per_sector:
                samples_length = audio_samples_per_sector
            if not samples_length:                                                # :483-484
                video_sectors_per_block += 1
            length = 0
            if samples_length:          # psx_audio_xa_encode writes nothing for sample_count 0 (adpcm.c:310)
                w_, audio_state = xa_encode(xa_settings, decoder.audio_samples, samples_length, lba=sector_count, state=audio_state)
                length = w_.size
                assert length == sector_size
                sector[:length] = w_
            if decoder.end_of_input and length >= 2336:                           # :492-493, adpcm.c:334-340
                sub = length - 2352 + 0x12
                sector[sub] |= 0x80
                sector[sub + 4] |= 0x80
            decoder.retire_av_data(samples_length * channels, 0)                  # :495
        out.append(sector[:sector_size].copy())                                   # :498
        sector_count += 1
    stream = np.stack(out) if out else np.zeros((0, sector_size), np.uint8)
    return stream, enc.quant_scale_sum, enc.frame_index
