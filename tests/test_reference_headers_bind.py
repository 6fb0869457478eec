"""The boundary check made by the COMPILER (`-m "not gpu"`, needs /root/reference, i.e. runs in the build container):

* a C translation unit that includes the reference's own `libpsxav/libpsxav.h` UNCHANGED, takes the address of all 13 public
  prototypes (libpsxav.h:73-101,174-176) and reads the struct sizes, linked against `libpsxav_hip.so` with
  `-Wl,--no-undefined` -- every symbol the reference's FFI would bind resolves, with the reference's own types;
* a second unit that includes the reference's `psxavenc/args.h` unchanged and this library's `psxav_mdec.h` through the
  header shim of INTEGRATION.md section 2, assigns each function to a pointer of the type `psxavenc/mdec.h:65-74` declares
  (written out here from the reference text) and checks `mdec_encoder_t`'s layout against the reference header's field list.
  The reference's `mdec.h` itself cannot be included: it includes <libavcodec/avdct.h>, absent from this image.
Nothing is executed on a device: the program only prints sizes and addresses."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_DIR = os.path.join(ROOT, "psxavenc_amd")

AUDIO_TU = r"""
#include <stdio.h>
#include "libpsxav.h"                 /* the reference's header, from /root/reference/libpsxav, unchanged */

typedef void (*fn_t)(void);
int main(void) {
	fn_t f[] = {
		(fn_t)psx_audio_xa_get_buffer_size, (fn_t)psx_audio_spu_get_buffer_size, (fn_t)psx_audio_xa_get_buffer_size_per_sector,
		(fn_t)psx_audio_xa_get_samples_per_sector, (fn_t)psx_audio_xa_get_sector_interleave,
		(fn_t)psx_audio_xa_encode, (fn_t)psx_audio_xa_encode_simple, (fn_t)psx_audio_spu_encode, (fn_t)psx_audio_spu_encode_simple,
		(fn_t)psx_audio_xa_encode_finalize,
		(fn_t)psx_cdrom_init_xa_subheader, (fn_t)psx_cdrom_init_sector, (fn_t)psx_cdrom_calculate_checksums,
	};
	/* calls with the reference's types that need no device: the size helpers */
	psx_audio_xa_settings_t s = {PSX_AUDIO_XA_FORMAT_XACD, true, PSX_AUDIO_XA_FREQ_DOUBLE, 4, 1, 0};
	unsigned n = 0;
	for (unsigned i = 0; i < sizeof f / sizeof f[0]; i++) n += f[i] != 0;
	printf("%u %zu %zu %zu %u %u %u %u\n", n, sizeof(psx_audio_xa_settings_t), sizeof(psx_audio_encoder_channel_state_t),
	       sizeof(psx_audio_encoder_state_t), psx_audio_xa_get_buffer_size_per_sector(s), psx_audio_xa_get_samples_per_sector(s),
	       psx_audio_xa_get_sector_interleave(s), psx_audio_spu_get_buffer_size(22050));
	return 0;
}
"""

# the shim of INTEGRATION.md section 2 ("psxavenc/mdec.h")
MDEC_SHIM = r"""
#pragma once
#include "args.h"              /* the reference's psxavenc/args.h, unchanged: format_t, bs_codec_t */
#define PSXAV_MDEC_NO_ENUMS    /* ... so the library header does not redefine them */
#include <psxav_mdec.h>
"""

MDEC_TU = r"""
#include <stddef.h>
#include <stdio.h>
#include "mdec.h"                     /* the shim; pulls the reference's args.h */

/* psxavenc/mdec.h:65-74, as the reference's callers see them */
static bool (*p_init)(mdec_encoder_t *encoder, bs_codec_t video_codec, int video_width, int video_height) = init_mdec_encoder;
static void (*p_destroy)(mdec_encoder_t *encoder) = destroy_mdec_encoder;
static void (*p_frame)(mdec_encoder_t *encoder, const uint8_t *video_frame) = encode_frame_bs;
static int (*p_sector)(mdec_encoder_t *encoder, format_t format, uint16_t str_video_id, const uint8_t *video_frames,
                       uint8_t *output) = encode_sector_str;
int main(void) {
	mdec_encoder_t e;
	printf("%d %zu %zu", (p_init != 0) + (p_destroy != 0) + (p_frame != 0) + (p_sector != 0), sizeof(mdec_encoder_t), sizeof(mdec_encoder_state_t));
#define OFF(f) printf(" %s=%zu", #f, offsetof(mdec_encoder_state_t, f))
	FIELDS
	printf(" | state=%zu video_codec=%zu video_width=%zu video_height=%zu\n", offsetof(mdec_encoder_t, state), offsetof(mdec_encoder_t, video_codec),
	       offsetof(mdec_encoder_t, video_width), offsetof(mdec_encoder_t, video_height));
	(void)e;
	/* the enumerators the callers pass come from the reference's args.h */
	return (int)FORMAT_STRCD - 7 + (int)BS_CODEC_V3DC - 2;
}
"""


def _cc():
    return shutil.which("gcc") or shutil.which("cc")


@pytest.mark.skipif(_cc() is None, reason="no C compiler")
def test_reference_libpsxav_header_binds_to_the_library(reference_root, tmp_path):
    src = tmp_path / "bind_audio.c"
    src.write_text(AUDIO_TU)
    exe = tmp_path / "bind_audio"
    r = subprocess.run([_cc(), "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(reference_root, "libpsxav"), str(src), "-o", str(exe),
                        "-L", LIB_DIR, "-lpsxav_hip", "-Wl,--no-undefined", "-Wl,-rpath," + LIB_DIR], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    n, s_set, s_ch, s_st, per_sector, sps, inter, spu = (int(v) for v in r.stdout.split())
    assert n == 13
    assert (s_set, s_ch, s_st) == (24, 24, 48)
    assert (per_sector, sps, inter) == (2352, 2016, 4)              # adpcm.c:235-260 through the library's implementation (x cd speed 2 = config strcd's 8)
    assert spu == 16 * ((22050 + 27) // 28)


@pytest.mark.skipif(_cc() is None, reason="no C compiler")
def test_reference_args_h_and_the_mdec_shim_bind_to_the_library(reference_root, tmp_path):
    # the field list of mdec_encoder_state_t, read from the reference's header text (mdec.h:32-55)
    text = open(os.path.join(reference_root, "psxavenc", "mdec.h")).read()
    body = text[text.index("typedef struct {"):text.index("} mdec_encoder_state_t;")]
    fields = []
    for line in body.splitlines()[1:]:
        line = line.split("//")[0].strip().rstrip(";")
        if not line:
            continue
        name = re.sub(r"\[.*\]", "", line.split()[-1].lstrip("*"))
        fields.append(name)
    assert fields[0] == "frame_index" and "dct_context" in fields and "dct_block_lists" in fields and len(fields) >= 20, fields
    shim_dir = tmp_path / "shim"
    shim_dir.mkdir()
    (shim_dir / "mdec.h").write_text(MDEC_SHIM)
    src = tmp_path / "bind_mdec.c"
    src.write_text(MDEC_TU.replace("FIELDS", " ".join("OFF(%s);" % f for f in fields)))
    exe = tmp_path / "bind_mdec"
    r = subprocess.run([_cc(), "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", str(shim_dir), "-I", os.path.join(reference_root, "psxavenc"),
                        "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", LIB_DIR, "-lpsxav_hip", "-Wl,--no-undefined",
                        "-Wl,-rpath," + LIB_DIR], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    head, tail = r.stdout.split("|")
    tok = head.split()
    assert (int(tok[0]), int(tok[1]), int(tok[2])) == (4, 168, 152)     # x86-64 sizes of the reference's structs
    offs = dict(t.split("=") for t in tok[3:])
    assert list(offs) == fields                                         # every reference field exists, by name
    vals = [int(offs[f]) for f in fields]
    assert vals == sorted(vals) and vals[0] == 0                        # ... in the reference's order
    # mdec_encoder_t (mdec.h:57-63): the order of its members as the reference's text has them, the state embedded by value
    outer = text[text.index("} mdec_encoder_state_t;"):text.index("} mdec_encoder_t;")]
    order = [ln.split("//")[0].strip().rstrip(";").split()[-1] for ln in outer.splitlines()[1:] if ln.split("//")[0].strip().endswith(";")]
    assert sorted(order) == ["state", "video_codec", "video_height", "video_width"], order
    t = {k: int(v) for k, v in (x.split("=") for x in tail.split())}
    assert sorted(order, key=lambda k: t[k]) == order and t[order[0]] == 0
    assert max(t.values()) + (152 if order[-1] == "state" else 4) <= 168
