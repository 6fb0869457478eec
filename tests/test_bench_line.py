"""The ONE line bench.py prints must stay inside the driver's 8 KB stdout tail (a 36 KB line was dropped and the
round's headline went unrecorded).  The line builder is run here on canned full records -- bloated on purpose -- without a GPU."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _child(i, fat=4000):
    return {"metric": "m%d" % i, "value": 1.5e6 + i, "unit": "sectors/s", "rc": 0, "roofline": {"frac": 0.05, "per_kernel_counters": {"k": "x" * fat}},
            "parity": {"bit_exact": True, "sectors_checked": 40}, "config": {"workload": "w" * 500, "legs": {"a": "y" * fat}}, "cpu_baseline": {"sample": "z" * 800}}


def _full(n_gpus=1, fat=4000):
    sec = {"noise_amp_8": {"two_lanes": {"frames_per_sec": 7.8e6}, "one_lane_in_order": {"frames_per_sec": 5.9e6}, "passes": {"hist": list(range(64))}},
           "mixed_content": {"two_lanes": {"frames_per_sec": 6.0e6}, "one_lane_in_order": {"frames_per_sec": 4.0e6}, "passes": {"passes_per_frame": 1.14}},
           "per_call_drop_in": {"encode_frame_bs_320x240_v2": {"us_per_call_median": 56.3, "frames_per_sec": 17700.0}, "blob": "q" * fat}}
    for i, name in enumerate(("sbs_v3_1250", "xacd_config5", "xacd_config5_white_noise", "xacd_config5_gated_tone", "strcd_config3", "rccl_world_size_1")):
        sec[name] = _child(i, fat)
    sec["strcd_config3"]["config"]["eight_streams_sectors_per_sec"] = 13.5e6
    return {
        "metric": "bs_v2_320x240_frames_per_sec", "value": 8.7e6, "unit": "frames/s", "n_gpus": n_gpus, "steps": 20, "warmup": 5, "ms_per_step": 551.2,
        "timed_region_s": 11.02, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "sbs v2: " + "w" * 900, "preset": "sbs_v2", "baseline_config": "sbs v2: 1000 synthetic 320x240 YCbCr frames, 1 GPU, bit-exact check",
                   "secondary_summary": bench._secondary_summary(sec), "quant_scale_hist": {"3": 4000}, "kernel_shape": {"a": 1}, "launch_order": "o" * 700,
                   "in_order_frames_per_sec": 7.8e6, "library": "psxav_hip 0.5 (gfx950, mdec-k4.0)"},
        "per_rank": [{"rank": r, "frames_per_sec": 8.7e6, "elapsed_s": 11.0, "kernel_ms": 0.127, "roofline_achieved_gbs": 970.0, "roofline_frac": 0.121,
                      "quant_scale_sum": 12000000, "results_sane": True} for r in range(n_gpus)],
        "roofline": {"bound": "hbm", "kernel": "mdec_encode_frames_kernel", "achieved": 969.5, "peak": 8000.0, "unit": "GB/s", "frac": 0.1212,
                     "traffic": 136800000, "traffic_source": "profiles/r05_a4_summary.txt", "traffic_key": "k" * 300, "kernel_ms": 0.1273,
                     "kernel_ms_stats": {"n": 5}, "kernel_ms_method": "m" * 600, "algorithmic_bytes_per_launch": 123392000,
                     "issue": {"valu_busy_frac": 0.796, "note": "n" * 600}, "overlapped": {"note": "n" * 900}},
        "cpu_baseline": {"value": 390.0, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "s" * 700,
                         "all_cores": {"value": 6200.0, "cores": 16, "nproc": 256, "cores_note": "c" * 400}},
        "secondary": sec, "parity": {"frames_checked": 64, "batches_checked": 4, "bit_exact": True}, "results_sane": True,
        "dist": {"backend": "nccl", "world_size": n_gpus, "nccl_version": "2.26.6"} if n_gpus > 1 else None,
        "predicted": {"value": 8.7e6 * n_gpus, "unit": "frames/s", "basis": "N x 8.7 M"},
    }


@pytest.mark.parametrize("n_gpus,fat", [(1, 4000), (8, 4000), (8, 40000)])
def test_compact_line_is_small_and_complete(n_gpus, fat):
    full = _full(n_gpus, fat)
    assert len(json.dumps(full)) > 30000          # the record that was dropped in round 5 was 36 KB
    line = bench.compact_line(full, "gpurun_out/bench_detail_sbs_v2_n%d.json" % n_gpus)
    s = json.dumps(line)
    assert len(s) <= bench.LINE_LIMIT < 8192
    back = json.loads(s)
    for k in REQUIRED:
        assert k in back, k
    assert back["roofline"]["bound"] == "hbm" and back["roofline"]["frac"] == 0.1212 and back["roofline"]["traffic"] == 136800000
    assert back["roofline"]["valu_busy_frac"] == 0.796
    assert back["cpu_baseline"]["kind"] == "port" and back["cpu_baseline"]["cores"] == 1 and back["cpu_baseline"]["all_cores"]["cores"] == 16
    assert back["parity"]["bit_exact"] is True
    assert "secondary" not in back and back["detail_file"].endswith(".json")
    summ = back["config"]["secondary_summary"]
    assert summ["strcd_config3"] == {"value": 1500004.0, "unit": "sectors/s", "frac": 0.05, "bit_exact": True, "s8": 13.5e6}
    assert all(len(json.dumps(v)) <= 150 for v in summ.values())
    if n_gpus > 1:
        assert len(back["per_rank"]) == n_gpus and back["per_rank"][3]["frames_per_sec"] == 8.7e6
        assert back["dist"]["world_size"] == n_gpus and back["predicted"]["value"] == 8.7e6 * n_gpus


def test_emit_prints_one_line_and_writes_the_detail(tmp_path, capsys):
    class A:
        detail_file = str(tmp_path / "d.json")
        config = "sbs_v2"
    bench._emit(_full(1), A)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and len(out[0]) <= bench.LINE_LIMIT
    assert json.loads(out[0])["metric"] == "bs_v2_320x240_frames_per_sec"
    assert json.load(open(A.detail_file))["secondary"]["xacd_config5"]["rc"] == 0
