#!/usr/bin/env python3
"""noise +-8, one 1000-frame launch at a time: what distinguishes the launches of the four-batch cycle (165 .. 230 us)?  The diagnostics
records of each batch after a warm-up: passes per frame, how many frames were stopped at the checkpoint, handed on, where the groups end."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_r05_diag as D
kind = sys.argv[1] if len(sys.argv) > 1 else "a8"
bb = D.batches(kind)
for warm in (4, 5, 6, 7, 8, 9):
    r = D.stats_run(bb, warm)
    print("batch", warm % 4, "frames", r["frames"], "passes/frame", r["passes_per_frame"], r["passes_hist_0_1_2_3_4_5plus"], "right", r["first_guess_right"], "ck", r["stopped_at_checkpoint"],
          "ends", r.get("group_end_us_min_p10_p50_p90_max"), "traces", r["pass_traces_of_frames_with_3_or_more_passes"], flush=True)
