#!/usr/bin/env python3
"""Turn the rocprofv3 results of tools/gpu_r05_strcd_pmc.sh (gpurun_out/prof_strcd_pmc_S<S>/) into profiles/<round>_strcd_S<S>_summary.txt
and an entry of profiles/pmc_index.json keyed by the workload key bench.py prints for config 3 (roofline.traffic_key).  A step of the
device-resident muxer is several kernels on two streams (frame kernel, video sector kernel, speculate, verify passes, sector assembly):
the entry's traffic is the SUM over the product's kernels of (average FETCH_SIZE x 2 + WRITE_SIZE) x launches, divided by the steps of
the profiled run (= launches of str_video_sector_kernel, one per step); per-kernel figures beside it.
usage: make_strcd_profile_summary.py <round> <streams>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = ("mdec_encode_frames_kernel", "str_video_sector_kernel", "adpcm_chunks_kernel", "xa_assemble_kernel", "mdec_stage_in_kernel")


def main():
    rnd, S = sys.argv[1], int(sys.argv[2])
    src = os.path.join(ROOT, "gpurun_out", "prof_strcd_pmc_S%d" % S)
    line = json.loads(open(os.path.join(src, "bench_line.json")).read())
    sj = json.load(open(os.path.join(src, "summary.json")))
    dst = os.path.join(ROOT, "profiles", "%s_strcd_S%d_summary.txt" % (rnd, S))
    with open(dst, "w") as fh:
        fh.write("# rocprofv3 summaries (tools/gpu_r05_strcd_pmc.sh %d): kernel-trace --stats, then one --pmc pass per counter group.\n" % S)
        fh.write("# bench line of the kernel-trace pass: %s\n" % json.dumps(line))
        fh.write(open(os.path.join(src, "summary.txt")).read())
    per = {}
    for path, v in sj.items():
        sub = path.split("/")[-2]
        for kn, cs in v["counters"].items():
            name = next((p for p in PRODUCT if p in kn), None)
            if not name:
                continue
            if "adpcm_chunks_kernel" in kn:
                name = "adpcm_chunks_kernel (speculate)" if "<false" in kn else "adpcm_chunks_kernel (verify)"
            e = per.setdefault(name, {})
            for cn, (avg, n) in cs.items():
                e[cn] = {"avg": round(avg, 2), "launches": n}
        if sub == "kt":
            for kn, (calls, avg) in v["kernels"].items():
                name = next((p for p in PRODUCT if p in kn), None)
                if not name:
                    continue
                if "adpcm_chunks_kernel" in kn:
                    name = "adpcm_chunks_kernel (speculate)" if "<false" in kn else "adpcm_chunks_kernel (verify)"
                per.setdefault(name, {})["kernel_trace"] = {"avg_ns": round(avg, 1), "launches": calls}
    steps = {c: per["str_video_sector_kernel"][c]["launches"] for c in ("FETCH_SIZE", "WRITE_SIZE")}
    fetch = sum(e["FETCH_SIZE"]["avg"] * e["FETCH_SIZE"]["launches"] for e in per.values() if "FETCH_SIZE" in e) * 1024 * 2 / steps["FETCH_SIZE"]
    write = sum(e["WRITE_SIZE"]["avg"] * e["WRITE_SIZE"]["launches"] for e in per.values() if "WRITE_SIZE" in e) * 1024 / steps["WRITE_SIZE"]
    valu = sum(e["SQ_INSTS_VALU"]["avg"] * e["SQ_INSTS_VALU"]["launches"] for e in per.values() if "SQ_INSTS_VALU" in e) / per["str_video_sector_kernel"]["SQ_INSTS_VALU"]["launches"]
    key = line["roofline"]["traffic_key"]
    idx_path = os.path.join(ROOT, "profiles", "pmc_index.json")
    idx = json.load(open(idx_path))
    idx[key] = {"source": "profiles/%s_strcd_S%d_summary.txt" % (rnd, S), "kernel": "whole step (all of the product's kernels, both streams)",
                "traffic_bytes_per_launch": int(fetch + write), "fetch_bytes_per_step": int(fetch), "write_bytes_per_step": int(write),
                "valu_insts_per_launch": int(valu), "steps_profiled": steps, "kernels": per, "fetch_correction": 2.0,
                "note": "per STEP: sum over the kernels of a step of average counter x launches, / steps of the profiled run; FETCH_SIZE doubled per MI355X_MICROARCH.md"}
    with open(idx_path, "w") as fh:
        json.dump(idx, fh, indent=1, sort_keys=True)
    print(key, int(fetch + write), "bytes per step; algorithmic", line["roofline"].get("algorithmic_bytes_per_launch"))


if __name__ == "__main__":
    main()
