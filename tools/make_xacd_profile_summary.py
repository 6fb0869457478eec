#!/usr/bin/env python3
"""Turn the rocprofv3 results of tools/gpu_r05_xacd_pmc.sh (gpurun_out/prof_xacd_<tag>/: summary.txt + summary.json, the databases stay
on the GPU box) into profiles/<round>_xacd_<tag>_summary.txt and an entry of profiles/pmc_index.json keyed by the ADPCM workload key
bench.py prints (roofline.traffic_key): per kernel (speculate = adpcm_chunks_kernel<false,..>, verify = <true,..>, xa_assemble_kernel)
the kernel-trace average duration, FETCH_SIZE / WRITE_SIZE (own passes, fetch doubled per MI355X_MICROARCH.md) and the SQ counters.
usage: make_xacd_profile_summary.py <round> <tag> [<tag> ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"speculate": "adpcm_chunks_kernel<false", "verify": "adpcm_chunks_kernel<true", "assemble": "xa_assemble_kernel"}


def main():
    rnd, tags = sys.argv[1], sys.argv[2:]
    idx_path = os.path.join(ROOT, "profiles", "pmc_index.json")
    idx = json.load(open(idx_path))
    for tag in tags:
        src = os.path.join(ROOT, "gpurun_out", "prof_xacd_" + tag)
        line = json.loads(open(os.path.join(src, "bench_line.json")).read())
        sj = json.load(open(os.path.join(src, "summary.json")))
        dst = os.path.join(ROOT, "profiles", "%s_xacd_%s_summary.txt" % (rnd, tag))
        with open(dst, "w") as fh:
            fh.write("# rocprofv3 summaries (tools/gpu_r05_xacd_pmc.sh %s): kernel-trace --stats, then one --pmc pass per counter group.\n" % tag)
            fh.write("# bench line of the kernel-trace pass: %s\n" % json.dumps(line))
            fh.write(open(os.path.join(src, "summary.txt")).read())
        kern = {}
        for role, frag in NAMES.items():
            e = {}
            for path, v in sj.items():
                sub = path.split("/")[-2]
                for kn, (calls, avg) in v["kernels"].items():
                    if frag in kn and sub == "kt":
                        e["calls_in_trace"], e["avg_ns"] = calls, round(avg, 1)
                for kn, cs in v["counters"].items():
                    if frag in kn:
                        for cn, (avg, n) in cs.items():
                            e[cn] = round(avg, 1)
            if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                e["traffic_bytes_per_launch"] = int(e["FETCH_SIZE"] * 1024 * 2 + e["WRITE_SIZE"] * 1024)
            if "SQ_INSTS_VALU" in e and "avg_ns" in e:
                e["valu_busy_frac"] = round(e["SQ_INSTS_VALU"] / (1024 * 2.4e9 / 4.0 * e["avg_ns"] * 1e-9), 4)
            kern[role] = e
        key = line["roofline"].get("traffic_key") or ("xacd %s" % tag)
        if len(sys.argv) > 3 and sys.argv[2].startswith("--key="):
            pass
        idx[key] = {"source": "profiles/%s_xacd_%s_summary.txt" % (rnd, tag), "kernel": "adpcm_chunks_kernel<false, 12> (speculate)",
                    "traffic_bytes_per_launch": kern["speculate"].get("traffic_bytes_per_launch"),
                    "valu_insts_per_launch": kern["speculate"].get("SQ_INSTS_VALU"), "kernels": kern, "fetch_correction": 2.0,
                    "note": "per launch of each kernel (verify: mean over all verify launches, most of which return at once); FETCH_SIZE doubled per "
                            "MI355X_MICROARCH.md; valu_busy_frac = SQ_INSTS_VALU / (1024 SIMDs x 2.4 GHz / 4 x kernel-trace duration)"}
        print(tag, key, json.dumps({r: {k: kern[r].get(k) for k in ("avg_ns", "traffic_bytes_per_launch", "valu_busy_frac")} for r in kern}))
    with open(idx_path, "w") as fh:
        json.dump(idx, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
