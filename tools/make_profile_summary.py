#!/usr/bin/env python3
"""Turn the rocprofv3 result databases of tools/gpu_rocprof_mdec.sh (gpurun_out/prof_<tag>/) into the committed
summaries under profiles/: <round>_<tag>_summary.txt (kernel-trace stats + every PMC pass), the bench line of the
profiled command, and an entry in profiles/pmc_index.json keyed by the workload key bench.py prints
(roofline.traffic_key), so that bench.py quotes HBM-side traffic only for the exact workload + library version measured.
usage: make_profile_summary.py <round> <tag> [<tag> ...]"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters_json(src, sub, like):
    """the same from summary.json (tools/rocpd_summary.py --json), when the databases did not travel back"""
    try:
        d = json.load(open(os.path.join(src, "summary.json")))
    except Exception:
        return {}
    key = like.strip("%")
    out = {}
    for path, v in d.items():
        if "/%s/" % sub not in path:
            continue
        for k, cs in v["counters"].items():
            if key in k:
                for cn, (avg, n) in cs.items():
                    out[cn] = (avg, n)
    return out


def counters(db, like):
    c = sqlite3.connect(db)
    out = {}
    try:
        for name, avg, n in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like ? "
                                      "group by counter_name", (like,)):
            out[name] = (avg, n)
    except sqlite3.Error:
        pass
    return out


def main():
    rnd, tags = sys.argv[1], sys.argv[2:]
    idx_path = os.path.join(ROOT, "profiles", "pmc_index.json")
    idx = json.load(open(idx_path)) if os.path.exists(idx_path) else {}
    for tag in tags:
        src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
        text = open(os.path.join(src, "summary.txt")).read()
        line = None
        for f in ("kt.log",):
            for l in open(os.path.join(src, f), errors="replace"):
                if l.startswith("{") and '"metric"' in l:
                    line = json.loads(l)
        dst = os.path.join(ROOT, "profiles", "%s_%s_summary.txt" % (rnd, tag))
        with open(dst, "w") as fh:
            fh.write("# rocprofv3 summaries (tools/gpu_rocprof_mdec.sh %s ...): kernel-trace --stats, then one --pmc pass per counter group.\n" % tag)
            fh.write("# bench line of the kernel-trace pass: %s\n" % json.dumps(line))
            fh.write(text)
        if not line:
            continue
        kernel = line["roofline"]["kernel"]
        like = "%" + kernel.split("(")[0].split("<")[0] + "%"
        fetch = write = None
        for db in glob.glob(os.path.join(src, "fetch", "**", "*.db"), recursive=True):
            fetch = counters(db, like).get("FETCH_SIZE")
        for db in glob.glob(os.path.join(src, "write", "**", "*.db"), recursive=True):
            write = counters(db, like).get("WRITE_SIZE")
        sq = {}
        for db in glob.glob(os.path.join(src, "sq", "**", "*.db"), recursive=True):
            sq = counters(db, like)
        if not fetch:
            fetch = counters_json(src, "fetch", like).get("FETCH_SIZE")
        if not write:
            write = counters_json(src, "write", like).get("WRITE_SIZE")
        if not sq:
            sq = counters_json(src, "sq", like)
        key = line["roofline"].get("traffic_key")
        if fetch and write and key:
            idx[key] = {
                "source": "profiles/%s_%s_summary.txt" % (rnd, tag),
                "kernel": kernel,
                "FETCH_SIZE_KB_per_launch": round(fetch[0], 1), "WRITE_SIZE_KB_per_launch": round(write[0], 1),
                "launches_sampled": fetch[1],
                "fetch_correction": 2.0,
                "traffic_bytes_per_launch": int(fetch[0] * 1024 * 2 + write[0] * 1024),
                "valu_insts_per_launch": int(sq["SQ_INSTS_VALU"][0]) if "SQ_INSTS_VALU" in sq else None,
                "salu_insts_per_launch": int(sq["SQ_INSTS_SALU"][0]) if "SQ_INSTS_SALU" in sq else None,
                "lds_insts_per_launch": int(sq["SQ_INSTS_LDS"][0]) if "SQ_INSTS_LDS" in sq else None,
                "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported. "
                        "L2<->fabric traffic, Infinity-Cache hits included.",
            }
            print(tag, key, idx[key]["traffic_bytes_per_launch"])
    with open(idx_path, "w") as fh:
        json.dump(idx, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
