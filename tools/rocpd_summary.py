#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace stats and PMC counters) as text.
usage: rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys


def summarise(path):
    c = sqlite3.connect(path)
    print("== %s" % path)
    try:
        rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                         "from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print("%-64s %8s %14s %14s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
        for r in rows:
            print("%-64s %8d %14d %14.1f %12d %12d %6.2f%%" % (r[0][:64], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    except sqlite3.Error as e:
        print("no kernel table:", e)
    try:
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        if cols:
            rows = c.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                             "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
            if rows:
                print("%-48s %-24s %8s %18s" % ("kernel", "counter", "samples", "avg per dispatch"))
                for r in rows:
                    print("%-48s %-24s %8d %18.1f" % (r[0][:48], r[1], r[2], r[3]))
    except sqlite3.Error as e:
        print("no counters:", e)


def as_json(path):
    """{"kernels": {name: [calls, avg_ns]}, "counters": {kernel: {counter: [avg per dispatch, samples]}}}"""
    c = sqlite3.connect(path)
    out = {"kernels": {}, "counters": {}}
    try:
        for name, n, avg in c.execute("select name, count(*), avg(end-start) from kernels group by name"):
            out["kernels"][name] = [n, avg]
    except sqlite3.Error:
        pass
    try:
        for k, cn, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            out["counters"].setdefault(k, {})[cn] = [avg, n]
    except sqlite3.Error:
        pass
    return out


args = sys.argv[1:]
json_out = None
if "--json" in args:          # also write the numbers as JSON (the databases are too large to travel back from the GPU box)
    i = args.index("--json")
    json_out = args[i + 1]
    del args[i:i + 2]
for p in args:
    summarise(p)
if json_out:
    import json
    with open(json_out, "w") as fh:
        json.dump({p: as_json(p) for p in args}, fh, indent=1)
