#!/usr/bin/env python3
"""Side-by-side PMC averages of the frame kernel from rocprofv3 rocpd databases: pmc_compare.py <dirA> <dirB> ..."""
import glob, os, sqlite3, sys
cols = {}
names = []
for d in sys.argv[1:]:
    names.append(os.path.basename(d.rstrip("/")))
    for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
        c = sqlite3.connect(db)
        try:
            for k, cn, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
                if "mdec_encode_frames" in k:
                    cols.setdefault(cn, {})[names[-1]] = avg
            for k, n, avg in c.execute("select name, count(*), avg(end-start) from kernels group by name"):
                if "mdec_encode_frames" in k:
                    cols.setdefault("kernel_ns(" + os.path.basename(os.path.dirname(db)) + ")", {})[names[-1]] = avg
        except sqlite3.Error as e:
            print("skip", db, e)
print("%-34s" % "counter" + "".join("%16s" % n[-15:] for n in names) + ("%9s" % "B/A" if len(names) > 1 else ""))
for cn in sorted(cols):
    v = [cols[cn].get(n) for n in names]
    r = ("%9.3f" % (v[1] / v[0])) if len(v) > 1 and v[0] and v[1] is not None else ""
    print("%-34s" % cn + "".join("%16.0f" % x if x is not None else "%16s" % "-" for x in v) + r)
