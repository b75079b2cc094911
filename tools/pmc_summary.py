#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters from result databases. usage: pmc_summary.py <dir> [kernel substring]"""
import glob, sqlite3, sys
src = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = {}
for db in sorted(glob.glob(src + "/**/*.db", recursive=True)):
    con = sqlite3.connect(db)
    try:
        for name, ctr, n, tot in con.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
            short = name.split("(")[0].replace("void ", "").replace("rspt::", "")
            if filt in short:
                rows.setdefault(short, {})[ctr] = (n, tot)
    except sqlite3.OperationalError:
        pass
for k in sorted(rows):
    print(k)
    for c in sorted(rows[k]):
        n, tot = rows[k][c]
        print("   %-32s launches %4d  sum %.4g  per-launch %.4g" % (c, n, tot, tot / max(n, 1)))
