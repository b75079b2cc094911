#!/usr/bin/env python3
"""Per-DISPATCH view of a rocprofv3 result database: the launches of the wavefront loop in launch order, each with its duration
(--kernel-trace) or its counter values (--pmc), so that the cost of a stage can be read per wavefront iteration (iteration 0 works on a
dense pixel-major queue, later ones on what is left of it) instead of as one per-kernel mean.
usage: per_dispatch.py <dir with *_results.db> [kernel substring, default k_]  -> text on stdout
Each line: ordinal within its kernel name, kernel, grid, duration in us (if present), counters (if present)."""
import glob
import sqlite3
import sys


def cols(con, view):
    try:
        return [r[1] for r in con.execute("pragma table_info(%s)" % view)]
    except sqlite3.OperationalError:
        return []


def short(name):
    return name.split("(")[0].replace("void ", "").replace("rspt::", "")


def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "k_"
    for db in sorted(glob.glob(src + "/**/*.db", recursive=True)):
        con = sqlite3.connect(db)
        kc, cc = cols(con, "kernels"), cols(con, "counters_collection")
        print("# %s" % db.split("/")[-1])
        print("# kernels columns: %s" % ", ".join(kc))
        print("# counters_collection columns: %s" % ", ".join(cc))
        rows = []
        if "start" in kc and "end" in kc:
            key = "dispatch_id" if "dispatch_id" in kc else "id"
            for did, name, gx, t0, t1 in con.execute("select %s, name, grid_x, start, end from kernels order by start" % key):
                rows.append([did, short(name), gx, t0, (t1 - t0) / 1e3, {}])
        ctr = {}
        if cc and "dispatch_id" in cc:
            namecol = "kernel_name" if "kernel_name" in cc else "name"
            for did, name, c, v in con.execute("select dispatch_id, %s, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name" % namecol):
                ctr.setdefault(did, [short(name), {}])[1][c] = v
        if rows:
            for r in rows:
                if r[0] in ctr:
                    r[5] = ctr[r[0]][1]
        elif ctr:
            for did in sorted(ctr):
                rows.append([did, ctr[did][0], None, None, None, ctr[did][1]])
        seen = {}
        for did, name, gx, t0, dur, cv in rows:
            if filt not in name:
                continue
            k = seen.get(name, 0)
            seen[name] = k + 1
            line = "%4d  %-58s" % (k, name[:58])
            if gx is not None:
                line += " grid %10d" % gx
            if dur is not None:
                line += "  %10.1f us" % dur
            for c in sorted(cv):
                line += "  %s %.5g" % (c, cv[c])
            print(line)


if __name__ == "__main__":
    main()
