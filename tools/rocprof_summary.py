#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats result database into a small text summary for profiles/.
usage: rocprof_summary.py <dir with *_results.db> <out.md> [title]"""
import glob
import sqlite3
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else src
    dbs = sorted(glob.glob(src + "/**/*.db", recursive=True))
    lines = ["# rocprofv3 --kernel-trace --stats: %s" % title, ""]
    for db in dbs:
        con = sqlite3.connect(db)
        lines += ["## %s" % db.split("/")[-1], "", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
        for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            short = name.split("(")[0].replace("void ", "")
            lines.append("| %s | %d | %.3f | %.1f | %.2f |" % (short, calls, total / 1e3, avg, pct))
        lines += ["", "| kernel | vgpr | sgpr | lds | scratch | workgroup | grid |", "|---|---|---|---|---|---|---|"]
        q = "select name, vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, min(grid_x), max(grid_x) from kernels group by name"
        for name, v, s, l, sc, wg, g0, g1 in con.execute(q):
            short = name.split("(")[0].replace("void ", "")
            lines.append("| %s | %s | %s | %s | %s | %s | %s..%s |" % (short, v, s, l, sc, wg, g0, g1))
        lines.append("")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
