#!/usr/bin/env python3
"""Turn the PMC pass summaries of tools/refresh_profiles.sh (gpurun_out/<tag>/pmc_<workload>_<pass>.txt, the output of
tools/pmc_summary.py) into pmc_traffic.json (what bench.py quotes as roofline.traffic) and pmc_trace_l1.md (the L1 /
issue-side counters of the trace kernels).  usage: pmc_to_json.py <dir>

Corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
16-B/lane reads (the trace kernels read dwordx4 per lane), so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  FETCH_SIZE counts
the L2's fabric-side requests: Infinity-Cache hits are included, so this is an upper bound of what reaches HBM."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(path):
    rows, cur = {}, None
    if not os.path.exists(path):
        return rows
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            rows[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+launches\s+(\d+)\s+sum\s+(\S+)", line)
            if m and cur:
                rows[cur][m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return rows


def main():
    d = sys.argv[1]
    from rs_pbrt_amd import lib
    out = {"_comment": __doc__.split("usage")[0].strip(), "source_hash": lib.source_hash(), "workloads": {}}
    for w in ("soup1m", "statue"):
        f, wr = parse("%s/pmc_%s_fetch.txt" % (d, w)), parse("%s/pmc_%s_write.txt" % (d, w))
        kernels, total = {}, 0.0
        for k in sorted(set(f) | set(wr)):
            fs = f.get(k, {}).get("FETCH_SIZE", (0, 0.0))
            ws = wr.get(k, {}).get("WRITE_SIZE", (0, 0.0))
            if not fs[0] and not ws[0]:
                continue
            b = (2.0 * fs[1] + ws[1]) * 1024.0
            kernels[k] = {"launches": fs[0] or ws[0], "FETCH_SIZE_KiB": fs[1], "WRITE_SIZE_KiB": ws[1], "bytes_corrected": b}
            if k.startswith("k_trace_w4"):
                total += b
        if kernels:
            out["workloads"][w] = {"kernels": kernels, "trace_traffic_bytes_per_step": total,
                                   "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --workload %s --steps 1 --warmup 0 "
                                             "--no-cpu-baseline --no-extra --no-count; sums over the k_trace_w4 launches of the one step" % w}
    json.dump(out, open(d + "/pmc_traffic.json", "w"), indent=1)
    lines = ["# L1 / issue-side PMC counters of one C2 step (soup1m), per kernel — rocprofv3 --kernel-trace --pmc <one block per run>", "",
             "source hash %s; command: `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-count`" % out["source_hash"], ""]
    for name in ("tcp", "sq", "ta", "lds"):
        rows = parse("%s/pmc_soup1m_%s.txt" % (d, name))
        if not rows:
            continue
        ctrs = sorted({c for k in rows for c in rows[k]})
        lines += ["## pass `%s`" % name, "", "| kernel | launches | " + " | ".join(ctrs) + " |", "|---|---|" + "---|" * len(ctrs)]
        for k in sorted(rows):
            if k.startswith("bvhdev::"):
                continue
            if name != "lds" and not (k.startswith("k_trace") or k.startswith("k_shade") or k.startswith("k_raygen") or k.startswith("k_film")):
                continue
            n = max((rows[k][c][0] for c in rows[k]), default=0)
            lines.append("| %s | %d | " % (k, n) + " | ".join("%.4g" % rows[k].get(c, (0, 0.0))[1] for c in ctrs) + " |")
        lines.append("")
        if name == "lds":
            fr = sorted(rows[k]["SQ_LDS_BANK_CONFLICT"][1] / max(rows[k]["SQ_LDS_IDX_ACTIVE"][1], 1.0) for k in rows
                        if k.startswith("k_trace_w4") and "SQ_LDS_BANK_CONFLICT" in rows[k] and "SQ_LDS_IDX_ACTIVE" in rows[k])
            if fr:
                lines += ["`SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE` of the traversal launches = %.1f-%.1f %%: the `stack[level][lane]` columns and the SoA copy of the"
                          % (100.0 * fr[0], 100.0 * fr[-1]), "root-side records are read without bank conflicts to speak of (DESIGN.md section 5).", ""]
    open(d + "/pmc_trace_l1.md", "w").write("\n".join(lines) + "\n")
    print(json.dumps(out["workloads"], indent=1)[:2000])


if __name__ == "__main__":
    main()
