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
    # limiter-side ratios of a C2 / C3 step (what bench.py leads its roofline blocks with).  Per kernel class:
    #   cycles            GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8
    #   l1_request_frac   TCP_TOTAL_CACHE_ACCESSES_sum / 256 CUs / cycles     (one L1 line probe per CU per cycle is the ceiling)
    #   ta_busy           TA_TA_BUSY_sum / 256 TAs / cycles;  ta_busy_avr = TA_BUSY_avr / cycles
    #   valu_busy         SQ_ACTIVE_INST_VALU (quad-cycles) * 4 / 1024 SIMDs / cycles
    #   wait_frac         SQ_WAIT_ANY / SQ_WAVE_CYCLES;  waves_per_simd = SQ_WAVE_CYCLES * 4 / 1024 / cycles
    for wname in ("soup1m", "statue"):
        tcp, sq, ta = (parse("%s/pmc_%s_%s.txt" % (d, wname, n)) for n in ("tcp", "sq", "ta"))
        derived = {}
        for label, key in (("trace_closest", "k_trace_w4<false"), ("trace_any", "k_trace_w4<true"), ("shade", "k_shade")):
            def pick(rows, ctr):
                return sum(v[ctr][1] for k, v in rows.items() if k.startswith(key) and ctr in v)
            cyc = pick(ta, "GRBM_GUI_ACTIVE") / 8.0
            if cyc <= 0:
                continue
            e = {"cycles": cyc, "ta_busy": pick(ta, "TA_TA_BUSY_sum") / 256.0 / cyc, "ta_busy_avr": pick(ta, "TA_BUSY_avr") / cyc}
            if pick(tcp, "TCP_TOTAL_CACHE_ACCESSES_sum"):
                e["l1_request_frac"] = pick(tcp, "TCP_TOTAL_CACHE_ACCESSES_sum") / 256.0 / cyc
                e["l1_to_l2_read_latency_cycles"] = pick(tcp, "TCP_TCC_READ_REQ_LATENCY_sum") / max(pick(tcp, "TCP_TCC_READ_REQ_sum"), 1.0)
                # round 4 (experiments/README.md): what the traversal kernel's time follows is its L1 MISSES — lines requested from L2 — and their latency
                e["l1_miss_lines"] = pick(tcp, "TCP_TCC_READ_REQ_sum")
                e["l1_hit_rate"] = 1.0 - pick(tcp, "TCP_TCC_READ_REQ_sum") / pick(tcp, "TCP_TOTAL_CACHE_ACCESSES_sum")
                e["l2_hit_rate"] = pick(tcp, "TCC_HIT_sum") / max(pick(tcp, "TCC_HIT_sum") + pick(tcp, "TCC_MISS_sum"), 1.0)
                # Little's law: misses in flight per CU = (lines / cycles) x latency / 256 CUs
                e["l1_misses_in_flight_per_cu"] = pick(tcp, "TCP_TCC_READ_REQ_LATENCY_sum") / cyc / 256.0
            if pick(sq, "SQ_WAVE_CYCLES"):
                e["valu_busy"] = pick(sq, "SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0 / cyc
                e["wait_frac"] = pick(sq, "SQ_WAIT_ANY") / pick(sq, "SQ_WAVE_CYCLES")
                e["waves_per_simd"] = pick(sq, "SQ_WAVE_CYCLES") * 4.0 / 1024.0 / cyc
            derived[label] = e
        if derived and wname in out["workloads"]:
            out["workloads"][wname]["limiters"] = derived
    for w in out["workloads"].values():   # the shade stage's fabric-side traffic, for its own roofline block
        w["shade_traffic_bytes_per_step"] = sum(v["bytes_corrected"] for k, v in w["kernels"].items() if k.startswith("k_shade") or k.startswith("k_texture") or k.startswith("k_bin_"))
    json.dump(out, open(d + "/pmc_traffic.json", "w"), indent=1)
    lines = ["# L1 / issue-side PMC counters of one step, per kernel — rocprofv3 --kernel-trace --pmc <one block per run>", "",
             "source hash %s; command: `python bench.py --workload <w> --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-count`" % out["source_hash"], ""]
    for wname, title in (("soup1m", "C2 (soup1m)"), ("statue", "C3 stand-in (statue)")):
      for name in ("tcp", "sq", "ta", "lds"):
        rows = parse("%s/pmc_%s_%s.txt" % (d, wname, name))
        if not rows:
            continue
        ctrs = sorted({c for k in rows for c in rows[k]})
        lines += ["## %s, pass `%s`" % (title, name), "", "| kernel | launches | " + " | ".join(ctrs) + " |", "|---|---|" + "---|" * len(ctrs)]
        for k in sorted(rows):
            if k.startswith("bvhdev::"):
                continue
            if name != "lds" and not (k.startswith("k_trace") or k.startswith("k_shade") or k.startswith("k_raygen") or k.startswith("k_film") or k.startswith("k_bin")):
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
