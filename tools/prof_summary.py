#!/usr/bin/env python3
"""Turns the rocprofv3 rocpd databases under gpurun_out/prof_* into the text summaries that are
committed under profiles/ (kernel-trace stats + per-kernel PMC averages) and into
profiles/hbm_traffic.json, which bench.py reads for roofline.traffic.

    python tools/prof_summary.py <round-tag>          # e.g. r01

HBM bytes per launch follow MI355X_MICROARCH.md §HBM: separate --pmc passes for FETCH_SIZE and
WRITE_SIZE (KiB); on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, so it is doubled:
    traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024   [bytes per launch, averaged over launches]
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


PREFIX = ""


def db_of(d):
    d = PREFIX + d.replace("prof_", "") if PREFIX else d
    f = glob.glob(os.path.join(OUT, d, "**", "*_results.db"), recursive=True)
    f.sort(key=os.path.getmtime)
    return sqlite3.connect(f[-1]) if f else None


def kernel_stats(tag):
    db = db_of("prof_kt")
    if db is None:
        return
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    det = db.execute("select name, min(duration), max(duration), avg(grid_x), avg(workgroup_x), avg(lds_size), avg(vgpr_count), avg(sgpr_count), "
                     "avg(scratch_size) from kernels group by name").fetchall()
    det = {r[0]: r[1:] for r in det}
    with open(os.path.join(PROF, "%s_kernel_trace_stats.txt" % tag), "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline   (MI355X, gfx950)\n")
        fh.write("# durations in us; grid = work-items\n")
        fh.write("%-46s %6s %12s %10s %10s %10s %6s %9s %6s %8s %5s %5s\n" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "grid", "wg", "lds_B", "vgpr", "sgpr"))
        for name, calls, total, avg, pct in rows:
            mn, mx, gx, wx, lds, vg, sg, scr = det.get(name, (0,) * 8)
            fh.write("%-46s %6d %12.1f %10.1f %10.1f %10.1f %6.2f %9d %6d %8d %5d %5d\n" %
                     (short(name)[:46], calls, total, avg, mn / 1e3, mx / 1e3, pct, gx, wx, lds, vg, sg))     # top_kernels is in us, kernels.duration in ns


def pmc(tag):
    per = {}
    for d in ("prof_fetch", "prof_write", "prof_l2", "prof_sq", "prof_sq2"):
        db = db_of(d)
        if db is None:
            continue
        for name, counter, val, n in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
            per.setdefault(short(name), {})[counter] = (val, n)
    if not per:
        return
    traffic = {}
    with open(os.path.join(PROF, "%s_pmc_summary.txt" % tag), "w") as fh:
        fh.write("# rocprofv3 --pmc <counters> --kernel-trace (one pass per counter group), averages per launch (MI355X, gfx950)\n")
        for k, cs in sorted(per.items()):
            if not k.startswith("k_"):
                continue
            fh.write("\n%s\n" % k)
            for c, (v, n) in sorted(cs.items()):
                fh.write("    %-28s %18.3f   (n=%d)\n" % (c, v, n))
            f, w = cs.get("FETCH_SIZE", (None, 0))[0], cs.get("WRITE_SIZE", (None, 0))[0]
            if f is not None and w is not None:
                t = 2.0 * f * 1024 + w * 1024
                fh.write("    %-28s %18.0f   bytes/launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024\n" % ("HBM traffic", t))
                key = {"k_interior<true, true, false>": "k_interior<AD>", "k_primary_edges<true, false>": "k_primary_edges",
                       "k_paths<true, true, false, 0>": "k_interior<AD>", "k_paths<false, true, false, 1>": "k_primary_edges",
                       "k_secondary_edges<true, false>": "k_secondary_edges", "k_secondary_edges<true, false, false>": "k_secondary_edges",
                       # scene classes (int template parameter since r01i): class 1 = the LDS kernels bench.py runs
                       "k_paths<true, 1, false, 0>": "k_interior<AD>", "k_paths<false, 1, false, 1>": "k_primary_edges",
                       "k_secondary_edges<1, false, false>": "k_secondary_edges"}.get(k, k)
                traffic[key] = t
            tc, ai = cs.get("SQ_THREAD_CYCLES_VALU", (None, 0))[0], cs.get("SQ_ACTIVE_INST_VALU", (None, 0))[0]
            if tc is not None and ai:
                fh.write("    %-28s %18.4f   SQ_THREAD_CYCLES_VALU / (64*SQ_ACTIVE_INST_VALU)\n" % ("VALU lane utilisation", tc / (64.0 * ai)))
            h, m = cs.get("TCC_HIT_sum", (None, 0))[0], cs.get("TCC_MISS_sum", (None, 0))[0]
            if h is not None and m is not None and h + m > 0:
                fh.write("    %-28s %18.4f   TCC_HIT/(TCC_HIT+TCC_MISS)\n" % ("L2 hit rate", h / (h + m)))
    if not PREFIX:          # only the bench.py profile (tools/profile.sh) feeds bench.py's roofline.traffic
        with open(os.path.join(PROF, "hbm_traffic.json"), "w") as fh:
            json.dump(traffic, fh, indent=1)


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    if len(sys.argv) > 2:
        PREFIX = sys.argv[2]
    os.makedirs(PROF, exist_ok=True)
    kernel_stats(tag)
    pmc(tag)
    print(open(os.path.join(PROF, "%s_kernel_trace_stats.txt" % tag)).read())
    p = os.path.join(PROF, "%s_pmc_summary.txt" % tag)
    if os.path.exists(p):
        print(open(p).read())
