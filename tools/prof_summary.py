#!/usr/bin/env python3
"""Turns the rocprofv3 rocpd databases under gpurun_out/prof_* into the text summaries that are
committed under profiles/ (kernel-trace stats + per-kernel PMC averages) and into
profiles/counters.json, which bench.py reads for roofline.traffic / roofline.issue (PMC counters cannot be
collected from inside the benchmarked process; the JSON names the summary file it was made from).

    python tools/prof_summary.py <round-tag>                                  # e.g. r01: the bench.py passes of tools/profile.sh
    python tools/prof_summary.py <tag> <dir prefix> [section] [workload]       # the passes of tools/profile_cmd.sh <prefix> ...; a section
                                                                               # name (e.g. config5) adds them to counters.json under that key

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
SECTION = None
WORKLOAD = None


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
        fh.write("# rocprofv3 --kernel-trace --stats   (MI355X, gfx950; command: tools/profile.sh, passes %s*)\n" % (PREFIX or "prof_"))
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
                traffic[k] = t
            tc, ai = cs.get("SQ_THREAD_CYCLES_VALU", (None, 0))[0], cs.get("SQ_ACTIVE_INST_VALU", (None, 0))[0]
            if tc is not None and ai:
                fh.write("    %-28s %18.4f   SQ_THREAD_CYCLES_VALU / (64*SQ_ACTIVE_INST_VALU)\n" % ("VALU lane utilisation", tc / (64.0 * ai)))
            h, m = cs.get("TCC_HIT_sum", (None, 0))[0], cs.get("TCC_MISS_sum", (None, 0))[0]
            if h is not None and m is not None and h + m > 0:
                fh.write("    %-28s %18.4f   TCC_HIT/(TCC_HIT+TCC_MISS)\n" % ("L2 hit rate", h / (h + m)))
    if not PREFIX:          # the bench.py profile (tools/profile.sh) feeds bench.py's roofline.traffic / roofline.issue ...
        write_counters_json(tag, per, traffic)
    elif SECTION:           # ... and a named section (config5) its config-5 object
        write_counters_json(tag, per, traffic, SECTION, WORKLOAD)


def bench_name(k):
    """kernel template instantiation -> the name bench.py prints (un-instrumented forward kernels only)"""
    import re
    m = re.match(r"k_paths<(\w+), (\d+), (\w+), (\d+)>", k)
    if m and m.group(3) == "false":
        return "k_primary_edges" if m.group(4) == "1" else ("k_interior<AD>" if m.group(1) == "true" else "k_interior")
    m = re.match(r"k_secondary_edges<(\d+), (\w+), (\w+)>", k)
    if m and m.group(2) == "false" and m.group(3) == "false":
        return "k_secondary_edges"
    return None


def write_counters_json(tag, per, traffic, section=None, workload=None):
    db = db_of("prof_kt")
    avg_us, refused = {}, {}
    if db is not None:
        spread = {short(r[0]): (r[1], r[2]) for r in db.execute("select name, min(duration), max(duration) from kernels group by name")}
        for name, avg in db.execute("select name, average from top_kernels"):
            mn, mx = spread.get(short(name), (0, 0))
            # a kernel whose launches differ by more than 10 % in one pass is not ONE kind of launch (launches that overlap on forked streams, skipping next to
            # non-skipping ones, a warm-up of another size): its average is no per-launch duration, and nothing is derived from it
            if mn > 0 and mx > 1.10 * mn:
                refused[short(name)] = "min %.1f us, max %.1f us: launches of this pass differ by more than 10 %%, no per-launch duration taken" % (mn / 1e3, mx / 1e3)
                sys.stderr.write("prof_summary: %s refused (%s)\n" % (short(name), refused[short(name)]))
                continue
            avg_us[short(name)] = avg
    out = {"source": "profiles/%s_pmc_summary.txt + profiles/%s_kernel_trace_stats.txt (rocprofv3 --pmc / --kernel-trace --stats passes, tools/profile.sh)" % (tag, tag), "kernels": {}}
    if workload:
        out["workload"] = workload
    for k, cs in per.items():
        b = bench_name(k)
        if b is None:
            continue
        rec = {"instantiation": k}
        if k in traffic:
            rec["hbm_bytes_per_launch"] = traffic[k]
        g = lambda c: cs.get(c, (None, 0))[0]
        iv, tc, ai, wc = g("SQ_INSTS_VALU"), g("SQ_THREAD_CYCLES_VALU"), g("SQ_ACTIVE_INST_VALU"), g("SQ_WAVE_CYCLES")
        h, m = g("TCC_HIT_sum"), g("TCC_MISS_sum")
        if h is not None and m is not None and h + m > 0:
            rec["l2_hit_rate"] = h / (h + m)
        if k in avg_us:
            rec["avg_launch_us"] = avg_us[k]
        if k in refused:
            rec["avg_launch_us_refused"] = refused[k]
        if iv is not None and k in avg_us:
            slots = 1024 * 2.4e9 / 2.0 * avg_us[k] * 1e-6          # 1024 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz
            rec["issue"] = {"SQ_INSTS_VALU": iv, "avg_launch_us": avg_us[k], "valu_issue_slots": slots, "valu_issue_frac": iv / slots,
                            "lane_utilisation": (tc / (64.0 * ai)) if (tc is not None and ai) else None, "SQ_WAVE_CYCLES": wc,
                            "SQ_WAIT_ANY": g("SQ_WAIT_ANY")}
        out["kernels"][b] = rec
    path = os.path.join(PROF, "counters.json")
    if section:
        try:
            whole = json.load(open(path))
        except Exception:
            whole = {}
        whole[section] = out
        out = whole
    else:
        try:                                   # keep the named sections of an earlier run until they are rewritten
            old = json.load(open(path))
            for key, val in old.items():
                if key not in ("source", "kernels", "workload"):
                    out[key] = val
        except Exception:
            pass
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    if len(sys.argv) > 2:
        PREFIX = sys.argv[2]
    SECTION = sys.argv[3] if len(sys.argv) > 3 else None
    WORKLOAD = sys.argv[4] if len(sys.argv) > 4 else None
    os.makedirs(PROF, exist_ok=True)
    kernel_stats(tag)
    pmc(tag)
    print(open(os.path.join(PROF, "%s_kernel_trace_stats.txt" % tag)).read())
    p = os.path.join(PROF, "%s_pmc_summary.txt" % tag)
    if os.path.exists(p):
        print(open(p).read())
