#!/bin/bash
# L1-path counters (TA / TD / TCP / SQ, two or three of a block per pass - more abort the profiler on this pool) of the BVH kernels on BASELINE config 5 at 512 x 512 x 16,
# terms serialised (PSDR_NO_FORK=1).  On the GPU box, from the repo root:
#     bash tools/l1_counters.sh <tag> [variant]        # variant: a library under _dev/variants (tools/variants.py), default: the shipped one
# -> gpurun_out/l1_<tag>.txt  (copy to profiles/ to keep)
set -u
TAG=$1; VAR=${2:-}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python bench.py --config 5 --res 512 --spp 16 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-backward --no-roofline --no-api --no-static-skip"
[ -n "$VAR" ] && RUN="python tools/variants.py run $VAR $RUN"
CMD="cd $REPO && PSDR_NO_FORK=1 $RUN"
run() { tag=$1; shift; rm -rf $OUT/x_$tag; timeout -s KILL 150 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/x_$tag -- bash -c "$CMD" > $OUT/x_$tag.log 2>&1 || echo "pass $tag failed"; }
run a GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum
run b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run c TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run e TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run f TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run g TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum
run h SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU
run i SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
run k SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES
cd $REPO
[ -n "$VAR" ] && python tools/variants.py restore
python - "$TAG" "$VAR" <<'PY' > $OUT/l1_$TAG.txt
import sqlite3, glob, os, sys
tag, var = sys.argv[1], sys.argv[2]
per, dur = {}, {}
for t in "abcefghik":
    f = glob.glob(os.path.join("gpurun_out", "x_" + t, "**", "*_results.db"), recursive=True)
    if not f: print("# pass", t, ": no database"); continue
    db = sqlite3.connect(f[-1])
    try:
        for name, counter, val, n in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
            per.setdefault(name.split("(")[0].replace("void ", "")[:44], {})[counter] = val
        for name, avg in db.execute("select name, avg(duration) from kernels group by name"):
            dur.setdefault(name.split("(")[0].replace("void ", "")[:44], []).append(avg / 1e3)
    except Exception as e:
        print("# pass", t, "error", e)
print("# rocprofv3 --kernel-trace --pmc <two or three counters per pass>, BASELINE config 5 at 512 x 512 x 16, terms serialised (PSDR_NO_FORK=1); library: %s" % (var or "shipped"))
print("# averages per launch, summed over the device's units (_sum); us = the kernel's mean duration over the passes")
for k, cs in sorted(per.items()):
    if "k_paths" not in k and "k_secondary" not in k: continue
    print("\n%s   %.1f us" % (k, sum(dur.get(k, [0])) / max(1, len(dur.get(k, [])))))
    for c in sorted(cs): print("    %-44s %.5g" % (c, cs[c]))
    g = lambda n: cs.get(n)
    if g("TCP_PENDING_STALL_CYCLES_sum") and g("TCP_GATE_EN1_sum"): print("    %-44s %.3f" % ("TCP_PENDING_STALL / TCP_GATE_EN1", g("TCP_PENDING_STALL_CYCLES_sum") / g("TCP_GATE_EN1_sum")))
    # (per-CU clocks of the launch: TCP_GATE_EN1_sum - the L1s are clocked for the whole launch; GRBM_GUI_ACTIVE is summed over the 8 XCDs, x 32 CUs each gives the same figure)
    if g("TD_TD_BUSY_sum") and g("TCP_GATE_EN1_sum"): print("    %-44s %.3f   %-26s %.3f" % ("TD_TD_BUSY / TCP_GATE_EN1", g("TD_TD_BUSY_sum") / g("TCP_GATE_EN1_sum"), "TA_TA_BUSY / TCP_GATE_EN1", (g("TA_TA_BUSY_sum") or 0) / g("TCP_GATE_EN1_sum")))
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"): print("    %-44s %.3f" % ("VALU lane utilisation", g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU"))))
    if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"): print("    %-44s %.3f" % ("SQ_WAIT_ANY / SQ_WAVE_CYCLES", g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")))
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("SQ_INSTS_VMEM_RD"): print("    %-44s %.2f" % ("cache-line accesses per vector load", g("TCP_TOTAL_CACHE_ACCESSES_sum") / g("SQ_INSTS_VMEM_RD")))
    if g("TCP_TCC_READ_REQ_sum") and g("TCP_TOTAL_CACHE_ACCESSES_sum"): print("    %-44s %.3f" % ("L1 miss rate (TCC_READ_REQ / accesses)", g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")))
    if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum"): print("    %-44s %.0f clocks" % ("mean L2 read latency", g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")))
PY
cat $OUT/l1_$TAG.txt | head -80
