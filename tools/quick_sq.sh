#!/bin/bash
# one kernel-trace pass + one SQ counter pass of the C3 timed loop: per-kernel time, registers, LDS, VALU instructions and lane utilisation (quick look between builds)
REPO=$(pwd); OUT=$REPO/gpurun_out/quick; rm -rf $OUT; mkdir -p $OUT
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-backward --no-config5"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- $BENCH > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3, glob, collections
for tag in ("kt", "sq"):
    for db in glob.glob("gpurun_out/quick/%s/**/*_results.db" % tag, recursive=True):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
        if tag == "kt":
            q = "select s.kernel_name, count(*), avg(d.end - d.start) / 1e3, min(s.arch_vgpr_count), min(s.sgpr_count), min(d.group_segment_size), min(d.private_segment_size) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (kd, ks)
            for r in con.execute(q):
                if "k_" in r[0]: print("%-60s n=%3d avg %9.1f us vgpr %s sgpr %s lds %s scratch %s" % (r[0][:60], r[1], r[2], r[3], r[4], r[5], r[6]))
        else:
            pm = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
            pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
            q = "select s.kernel_name, i.name, avg(e.value) from %s e join %s i on e.pmc_id = i.id join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.kernel_name, i.name" % (pm, pi, kd, ks)
            acc = collections.defaultdict(dict)
            for k, n, v in con.execute(q): acc[k][n] = v
            for k, d in acc.items():
                if "k_" not in k: continue
                print(k[:60], {n: "%.4g" % v for n, v in d.items()}, "util %.3f" % (d.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * d.get("SQ_ACTIVE_INST_VALU", 1))))
PY
