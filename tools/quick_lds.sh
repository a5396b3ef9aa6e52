#!/bin/bash
# LDS pressure of the C3 kernels: instructions, active / conflict / wait cycles (two PMC passes of the timed loop)
REPO=$(pwd); OUT=$REPO/gpurun_out/quick_lds; rm -rf $OUT; mkdir -p $OUT
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-backward --no-config5"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $OUT/a -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES -d $OUT/b -- $BENCH > $OUT/b.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3, glob, collections
for tag in ("a", "b"):
    for db in glob.glob("gpurun_out/quick_lds/%s/**/*_results.db" % tag, recursive=True):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
        pm = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
        pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
        q = "select s.kernel_name, i.name, avg(e.value), count(*) from %s e join %s i on e.pmc_id = i.id join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.kernel_name, i.name" % (pm, pi, kd, ks)
        acc = collections.defaultdict(dict)
        for k, n, v, c in con.execute(q):
            if c > 2: acc[k][n] = v
        for k, d in acc.items():
            if "k_" in k: print(k[:44], {n: "%.4g" % v for n, v in sorted(d.items())})
PY
