#!/bin/bash
# per-kernel times of the C3 timed loop under settings of ONE environment variable (same box):   bash tools/ab_env.sh NAME VALUE [VALUE ...]
REPO=$(pwd); OUT=$REPO/gpurun_out/ab_env; rm -rf $OUT; mkdir -p $OUT
B="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-backward --no-config5 --no-roofline"
export TMPDIR=/tmp
NAME=$1; shift
for v in "$@"; do
    env $NAME=$v rocprofv3 --kernel-trace -d $OUT/$v -- $B > $OUT/$v.log 2>&1
    python - $v <<'PY'
import sqlite3, glob, sys
v = sys.argv[1]
for db in glob.glob("gpurun_out/ab_env/%s/**/*_results.db" % v, recursive=True):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = con.execute("select s.kernel_name, count(*), avg(d.end - d.start) / 1e3 from %s d join %s s on d.kernel_id = s.id group by 1 having count(*) > 5 order by 3 desc" % (kd, ks)).fetchall()
    print("%-10s" % v, "  ".join("%s %.1f us" % (("primary" if "ELi1EEv" in k and "k_paths" in k else ("interior" if "k_paths" in k else ("secondary" if "secondary" in k else k[:12]))), t) for k, n, t in rows if "k_" in k))
PY
done
