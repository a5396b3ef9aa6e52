#!/bin/bash
# round-4 hunt for the codegen-dependent result of the class-2 reverse sweep: sweep vs probe per variant, then the bisect dump of the failing one
O=gpurun_out/ub1; mkdir -p $O
for v in "$@"; do
  timeout 300 python tools/variants.py run $v python tools/dp_env_case.py > $O/$v.sweep.log 2>&1
  PSDR_ADJ_PROBE=1 timeout 300 python tools/variants.py run $v python tools/dp_env_case.py > $O/$v.probe.log 2>&1
  echo "== $v"; grep -h "^box_x\|^albedo" $O/$v.sweep.log; echo "-- probe"; grep -h "^box_x\|^albedo" $O/$v.probe.log
done
v=$1
timeout 600 python tools/variants.py run $v python tools/adj_bisect.py dump $O/${v}_sweep.npz box_x > $O/bisect_sweep.log 2>&1
PSDR_ADJ_PROBE=1 timeout 900 python tools/variants.py run $v python tools/adj_bisect.py dump $O/${v}_probe.npz box_x > $O/bisect_probe.log 2>&1
python tools/adj_bisect.py cmp $O/${v}_sweep.npz $O/${v}_probe.npz > $O/bisect_cmp.txt 2>&1
tail -5 $O/bisect_sweep.log $O/bisect_probe.log
cat $O/bisect_cmp.txt | head -150
python tools/variants.py restore
