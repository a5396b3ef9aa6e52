#!/bin/bash
O=gpurun_out/ub5; mkdir -p $O
for v in brk_dump def_dump; do
  timeout 600 python tools/variants.py run $v python tools/adj_dump.py 2 99 20 19 $O/$v.r2.npy > $O/$v.r2.log 2>&1
  timeout 600 python tools/variants.py run $v python tools/adj_dump.py 2 732 13 12 $O/$v.r18.npy > $O/$v.r18.log 2>&1
done
timeout 600 python tools/variants.py run brk python tools/adj_mask.py 2 80 20 > $O/brk_mask.log 2>&1
python tools/variants.py restore
for v in brk_dump def_dump; do echo "=== $v"; cat $O/$v.r2.log $O/$v.r18.log; done
grep -c DIFFERS $O/brk_mask.log
