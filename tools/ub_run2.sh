#!/bin/bash
O=gpurun_out/ub2; mkdir -p $O
timeout 600 python tools/variants.py run brk python tools/adj_subset.py 2 2 > $O/subset_d2_r2.log 2>&1
timeout 600 python tools/variants.py run brk python tools/adj_subset.py 2 18 > $O/subset_d2_r18.log 2>&1
timeout 600 python tools/variants.py run brk python tools/adj_subset.py 3 2 > $O/subset_d3_r2.log 2>&1
python tools/variants.py restore
tail -n 30 $O/subset_d2_r2.log $O/subset_d2_r18.log $O/subset_d3_r2.log
