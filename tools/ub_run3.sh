#!/bin/bash
O=gpurun_out/ub3; mkdir -p $O
timeout 900 python tools/variants.py run brk python tools/adj_pairs.py 2 100 > $O/pairs_100.log 2>&1
timeout 900 python tools/variants.py run brk python tools/adj_pairs.py 2 748 > $O/pairs_748.log 2>&1
python tools/variants.py restore
cat $O/pairs_100.log $O/pairs_748.log
