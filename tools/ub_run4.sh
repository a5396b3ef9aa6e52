#!/bin/bash
O=gpurun_out/ub4; mkdir -p $O
timeout 600 python tools/variants.py run brk python tools/adj_mask.py 2 80 20 > $O/mask_r2.log 2>&1
timeout 600 python tools/variants.py run brk python tools/adj_mask.py 2 720 13 > $O/mask_r18.log 2>&1
python tools/variants.py restore
cat $O/mask_r2.log $O/mask_r18.log
