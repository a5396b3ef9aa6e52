#!/bin/bash
# (1) the sweep check on the failing and the re-allocated variants, (2) timing of the whole library with the scalar basic allocator
O=gpurun_out/ub7; mkdir -p $O
for v in brk brk_sb brk_dump3 brk_dump3_sb def4; do
  top=$(python tools/variants.py stage $v)
  timeout 600 python tools/sweep_check.py --pkg $top > $O/check_$v.log 2> $O/check_$v.err; echo "== $v rc $?"; grep '"case"' $O/check_$v.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['case'], 'ok' if d['ok'] else 'FAIL', 'dot %.2e' % d['dot_err'], {k[15:]: float('%.1e' % v) for k, v in d.items() if k.startswith('sweep_vs_probe')})"
done
bash tools/variants_run_c3.sh cur sbasic cur sbasic
bash tools/variants_run_bwd.sh c3 cur sbasic
bash tools/variants_run_bwd.sh c5 cur sbasic
