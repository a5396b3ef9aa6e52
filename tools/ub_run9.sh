O=gpurun_out/ub9; mkdir -p $O
for v in "$@"; do
  top=$(python tools/variants.py stage $v)
  timeout 600 python tools/sweep_check.py --pkg $top > $O/check_$v.log 2> $O/check_$v.err; echo "== $v rc $?"; grep '"case"' $O/check_$v.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['case'], 'ok' if d['ok'] else 'FAIL', 'dot %.2e' % d['dot_err'], {k[15:]: float('%.1e' % v) for k, v in d.items() if k.startswith('sweep_vs_probe')})"
done
