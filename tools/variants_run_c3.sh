for v in "$@"; do
  python tools/variants.py run $v python bench.py --no-cpu-baseline --no-parity --no-backward --no-config5 > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/var_%s.json"%v).read().strip().splitlines()[-1])
    print(v, d["ms_per_step"], [(k["kernel"], round(k["avg_launch_ms"],3)) for k in d["roofline"]["kernels"]])
except Exception as e:
    print(v, "failed", e, open("gpurun_out/var_%s.err"%v).read()[-300:])
PY
done
