# on the GPU box: times the named variants (tools/variants.py build NAME "...") on config 5 at full size, each under a timeout
for v in "$@"; do
  timeout 150 python tools/variants.py run $v python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-backward > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
  echo "rc $?"
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/var_%s.json"%v).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(v, d["ms_per_step"], "nodes/ray %.2f tris/ray %.2f" % (r["nodes_per_ray"], r["tris_per_ray"]), [(k["kernel"], round(k["avg_launch_ms"],2)) for k in r["kernels"]])
except Exception as e:
    print(v, "failed", e, open("gpurun_out/var_%s.err"%v).read()[-300:])
PY
done
python tools/variants.py restore
