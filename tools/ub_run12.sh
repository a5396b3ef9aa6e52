# what does the AD interior kernel of C3 fetch?  FETCH_SIZE / WRITE_SIZE / L2 hits of the kernel alone, with and without its output atomics (variants c1 / c1_noat)
O=$(pwd)/gpurun_out/ub12; mkdir -p $O; REPO=$(pwd)
for v in c1 c1_noat; do
  top=$(python tools/variants.py stage $v)
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "TCC_ATOMIC_sum TCC_EA_ATOMIC_sum"; do
    d=$O/${v}_$(echo $c | tr ' ' '_'); rm -rf $d
    (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc $c -d $d -- python $REPO/tools/terms_only.py --pkg $top 1 5 > $d.log 2>&1)
  done
done
python - <<'PY'
import glob, csv, os, collections
O = "gpurun_out/ub12"
for d in sorted(glob.glob(O + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void psdr::k_paths") or "k_paths" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(os.path.basename(d.rstrip("/")), k, "avg per launch %.4g (n=%d)" % (sum(v) / len(v), len(v)))
PY
