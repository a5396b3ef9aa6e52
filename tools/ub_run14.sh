# C3 primary-edge kernel alone: are the ~0.9 GB of HBM-side traffic per launch spills or the float atomics of the derivative image?
O=$(pwd)/gpurun_out/ub14; mkdir -p $O; REPO=$(pwd)
for c in "WRITE_SIZE" "FETCH_SIZE" "TCC_ATOMIC_sum TCC_EA_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum"; do
  d=$O/$(echo $c | tr ' ' '_'); rm -rf $d
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc $c -d $d -- python $REPO/tools/terms_only.py 2 5 > $d.log 2>&1)
done
