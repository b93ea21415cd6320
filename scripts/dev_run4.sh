timeout 600 python -m pytest tests/test_gpu_iaf.py -x -q -k "golden or groups or batch8 or ragged" 2>&1 | tail -3
bash scripts/dev_ab.sh -r 3 tilemajor base 2>&1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for t in tilemajor base; do
  if [ $t = base ]; then unset WN_LIB_PATH; else export WN_LIB_PATH=$GRAFT_REPO_ROOT/vlibs/lib_$t.so; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fetch_$t -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/fetch_$t/*counter_collection.csv")[0]
tot, n = 0.0, 0
for row in csv.DictReader(open(f)):
    if 'iaf_cond_h_kernel' in row['Kernel_Name'] and row['Counter_Name'] == 'FETCH_SIZE':
        tot += float(row['Counter_Value']); n += 1
print("$t", 'iaf_cond_h_kernel FETCH_SIZE per launch (KiB)', round(tot / max(n, 1), 1), 'over', n, 'launches  -> x2 =', round(2 * tot / max(n, 1) / 1024, 1), 'MB')
PY
done
