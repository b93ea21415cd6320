#!/bin/bash
# dev: average duration of one kernel ($1 = name substring) under rocprofv3 for several library variants (vlibs/lib_<tag>.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; KERN=$1; shift; ARGS=$1; shift
for t in "$@"; do
  WN_LIB_PATH=$R/vlibs/lib_$t.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vk_$t -o ks -- python $R/bench.py --steps 20 --warmup 3 --ramp-steps 0 --no-cpu-baseline --no-extras $ARGS > /dev/null 2>&1
  python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/vk_$t/ks_kernel_stats.csv")))
for r in rows:
    if "$KERN" in r["Name"]:
        print("%-8s %-40s %6s calls %9.2f us avg (min %.2f max %.2f)" % ("$t", r["Name"].replace("(anonymous namespace)::","")[:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  rm -rf $R/gpurun_out/vk_$t
done
