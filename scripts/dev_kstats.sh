#!/bin/bash
# dev: per-kernel average duration of one bench run under rocprofv3 ($1 = tag, rest = env assignments / bench args)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$TAG -o ks -- python $R/bench.py --steps 20 --warmup 3 --ramp-steps 0 --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/ks_$TAG/ks_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows if not r["Name"].startswith("void at::") and "rocclr" not in r["Name"])
for r in rows[:9]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    print("$TAG", n.ljust(44), r["Calls"].rjust(5), "%9.1f us avg" % (float(r["AverageNs"]) / 1e3), "%5.1f%%" % (100 * float(r["TotalDurationNs"]) / tot))
print("$TAG total per step us:", round(tot / 23 / 1e3, 1))
PY
