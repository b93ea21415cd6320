#!/bin/bash
# dev: shader clock / power the part holds under a sustained run of one workload (rocm-smi sampled beside it)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
ls /sys/class/drm/ | head; for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $h; ls $h | tr '\n' ' '; echo; done 2>/dev/null | head -20
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -v "^$" | head -30
run() {  # tag, command...
  tag=$1; shift
  "$@" > gpurun_out/pw_$tag.log 2>&1 &
  pid=$!
  sleep 4
  for i in 1 2 3 4 5 6; do
    rocm-smi -d 0 --showpower --showclocks --json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=next(iter(d.values()))
print('$tag', {k:v for k,v in c.items() if 'sclk' in k.lower() or 'power' in k.lower() or 'mclk' in k.lower()})"
    sleep 0.5
  done
  wait $pid
  tail -1 gpurun_out/pw_$tag.log | cut -c1-200
}
run default python scripts/dev_abl_bench.py --tag default --steps 8000
WN_PRECISION=f32 run f32 python scripts/dev_abl_bench.py --tag f32 --steps 2500
WN_NO_GROUPS=1 run perlayer python scripts/dev_abl_bench.py --tag perlayer --steps 7000
run b8 python scripts/dev_abl_bench.py --tag b8 --batch 8 --steps 1000
