#!/bin/bash
# dev: ms per generate call of several library variants (vlibs/lib_<tag>.so) through WN_LIB_PATH; $1 = bench.py arguments
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for t in "$@"; do
  WN_LIB_PATH=$GRAFT_REPO_ROOT/vlibs/lib_$t.so python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-8s' % '$t', round(d['value']/1e6,2),'Ms/s', round(d['ms_per_step'],4),'ms  kernel us', round(r['avg_launch_us'],2))"
done
