#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
for t in g0 g16 nh3 nh3g16; do
  WN_LIB_PATH=$GRAFT_REPO_ROOT/vlibs/lib_$t.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --precision f32 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$t: %.3f ms, path frac %.3f, layer %.1f us' % (d['ms_per_step'], d['config']['path_achieved_tflops']/157.3, r['avg_launch_us']))"
done
bash scripts/dev_vlib_kstats.sh "f32" "--precision f32" g0 g16
bash scripts/dev_vlib_kstats.sh "iaf_layer" "--precision f32" nh3
