#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
run() { # tag env lib
  env $2 WN_LIB_PATH=$GRAFT_REPO_ROOT/vlibs/lib_$3.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --precision f32 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-34s %.3f ms, path frac %.3f, layer %.1f us' % ('$1', d['ms_per_step'], d['config']['path_achieved_tflops']/157.3, r['avg_launch_us']))"
}
for rep in 1 2; do
run "one GEMM, nt C"            "A=1" pf
run "per-flow GEMMs, nt C"      "WN_F32_PERFLOW=1" pf
run "one GEMM, cached C"        "A=1" pfc
run "per-flow GEMMs, cached C"  "WN_F32_PERFLOW=1" pfc
done
