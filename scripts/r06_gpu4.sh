#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
timeout 900 python -m pytest tests/test_gpu_iaf.py::test_golden_vectors "tests/test_ref_float.py::test_engine_student_against_the_reference_code" -x -q -m gpu -k "f32" 2>&1 | tail -3
bash scripts/dev_vlib_kstats.sh "iaf_" "--precision f32" base nh1
for t in st1 st1n1; do echo "== $t"; WN_LIB_PATH=$GRAFT_REPO_ROOT/vlibs/lib_$t.so python scripts/dev_f32_stamps.py 2>&1 | tail -9; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --precision f32 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('f32: value %.2f M, %.3f ms, path frac of f32 peak %.3f, layer %.1f us (frac %.3f)' % (d['value']/1e6, d['ms_per_step'], d['config']['path_achieved_tflops']/157.3, r['avg_launch_us'], r['frac']))"
