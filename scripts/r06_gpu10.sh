#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
timeout 900 python -m pytest tests/test_gpu_iaf.py::test_golden_vectors "tests/test_ref_float.py::test_engine_student_against_the_reference_code" tests/test_ref_float.py::test_fp32_engine_equals_the_reference_code_at_full_size tests/test_gpu_iaf.py::test_fp32_forms_hoisted_against_fused_and_the_frame_axis_upsampler_against_the_phase_major_one -x -q -m gpu -k "f32 or fp32" 2>&1 | tail -3
for t in g0 pipe g0 pipe; do
  WN_LIB_PATH=$GRAFT_REPO_ROOT/vlibs/lib_$t.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --precision f32 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$t: %.3f ms, path frac %.3f, layer %.1f us' % (d['ms_per_step'], d['config']['path_achieved_tflops']/157.3, r['avg_launch_us']))"
done
bash scripts/dev_vlib_kstats.sh "iaf_layer" "--precision f32" g0 pipe
