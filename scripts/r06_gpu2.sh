#!/bin/bash
# round 6, second GPU pass: the fp32 hoisted form -- parity first, then time and the kernel breakdown
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 1500 python -m pytest tests/test_gpu_iaf.py::test_golden_vectors "tests/test_ref_float.py::test_engine_student_against_the_reference_code" tests/test_gpu_teacher.py::test_forward_with_fp32_upsampler_handle -x -q -m gpu -k "f32" > $O/tests_f32.log 2>&1
echo "f32 tests exit $?" >> $O/tests_f32.log
tail -15 $O/tests_f32.log
for prec in f32 f32-fused; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --precision $prec > $O/bench_$prec.json 2> $O/bench_$prec.err
  python - <<PY
import json
d=json.loads(open('$O/bench_$prec.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$prec', 'value %.2f M' % (d['value']/1e6), 'ms %.3f' % d['ms_per_step'], 'path frac of f32 peak %.3f' % (d['config']['path_achieved_tflops']/157.3), 'layer us %.1f' % r['avg_launch_us'], 'frac %.3f' % r['frac'])
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_f32 -o f32 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --precision f32 > $GRAFT_REPO_ROOT/$O/prof_f32.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_f32 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_f32.csv
head -12 $O/kernel_stats_f32.csv | cut -c1-170
rm -rf $O/prof_f32
timeout 100 scripts/ubench/mfma_f32_power auto 2.5 > $O/mfma_f32_power.txt 2>&1
cat $O/mfma_f32_power.txt
