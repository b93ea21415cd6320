#!/bin/bash
# round 6: the whole GPU suite + the driver-protocol bench line (all extras)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out/r06d
O=gpurun_out/r06d
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1
echo "exit $?" >> $O/gpu_tests.log
tail -8 $O/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
tail -3 $O/bench_driver.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06d/bench_driver.json').read().strip().splitlines()[-1])
print('value %.2f M  ms %.4f  ramp %s' % (d['value']/1e6, d['ms_per_step'], d['ramp_steps']))
print('sustained', d.get('power',{}).get('ms_per_step_sustained'))
r=d.get('roofline_f32',{}); print('f32: ms %.3f frac %.3f layer us %.1f' % (r.get('ms_per_step',0), r.get('path_frac_of_f32_mfma_peak',0), r.get('avg_launch_us',0)))
for k in ('ar_b1','ar_b64','ar_b256','ar_b64_s2'):
    v=d.get(k,{}); print(k, v.get('samples_per_sec'), v.get('us_per_sample_step'), v.get('error'))
print('cli_e2e', json.dumps(d.get('cli_e2e'))[:600])
for v in d.get('roofline_f16x2',{}).get('variants',[]):
    print('f16x2', v.get('library'), 'err full', v.get('full_size_max_abs_err'), 'flips', v.get('full_size_index_flips'), 'ms', v.get('ms_per_call_sustained'), 'J', v.get('J_per_call'), 'unit-case', (v.get('small_cases') or {}).get('iaf_logistic_unit'), v.get('error'))
print('b8', d.get('roofline_b8',{}).get('samples_per_sec'))
print('cpu', d.get('cpu_baseline'))
PY
