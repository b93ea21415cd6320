#!/bin/bash
# round 6, first GPU pass: the new tests, the driver-protocol bench line with the clock settle, the fp32 MFMA power ubench,
# AR utterance groups on streams
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out/r06a
O=gpurun_out/r06a
timeout 900 python -m pytest tests/test_gpu_threads.py tests/test_gpu_teacher.py::test_scoring_picks_the_right_class_on_both_sides_of_every_mu_law_bin_edge "tests/test_gpu_iaf.py::test_part_timing_aid_accounts_for_the_call_and_leaves_results_alone" tests/test_integration_stub.py -x -q -m gpu -s > $O/tests_new.log 2>&1
echo "new tests exit $?" >> $O/tests_new.log
tail -5 $O/tests_new.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06a/bench_driver.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'ramp', d['ramp_steps'], d['ramp_reason'])
print('sustained', d.get('power',{}).get('ms_per_step_sustained'))
print('f32', d.get('roofline_f32',{}).get('ms_per_step'), d.get('roofline_f32',{}).get('path_frac_of_f32_mfma_peak'))
print('ar_b64', d.get('ar_b64',{}).get('samples_per_sec'), d.get('ar_b64',{}).get('us_per_sample_step'))
PY
HW=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
echo "hwmon $HW"
timeout 120 scripts/ubench/mfma_f32_power $HW 2.5 > $O/mfma_f32_power.txt 2>&1
cat $O/mfma_f32_power.txt
for cfgs in "64 1" "64 2" "64 4" "256 1" "256 4" "256 8" "16 4" "8 2" "8 8"; do
  set -- $cfgs
  timeout 300 python bench_aux.py --workload ar --batch $1 --streams $2 --samples 800 --steps 3 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    ln=ln.strip()
    if ln.startswith('{'):
        d=json.loads(ln); c=d['config']
        print('ar batch %s streams %s: %.0f samples/s aggregate, %.1f us per sample step, %.2fx RT aggregate' % ('$1','$2', d['value'], c['us_per_sample_step'], d['value']/16000.0))
" | tee -a $O/ar_streams.txt
done
