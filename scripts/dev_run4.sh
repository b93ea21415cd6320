timeout 600 python -m pytest tests/test_gpu_ar.py -x -q 2>&1 | tail -3
for r in 1 2 3; do
  WN_AR_TAIL3=1 python bench_aux.py --workload ar --batch 1 --samples 1600 --steps 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ar B=1 three-launch tail', round(d['config']['us_per_sample_step'],2), 'us/step')"
  python bench_aux.py --workload ar --batch 1 --samples 1600 --steps 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ar B=1 merged tail      ', round(d['config']['us_per_sample_step'],2), 'us/step')"
done
