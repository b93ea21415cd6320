timeout 800 python -m pytest tests/test_gpu_iaf.py -x -q 2>&1 | tail -3
for nh in 1 0 1 0; do
for b in 1 8; do
if [ $nh = 1 ]; then export WN_NO_HEADFUSE=1; else unset WN_NO_HEADFUSE; fi
python bench.py --batch-per-gpu $b --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('noheadfuse=$nh B=$b', d['value']/1e6, d['ms_per_step'])"
done; done
