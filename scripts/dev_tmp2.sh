for mr in 2 3 4 5; do
for b in 1 2; do
WN_PAIR_MINRUN=$mr python bench.py --precision f16x3-hoisted --batch-per-gpu $b --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('minrun=$mr B=$b', d['value']/1e6, d['ms_per_step'])"
done
done
WN_NO_PAIR=1 python bench.py --precision f16x3-hoisted --batch-per-gpu 1 --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('nopair B=1', d['value']/1e6, d['ms_per_step'])"
python bench.py --batch-per-gpu 1 --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fused B=1', d['value']/1e6, d['ms_per_step'])"
