cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in 1 8; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pair$b -- python $R/bench.py --precision f16x3-hoisted --batch-per-gpu $b --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pair$b.log 2>&1
done
cd $R
python - <<'PY'
import csv,glob
for b in (1,8):
    f=sorted(glob.glob(f'gpurun_out/pair{b}/**/*kernel_stats.csv',recursive=True))[-1]
    print('B',b)
    for r in list(csv.DictReader(open(f)))[:8]:
        n=r['Name']; n=n[n.find('::')+2:][:28]
        print(f"  {n:30s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
for b in 1 2 8; do
python bench.py --precision f16x3-hoisted --batch-per-gpu $b --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('B=$b', d['value']/1e6, d['ms_per_step'], d['roofline']['frac'])"
done
timeout 600 python -m pytest tests/test_gpu_iaf.py -x -q 2>&1 | tail -2
