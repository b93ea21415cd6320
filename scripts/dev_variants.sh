timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for b in 1 8; do for p in f16x3-fused f16x3-hoisted; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch-per-gpu $b --precision $p 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$b $p', round(d['value']/1e6,2),'Ms/s', round(d['ms_per_step'],3),'ms', d['roofline']['kernel'][:20], round(d['roofline']['achieved']), round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_us'],2))"
done; done
