timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 > gpurun_out/bench_r01_f16x3.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --precision f32 2>&1 | tail -1 > gpurun_out/bench_r01_f32.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch-per-gpu 8 2>&1 | tail -1 > gpurun_out/bench_r01_f16x3_b8.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r01n -o r01n -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/r01n.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r01n8 -o r01n8 -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch-per-gpu 8 > /root/repo/gpurun_out/r01n8.log 2>&1
cd /root/repo; for f in gpurun_out/bench_r01_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f', round(d['value']/1e6,2),'Ms/s', round(d['ms_per_step'],3),'ms', r['bound'], round(r['achieved'],1), round(r['frac'],3), round(r['avg_launch_us'],2), r.get('traffic'), d.get('cpu_baseline',{}).get('value'))"; done
