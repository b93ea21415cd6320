# final measurement pass of a round (run through gpurun): tests, bench lines, kernel stats, PMC passes
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -rf 2>&1 | tail -8 | tee gpurun_out/gputest_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke_final.txt
timeout 400 python bench.py 2>&1 | tail -1 > gpurun_out/bench_f16x3.json
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_f16x3_driver_protocol.json
WN_NO_GROUPS=1 timeout 300 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 > gpurun_out/bench_f16x3_per_layer.json
for p in f16x3-fused f32; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 5 --precision $p 2>&1 | tail -1 > gpurun_out/bench_$p.json
done
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --batch-per-gpu 8 2>&1 | tail -1 > gpurun_out/bench_f16x3_b8.json
for b in 1 8 64 256; do timeout 300 python bench_aux.py --workload ar --batch $b 2>&1 | tail -1 > gpurun_out/bench_ar_b$b.json; done
timeout 300 python bench_aux.py --workload teacher 2>&1 | tail -1 > gpurun_out/bench_teacher.json
bash scripts/pmc_layer.sh fin "--no-extras" > gpurun_out/pmc_fin.log 2>&1
bash scripts/pmc_layer.sh fused "--no-extras --precision f16x3-fused" > gpurun_out/pmc_fused.log 2>&1
bash scripts/pmc_layer.sh b8 "--no-extras --batch-per-gpu 8" > gpurun_out/pmc_b8.log 2>&1
bash scripts/pmc_layer.sh f32 "--no-extras --precision f32" > gpurun_out/pmc_f32.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin1 -o fin1 -- python $R/bench.py --steps 20 --warmup 3 --ramp-steps 0 --no-cpu-baseline --no-extras > $R/gpurun_out/fin1.log 2>&1
WN_NO_GROUPS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/finl -o finl -- python $R/bench.py --steps 20 --warmup 3 --ramp-steps 0 --no-cpu-baseline --no-extras > $R/gpurun_out/finl.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/finf -o finf -- python $R/bench.py --steps 20 --warmup 3 --ramp-steps 0 --no-cpu-baseline --no-extras --precision f16x3-fused > $R/gpurun_out/finf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin32 -o fin32 -- python $R/bench.py --steps 20 --warmup 3 --ramp-steps 0 --no-cpu-baseline --no-extras --precision f32 > $R/gpurun_out/fin32.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin8 -o fin8 -- python $R/bench.py --steps 10 --warmup 3 --ramp-steps 0 --no-cpu-baseline --no-extras --batch-per-gpu 8 > $R/gpurun_out/fin8.log 2>&1
cd $R; for f in gpurun_out/bench_*.json; do python -c "
import json; d=json.load(open('$f')); r=d['roofline']; print('$f', round(d['value']/1e6,3),'Ms/s', round(d['ms_per_step'],3),'ms', r['bound'], round(r['achieved'],1), round(r['frac'],3), r.get('traffic'), d.get('cpu_baseline',{}).get('value'))"; done
# package power / shader clock / energy per part, and the bare matrix pipe's sustained rate
python scripts/dev_power.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/power_per_part.txt
python scripts/dev_power.py --batch 8 --seconds 2 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/power_per_part.txt
HW=$(grep -m1 "^hwmon" gpurun_out/power_per_part.txt | awk '{print $2}')
[ -x scripts/ubench/mfma_power ] && scripts/ubench/mfma_power $HW 3 | tee gpurun_out/mfma_power.txt
[ -x scripts/ubench/mfma_f32_power ] && timeout 120 scripts/ubench/mfma_f32_power auto 2.5 | tee gpurun_out/mfma_f32_power.txt
(cd scripts/ubench && timeout 200 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o /tmp/mfma_f32_issue mfma_f32_issue.hip && timeout 120 /tmp/mfma_f32_issue) | tee gpurun_out/mfma_f32_issue.txt
python scripts/dev_ramp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ramp.txt
# batch sweep of the two launch structures (policy of wn_iaf_use_groups): ms per call
for b in 1 2 4 6 8 12 16; do
  WN_GROUPS=1 python scripts/dev_abl_bench.py --tag groups --batch $b --steps 40 2>/dev/null | tail -1
  WN_NO_GROUPS=1 python scripts/dev_abl_bench.py --tag per-layer --batch $b --steps 40 2>/dev/null | tail -1
done | tee gpurun_out/batch_sweep.txt
