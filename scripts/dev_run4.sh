timeout 900 python -m pytest tests/test_gpu_iaf.py -x -q -k "golden or phase_group or ragged or full_size" 2>&1 | tail -4
for r in 1 2; do
WN_DC_NO_PG=1 python scripts/dev_abl_bench.py --tag nopg --steps 50 2>&1 | tail -1 | tee -a gpurun_out/r5_ab_pg2.txt
python scripts/dev_abl_bench.py --tag pg_coalesced --steps 50 2>&1 | tail -1 | tee -a gpurun_out/r5_ab_pg2.txt
done
python scripts/dev_abl_bench.py --tag pg_coalesced_b8 --batch 8 --steps 20 2>&1 | tail -1 | tee -a gpurun_out/r5_ab_pg2.txt
WN_DC_NO_PG=1 python scripts/dev_abl_bench.py --tag nopg_b8 --batch 8 --steps 20 2>&1 | tail -1 | tee -a gpurun_out/r5_ab_pg2.txt
