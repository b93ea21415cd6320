timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/var_dc -o v -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/var_dc.log 2>&1
grep -E "deconv|interleave" /root/repo/gpurun_out/var_dc/v_kernel_stats.csv | sed -E 's/\(unsigned[^"]*"/"/;s/\(float[^"]*"/"/' | cut -d, -f1,2,4
tail -1 /root/repo/gpurun_out/var_dc.log | cut -c1-160
