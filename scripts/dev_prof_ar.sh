cd /tmp && export TMPDIR=/tmp
for b in 1 3; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ar_b$b -o v -- python /root/repo/scripts/dev_time_ar.py 800 $b > /dev/null 2>&1
echo "== B=$b"; head -9 /root/repo/gpurun_out/ar_b$b/v_kernel_stats.csv | sed -E 's/\(anonymous namespace\):://g' | cut -d'(' -f1,2 | cut -c1-60 | paste - <(head -9 /root/repo/gpurun_out/ar_b$b/v_kernel_stats.csv | awk -F'",' '{print $2}' | cut -d, -f1,3)
done
