#!/bin/bash
# dev: kernel timeline of the fp32 form with the last flow's conditioning GEMM on the side stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WN_F32_BG=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bgtr -o t -- python $R/bench.py --steps 6 --warmup 3 --ramp-steps 0 --no-cpu-baseline --no-extras --precision f32 --layer-events-every 1000000 > /dev/null 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/bgtr/t_kernel_trace.csv")))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'].replace('(anonymous namespace)::','')[:46]) for r in rows)
# last call: from the last prologue kernel on
i0 = max(i for i, k in enumerate(ks) if 'iaf_prologue' in k[3])
call = ks[i0:]
t0 = call[0][0]
gemm = [k for k in call if 'gemm_f32_kernel<4, false>' in k[3]]
print('kernels in the last call:', len(call), ' span %.1f us' % ((max(k[1] for k in call) - t0) / 1e3))
for k in call[:14]:
    print('  %8.1f -> %8.1f us  q%s  %s' % ((k[0]-t0)/1e3, (k[1]-t0)/1e3, k[2], k[3]))
if gemm:
    g = gemm[0]
    inside = [k for k in call if 'iaf_layer_kernel' in k[3] and k[0] >= g[0] and k[1] <= g[1]]
    after = [k for k in call if 'iaf_layer_kernel' in k[3] and k[0] > g[1]]
    print('background GEMM: %.1f -> %.1f us (%.1f us)' % ((g[0]-t0)/1e3, (g[1]-t0)/1e3, (g[1]-g[0])/1e3))
    if inside: print('layer launches inside it: %d, average %.1f us' % (len(inside), sum(k[1]-k[0] for k in inside)/len(inside)/1e3))
    if after: print('layer launches after it : %d, average %.1f us' % (len(after), sum(k[1]-k[0] for k in after)/len(after)/1e3))
PY
rm -rf $R/gpurun_out/bgtr
