# Ablation of the conditioning GEMM (iaf_cond_h_kernel, -DWN_CK_ABL) and of the layer-2 upsampler GEMM
# (deconv_mfma_hs_kernel, -DWN_DC_ABL): kernel time with parts of the work removed.
# Build the variants first:  for a in 0 1 2 4 8 16 15 31; do WN_EXTRA_FLAGS=-DWN_CK_ABL=$a python -m nsynth_wavenet_amd.build;
#                            cp nsynth_wavenet_amd/lib/libwnhip.so vlibs/lib_abl$a.so; done   (results are wrong when a != 0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for bsz in 1 8; do
for a in ${@:-0 1 2 4 8 16 15 31}; do
  cp $R/vlibs/lib_abl$a.so $R/nsynth_wavenet_amd/lib/libwnhip.so
  rm -rf /tmp/abl; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o abl -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --batch-per-gpu $bsz > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('/tmp/abl/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    for kn in ('iaf_cond_h_kernel', 'deconv_mfma_hs_kernel'):
        if kn in r['Name']:
            print('B=$bsz ablation %3d  %s  calls %s  avg %.1f us' % ($a, kn, r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
done
