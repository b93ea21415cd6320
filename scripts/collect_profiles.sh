# Copies what scripts/final_profile.sh left under gpurun_out/ into profiles/ (tracked) under this round's names and
# rebuilds the PMC summaries (with the source hash of the kernels they were measured on).  Usage: collect_profiles.sh r02
set -e
R=${1:-r06}
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/bench_f16x3.json profiles/${R}_bench_f16x3.json
cp $G/bench_f16x3_b8.json profiles/${R}_bench_f16x3_batch8.json
cp $G/bench_f16x3_driver_protocol.json profiles/${R}_bench_f16x3_driver_protocol.json
cp $G/bench_f16x3_per_layer.json profiles/${R}_bench_f16x3_per_layer_launches.json
cp $G/bench_f16x3-fused.json profiles/${R}_bench_f16x3_fused.json
cp $G/bench_f32.json profiles/${R}_bench_fp32.json
for b in 1 8 64 256; do cp $G/bench_ar_b$b.json profiles/${R}_bench_ar_batch$b.json; done
cp $G/bench_teacher.json profiles/${R}_bench_teacher_forward.json
cp $G/fin1/fin1_kernel_stats.csv profiles/${R}_kernel_stats_f16x3.csv
cp $G/finf/finf_kernel_stats.csv profiles/${R}_kernel_stats_f16x3_fused.csv
cp $G/finl/finl_kernel_stats.csv profiles/${R}_kernel_stats_f16x3_per_layer_launches.csv
cp $G/fin8/fin8_kernel_stats.csv profiles/${R}_kernel_stats_f16x3_batch8.csv
cp $G/fin32/fin32_kernel_stats.csv profiles/${R}_kernel_stats_f32.csv
[ -f $G/mfma_f32_power.txt ] && cp $G/mfma_f32_power.txt profiles/${R}_mfma_f32_power_ubench.txt
[ -f $G/mfma_f32_issue.txt ] && cp $G/mfma_f32_issue.txt profiles/${R}_mfma_f32_issue_ubench.txt
cp $G/power_per_part.txt profiles/${R}_power_per_part.txt
[ -f $G/mfma_power.txt ] && cp $G/mfma_power.txt profiles/${R}_mfma_power_ubench_final.txt
cp $G/ramp.txt profiles/${R}_clock_ramp.txt
cp $G/batch_sweep.txt profiles/${R}_batch_sweep_final.txt
python scripts/pmc_summary.py $G/pmc_fin profiles/${R}_pmc_summary_f16x3.json 1 "--no-extras" $G/fin1/fin1_kernel_stats.csv 23
python scripts/pmc_summary.py $G/pmc_fused profiles/${R}_pmc_summary_f16x3_fused.json 1 "--no-extras --precision f16x3-fused" $G/finf/finf_kernel_stats.csv 23
python scripts/pmc_summary.py $G/pmc_b8 profiles/${R}_pmc_summary_f16x3_batch8.json 8 "--no-extras --batch-per-gpu 8" $G/fin8/fin8_kernel_stats.csv 13
python scripts/pmc_summary.py $G/pmc_f32 profiles/${R}_pmc_summary_f32.json 1 "--no-extras --precision f32" $G/fin32/fin32_kernel_stats.csv 23
python - <<PY
import json
from nsynth_wavenet_amd import build
h = build.source_hash()
for t in ('f16x3', 'f16x3_fused', 'f16x3_batch8', 'f32'):
    d = json.load(open('profiles/${R}_pmc_summary_%s.json' % t))
    print(t, 'source_hash matches the tree:', d.get('source_hash') == h)
PY
