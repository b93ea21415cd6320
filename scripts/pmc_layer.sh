#!/bin/bash
# PMC passes for the layer kernels (run on the GPU box through gpurun). $1 = tag, $2 = extra bench.py arguments
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
# the hash of the kernel sources these counters are measured on, stamped NOW (pmc_summary.py copies it and refuses a
# directory without one: a summary regenerated later from old counters must not claim the current sources)
(cd $GRAFT_REPO_ROOT && python -c "from nsynth_wavenet_amd import build; print(build.source_hash())") > $OUT/source_hash.txt
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --ramp-steps 0 --no-cpu-baseline --layer-events-every 1000 ${2:-}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/inst -o inst -- $CMD > $OUT/inst.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -o tcc -- $CMD > $OUT/tcc.log 2>&1
ls -R $OUT | head -30
python - <<PY
import csv, glob, collections, os
out = "$OUT"
for sub in ("sq", "fetch", "write", "inst", "tcc"):
    files = glob.glob(os.path.join(out, sub, "*counter_collection.csv"))
    if not files:
        print(sub, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"].split("(")[0][-40:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
        cnt[(k, row["Counter_Name"])] += 1
    for k in agg:
        if "iaf_layer" in k or "deconv_mfma" in k or "iaf_head" in k or "iaf_cond" in k or "iaf_pair" in k or "iaf_group" in k or "gemm_f32" in k:
            print(sub, k, {c: round(v / cnt[(k, c)], 1) for c, v in agg[k].items()}, "dispatches", max(cnt[(k, c)] for c in agg[k]))
PY
