#!/bin/bash
# round 6: the autoregressive step with independent utterance groups on streams -- timelines (rocprofv3 kernel trace) and a
# batch sweep of the single-stream step
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out/r06ar
O=$GRAFT_REPO_ROOT/gpurun_out/r06ar
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for G in 1 2 4; do
  rocprofv3 --kernel-trace -d $O/tr_$G -o t -- python $R/bench_aux.py --workload ar --batch 64 --streams $G --samples 160 --steps 1 > $O/tr_$G.log 2>&1
  f=$(find $O/tr_$G -name "*kernel_trace.csv" | head -1)
  echo "== batch 64, $G stream(s): $(tail -1 $O/tr_$G.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f samples/s, %.1f us per sample step' % (d['value'], d['config']['us_per_sample_step']))")" | tee -a $O/ar_timeline.txt
  python $R/scripts/dev_ar_trace.py $f | tee -a $O/ar_timeline.txt
  head -3 $f | cut -c1-300 > $O/trace_head_$G.txt
  rm -rf $O/tr_$G
done
cd $R
for B in 64 128 256 512 1024; do
  timeout 300 python bench_aux.py --workload ar --batch $B --samples 400 --steps 2 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    ln=ln.strip()
    if ln.startswith('{'):
        d=json.loads(ln); c=d['config']
        print('ar batch %4d, one stream: %8.0f samples/s aggregate = %6.2fx real time, %7.1f us per sample step, %.3f of the weight-stream bound' % ($B, d['value'], d['value']/16000.0, c['us_per_sample_step'], d['roofline']['frac']))
" | tee -a $O/ar_batch_sweep.txt
done
