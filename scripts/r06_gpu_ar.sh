#!/bin/bash
# round 6: the autoregressive step with independent utterance groups on streams -- a batch sweep of the single-stream step,
# groups on streams (hipGraph replay from one thread / plain launches from G threads), and kernel timelines (rocprofv3)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
mkdir -p gpurun_out/r06ar
O=$GRAFT_REPO_ROOT/gpurun_out/r06ar
R=$GRAFT_REPO_ROOT
line() { python -c "
import sys,json
for ln in sys.stdin:
    ln=ln.strip()
    if ln.startswith('{'):
        d=json.loads(ln); c=d['config']
        print('%-44s %8.0f samples/s aggregate = %6.2fx real time, %7.1f us per sample step' % ('$1', d['value'], d['value']/16000.0, c['us_per_sample_step']))
"; }
for B in 64 128 256 512 1024; do
  timeout 200 python bench_aux.py --workload ar --batch $B --samples 320 --steps 2 2>&1 | line "batch $B, one stream" | tee -a $O/ar_batch_sweep.txt
done
for spec in "64 2" "64 4" "256 4"; do
  set -- $spec
  timeout 200 python bench_aux.py --workload ar --batch $1 --streams $2 --samples 320 --steps 2 2>&1 | line "batch $1, $2 streams, graphs, 1 thread" | tee -a $O/ar_streams.txt
  timeout 200 python bench_aux.py --workload ar --batch $1 --streams $2 --threads --samples 320 --steps 2 2>&1 | line "batch $1, $2 streams, launches, $2 threads" | tee -a $O/ar_streams.txt
done
cd /tmp && export TMPDIR=/tmp
for G in 1 2; do
  EXTRA=""; [ $G -gt 1 ] && EXTRA="--streams $G --threads"
  timeout 240 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$G -o t -- python $R/bench_aux.py --workload ar --batch 64 $EXTRA --samples 48 --steps 1 > $O/tr_$G.log 2>&1
  f=$(find $O/tr_$G -name "*kernel_trace.csv" | head -1)
  echo "== batch 64, $G stream(s), plain launches from $G host thread(s): $(grep '^{' $O/tr_$G.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f samples/s, %.1f us per sample step (under the profiler)' % (d['value'], d['config']['us_per_sample_step']))")" | tee -a $O/ar_timeline.txt
  [ -n "$f" ] && python $R/scripts/dev_ar_trace.py $f | tee -a $O/ar_timeline.txt
  [ -n "$f" ] && head -2 $f | cut -c1-400 > $O/trace_head_$G.txt
  rm -rf $O/tr_$G
done
