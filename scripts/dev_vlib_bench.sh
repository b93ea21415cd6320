# dev: bench lines of several prebuilt library variants (vlibs/lib_<tag>.so), B=1 and B=8
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  cp vlibs/lib_$t.so nsynth_wavenet_amd/lib/libwnhip.so
  for b in 1 8; do
    python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 10 --batch-per-gpu $b 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$t B=$b', round(d['value']/1e6,2),'Ms/s', round(d['ms_per_step'],3),'ms  kernel frac', round(r['frac'],3), 'achieved', round(r['achieved'],1))"
  done
done
