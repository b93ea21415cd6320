# dev: bench lines under different environment settings: dev_env_bench.sh "VAR=1" "VAR=0 OTHER=2" ...
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  for b in 1 8; do
    env $e python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 10 --batch-per-gpu $b 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$e B=$b', round(d['value']/1e6,2),'Ms/s', round(d['ms_per_step'],3),'ms  kernel frac', round(r['frac'],3), 'achieved', round(r['achieved'],1))"
  done
done
