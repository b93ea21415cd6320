#!/bin/bash
# dev: time the flow-pipeline form over batch / length (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1 384" "1 768" "2 384" "4 384" "8 384"; do
  set -- $cfg
  timeout 200 python bench.py --precision f16x3-pipe --batch-per-gpu $1 --frames $2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=$1 F=$2', 'ms/step %.3f'%d['ms_per_step'], 'Msamp/s %.1f'%(d['value']/1e6), 'kern_us %.1f'%d['roofline'].get('avg_launch_us',0))"
done
