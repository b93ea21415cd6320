#!/bin/bash
# round 6: timing-only upper bounds for the "joule-removing" candidates on the default (split-fp16) path
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
O=gpurun_out/r06e; mkdir -p $O
for rep in 1 2; do
for t in e0 e1 e2 e3; do
  for b in 1 8; do
    WN_LIB_PATH=$GRAFT_REPO_ROOT/vlibs/lib_$t.so python scripts/dev_abl_bench.py --tag $t --batch $b --steps $((400/b)) 2>/dev/null | tail -1
  done
done
done | tee $O/epi_abl.txt
