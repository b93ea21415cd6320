#!/bin/bash
# dev: the working tree's library against the round's committed group kernel (vlibs/lib_head.so): GPU suite, then timing
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r04/gputest_mid.txt
for rep in 1 2; do
  WN_LIB_PATH=$PWD/vlibs/lib_head.so timeout 200 python scripts/dev_abl_bench.py --tag head 2>&1 | tail -1
  timeout 200 python scripts/dev_abl_bench.py --tag new 2>&1 | tail -1
done | tee gpurun_out/r04/mid_bench.txt
