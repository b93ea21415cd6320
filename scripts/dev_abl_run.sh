#!/bin/bash
# dev: run scripts/dev_abl_bench.py for every vlibs/lib_<tag>.so named on the command line (twice, interleaved)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for t in "$@"; do
  WN_LIB_PATH=$PWD/vlibs/lib_$t.so python scripts/dev_abl_bench.py 2>&1 | tail -1
done
done
