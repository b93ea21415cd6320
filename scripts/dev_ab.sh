#!/bin/bash
# dev: alternating A/B timing of library variants on ONE box.  scripts/dev_ab.sh [-r reps] [-b batch] tag1 tag2 ...
# tag "base" = nsynth_wavenet_amd/lib/libwnhip.so, any other tag = vlibs/lib_<tag>.so (scripts/dev_gk_variants.sh)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
reps=3; batch=1
while getopts "r:b:" o; do case $o in r) reps=$OPTARG;; b) batch=$OPTARG;; esac; done
shift $((OPTIND-1))
for r in $(seq $reps); do
  for t in "$@"; do
    if [ "$t" = base ]; then unset WN_LIB_PATH; else export WN_LIB_PATH=$PWD/vlibs/lib_$t.so; fi
    timeout 300 python scripts/dev_abl_bench.py --tag $t --batch $batch --steps 100 2>&1 | tail -1
  done
done
