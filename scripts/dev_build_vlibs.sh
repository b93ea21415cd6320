#!/bin/bash
# dev: build library variants vlibs/lib_<tag>.so from "tag=flags" arguments (WN_EXTRA_FLAGS), e.g. abl1="-DWN_F32_ABL=1"
cd $(dirname "$0")/..
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  WN_EXTRA_FLAGS="$flags" WN_LIB_NAME=libwnhip_v.so python -m nsynth_wavenet_amd.build --force > /tmp/vbuild_$tag.log 2>&1 || { tail -5 /tmp/vbuild_$tag.log; exit 1; }
  mv nsynth_wavenet_amd/lib/libwnhip_v.so vlibs/lib_$tag.so
  rm -f nsynth_wavenet_amd/lib/*_v.o nsynth_wavenet_amd/lib/libwnhip_v.sha256
  echo built vlibs/lib_$tag.so
done
