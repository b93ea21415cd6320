#!/bin/bash
# dev: variant builds of ONE kernel source linked against the current objects of the others.
#   scripts/dev_gk_variants.sh <source.hip> tag1="-DFLAG=1" tag2="-DFLAG=2 ..." ...   ->  vlibs/lib_<tag>.so
# (vlibs/ is git-ignored; the .so files travel to the GPU box with gpurun.  Run them with WN_LIB_PATH=vlibs/lib_<tag>.so)
set -e
cd "$(dirname "$0")/.."
src=$1; shift
base=$(basename "$src" .hip)
mkdir -p vlibs
python -m nsynth_wavenet_amd.build > /dev/null
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -Wall -Wno-unused-function -I include -I nsynth_wavenet_amd/csrc \
      $flags -c nsynth_wavenet_amd/csrc/$base.hip -o vlibs/${base}_$tag.o &
done
wait
for spec in "$@"; do
  tag=${spec%%=*}
  objs=$(ls nsynth_wavenet_amd/lib/*.o | grep -v "_v.o" | grep -v "/$base.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs vlibs/${base}_$tag.o -o vlibs/lib_$tag.so
  echo "vlibs/lib_$tag.so"
done
