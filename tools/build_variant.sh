#!/bin/bash
# Builds pysco_amd/_ab/lib<tag>.so from the working tree with extra compiler flags (development A/B of kernel variants):
#   bash tools/build_variant.sh <tag> -DPM_EXP_SOMETHING=1     then   PMSTEP_LIB=$PWD/pysco_amd/_ab/lib<tag>.so python bench.py ...
set -e
TAG=$1; shift
R=/tmp/abv_$TAG; rm -rf $R; mkdir -p $R pysco_amd/_ab
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -munsafe-fp-atomics -I/opt/rocm/include $@"
objs=""
for f in pysco_amd/csrc/*.hip; do
  o=$R/$(basename $f .hip).o
  /opt/rocm/bin/hipcc $F -c $f -o $o 2>/dev/null &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pysco_amd/_ab/lib$TAG.so $objs -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
ls -la pysco_amd/_ab/lib$TAG.so
