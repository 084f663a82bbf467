#!/bin/bash
# Builds pysco_amd/_ab/libA.so from the csrc of git HEAD (the "A" of tools/ab.sh; "B" is the working tree's
# libpmstep.so).  Run from the repo root after `python -m pysco_amd.build`.
set -e
R=/tmp/abr; rm -rf $R; mkdir -p $R/pysco_amd/csrc $R/include pysco_amd/_ab
git show HEAD:include/pmstep.h > $R/include/pmstep.h
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -munsafe-fp-atomics -I/opt/rocm/include"
objs=""
for f in $(git ls-tree --name-only HEAD pysco_amd/csrc/); do git show HEAD:$f > $R/$f; done
for f in $R/pysco_amd/csrc/*.hip; do
  o=$R/$(basename $f .hip).o
  /opt/rocm/bin/hipcc $F -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pysco_amd/_ab/libA.so $objs -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
ls -la pysco_amd/_ab/libA.so
