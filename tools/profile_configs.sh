#!/bin/bash
# Secondary BASELINE configs on the GPU box: the bench line (with cpu_baseline) and a rocprofv3 kernel-trace summary each.
#   bash tools/profile_configs.sh   -> gpurun_out/prof_cfg/<config>.json, <config>_kernel_stats.csv
set -e
OUT=$PWD/gpurun_out/prof_cfg
mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in newton_mg fr mond f64; do
  args="--config $c"
  [ "$c" = f64 ] && args="--ncells 512 --dtype f64"   # (BASELINE config 2 as stated: the fp64 device mode)
  python3 $R/bench.py $args --steps 10 --warmup 3 --no-secondary > $OUT/$c.log 2>&1
  grep '^{' $OUT/$c.log | tail -1 > $OUT/$c.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c -- python3 $R/bench.py $args --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/${c}_prof.log 2>&1
  cp $OUT/$c/*/*_kernel_stats.csv $OUT/${c}_kernel_stats.csv
  rm -rf $OUT/$c
  echo "$c done"
done
