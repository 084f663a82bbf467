#!/bin/bash
# Round profile of bench.py on the GPU box: kernel-trace stats + the two PMC passes (FETCH_SIZE, WRITE_SIZE).
#   bash tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{stats,fetch,write}/...
set -e
TAG=${1:-x}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
rm -f $OUT/*/*/*kernel_trace.csv   # large; the stats and counter tables are what is kept
ls $OUT/*/*
