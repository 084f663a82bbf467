#!/bin/bash
# Round profile of bench.py on the GPU box with the driver's step counts: kernel-trace stats + the two PMC
# passes (FETCH_SIZE, WRITE_SIZE), all three with the SAME command line.
#   bash tools/profile_round.sh <tag> [steps] [warmup] [extra bench args]  -> gpurun_out/prof_<tag>/{stats,fetch,write}/...
set -e
TAG=${1:-x}
STEPS=${2:-20}
WARM=${3:-5}
shift 3 || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-secondary "$@" > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
if [ -z "$NO_PMC" ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline --no-secondary "$@" > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline --no-secondary "$@" > $OUT/write.log 2>&1
rm -f $OUT/fetch/*/*kernel_trace.csv $OUT/write/*/*kernel_trace.csv
fi
ls $OUT/*/*
