#!/bin/bash
# SQ counters of the step's kernels (separate --pmc passes, kernel-trace only): VALU / LDS activity and LDS bank conflicts.
#   bash tools/profile_counters.sh <tag>   -> gpurun_out/cnt_<tag>/{valu,lds}/..., gpurun_out/cnt_<tag>/table.txt
set -e
TAG=${1:-x}
OUT=$PWD/gpurun_out/cnt_$TAG
mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/valu -- python3 $R/bench.py --gpus 1 --steps 3 --warmup 9 --no-cpu-baseline --no-secondary > $OUT/valu.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/lds -- python3 $R/bench.py --gpus 1 --steps 3 --warmup 9 --no-cpu-baseline --no-secondary > $OUT/lds.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/wait -- python3 $R/bench.py --gpus 1 --steps 3 --warmup 9 --no-cpu-baseline --no-secondary > $OUT/wait.log 2>&1 || true
for p in valu lds wait; do
  f=$(ls $OUT/$p/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_table.py $f "interp_ranges_kernel<3, 5, true, false, true>" "kdw_deposit_kernel<3, true>" "fftp_x_green" >> $OUT/table.txt
done
cat $OUT/table.txt
