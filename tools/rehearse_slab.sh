# 2-rank rehearsal of the slab path on ONE GPU (gloo stages the exchanges through the host: the communication times
# mean nothing, the per-rank kernel times and straggler fractions do):  NCELLS=512 STEPS=15 bash tools/rehearse_slab.sh [0|1]
NCELLS=${NCELLS:-512}; STEPS=${STEPS:-15}; WARMUP=${WARMUP:-3}; RANKS=${RANKS:-2}
PMSTEP_SLAB_REBUCKET=${1:-1} PMSTEP_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $RANKS \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $RANKS --ncells $NCELLS --steps $STEPS --warmup $WARMUP --no-cpu-baseline 2>&1 | grep '^{' | tail -1
