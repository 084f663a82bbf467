# same-box A/B of bench.py stage times: A = pysco_amd/_ab/libA.so (tools/build_ab.sh: git HEAD), B = working tree
#   NCELLS=1024 STEPS=5 bash tools/ab_bench.sh
NCELLS=${NCELLS:-1024}; STEPS=${STEPS:-5}
for i in 1 2; do
for lib in pysco_amd/_ab/libA.so pysco_amd/libpmstep.so; do
PMSTEP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --ncells $NCELLS --steps $STEPS --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st={s['stage']:s['avg_ms'] for s in d['roofline']['stages'][:3]}
print('$lib', round(d['ms_per_step'],3), st)"
done; done
