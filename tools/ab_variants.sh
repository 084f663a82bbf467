# same-box comparison of bench.py stage times over library variants: bash tools/ab_variants.sh tagA tagB ...   ("base" = the tree's libpmstep.so)
NCELLS=${NCELLS:-1024}; STEPS=${STEPS:-8}; WARMUP=${WARMUP:-3}; ROUNDS=${ROUNDS:-2}
for i in $(seq $ROUNDS); do
for tag in "$@"; do
lib=pysco_amd/_ab/lib$tag.so; [ "$tag" = base ] && lib=pysco_amd/libpmstep.so
PMSTEP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --ncells $NCELLS --steps $STEPS --warmup $WARMUP --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st={s['stage'][:28]:round(s['avg_ms'],2) for s in d['roofline']['stages'][:3]}
print('$tag', round(d['ms_per_step'],3), st)"
done; done
