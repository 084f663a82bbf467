# same-box comparison of bench.py stage times over library variants: bash tools/ab_variants.sh tagA tagB ...   ("base" = the tree's libpmstep.so)
NCELLS=${NCELLS:-1024}; STEPS=${STEPS:-8}; WARMUP=${WARMUP:-3}; ROUNDS=${ROUNDS:-2}
for i in $(seq $ROUNDS); do
for tag in "$@"; do
lib=pysco_amd/_ab/lib$tag.so; [ "$tag" = base ] && lib=pysco_amd/libpmstep.so
PMSTEP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --ncells $NCELLS --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-secondary $EXTRA 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st={s['stage'][3:24]:round(s['avg_ms'],2) for s in d['roofline']['stages'][:3]}
ps=d.get('per_step',[])
def g(p):
    return {k[:12]:v for k,v in p['ms'].items()}
print('$tag', round(d['ms_per_step'],3), st, 'first', g(ps[0]) if ps else None, 'last', g(ps[-1]) if ps else None, 'strag', d.get('straggler_series',[None])[-1])"
done; done
