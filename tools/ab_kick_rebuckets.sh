for i in 1 2; do for kr in 1 0; do
PMSTEP_KICK_REBUCKETS=$kr timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st={s['stage'][3:40]:round(s['avg_ms'],2) for s in d['roofline']['stages'][:3]}
ps=d['per_step']
print('KR=$kr', round(d['ms_per_step'],3), st, 'first', ps[0]['wall_ms'], 'last', ps[-1]['wall_ms'], 'strag', d['straggler_series'][-1])"
done; done
