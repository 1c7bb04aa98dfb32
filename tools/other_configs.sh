# The parity-test configurations of SURVEY section 8 on the current build (not the metric).
set -e
O=gpurun_out/other; mkdir -p $O
timeout -k 10 200 python bench.py --rays 1024 --coarse 480 --steps 20 --warmup 5 --no-cpu-baseline > $O/c1.json 2>/dev/null
timeout -k 10 300 python bench.py --rays 16384 --coarse 480 --steps 10 --warmup 3 --no-cpu-baseline --no-render > $O/c4.json 2>/dev/null
timeout -k 10 200 python bench.py --table-log2 19 --steps 20 --warmup 5 --no-cpu-baseline --no-render > $O/t19.json 2>/dev/null
NERF_TABLE_BINNED=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-render > $O/classic.json 2>/dev/null
timeout -k 10 200 python bench.py --levels 16 --table-log2 19 --steps 20 --warmup 5 --no-cpu-baseline > $O/l16t19.json 2>/dev/null
python - <<'P'
import json
out = {}
for k, what in (('c1', 'C1: 1024 rays, 480 coarse + 128 fine'), ('c4', 'C4: 16384 rays, 480 + 128'), ('t19', 'light table T=2^19'),
                ('classic', 'metric config with NERF_TABLE_BINNED=0 (atomic scatter -> Adam pipeline)'),
                ('l16t19', "SURVEY's light variant: 16 levels, T = 2^19 (46 MB table)")):
    d = json.loads(open('gpurun_out/other/%s.json' % k).read().strip().splitlines()[-1])
    out['bench_' + k] = {'what': what, 'ms_per_step': d['ms_per_step'], 'rays_per_s': d['value'], 'render': d.get('render'),
                         'config': d['config']}
    print(k, round(d['ms_per_step'], 3), round(d['value']), (d.get('render') or {}).get('rays_per_s'))
json.dump(out, open('gpurun_out/other/other_configs.json', 'w'), indent=1)
P
