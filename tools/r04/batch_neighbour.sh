# what a neighbour that takes ONLY one resource costs the two-stream f32 job (configs[1]): matrix-pipe cycles, HBM, L2 hits, Infinity Cache
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4nb3; mkdir -p $O; cd $R
B="timeout 200 python3 bench.py --no-cpu-baseline --no-extras --check-rows 0 --steps 60 --warmup 5"
$B > $O/base.json 2>/dev/null
for k in mfma:1024 load:1024 l2:256 l2:1024 mall:256 mall:1024; do $B --neighbour $k:80 > $O/$(echo $k | tr : _).json 2>/dev/null; done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); nb=d.get('neighbour') or {}
print('$f'.split('/')[-1].ljust(16), d['value'], 'p50', d['latency_ms']['p50_model_chunk'], 'cell_us', d['roofline'].get('launch_us'), round(nb.get('achieved') or 0, 1), nb.get('unit'), round(nb.get('timed_region_ms') or 0, 1))"; done
