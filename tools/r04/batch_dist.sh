#!/bin/bash
# Round 4: the N-rank line's diagnostics on hardware at world 1 -- RCCL process group (lazy / eager init) against
# no process group, 40 steps; the 2-rank gloo dry run on ONE GPU (launcher path, new per_rank fields)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
B="python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras"
$B > $O/bench_nodist_40.json 2> $O/bench_nodist_40.err
LASR_BENCH_FORCE_DIST=1 $B > $O/bench_rccl_world1_lazy_40.json 2> $O/bench_rccl_world1_lazy_40.err
LASR_BENCH_FORCE_DIST=1 LASR_BENCH_EAGER_RCCL=1 $B > $O/bench_rccl_world1_eager_40.json 2> $O/bench_rccl_world1_eager_40.err
$B > $O/bench_nodist_40_b.json 2> $O/bench_nodist_40_b.err
LASR_BENCH_BACKEND=gloo LASR_BENCH_SAME_GPU=1 timeout 300 python bench.py --gpus 2 --no-cpu-baseline --no-extras > $O/bench_2rank_dryrun.json 2> $O/bench_2rank_dryrun.err; echo rc=$? >> $O/bench_2rank_dryrun.err
python tools/r04/summ.py $O/*.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4i/*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], d.get("dist"), [ (p["rank"], p["value"], p["overlap_probe"], p["streams_overlap"]) for p in d["per_rank"]])
    except Exception as e: print(f, e)
PY
