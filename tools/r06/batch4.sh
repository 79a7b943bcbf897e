# round 6, batch 4: group size of the beam's continuous loop on configs[4] (G = 1 since round 3: re-checked under the pump and the deep pipeline) and configs[2]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6i; mkdir -p $O; cd $R
B="timeout 300 python3 bench.py --no-cpu-baseline --no-extras --sustained-s 0 --other-configs 0"
for g in 1 2 3 4; do
  LASR_PUMP_G=$g $B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 4 --warmup 1 > $O/cfg5_G$g.json 2>/dev/null
  LASR_PUMP_G=$g $B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 4 --warmup 1 --depth 6 > $O/cfg5_d6_G$g.json 2>/dev/null
done
for g in 2 4 6 8; do
  LASR_PUMP_G=$g $B --dtype bf16 --beam 4 --steps 4 --warmup 1 > $O/cfg2b4_G$g.json 2>/dev/null
done
python3 - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6i"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f))
        print(os.path.basename(f), d["value"], "p50",d["latency_ms"]["p50_model_chunk"],"p95",d["latency_ms"]["p95_model_chunk"],"rounds/step",d.get("iterations_per_model_step"))
    except Exception as e: print(f,e)
PY
