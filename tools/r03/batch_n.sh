R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O
cd $R
timeout 900 python -u -m pytest tests -m gpu -x -q 2>&1 | tail -12
B="timeout 150 python3 bench.py --no-cpu-baseline --no-extras --steps 40"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"], "chk", d.get("tokens_checked"), d.get("tokens_equal"))
except Exception as e: print("$name ERR", e, open("$O/$name.err").read()[-300:])
PY
}
run fold $B
LASR_PRED_FOLD=0 run nofold $B --check-rows 0
run fold2 $B --check-rows 0
LASR_PRED_FOLD=0 run nofold2 $B --check-rows 0
run bf16_fold $B --dtype bf16 --check-rows 0
LASR_PRED_FOLD=0 run bf16_nofold $B --dtype bf16 --check-rows 0
