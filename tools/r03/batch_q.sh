R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O
cd $R
B="timeout 150 python3 bench.py --no-cpu-baseline --no-extras --check-rows 0 --steps 40"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"])
except Exception as e: print("$name ERR", e, open("$O/$name.err").read()[-300:])
PY
}
run base $B
LASR_SEL_PRIO=3 run selprio3 $B
LASR_SEL_PRIO=1 run selprio1 $B
run base2 $B
LASR_SEL_PRIO=3 run selprio3b $B
LASR_SEL_PRIO=3 LASR_DEC_PRIO=2 run selprio3_dec2 $B
run bf16 $B --dtype bf16
LASR_SEL_PRIO=3 run bf16_selprio3 $B --dtype bf16
