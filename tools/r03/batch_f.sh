R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
B="timeout 120 python3 bench.py --no-cpu-baseline --check-rows 0 --no-extras"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"], "ev", r.get("launch_us_events"))
except Exception as e: print("$name ERR", e)
PY
}
run base $B
for v in 0 1 2 3 4 5 6; do LASR_DEC_NW4=$v run decnw4_$v $B; done
for v in 0 2 3; do LASR_DEC_PRIO=$v run decprio_$v $B; done
for v in 1 3; do LASR_LOOKAHEAD=$v run la_$v $B; done
for v in 1 4; do LASR_LOGITS_MT=$v run logitsmt_$v $B; done
LASR_KICK=2 run kick2 $B
LASR_KICK=4 run kick4 $B
LASR_GROUP=2 run group2 $B
LASR_CELL_NW=8 run cellnw8 $B
LASR_ENC_WAVE=1 run encwave $B
LASR_CELL_PRIO=1 LASR_DEC_PRIO=2 run prio12 $B
run base2 $B
for v in 0 7; do LASR_DEC_NW4=$v run bf16_decnw4_$v $B --dtype bf16; done
LASR_DEC_PRIO=0 run bf16_decprio0 $B --dtype bf16
