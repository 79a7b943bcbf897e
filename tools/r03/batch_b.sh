# round 3, GPU batch B: marginal cost of a microsecond on either stream (delay kernels), cheap cell timers
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
B="timeout 200 python3 bench.py --no-cpu-baseline --no-extras --check-rows 0"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"], "ev", r.get("launch_us_events"), "prof", r.get("value_profiled"), "host", d.get("per_rank",[{}])[0].get("host_us_per_model_step"))
except Exception as e: print("$name ERR", e)
PY
}
run base $B
run prof1 $B --cell-prof-in-timed 1
run prof2 $B --cell-prof-in-timed 2
for d in 10 20 40; do LASR_DELAY_MAIN_US=$d run dmain$d $B --prof-steps 0; done
for d in 5 10 20; do LASR_DELAY_DEC_US=$d run ddec$d $B --prof-steps 0; done
run base2 $B --prof-steps 0
for dp in 6 9 15; do run depth$dp $B --prof-steps 0 --depth $dp; done
LASR_DELAY_MAIN_US=20 run bf16_dmain20 $B --prof-steps 0 --dtype bf16
LASR_DELAY_DEC_US=10 run bf16_ddec10 $B --prof-steps 0 --dtype bf16
run bf16_base $B --prof-steps 0 --dtype bf16
