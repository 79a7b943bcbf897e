R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
B="timeout 120 python3 bench.py --no-cpu-baseline --check-rows 0 --no-extras"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"], "ev", r.get("launch_us_events"), "iso", r.get("launch_us_isolated"))
except Exception as e: print("$name ERR", e, open("$O/$name.err").read()[-300:])
PY
}
run base $B
for v in 49152 65536 81920 98304 126976; do LASR_CELL_LDS_PAD=$v run pad_$v $B; done
run base2 $B
LASR_CELL_LDS_PAD=65536 run bf16_pad65536 $B --dtype bf16
LASR_CELL_LDS_PAD=98304 run bf16_pad98304 $B --dtype bf16
run bf16_base $B --dtype bf16
LASR_ENC_WAVE=0 run bf16_nowave $B --dtype bf16
LASR_ENC_WAVE=0 LASR_CELL_LDS_PAD=65536 run bf16_nowave_pad $B --dtype bf16
LASR_CELL_LDS_PAD=65536 timeout 200 python -m pytest tests/test_gpu_round3.py -x -q 2>&1 | tail -3
