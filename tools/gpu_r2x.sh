R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2x; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --no-cpu-baseline $EXTRA > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"], "cpl", d["roofline"].get("cells_per_launch"), "frac", d["roofline"]["frac"], "offline", d.get("offline",{}).get("audio_sec_per_sec"))
except Exception as e: print("$name ERR", e)
PY
}
run wave A=1
run nowave LASR_ENC_WAVE=0
run wave2 A=1
run nowave2 LASR_ENC_WAVE=0
EXTRA="--dtype bf16" run bf_wave A=1
EXTRA="--dtype bf16" run bf_nowave LASR_ENC_WAVE=0
EXTRA="--model cfg5 --dtype bf16 --streams 128 --depth 6 --no-extras" run cfg5_wave A=1
EXTRA="--model cfg5 --dtype bf16 --streams 128 --depth 6 --no-extras" run cfg5_nowave LASR_ENC_WAVE=0
