R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 200 python3 bench.py --no-cpu-baseline --no-extras $EXTRA > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"], "tok", d["tokens_per_frame"])
except Exception as e: print("$name ERR", e)
PY
}
timeout 200 python3 bench.py --no-cpu-baseline --no-extras --trace $O/trace.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace.json
LASR_KICK=2 timeout 200 python3 bench.py --no-cpu-baseline --no-extras --trace $O/trace_k2.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace_k2.json
LASR_KICK=1 timeout 200 python3 bench.py --no-cpu-baseline --no-extras --trace $O/trace_k1.json > /dev/null 2>&1
python3 tools/stream_timeline.py $O/trace_k1.json
run base A=1
run sprio1 LASR_DEC_STREAM_PRIO=1
run sprio0 LASR_DEC_STREAM_PRIO=-1
run dprio3 LASR_DEC_PRIO=3
run dprio1 LASR_DEC_PRIO=1
run cprio3 LASR_CELL_PRIO=3
run sprio1_dprio3 LASR_DEC_STREAM_PRIO=1 LASR_DEC_PRIO=3
run k1 LASR_KICK=1
