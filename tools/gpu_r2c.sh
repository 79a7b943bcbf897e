R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
run() { name=$1; shift; env "$@" timeout 200 python3 bench.py --no-cpu-baseline --no-extras $EXTRA > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"])
except Exception as e: print("$name ERR", e)
PY
}
run base A=1
run k2 LASR_KICK=2
run k4 LASR_KICK=4
run k5 LASR_KICK=5
run k3g2 LASR_KICK=3 LASR_GROUP=2
run k4g2 LASR_KICK=4 LASR_GROUP=2
EXTRA="--depth 3" run d3 A=1
EXTRA="--depth 4" run d4 A=1
EXTRA="--depth 7" run d7 A=1
EXTRA="--dtype bf16" run bf16 A=1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/kt.json 2>$O/kt.err
