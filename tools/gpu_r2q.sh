R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2q; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --no-cpu-baseline $EXTRA > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", d["roofline"]["launch_us"], "offline", d.get("offline",{}).get("audio_sec_per_sec"), "pcie", d.get("pcie_inclusive",{}).get("value"))
except Exception as e: print("$name ERR", e)
PY
}
run f32 A=1
EXTRA="--dtype bf16" run bf16 A=1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $R; python3 tools/rocpd_stats.py $O/kt/kt_results.db | grep -E "frontend|logmel|push"
