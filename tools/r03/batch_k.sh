R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; mkdir -p $O
cd $R
timeout 500 python -u -m pytest tests/test_gpu_beam.py tests/test_gpu_round2.py -x -q 2>&1 | tail -8
B="timeout 200 python3 bench.py --no-cpu-baseline --no-extras --check-rows 0"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"])
except Exception as e: print("$name ERR", e, open("$O/$name.err").read()[-400:])
PY
}
run cfg2_beam4 $B --dtype bf16 --beam 4 --steps 10 --warmup 3
run cfg2_beam4_d12 $B --dtype bf16 --beam 4 --steps 10 --warmup 3 --depth 12
run cfg5_beam8 $B --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2
run cfg2_beam4_f32 $B --beam 4 --steps 10 --warmup 3
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras --check-rows 0 --dtype bf16 --beam 4 --steps 6 --warmup 2 > /dev/null 2>&1
cd $R; python3 tools/rocpd_stats.py $O/kt/kt_results.db $O/kt_beam4_stats.txt > /dev/null 2>&1; head -12 $O/kt_beam4_stats.txt; rm -rf $O/kt
