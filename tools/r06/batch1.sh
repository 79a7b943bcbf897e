# round 6, batch 1: widened parity tests, the default bench line with its other_configs legs, depth sweep for the p95 target
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round2.py -m gpu -q -k "token_error or greedy_streaming_exact or config2 or config4" 2>&1 | grep -v "^    \|^$" | cut -c1-700 | tail -80 > $O/pytest_parity.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
for d in 14 16 18 20; do
  timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --depth $d --no-cpu-baseline --no-extras --other-configs 0 --sustained-s 4 > $O/bench_depth$d.json 2>/dev/null
done
python3 - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6e"
for f in sorted(glob.glob(O+"/bench_depth*.json")):
    try:
        d=json.load(open(f)); s=d.get("sustained",{})
        print(os.path.basename(f), d["value"], "p50",d["latency_ms"]["p50_model_chunk"],"p95",d["latency_ms"]["p95_model_chunk"],"sustained",s.get("value"),s.get("p50_model_chunk_ms"),s.get("p95_model_chunk_ms"))
    except Exception as e: print(f,e)
try:
    d=json.load(open(O+"/bench_default.json"))
    print("default", d["value"], d["roofline"]["frac"], d["roofline"].get("frac_rocprof"), d.get("pcie_inclusive"))
    for l in d.get("other_configs",[]): print(json.dumps(l)[:1500])
except Exception as e: print("default",e)
PY
cat $O/bench_default.time; tail -30 $O/pytest_parity.txt
