# round 6, batch 3: x-shared encoder cell pairs (LASR_ENC_XS) -- parity against the reference's goldens, then the A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6g; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_gpu_round6.py -m gpu -q -x -k "not native and not grpc and not front and not switch" 2>&1 | tail -15 > $O/pytest_xs.txt
B="timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --other-configs 0 --sustained-s 3"
for i in 1 2; do
  LASR_ENC_XS=0 $B > $O/bench_xs0_$i.json 2>/dev/null
  LASR_ENC_XS=1 $B > $O/bench_xs1_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r6g"
for f in sorted(glob.glob(O+"/bench_xs*.json")):
    try:
        d=json.load(open(f)); s=d.get("sustained",{})
        print(os.path.basename(f), d["value"], "eq",d.get("tokens_equal"),"p50",d["latency_ms"]["p50_model_chunk"],"sustained",s.get("value"),s.get("p95_model_chunk_ms"),"cell_us",d["roofline"]["launch_us"],s.get("cell_launch_us"),"iters",d.get("iterations_per_model_step"),d["config"]["engine"])
    except Exception as e: print(f,e)
PY
cat $O/pytest_xs.txt
