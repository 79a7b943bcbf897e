R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O
cd $R
timeout 300 python -u -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
B="timeout 150 python3 bench.py --no-cpu-baseline --check-rows 0 --steps 40"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"], "pcie", d.get("pcie_inclusive",{}).get("value"), d.get("pcie_inclusive",{}).get("pinned_nocopy",{}).get("value"))
except Exception as e: print("$name ERR", e, open("$O/$name.err").read()[-300:])
PY
}
run dma $B
LASR_PUSH_ZEROCOPY=1 run zerocopy $B
LASR_PUSH_THREADS=0 run dma_nothreads $B
run dma2 $B
run bf16_dma $B --dtype bf16
