R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O
cd $R
rm -f gpurun_out/parity_counts.json gpurun_out/served_rate.json
timeout 700 python -u -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^\[" | tail -40 | tee $O/pytest.log
cp gpurun_out/parity_counts.json gpurun_out/served_rate.json $O/ 2>/dev/null
B="timeout 120 python3 bench.py --no-cpu-baseline --check-rows 0"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"], "ev", r.get("launch_us_events"), "pcie", d.get("pcie_inclusive",{}).get("value"), d.get("pcie_inclusive",{}).get("pinned_nocopy",{}).get("value"))
except Exception as e: print("$name ERR", e, open("$O/$name.err").read()[-300:])
PY
}
run lntile $B --no-extras
LASR_LN_TILE=0 run lnold $B --no-extras
run lntile2 $B --no-extras
run bf16_lntile $B --no-extras --dtype bf16
LASR_LN_TILE=0 run bf16_lnold $B --no-extras --dtype bf16
run extras $B
