# round 3, GPU batch D: round-3 tests (verbose, bounded), kernel trace of the split front-end, staging-copy cost
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O
cd $R
timeout 420 python -u -m pytest tests/test_gpu_round3.py -x -v 2>&1 | tail -25 | tee $O/pytest_r3.log
B="timeout 120 python3 bench.py --no-cpu-baseline --check-rows 0"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    r=d["roofline"]
    print("$name", d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["stage_ms_per_model_step"]["decode_iters"], "cell", r["launch_us"], "ev", r.get("launch_us_events"), "pcie", d.get("pcie_inclusive",{}).get("value"), d.get("pcie_inclusive",{}).get("pinned_nocopy",{}).get("value"), "host", d.get("per_rank",[{}])[0].get("host_us_per_model_step"))
except Exception as e: print("$name ERR", e)
PY
}
LASR_DBG_NOSTAGECOPY=1 run extras_nostagecopy $B
run extras $B
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_split -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --check-rows 0 > $O/kt_split.json 2> $O/kt_split.err
LASR_FE_MODE=0 timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_one -o kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --check-rows 0 --split-push > $O/kt_one.json 2> $O/kt_one.err
cd $R
for v in split one; do python3 tools/rocpd_gaps.py $O/kt_$v/kt_results.db > $O/kt_${v}_gaps.txt 2>&1; python3 tools/rocpd_stats.py $O/kt_$v/kt_results.db $O/kt_${v}_stats.txt > /dev/null 2>&1; head -14 $O/kt_${v}_stats.txt; grep -A12 "queue 1" $O/kt_${v}_gaps.txt | head -24; done
rm -rf $O/kt_split/*.db $O/kt_one/*.db
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/full.json 2> $O/full.err; tail -c 1500 $O/full.json
