#!/bin/bash
# Round 5, batch K: lasr_step_wait sleeping on the pump's progress counter (after a short spin) against the pure spin (previous build via LASR_LIB)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
export TMPDIR=/tmp
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err || echo "rc $? $n" >> $O/failures.txt; }
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --steps 40 --warmup 5 --sustained-s 0"
OLD=$GRAFT_REPO_ROOT/libreasr_amd/csrc/liblasr_hip_spin.so
for i in a b c; do
  LASR_LIB=$OLD run f32_spin_$i $B
  run f32_sleep_$i $B
done
LASR_LIB=$OLD run bf16_spin $B --dtype bf16
run bf16_sleep $B --dtype bf16
C5="python bench.py --gpus 1 --no-cpu-baseline --no-extras --sustained-s 0 --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 12 --warmup 3"
LASR_LIB=$OLD run cfg5b8_spin $C5
run cfg5b8_sleep $C5
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r5k/*.json")):
    try: d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception: continue
    r = d["per_rank"][0]
    print(p.split("/")[-1], d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "cores", r["host_cores_busy"], r["host_us_per_model_step"], d.get("tokens_equal"))
PY
