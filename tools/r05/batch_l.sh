#!/bin/bash
# Round 5, batch L: joint encoder-half GEMM on the DECODE stream behind the step's encoder event (LASR_J_DEC) against on the main stream
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5l; mkdir -p $O
export TMPDIR=/tmp
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err || echo "rc $? $n" >> $O/failures.txt; }
LASR_J_DEC=1 timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_round2.py -q -x 2>&1 | tail -3
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --steps 40 --warmup 5 --sustained-s 0"
for i in a b c; do
  LASR_J_DEC=0 run f32_jmain_$i $B
  LASR_J_DEC=1 run f32_jdec_$i $B
done
for d in 12 25; do LASR_J_DEC=0 run f32_jmain_d$d $B --depth $d; LASR_J_DEC=1 run f32_jdec_d$d $B --depth $d; done
C5="python bench.py --gpus 1 --no-cpu-baseline --no-extras --sustained-s 0 --model cfg5 --dtype bf16 --streams 128 --depth 6 --steps 16 --warmup 4"
LASR_J_DEC=0 run cfg5_jmain $C5
LASR_J_DEC=1 run cfg5_jdec $C5
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r5l/*.json")):
    try: d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception: continue
    print(p.split("/")[-1], d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "iters", d["iterations_per_model_step"], d["config"]["engine"].get("j_dec"), d.get("tokens_equal"))
PY
cat $O/failures.txt 2>/dev/null; true
