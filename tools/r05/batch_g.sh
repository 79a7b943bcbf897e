#!/bin/bash
# Round 5, batch G: (i) f32 main-stream hipGraph again, now that no event pair sits around the cell sequence; (ii) steps in flight
# for configs[4] (default 3: 18.7 k at p50 3.0 ms; 6: 24 k at 4.9 ms) and for configs[1]; (iii) full GPU suite on the final build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g; mkdir -p $O
export TMPDIR=/tmp
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err || echo "rc $? $n" >> $O/failures.txt; }
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --steps 40 --warmup 5 --sustained-s 0"
for i in a b; do
  LASR_MAIN_GRAPH=0 run f32_graph0_$i $B
  LASR_MAIN_GRAPH=1 run f32_graph1_$i $B
done
for d in 9 10 14 15; do run f32_depth$d $B --depth $d; done
C5="python bench.py --gpus 1 --no-cpu-baseline --no-extras --sustained-s 0 --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 12 --warmup 3"
for d in 3 4 5 6; do run cfg5_beam8_depth$d $C5 --depth $d; done
C2="python bench.py --gpus 1 --no-cpu-baseline --no-extras --sustained-s 0 --dtype bf16 --beam 4 --steps 12 --warmup 3"
for d in 6 8 10; do run cfg2_beam4_depth$d $C2 --depth $d; done
python tools/r05/summ.py $O/*.json | tee $O/summary.txt
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r5g/*.json")):
    try: d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception: continue
    print(p.split("/")[-1], d["value"], "p50", d["latency_ms"]["p50_model_chunk"], "p95", d["latency_ms"]["p95_model_chunk"])
PY
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
