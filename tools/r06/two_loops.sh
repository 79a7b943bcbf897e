#!/bin/bash
# Would two independent decode loops per GPU pay?  Zero-code probe: two ranks on ONE GPU (LASR_BENCH_SAME_GPU), half the streams each,
# against one rank with all of them.  usage: two_loops.sh <out_dir>
OUT=$1; mkdir -p $OUT
C="--no-cpu-baseline --no-extras --sustained-s 0 --other-configs 0 --check-rows 8"
for i in 1 2; do
timeout 300 python bench.py $C --model cfg5 --dtype bf16 --beam 8 --streams 128 --steps 4 --warmup 1 2>/dev/null | tail -1 > $OUT/cfg5b8_1x128_$i.json
LASR_BENCH_BACKEND=gloo LASR_BENCH_SAME_GPU=1 timeout 300 python bench.py --gpus 2 $C --model cfg5 --dtype bf16 --beam 8 --streams 64 --steps 4 --warmup 1 2>/dev/null | tail -1 > $OUT/cfg5b8_2x64_$i.json
timeout 300 python bench.py $C --model cfg2 --dtype bf16 --beam 4 --streams 64 --steps 4 --warmup 1 2>/dev/null | tail -1 > $OUT/cfg2b4_1x64_$i.json
LASR_BENCH_BACKEND=gloo LASR_BENCH_SAME_GPU=1 timeout 300 python bench.py --gpus 2 $C --model cfg2 --dtype bf16 --beam 4 --streams 32 --steps 4 --warmup 1 2>/dev/null | tail -1 > $OUT/cfg2b4_2x32_$i.json
timeout 300 python bench.py $C 2>/dev/null | tail -1 > $OUT/f32_1x64_$i.json
LASR_BENCH_BACKEND=gloo LASR_BENCH_SAME_GPU=1 timeout 300 python bench.py --gpus 2 $C --streams 32 2>/dev/null | tail -1 > $OUT/f32_2x32_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        j=json.loads(open(f).read()); print(f.split("/")[-1], round(j["value"]), j["n_gpus"], j["latency_ms"]["p50_model_chunk"], [round(r["value"]) for r in j.get("per_rank",[])])
    except Exception as e: print(f, "?", str(e)[:60])
PY
