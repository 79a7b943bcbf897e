#!/bin/bash
# fused logits + select: bf16, sync protocol, stream timelines
mkdir -p gpurun_out/fuse
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["roofline"].get("launch_us"), d["stage_ms_per_model_step"]["decode_iters"], d["latency_ms"]["p50_model_chunk"], d.get("tokens_equal"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for rep in 1 2; do
  for f in 1 0; do
    LASR_FUSE_SELECT=$f timeout 120 python bench.py --steps 40 --dtype bf16 --no-cpu-baseline --no-extras > gpurun_out/fuse/bf16_f${f}_r${rep}.json 2> gpurun_out/fuse/bf16_f${f}_r${rep}.err
    show "bf16 fuse=$f rep=$rep" gpurun_out/fuse/bf16_f${f}_r${rep}.json
  done
done
for f in 1 0; do
  LASR_FUSE_SELECT=$f timeout 120 python bench.py --steps 40 --no-pipeline --no-cpu-baseline --no-extras > gpurun_out/fuse/sync_f${f}.json 2> gpurun_out/fuse/sync_f${f}.err
  show "f32 sync fuse=$f" gpurun_out/fuse/sync_f${f}.json
  LASR_FUSE_SELECT=$f timeout 120 python bench.py --steps 16 --no-cpu-baseline --no-extras --trace gpurun_out/fuse/trace_f${f}.txt > gpurun_out/fuse/tr_f${f}.json 2> gpurun_out/fuse/tr_f${f}.err
  python tools/stream_timeline.py gpurun_out/fuse/trace_f${f}.txt 2>&1 | head -4
done
