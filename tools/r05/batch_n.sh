#!/bin/bash
# Round 5, batch N: deferred ring append (lasr_push_submit, LASR_PUSH_DEVICE_STABLE / host pushes): parity test, then the A/B on
# configs[1] f32 and bf16 (interleaved legs on one box), host-push leg included
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -x -k "deferred or long or front_grpc or decode_tail or irregular" > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -3 $O/pytest_sel.txt
for i in 1 2 3; do
  for ds in 1 0; do
    timeout 300 python bench.py --steps 40 --warmup 8 --device-stable $ds --sustained-s 0 --check-rows 8 2>/dev/null | tail -1 > $O/f32_ds${ds}_$i.json
    timeout 300 python bench.py --dtype bf16 --steps 40 --warmup 8 --device-stable $ds --sustained-s 0 --check-rows 8 2>/dev/null | tail -1 > $O/bf16_ds${ds}_$i.json
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5n/*.json")):
    try:
        j = json.loads(open(f).read())
        print(f.split("/")[-1], j["value"], j.get("ms_per_step"), j.get("latency_ms", {}).get("p50") if isinstance(j.get("latency_ms"), dict) else "")
    except Exception as e:
        print(f, "ERR", e)
PY
