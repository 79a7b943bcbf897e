#!/bin/bash
# Round 5, batch O: front-end ahead (LASR_FE_AHEAD: k_fe_mel + k_ln_tile of step k+1 on their own stream beside the cells of step
# k): parity tests, then the A/B on configs[1] f32 / bf16 (interleaved legs on one box) and one host-push leg each
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5o; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_server.py tests/test_gpu_round3.py -m gpu -q -x > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -3 $O/pytest_sel.txt
for i in 1 2 3; do
  for fa in 1 0; do
    LASR_FE_AHEAD=$fa timeout 300 python bench.py --steps 40 --warmup 8 --sustained-s 0 --check-rows 8 2>/dev/null | tail -1 > $O/f32_fa${fa}_$i.json
    LASR_FE_AHEAD=$fa timeout 300 python bench.py --dtype bf16 --steps 40 --warmup 8 --sustained-s 0 --check-rows 8 2>/dev/null | tail -1 > $O/bf16_fa${fa}_$i.json
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5o/*.json")):
    try:
        j = json.loads(open(f).read())
        print(f.split("/")[-1], j["value"], j.get("ms_per_step"), j.get("tokens_equal"), j.get("iterations_per_model_step"), (j.get("pcie_inclusive") or {}).get("pageable", {}).get("value") if isinstance(j.get("pcie_inclusive"), dict) else "", j["config"].get("engine", {}).get("fe_ahead"))
    except Exception as e:
        print(f, "ERR", e)
PY
