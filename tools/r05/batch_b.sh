#!/bin/bash
# Round 5, batch B: x side of a layer's frames as ONE GEMM per model step (LASR_ENC_XG=1) against the fused [x, h] cell:
# the whole GPU suite under the switch, then A/B on configs[1] (f32), bf16 greedy, cfg5 (bf16, 128 streams) greedy / beam 8
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err || echo "rc $? $n" >> $O/failures.txt; }
timeout 600 python -m pytest tests/test_gpu_round5.py -q -x > $O/pytest_round5.txt 2>&1; tail -3 $O/pytest_round5.txt
LASR_ENC_XG=1 timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_xg1.txt 2>&1; tail -5 $O/pytest_xg1.txt
B="python bench.py --gpus 1 --no-cpu-baseline --steps 40 --warmup 5 --sustained-s 0"
for i in a b; do
  LASR_ENC_XG=0 run f32_xg0_$i $B
  LASR_ENC_XG=1 run f32_xg1_$i $B
done
LASR_ENC_XG=0 run bf16_xg0 $B --dtype bf16
LASR_ENC_XG=1 run bf16_xg1 $B --dtype bf16
LASR_ENC_XG=0 LASR_ENC_WAVE=0 run bf16_xg0_wave0 $B --dtype bf16
C5="python bench.py --gpus 1 --no-cpu-baseline --no-extras --model cfg5 --dtype bf16 --streams 128 --steps 16 --warmup 4 --sustained-s 0"
LASR_ENC_XG=0 run cfg5_xg0 $C5
LASR_ENC_XG=1 run cfg5_xg1 $C5
LASR_ENC_XG=0 run cfg5_beam8_xg0 $C5 --beam 8 --steps 8 --warmup 2
LASR_ENC_XG=1 run cfg5_beam8_xg1 $C5 --beam 8 --steps 8 --warmup 2
LASR_ENC_XG=0 run cfg2_beam4_xg0 $B --no-extras --dtype bf16 --beam 4 --steps 10 --warmup 3
LASR_ENC_XG=1 run cfg2_beam4_xg1 $B --no-extras --dtype bf16 --beam 4 --steps 10 --warmup 3
python tools/r05/summ.py $O/*.json | tee $O/summary.txt
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r5b/*.json")):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        continue
    r = d.get("roofline", {})
    print(p.split("/")[-1], "isolated", r.get("launch_us_isolated"), "offline", (d.get("offline") or {}).get("audio_sec_per_sec"), "pcie", (d.get("pcie_inclusive") or {}).get("value"))
PY
cat $O/failures.txt 2>/dev/null
# the weight-stationary / step-persistent recurrent cell against launch-per-step: stand-alone probe
timeout 120 tools/probe/persist_cell_probe 64 5 > $O/persist_cell_probe.jsonl 2> $O/persist_cell_probe.err; echo "probe rc $?"; cat $O/persist_cell_probe.jsonl
