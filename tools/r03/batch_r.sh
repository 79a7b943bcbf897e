R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3r; mkdir -p $O
cd $R
B="timeout 150 python3 bench.py --no-cpu-baseline --check-rows 0 --steps 40"
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
    print("$name", d["value"], "pageable", d.get("pcie_inclusive",{}).get("value"), "pinned_nocopy", d.get("pcie_inclusive",{}).get("pinned_nocopy",{}).get("value"))
except Exception as e: print("$name ERR", e, open("$O/$name.err").read()[-300:])
PY
}
run dma $B
LASR_PUSH_ZEROCOPY=1 run zerocopy $B
run dma2 $B
LASR_PUSH_ZEROCOPY=1 run zerocopy2 $B
