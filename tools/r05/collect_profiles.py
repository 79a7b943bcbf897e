"""gpurun_out/r5final (tools/profile_round5.sh) -> profiles/r05/ + profiles/cell_rocprof.json + profiles/cell_pmc.json.

cell_rocprof.json: per workload, the rocprofv3 --kernel-trace average duration of the dominant kernel (the encoder cell) in the
traced bench run; bench.py puts it beside its own in-kernel clocks (`roofline.launch_us_rocprof`) and computes `frac` from the
larger of the two.  cell_pmc.json: per workload, the PMC passes of the isolated cell (tools/cellbench.py): HBM bytes per launch
(FETCH_SIZE x 2 per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE), L2 hit rate, requests through L2, MFMA busy."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r5final")
DST = os.path.join(ROOT, "profiles", "r05")
os.makedirs(DST, exist_ok=True)


def copy(src, dst):
    p = os.path.join(SRC, src)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, dst))
        return True
    print("missing", src)
    return False


for src, dst in [
    ("bench_driverform.json", "r05_bench_driverform.json"), ("bench_f32_steps100.json", "r05_bench_f32_steps100.json"),
    ("bench_bf16.json", "r05_bench_bf16.json"), ("bench_bf16_beam4.json", "r05_bench_bf16_beam4.json"),
    ("bench_cfg5_bf16.json", "r05_bench_cfg5_bf16.json"), ("bench_cfg5_bf16_beam8.json", "r05_bench_cfg5_bf16_beam8.json"),
    ("bench_cfg5_bf16_beam8_depth6.json", "r05_bench_cfg5_bf16_beam8_depth6.json"), ("bench_f32_sync.json", "r05_bench_f32_sync.json"),
    ("bench_8rank_dryrun.json", "r05_bench_8rank_dryrun.json"), ("bench_rccl_world1.json", "r05_bench_rccl_world1.json"),
    ("kt_driverform.json", "r05_bench_driverform_under_rocprof.json"),
    ("kernel_stats_driver.txt", "r05_bench_driverform_kernel_stats.txt"), ("kernel_gaps_driver.txt", "r05_bench_driverform_kernel_gaps.txt"),
    ("kernel_stats_steady.txt", "r05_bench_f32_noextras_kernel_stats.txt"), ("kernel_steady_f32.txt", "r05_bench_f32_steady_window.txt"),
    ("kernel_stats_bf16.txt", "r05_bench_bf16_kernel_stats.txt"), ("kernel_stats_beam.txt", "r05_bench_bf16_beam4_kernel_stats.txt"),
    ("kernel_stats_cfg5.txt", "r05_bench_cfg5_bf16_beam8_kernel_stats.txt"), ("kernel_stats_cfg5g.txt", "r05_bench_cfg5_bf16_kernel_stats.txt"),
    ("timeline_f32.txt", "r05_stream_timeline_f32.txt"), ("cell_pmc.txt", "r05_cell_pmc.txt"), ("pmc_counter_names.txt", "r05_pmc_counter_names.txt"),
]:
    copy(src, dst)


def stats_avg(path, pattern):
    """(avg_us, calls, name) of the first kernel of a rocpd_stats table whose name matches."""
    if not os.path.exists(path):
        return None
    for ln in open(path).read().splitlines()[1:]:
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if m and re.search(pattern, m.group(1)):
            return float(m.group(4)), int(m.group(2)), m.group(1).strip()
    return None


rocprof = {}
for key, f, pat, cmd in [
    ("cfg2_f32_64_beam1", "kernel_stats_driver.txt", r"EpiLSTM<OpsF32, false, false, 8", "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustained-s 0"),
    ("cfg2_bf16_64_beam1", "kernel_stats_bf16.txt", r"k_gemm_multi<OpsBF16, EpiLSTM<OpsBF16, false, false, 8", "python bench.py --no-cpu-baseline --no-extras --sustained-s 0 --dtype bf16"),
    ("cfg2_bf16_64_beam4", "kernel_stats_beam.txt", r"k_gemm_multi<OpsBF16, EpiLSTM<OpsBF16, false, false, 8", "... --dtype bf16 --beam 4 --steps 10 --warmup 3"),
    ("cfg5_bf16_128_beam8", "kernel_stats_cfg5.txt", r"EpiLSTMe<OpsBF16, 12", "... --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 8 --warmup 2"),
    ("cfg5_bf16_128_beam1", "kernel_stats_cfg5g.txt", r"EpiLSTMe<OpsBF16, 12", "... --model cfg5 --dtype bf16 --streams 128 --depth 6"),
]:
    r = stats_avg(os.path.join(SRC, f), pat)
    if r:
        rocprof[key] = {"avg_us": round(r[0], 3), "calls": r[1], "kernel": r[2], "command": "rocprofv3 --kernel-trace --stats -- " + cmd,
                        "file": "profiles/r05/" + {"kernel_stats_driver.txt": "r05_bench_driverform_kernel_stats.txt", "kernel_stats_bf16.txt": "r05_bench_bf16_kernel_stats.txt",
                                                   "kernel_stats_beam.txt": "r05_bench_bf16_beam4_kernel_stats.txt", "kernel_stats_cfg5.txt": "r05_bench_cfg5_bf16_beam8_kernel_stats.txt",
                                                   "kernel_stats_cfg5g.txt": "r05_bench_cfg5_bf16_kernel_stats.txt"}[f],
                        "note": "average over every launch of the traced process (warm-up, profiled region, offline / PCIe legs included); the tracer slows the host"}
json.dump(rocprof, open(os.path.join(ROOT, "profiles", "cell_rocprof.json"), "w"), indent=1)
print(json.dumps(rocprof, indent=1))

# PMC passes
txt = open(os.path.join(SRC, "cell_pmc.txt")).read() if os.path.exists(os.path.join(SRC, "cell_pmc.txt")) else ""
vals = {}
cur_db, cur_k = None, None
for ln in txt.splitlines():
    if ln.startswith("== "):
        cur_db = re.search(r"pmc_(f32|bf16|cfg5)_\d+", ln).group(1)
    elif ln.startswith("k_gemm"):
        cur_k = ln.strip()
    else:
        m = re.match(r"\s+(\S+)\s+per-dispatch\s+([\d.]+)\s+\(dispatches (\d+)\)", ln)
        if m and int(m.group(3)) > 4 and ("false, false, 8" in cur_k or "EpiLSTMe" in cur_k):
            vals.setdefault(cur_db, {"kernel": cur_k})[m.group(1)] = float(m.group(2))
pmc = {}
shape = {"f32": ("cfg2_f32_64_beam1", 4.0 * 4 * 1024 * 2048, 13.6), "bf16": ("cfg2_bf16_64_beam1", 2.0 * 4 * 1024 * 2048, 5.7),
         "cfg5": ("cfg5_bf16_128_beam1", 2.0 * 4 * 1536 * 3072, 13.0)}
for k, v in vals.items():
    key, wbytes, iso_us = shape[k]
    hbm = (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0
    e = {"kernel": v["kernel"], "fetch_size_kib": v.get("FETCH_SIZE"), "fetch_correction": 2.0, "write_size_kib": v.get("WRITE_SIZE"),
         "hbm_bytes_per_launch": int(hbm), "algorithmic_weight_bytes": int(wbytes), "hbm_over_weights": round(hbm / wbytes, 3),
         "tcc_hit_rate": round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 3) if "TCC_HIT_sum" in v else None,
         "l2_requests": v.get("TCC_REQ_sum"), "l1_to_l2_read_requests": v.get("TCP_TCC_READ_REQ_sum"),
         "l2_bytes_at_128B_per_request": int(v["TCC_REQ_sum"] * 128) if "TCC_REQ_sum" in v else None,
         "l1_to_l2_read_bytes_at_128B": int(v["TCP_TCC_READ_REQ_sum"] * 128) if "TCP_TCC_READ_REQ_sum" in v else None,
         "mfma_busy_cycles_per_simd": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0, 1) if "SQ_VALU_MFMA_BUSY_CYCLES" in v else None,
         "mfma_busy_frac_of_isolated_launch_at_2p4GHz": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (iso_us * 2400.0), 3) if "SQ_VALU_MFMA_BUSY_CYCLES" in v else None,
         "wave_cycles": {n: v.get(n) for n in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if n in v},
         "file": "profiles/r05/r05_cell_pmc.txt",
         "how": "rocprofv3 --kernel-trace --pmc <one counter group per pass> -- python tools/cellbench.py (the cell alone, 33 launches per pass); "
                "HBM bytes = FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE; requests counted at 128 B (1 KiB wave loads = 8 lines): "
                "the L1->L2 read bytes of the f32 cell come out at the tiling's arithmetic (256 workgroups x 524 KB = 134 MB)"}
    pmc[key] = e
    if k == "bf16":
        pmc["cfg2_bf16_64_beam4"] = e
    if k == "cfg5":
        pmc["cfg5_bf16_128_beam8"] = e
if pmc:          # (a run with SKIP_PMC=1 keeps the committed passes: the cell kernels have not changed since)
    json.dump(pmc, open(os.path.join(ROOT, "profiles", "cell_pmc.json"), "w"), indent=1)
print(json.dumps({k: {q: v[q] for q in ("hbm_bytes_per_launch", "hbm_over_weights", "tcc_hit_rate", "l2_bytes_at_128B_per_request", "mfma_busy_frac_of_isolated_launch_at_2p4GHz")} for k, v in pmc.items()}, indent=1))
for extra in ("parity_counts.json", "served_rate.json", "pytest_gpu.txt"):
    p = os.path.join(ROOT, "gpurun_out", "r5d", extra)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, extra if extra != "pytest_gpu.txt" else "r05_pytest_gpu_tail.txt"))
