"""gpurun_out/r6final (tools/profile_round6.sh) -> profiles/r06/ + profiles/cell_rocprof.json.

cell_rocprof.json: per workload, the rocprofv3 --kernel-trace duration of the dominant kernel (the encoder cell) in the traced
bench run: `avg_us` = mean WITHOUT the slowest 1 % of its launches (`trim_us` of tools/rocpd_stats.py; one first-touch launch of
25-32 ms among a few thousand turned round 5's cfg5 cell into "34 us" where the median is 25), beside the plain mean and the median.
bench.py reports it as `roofline.launch_us_rocprof` / `frac_rocprof` BESIDE the figure of its own run."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r6final")
DST = os.path.join(ROOT, "profiles", "r06")
os.makedirs(DST, exist_ok=True)

NAMES = {
    "bench_driverform.json": "r06_bench_driverform.json", "bench_bf16.json": "r06_bench_bf16.json",
    "bench_bf16_beam4.json": "r06_bench_bf16_beam4.json", "bench_cfg5_bf16.json": "r06_bench_cfg5_bf16.json",
    "bench_cfg5_bf16_beam8.json": "r06_bench_cfg5_bf16_beam8.json", "bench_f32_sync.json": "r06_bench_f32_sync.json",
    "bench_8rank_dryrun.json": "r06_bench_8rank_dryrun.json", "bench_8rank_dryrun_nap0.json": "r06_bench_8rank_dryrun_nap0_nopin.json",
    "bench_rccl_world1.json": "r06_bench_rccl_world1.json", "kt_driverform.json": "r06_bench_driverform_under_rocprof.json",
    "kernel_stats_driver.txt": "r06_bench_driverform_kernel_stats.txt", "kernel_gaps_driver.txt": "r06_bench_driverform_kernel_gaps.txt",
    "kernel_stats_bf16.txt": "r06_bench_bf16_kernel_stats.txt", "kernel_stats_beam.txt": "r06_bench_bf16_beam4_kernel_stats.txt",
    "kernel_stats_cfg5.txt": "r06_bench_cfg5_bf16_beam8_kernel_stats.txt", "timeline_f32.txt": "r06_stream_timeline_f32.txt",
    "cell_pmc_cfg5_job_vs_isolated.txt": "r06_cell_pmc_cfg5_job_vs_isolated.txt",
}
for src, dst in NAMES.items():
    p = os.path.join(SRC, src)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, dst))
    else:
        print("missing", src)


def stats_row(path, pattern):
    if not os.path.exists(path):
        return None
    for ln in open(path).read().splitlines()[1:]:
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if m and re.search(pattern, m.group(1)):
            return dict(kernel=m.group(1).strip(), calls=int(m.group(2)), mean_us=float(m.group(4)), med_us=float(m.group(8)), p95_us=float(m.group(9)), trim_us=float(m.group(10)))
    return None


rocprof = {}
try:
    rocprof = json.load(open(os.path.join(ROOT, "profiles", "cell_rocprof.json")))      # (workloads not re-traced keep their earlier entry)
except Exception:
    pass
for key, f, pat, cmd in [
    ("cfg2_f32_64_beam1", "kernel_stats_driver.txt", r"EpiLSTM<OpsF32, false, false, 8", "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustained-s 0 --other-configs 0"),
    ("cfg2_bf16_64_beam1", "kernel_stats_bf16.txt", r"k_gemm_multi<OpsBF16, EpiLSTM<OpsBF16, false, false, 8", "... --no-extras --dtype bf16"),
    ("cfg2_bf16_64_beam4", "kernel_stats_beam.txt", r"k_gemm_multi<OpsBF16, EpiLSTM<OpsBF16, false, false, 8", "... --no-extras --dtype bf16 --beam 4 --steps 4 --warmup 1"),
    ("cfg5_bf16_128_beam8", "kernel_stats_cfg5.txt", r"EpiLSTMe<OpsBF16, 12", "... --no-extras --model cfg5 --dtype bf16 --streams 128 --beam 8 --steps 4 --warmup 1"),
]:
    r = stats_row(os.path.join(SRC, f), pat)
    if r:
        rocprof[key] = {"avg_us": round(r["trim_us"], 3), "mean_all_launches_us": r["mean_us"], "median_us": r["med_us"], "p95_us": r["p95_us"],
                        "calls": r["calls"], "kernel": r["kernel"], "command": "rocprofv3 --kernel-trace --stats -- " + cmd,
                        "file": "profiles/r06/" + NAMES[f],
                        "note": "avg_us = mean without the slowest 1 % of the launches of the traced process (warm-up, profiled region and extra legs "
                                "included; the tracer slows the host); mean_all_launches_us keeps them (first-touch outliers of tens of ms)"}
json.dump(rocprof, open(os.path.join(ROOT, "profiles", "cell_rocprof.json"), "w"), indent=1)
print(json.dumps({k: (v["avg_us"], v.get("median_us"), v["file"]) for k, v in rocprof.items()}, indent=1))
