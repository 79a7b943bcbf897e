#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls, total,
average, min, max duration -- the `--stats` table as text, for committing under profiles/."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void lasr::", "").replace("lasr::", "")
    return name[:110]


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if {"name", "start", "end"} <= set(cols) else []
    agg, durs = {}, {}
    for name, s, e in rows:
        d = (e - s) / 1e3
        a = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        durs.setdefault(short(name), []).append(d)
    tot = sum(a[1] for a in agg.values()) or 1.0
    # med_us / p95_us / trim_us (mean without the slowest 1 % of the launches): one first-touch launch of 25 ms in 2 591 turned a
    # 24 us kernel into "34 us on average" in round 5's table (profiles/r05/r05_bench_cfg5_bf16_beam8_kernel_stats.txt)
    lines = [f"{'kernel':<112} {'calls':>7} {'total_us':>12} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6} {'med_us':>9} {'p95_us':>9} {'trim_us':>9}"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        v = sorted(durs[k])
        keep = v[:max(1, len(v) - max(1, len(v) // 100))] if len(v) >= 20 else v
        lines.append(f"{k:<112} {a[0]:>7} {a[1]:>12.1f} {a[1]/a[0]:>9.2f} {a[2]:>9.2f} {a[3]:>9.2f} {100*a[1]/tot:>6.2f} "
                     f"{v[len(v) // 2]:>9.2f} {v[min(len(v) - 1, int(0.95 * len(v)))]:>9.2f} {sum(keep) / len(keep):>9.2f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
