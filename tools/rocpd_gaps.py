#!/usr/bin/env python
"""Per-queue view of a rocprofv3 kernel trace (rocpd sqlite): for every (kernel -> next kernel on the same queue) pair of the
steady-state middle third, the gap between the end of the first and the start of the second, and the per-kernel durations.
Answers "is the time between kernels, or inside them?" for the two streams of the pipelined protocol."""
import collections
import re
import sqlite3
import sys

import numpy as np


def short(n):
    n = n.replace("void lasr::", "").replace("lasr::", "")
    return re.sub(r"\(.*$", "", n)[:52]


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select start, end, name, queue_id from kernels order by start").fetchall()
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[3]].append(r)
    for q, rs in sorted(byq.items()):
        rs = rs[len(rs) // 3: 2 * len(rs) // 3]
        if len(rs) < 8:
            continue
        span = (rs[-1][1] - rs[0][0]) / 1e3
        busy = sum(r[1] - r[0] for r in rs) / 1e3
        print(f"queue {q}: {len(rs)} kernels over {span:.0f} us, in kernels {busy:.0f} us ({100 * busy / span:.0f} %)")
        gaps = collections.defaultdict(list)
        for a, b in zip(rs[:-1], rs[1:]):
            gaps[(short(a[2]), short(b[2]))].append((b[0] - a[1]) / 1e3)
        for k, v in sorted(gaps.items(), key=lambda kv: -np.sum(kv[1]))[:12]:
            print(f"   {k[0]:52s} -> {k[1]:52s} n={len(v):4d} gap mean {np.mean(v):6.2f} p50 {np.median(v):6.2f} "
                  f"p90 {np.percentile(v, 90):6.2f} us")
        dur = collections.defaultdict(list)
        for r in rs:
            dur[short(r[2])].append((r[1] - r[0]) / 1e3)
        for k, v in sorted(dur.items(), key=lambda kv: -np.sum(kv[1]))[:10]:
            print(f"   dur {k:52s} n={len(v):4d} mean {np.mean(v):6.2f} p50 {np.median(v):6.2f} min {np.min(v):6.2f} us")


if __name__ == "__main__":
    main(sys.argv[1])
