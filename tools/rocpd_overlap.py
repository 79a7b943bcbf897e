#!/usr/bin/env python
"""GPU busy time (union of kernel intervals) vs sum of kernel durations vs span: how much do the
kernels of the two HIP streams overlap?"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = sorted(cur.execute("select start, end, name from kernels").fetchall())
skip = int(len(rows) * 0.3)                      # drop warm-up / model load
rows = rows[skip:]
tot = sum(e - s for s, e, _ in rows)
union, cs, ce = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > ce:
        union += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
span = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)}  span {span/1e3:.0f} us  sum of durations {tot/1e3:.0f} us ({100*tot/span:.0f}% of span)  "
      f"GPU busy (union) {union/1e3:.0f} us ({100*union/span:.0f}% of span)  overlap factor {tot/union:.2f}")
