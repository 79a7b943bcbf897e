#!/usr/bin/env python
"""Steady-state view of a bench.py kernel trace: per model step (two k_push_pcm launches), wall time,
GPU busy time (union of kernel intervals), per-queue busy time and the idle gaps."""
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select start, end, name, queue_id from kernels order by start").fetchall()
push = [i for i, r in enumerate(rows) if "k_push_pcm" in r[2]]
lo, hi = push[int(len(push) * 0.35)], push[int(len(push) * 0.85)]
n_chunks = int(len(push) * 0.85) - int(len(push) * 0.35)
win = rows[lo:hi]
span = win[-1][0] - win[0][0]
def union(iv):
    iv = sorted(iv); u = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: u += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return u + ce - cs
tot = sum(e - s for s, e, _, _ in win)
u = union([(s, e) for s, e, _, _ in win])
print(f"{n_chunks} chunks ({n_chunks/2:.0f} model steps): wall {span/n_chunks*2/1e3:.1f} us/model step; "
      f"kernel time {tot/n_chunks*2/1e3:.1f} us/step; GPU busy (union) {u/n_chunks*2/1e3:.1f} us/step "
      f"= {100*u/span:.0f}% of wall; overlap factor {tot/u:.2f}")
byq = collections.defaultdict(list)
for s, e, n, q in win: byq[q].append((s, e))
for q, iv in sorted(byq.items()):
    print(f"  queue {q}: {len(iv)/n_chunks*2:.1f} kernels/step, busy {union(iv)/n_chunks*2/1e3:.1f} us/step")
byn = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in win: k = n.replace("lasr::", "").replace("void ", "").split("(")[0][:80]; byn[k][0] += e - s; byn[k][1] += 1
for n, (t, c) in sorted(byn.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {t/n_chunks*2/1e3:7.1f} us/step  {c/n_chunks*2:5.1f} calls/step  {t/c/1e3:6.2f} us  {n}")
