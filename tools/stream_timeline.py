#!/usr/bin/env python
"""Two-stream timeline of the pipelined protocol from a `bench.py --trace FILE` dump (lasr_trace marks, no profiler
in the process).  Main stream: 1 push, 3 first cell, 4 cells done, 5 model step enqueued.  Decode stream: 10 group
reached, 11+100G(+1000 admitted) admission done, 12 group done."""
import json, sys
import numpy as np
d = json.load(open(sys.argv[1]))
notes = [us for t, us in d["marks"] if t == 20][2:]
m = [(t, us) for t, us in d["marks"] if us >= 0 and t != 20]
n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 0
main = [(t, us) for t, us in m if t < 10]
dec = [(t, us) for t, us in m if t >= 10]
# main stream: per model step  [1 .. 1 .. 3 .. 4 .. 5]
steps = []
cur = {}
for t, us in main:
    if t == 1: cur.setdefault("push", []).append(us)
    elif t == 3: cur["c0"] = us
    elif t == 4: cur["c1"] = us
    elif t == 5:
        cur["end"] = us
        if "c0" in cur and len(cur.get("push", [])) >= 2: steps.append(cur)
        cur = {}
steps = steps[len(steps) // 5:]
period = np.diff([s["end"] for s in steps])
fe = [s["c0"] - s["push"][-2] for s in steps]
cells = [s["c1"] - s["c0"] for s in steps]
tail = [s["end"] - s["c1"] for s in steps]
idle = [b["push"][-2] - a["end"] for a, b in zip(steps[:-1], steps[1:])]
print(f"main stream, {len(steps)} model steps: period {np.mean(period):.1f} us (p50 {np.median(period):.1f}); "
      f"FE (first push -> first cell) {np.mean(fe):.1f}; cells {np.mean(cells):.1f}; tail (pe GEMM, advance) {np.mean(tail):.1f}; "
      f"idle before the next push {np.mean(idle):.1f} (p50 {np.median(idle):.1f})")
# decode stream: groups
groups = []
g = {}
for t, us in dec:
    if t == 10: g = {"reach": us}
    elif t % 100 == 11: g["adm"] = us; g["G"] = (t % 1000) // 100; g["admitted"] = t >= 1000
    elif t == 12:
        g["end"] = us
        if "adm" in g and "reach" in g: groups.append(g)
        g = {}
groups = [q for q in groups if q["reach"] >= steps[0]["push"][0]]
span = groups[-1]["end"] - groups[0]["reach"]
busy = sum(q["end"] - q["adm"] for q in groups)
waitadm = sum(q["adm"] - q["reach"] for q in groups)
gaps = sum(max(0.0, b["reach"] - a["end"]) for a, b in zip(groups[:-1], groups[1:]))
iters = sum(q["G"] for q in groups)
nsteps = span / np.mean(period)
print(f"decode stream over {span:.0f} us (~{nsteps:.1f} model steps): {len(groups)} groups, {iters} iterations "
      f"({iters / nsteps:.2f} per step); iterating {100 * busy / span:.0f} %, waiting for an encoder (admission) "
      f"{100 * waitadm / span:.0f} %, idle between groups (host) {100 * gaps / span:.0f} %; "
      f"{busy / iters:.1f} us per iteration")
if notes:
    need = np.array([int(v // 1000) for v in notes]); G = np.array([int(v % 1000) // 100 for v in notes])
    left = np.array([int(v % 100) for v in notes]); rows = np.array([round((v % 1) * 1000) for v in notes])
    print(f"groups consumed {len(notes)}: iterations launched {G.sum()}, needed (max decisions of a row) {need.sum()} "
          f"= {100 * need.sum() / G.sum():.0f} %; groups that ended with frames left {100 * np.mean(left > 0):.0f} %; "
          f"rows that moved per group {rows.mean():.1f}; by G: " +
          ", ".join(f"G={g}: n={np.sum(G == g)} need {need[G == g].mean():.2f} left>0 {100 * np.mean(left[G == g] > 0):.0f}%" for g in sorted(set(G))))
if n_show:
    t0 = steps[0]["push"][0]
    ev = sorted([(us, "M", t) for t, us in main if us >= t0] + [(us, "D", t) for t, us in dec if us >= t0])[:n_show]
    for us, s, t in ev:
        print(f"{us - t0:9.1f}  {'main  ' if s == 'M' else '            dec '} {t}")
