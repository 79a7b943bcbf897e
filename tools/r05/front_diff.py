"""Bisect: native front with the reset rule, depth 12 (early verdicts on / off) against depth 1."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from libreasr_amd import synth
from libreasr_amd.engine import Engine
from libreasr_amd.front import bench_native_producers
cfg = synth.model_cfg("cfg2"); sd = synth.synth_state_dict(cfg, seed=0)
B = 64
eng = Engine(sd, cfg, max_streams=B)
base = np.stack([synth.synth_pcm(1, 64 * 1280, seed=1234 + s)[0] for s in range(B)])
pcm = np.concatenate([base] * 4, axis=1)
def run(depth, rule):
    t, sec, st = bench_native_producers(eng, pcm, depth=depth, reset_steps=rule, cap=8192)
    return t, st
t1, s1 = run(1, 25)
t1b, s1b = run(1, 25)
print("depth1 repeat equal:", t1 == t1b, s1, s1b)
for name, d, r in (("d12_early", 12, 25), ("d12_noearly", 12, -25), ("d6_early", 6, 25), ("d2_noearly", 2, -25), ("d12_early_again", 12, 25)):
    t, st = run(d, r)
    bad = [i for i in range(B) if t[i] != t1[i]]
    print(name, "equal" if not bad else f"differ in streams {bad}", st)
    for i in bad[:2]:
        a, b = t[i], t1[i]
        k = next((j for j in range(min(len(a), len(b))) if a[j] != b[j]), min(len(a), len(b)))
        print("   stream", i, "len", len(a), len(b), "first diff at", k, a[max(0,k-3):k+5], b[max(0,k-3):k+5])
eng.close()
