"""Where does the wall time of a streaming step go?  (run on the GPU box)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libreasr_amd import synth
from libreasr_amd.engine import Engine
cfg = synth.model_cfg("cfg2"); sd = synth.synth_state_dict(cfg)
B = 64
eng = Engine(sd, cfg, max_streams=B)
n = 64
pcm = torch.as_tensor(np.stack([synth.synth_pcm(1, n * 1280, seed=1234 + s)[0] for s in range(B)]).reshape(B, n, 1280).transpose(1, 0, 2).copy()).cuda()
slots = [eng.open() for _ in range(B)]
T = {"index": [], "push": [], "step_odd": [], "step_model": [], "fetch": []}
for k in range(n):
    t0 = time.perf_counter(); x = pcm[k]; t1 = time.perf_counter()
    eng.push(slots, x); t2 = time.perf_counter()
    ran = eng.step(slots); t3 = time.perf_counter()
    if ran:
        for s in slots: eng.fetch(s, cap=256)
    t4 = time.perf_counter()
    if k > 8:
        T["index"].append(t1 - t0); T["push"].append(t2 - t1)
        (T["step_model"] if ran else T["step_odd"]).append(t3 - t2)
        if ran: T["fetch"].append(t4 - t3)
for k, v in T.items():
    print(f"{k:12s} median {1e6*np.median(v):8.1f} us  mean {1e6*np.mean(v):8.1f} us  n={len(v)}")
print(eng.stats())
