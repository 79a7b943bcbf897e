"""Where does the HOST time of a pipelined streaming step go?  (run on the GPU box)
Per 80 ms chunk: push, submit, and every `depth` steps wait + fetch_many.  If the time inside wait()
(mostly spinning on the pinned flag) is small, the loop is host-bound, not GPU-bound."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libreasr_amd import synth
from libreasr_amd.engine import Engine
cfg = synth.model_cfg("cfg2"); sd = synth.synth_state_dict(cfg)
B, depth = 64, int(os.environ.get("DEPTH", "6"))
eng = Engine(sd, cfg, max_streams=B, dtype=os.environ.get("LASR_DTYPE", "f32"))
n = 260
pcm = torch.as_tensor(np.stack([synth.synth_pcm(1, n * 1280, seed=1234 + s)[0] for s in range(B)]).reshape(B, n, 1280).transpose(1, 0, 2).copy()).cuda()
slots = [eng.open() for _ in range(B)]
T = {"index": [], "push": [], "submit_odd": [], "submit_model": [], "wait": [], "fetch": []}
t_start = None
for k in range(n):
    if k == 40: t_start = time.perf_counter()
    t0 = time.perf_counter(); x = pcm[k]; t1 = time.perf_counter()
    eng.push(slots, x); t2 = time.perf_counter()
    before = eng.pending(); eng.submit(slots); t3 = time.perf_counter()
    ran = eng.pending() > before
    tw = tf = 0.0
    if eng.pending() >= depth:
        t4 = time.perf_counter(); eng.wait(); t5 = time.perf_counter(); eng.fetch_many(slots, cap=64); t6 = time.perf_counter()
        tw, tf = t5 - t4, t6 - t5
    if k >= 40:
        T["index"].append(t1 - t0); T["push"].append(t2 - t1)
        (T["submit_model"] if ran else T["submit_odd"]).append(t3 - t2)
        if tw: T["wait"].append(tw); T["fetch"].append(tf)
total = time.perf_counter() - t_start
while eng.pending(): eng.wait()
steps = (n - 40) / 2
print(f"wall per model step {1e6*total/steps:.1f} us  ({(n-40)*B*0.08/total:.0f} audio-s/s)")
for k, v in T.items():
    print(f"{k:13s} median {1e6*np.median(v):7.1f} us  mean {1e6*np.mean(v):7.1f} us  per model step {1e6*np.sum(v)/steps:7.1f} us  n={len(v)}")
