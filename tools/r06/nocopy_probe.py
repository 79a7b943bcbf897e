"""Why is LASR_PUSH_PINNED_NOCOPY slower than the copying path (39 k against 52 k)?  Host time per push_submit
call and whole-leg rate for: pageable numpy (copied), torch-pinned (copied), torch-pinned NOCOPY, one hipHostMalloc'ed block NOCOPY
(torch.empty(pin_memory=True) slices vs. a single registered buffer), in both orders."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from libreasr_amd import synth
from libreasr_amd.engine import Engine

cfg = synth.model_cfg("cfg2"); sd = synth.synth_state_dict(cfg, seed=0)
B, CH, N = 64, 1280, 320
eng = Engine(sd, cfg, max_streams=B)
slots = [eng.open() for _ in range(B)]
pcm = np.stack([synth.synth_pcm(1, N * CH, seed=1234 + s)[0] for s in range(B)]).reshape(B, N, CH).transpose(1, 0, 2).copy()
pinned = torch.from_numpy(pcm).pin_memory()
pinned_list = [torch.from_numpy(pcm[k].copy()).pin_memory() for k in range(N)]       # one allocation per chunk
res = {}
def leg(name, src, nocopy, K=320, depth=18):
    t_call = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        a = time.perf_counter()
        eng.push_submit(slots, src(k % N), pinned_nocopy=nocopy)
        t_call += time.perf_counter() - a
        while eng.pending() >= depth:
            eng.wait(); eng.fetch_many(slots, 64)
    while eng.pending():
        eng.wait(); eng.fetch_many(slots, 64)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res.setdefault(name, []).append({"audio_s_per_s": round(K * B * CH / 16000 / dt, 1), "host_us_per_push": round(1e6 * t_call / K, 1)})
for rep in range(2):
    order = [("pageable_copied", lambda k: pcm[k], False), ("pinned_copied", lambda k: pinned[k], False),
             ("pinned_view_nocopy", lambda k: pinned[k], True), ("pinned_own_alloc_nocopy", lambda k: pinned_list[k], True)]
    if rep: order = order[::-1]
    for name, src, nc in order:
        leg(name, src, nc, K=64)          # untimed warm-up of the mode
        leg(name + "_timed", src, nc)
print(json.dumps({k: v for k, v in res.items() if k.endswith("_timed")}, indent=1))
eng.close()
