import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libreasr_amd import synth
from libreasr_amd.engine import Engine
name, W, B, n_chunks, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cfg = synth.model_cfg(name); sd = synth.synth_state_dict(cfg, seed=0)
L = cfg["enc_layers"]
pcm = np.stack([synth.synth_pcm(1, n_chunks * 1280, seed=1234 + s)[0] for s in range(B)])
eng = Engine(sd, cfg, max_streams=B, dtype="bf16", beam=W)
slots = [eng.open() for _ in range(B)]
def run(mode):
    for s in slots: eng.reset(s, 15)
    for k in range(n_chunks):
        if mode == "sync":
            eng.push(slots, pcm[:, k * 1280:(k + 1) * 1280])
            if eng.step(slots): eng.fetch_many(slots, 8192)
            continue
        eng.push_submit(slots, pcm[:, k * 1280:(k + 1) * 1280])
        while eng.pending() >= 6:
            if eng.wait(): eng.fetch_many(slots, 8192)
    while eng.pending():
        if eng.wait(): eng.fetch_many(slots, 8192)
    return [eng.debug_read("x0", 0), eng.debug_read("x0", 1)] + [eng.debug_read("enc_h", l) for l in range(L)] + [eng.debug_read("enc_c", l) for l in range(L)] + [eng.debug_read("pend")]
ref = run("sync")
bad = 0
for r in range(N):
    cur = run("pipe")
    d = [float(np.abs(a[:B] - b[:B]).max()) for a, b in zip(ref, cur)]
    if max(d) > 0:
        bad += 1
        rows = sorted(set(int(i) for a, b in zip(ref, cur) for i in np.nonzero(np.abs(a[:B] - b[:B]).max(1))[0]))
        print(f"run {r}: x0 {d[0]:.3g} {d[1]:.3g} enc_h {[round(x, 5) for x in d[2:2 + L]]} enc_c {[round(x, 6) for x in d[2 + L:2 + 2 * L]]} pend {d[-1]:.3g} rows {rows[:10]}", flush=True)
print("runs that differ:", bad, "of", N)
eng.close()
