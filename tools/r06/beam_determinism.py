"""Is the pipelined beam path deterministic?  cfg5 bf16 beam 8, 128 streams x 48 chunks, N runs in one process (fresh engine each):
best hypothesis after every model step and the final score of every stream must be identical between runs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libreasr_amd import synth
from libreasr_amd.engine import Engine
name, W, B = sys.argv[1] if len(sys.argv) > 1 else "cfg5", int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 128
n_chunks = int(os.environ.get("CHUNKS", "48"))
N = int(sys.argv[4]) if len(sys.argv) > 4 else 8
MODE = os.environ.get("MODE", "pipe"); DEPTH = int(os.environ.get("DEPTH", "6"))
cfg = synth.model_cfg(name); sd = synth.synth_state_dict(cfg, seed=0)
pcm = np.stack([synth.synth_pcm(1, n_chunks * 1280, seed=1234 + s)[0] for s in range(B)])
ref = None
for run in range(N):
    eng = Engine(sd, cfg, max_streams=B, dtype="bf16", beam=W)
    slots = [eng.open() for _ in range(B)]
    hist = [[] for _ in range(B)]; score = [0.0] * B
    def take():
        for i in range(B):
            t, nl, _ = eng.fetch(slots[i])
            hist[i].append(t if t else (hist[i][-1] if hist[i] else []))
            score[i] = -nl
    for k in range(n_chunks):
        if MODE == "sync":
            eng.push(slots, pcm[:, k * 1280:(k + 1) * 1280])
            if eng.step(slots):
                take()
            continue
        eng.push_submit(slots, pcm[:, k * 1280:(k + 1) * 1280])
        while eng.pending() >= DEPTH:
            if eng.wait():
                take()
    while eng.pending():
        if eng.wait():
            take()
    eng.close()
    cur = (hist, score)
    if ref is None:
        ref = cur
    else:
        dh = [i for i in range(B) if hist[i] != ref[0][i]]
        ds = [(i, score[i], ref[1][i]) for i in range(B) if score[i] != ref[1][i]]
        print(f"run {run}: streams with another history {dh[:8]} ({len(dh)}), another final score {ds[:4]} ({len(ds)})", flush=True)
print("done")
