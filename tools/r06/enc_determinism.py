import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libreasr_amd import synth
from libreasr_amd.engine import Engine
name, W, B, n_chunks, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), 40, 6
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
cfg = synth.model_cfg(name); sd = synth.synth_state_dict(cfg, seed=0)
L = cfg["enc_layers"]
pcm = np.stack([synth.synth_pcm(1, n_chunks * 1280, seed=1234 + s)[0] for s in range(B)])
def run(mode):
    eng = Engine(sd, cfg, max_streams=B, dtype=dtype, beam=W)
    slots = [eng.open() for _ in range(B)]
    for k in range(n_chunks):
        if mode == "sync":
            eng.push(slots, pcm[:, k * 1280:(k + 1) * 1280])
            if eng.step(slots): eng.fetch_many(slots, 8192 if W > 1 else 64)
            continue
        eng.push_submit(slots, pcm[:, k * 1280:(k + 1) * 1280])
        while eng.pending() >= 6:
            if eng.wait(): eng.fetch_many(slots, 8192 if W > 1 else 64)
    while eng.pending():
        if eng.wait(): eng.fetch_many(slots, 8192 if W > 1 else 64)
    st = [eng.debug_read("enc_h", l) for l in range(L)] + [eng.debug_read("enc_c", l) for l in range(L)] + [eng.debug_read("pred_h", 0), eng.debug_read("pp")]
    eng.close()
    return st
ref = run("sync")
for r in range(N):
    cur = run("pipe")
    d = [float(np.abs(a[:B] - b[:B]).max()) for a, b in zip(ref, cur)]
    rows = sorted(set(int(i) for a, b in zip(ref[:2 * L], cur[:2 * L]) for i in np.nonzero(np.abs(a[:B] - b[:B]).max(1))[0]))
    print(f"pipelined run {r} vs sync: max |diff| enc_h per layer {[round(x, 6) for x in d[:L]]} enc_c {[round(x, 6) for x in d[L:2 * L]]} pred_h0 {d[-2]:.4g} pp {d[-1]:.4g}; encoder rows that differ {rows[:12]}", flush=True)
