import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libreasr_amd import synth
from libreasr_amd.engine import Engine
name, W, B, n_chunks, N = "cfg5", 8, 128, 48, 6
cfg = synth.model_cfg(name); sd = synth.synth_state_dict(cfg, seed=0)
pcm = np.stack([synth.synth_pcm(1, n_chunks * 1280, seed=1234 + s)[0] for s in range(B)])
DEV = os.environ.get('DEV') == '1'; SYNCP = os.environ.get('SYNCP') == '1'
pcmd = [torch.as_tensor(np.ascontiguousarray(pcm[:, k * 1280:(k + 1) * 1280])).cuda() for k in range(n_chunks)] if DEV else None
def run(mode):
    eng = Engine(sd, cfg, max_streams=B, dtype="bf16", beam=W)
    slots = [eng.open() for _ in range(B)]
    steps = []
    def take():
        steps.append([(tuple(t), -nl) for t, nl, _ in (eng.fetch(s) for s in slots)])
    for k in range(n_chunks):
        if mode == "sync":
            eng.push(slots, pcm[:, k * 1280:(k + 1) * 1280])
            if eng.step(slots): take()
            continue
        eng.push_submit(slots, pcmd[k] if DEV else pcm[:, k * 1280:(k + 1) * 1280])
        if SYNCP: torch.cuda.current_stream().synchronize()
        while eng.pending() >= 6:
            if eng.wait(): take()
    while eng.pending():
        if eng.wait(): take()
    eng.close()
    return steps
ref = run("sync")
for r in range(N):
    cur = run("pipe")
    out = []
    for i in range(B):
        for j in range(len(ref)):
            a, b = ref[j][i], cur[j][i]
            if a != b:
                out.append((i, j, "tokens" if a[0] != b[0] else "score", a[1], b[1]))
                break
    print(f"pipelined run {r} vs sync: first differences (stream, model step, what, sync score, pipelined score): {out[:6]} ... {len(out)} streams", flush=True)
