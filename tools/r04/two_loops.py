"""Experiment: N engine contexts in ONE process, each with 64 / N streams, driven by one host thread (every context has its own decode
stream and pump thread): do several independent decode loops overlap on the GPU where one loop is latency-bound (beam configs)?"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import __graft_entry__ as g
g.build()
from libreasr_amd import synth
from libreasr_amd.engine import Engine

model, dtype, beam, total, n_ctx, depth, n_chunks = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), 160
cfg = synth.model_cfg(model); sd = synth.synth_state_dict(cfg, seed=0)
B = total // n_ctx
engs = [Engine(sd, cfg, max_streams=B, dtype=dtype, beam=beam) for _ in range(n_ctx)]
pcm = np.stack([synth.synth_pcm(1, n_chunks * 1280, seed=1234 + s)[0] for s in range(total)])
dev = torch.as_tensor(pcm.reshape(total, n_chunks, 1280).transpose(1, 0, 2).copy()).cuda()
slots = [[e.open() for _ in range(B)] for e in engs]
cap = 64 if beam == 1 else 8192
ntok = 0
def run(k0, k1):
    global ntok
    for k in range(k0, k1):
        for i, e in enumerate(engs):
            e.push_submit(slots[i], dev[k, i * B:(i + 1) * B])
        for i, e in enumerate(engs):
            while e.pending() >= depth:
                if e.wait():
                    ntok += sum(len(t) for t in e.fetch_many(slots[i], cap))
    for i, e in enumerate(engs):
        while e.pending():
            if e.wait():
                ntok += sum(len(t) for t in e.fetch_many(slots[i], cap))
run(0, 32)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(32, n_chunks)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{model} {dtype} beam {beam}: {n_ctx} context(s) x {B} streams, depth {depth}: {total * (n_chunks - 32) * 0.08 / dt:9.1f} audio-s/s")
