"""Phase stamps of k_select's workgroup 0 in the pipelined job (LASR_DBG_TIMING=1: plain launches instead of graph replays)."""
import os, sys
os.environ["LASR_DBG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C, torch
from libreasr_amd import synth
from libreasr_amd.engine import Engine
cfg = synth.model_cfg("cfg2"); sd = synth.synth_state_dict(cfg)
eng = Engine(sd, cfg, max_streams=64)
slots = [eng.open() for _ in range(64)]
n = 120
pcm = torch.as_tensor(np.stack([synth.synth_pcm(1, n * 1280, seed=1234 + s)[0] for s in range(64)]).reshape(64, n, 1280).transpose(1, 0, 2).copy()).cuda()
buf = np.zeros(5 * 4096 * 16, dtype=np.uint64)
rows = []
for k in range(n):
    eng.push_submit(slots, pcm[k])
    if eng.pending() >= 12:
        eng.wait(); eng.fetch_many(slots, 64)
        if k > 60:
            eng.lib.lasr_debug_timing(eng.ctx, buf.ctypes.data_as(C.c_void_p))
            d = buf.reshape(5, 4096, 16)[4, 4095, :6].astype(np.int64)
            rows.append(d - d[0])
while eng.pending():
    eng.wait()
a = np.array(rows)
print("k_select workgroup 0, 10 ns ticks since entry: [entry, state loaded, statistics done, decisions done, state stored, published]")
print("median", np.median(a, 0)); print("p90   ", np.percentile(a, 90, 0))
