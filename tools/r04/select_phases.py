"""Phase stamps of k_beam_select's workgroup 0 (LASR_DBG_TIMING; wall clock, 10 ns ticks): entry, state loaded, statistics done,
wave-level top-W done [9], merge done [3], bookkeeping done [4].  argv: model beam streams"""
import os, sys
os.environ["LASR_DBG_TIMING"] = "1"
sys.path.insert(0, ".")
import numpy as np, ctypes as C
from libreasr_amd import synth
from libreasr_amd.engine import Engine
model, W, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = synth.model_cfg(model); sd = synth.synth_state_dict(cfg)
eng = Engine(sd, cfg, max_streams=B, beam=W, dtype="bf16")
slots = [eng.open() for _ in range(B)]
pcm = np.stack([synth.synth_pcm(1, 12 * 1280, seed=1234 + s)[0] for s in range(B)]).reshape(B, 12, 1280)
acc = []
for k in range(12):
    eng.push(slots, pcm[:, k])
    if eng.step(slots):
        buf = np.zeros(5 * 4096 * 16, dtype=np.uint64)
        eng.lib.lasr_debug_timing(eng.ctx, buf.ctypes.data_as(C.c_void_p))
        d = buf.reshape(5, 4096, 16)[4, 0, :10].astype(np.int64)
        if d[4] > d[0] > 0:
            acc.append([d[1] - d[0], d[2] - d[1], d[9] - d[2], d[3] - d[9], d[4] - d[3], d[4] - d[0]])
a = np.array(acc, dtype=np.float64) / 100.0
print(f"{model} W={W} {B} streams: state {a[:,0].mean():.2f} | statistics {a[:,1].mean():.2f} | wave top-W {a[:,2].mean():.2f} | merge {a[:,3].mean():.2f} | bookkeeping {a[:,4].mean():.2f} | total {a[:,5].mean():.2f} us  (n={len(a)})")
