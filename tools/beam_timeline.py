import os, sys
os.environ["LASR_DBG_TIMING"] = "1"
sys.path.insert(0, ".")
import numpy as np, ctypes as C
from libreasr_amd import synth
from libreasr_amd.engine import Engine
cfg = synth.model_cfg("cfg2"); sd = synth.synth_state_dict(cfg)
eng = Engine(sd, cfg, max_streams=64, beam=4, dtype="bf16")
slots = [eng.open() for _ in range(64)]
pcm = np.stack([synth.synth_pcm(1, 12 * 1280, seed=1234 + s)[0] for s in range(64)]).reshape(64, 12, 1280)
for k in range(12):
    eng.push(slots, pcm[:, k]); eng.step(slots)
buf = np.zeros(5 * 4096 * 16, dtype=np.uint64)
eng.lib.lasr_debug_timing(eng.ctx, buf.ctypes.data_as(C.c_void_p))
d = buf.reshape(5, 4096, 16)[4, 0, :10].astype(np.int64)
print("ticks(10ns):", d - d[0])
