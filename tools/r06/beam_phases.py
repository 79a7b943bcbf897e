"""Phase stamps of k_beam_select_rw (workgroup 0) for beam 4 (cfg2 bf16) and beam 8 (cfg5 bf16, 128 streams): entry -> state loaded [1]
-> statistics [2] -> per-row top-W done + barrier [9] -> merge [3] -> bookkeeping [4]; 10 ns ticks."""
import os, sys
os.environ["LASR_DBG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, ctypes as C
from libreasr_amd import synth
from libreasr_amd.engine import Engine
for name, B, W in (("cfg2", 64, 4), ("cfg5", 128, 8)):
    cfg = synth.model_cfg(name); sd = synth.synth_state_dict(cfg)
    eng = Engine(sd, cfg, max_streams=B, beam=W, dtype="bf16")
    slots = [eng.open() for _ in range(B)]
    pcm = np.stack([synth.synth_pcm(1, 16 * 1280, seed=1234 + s)[0] for s in range(B)]).reshape(B, 16, 1280)
    rows = []
    for k in range(16):
        eng.push(slots, pcm[:, k])
        if eng.step(slots):
            buf = np.zeros(5 * 4096 * 16, dtype=np.uint64)
            eng.lib.lasr_debug_timing(eng.ctx, buf.ctypes.data_as(C.c_void_p))
            d = buf.reshape(5, 4096, 16)[4, 0, :10].astype(np.int64)
            rows.append([(d[i] - d[0]) / 100.0 for i in (1, 2, 9, 3, 4)])
    r = np.array(rows)
    print(name, "beam", W, "us since entry: state", r[:, 0].mean().round(2), "stats", r[:, 1].mean().round(2), "row top-W + barrier", r[:, 2].mean().round(2),
          "merge", r[:, 3].mean().round(2), "bookkeeping", r[:, 4].mean().round(2), flush=True)
    eng.close()
