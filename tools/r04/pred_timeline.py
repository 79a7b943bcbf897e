"""Phase stamps (LASR_DBG_TIMING + LASR_DBG_BEAM) of the beam round's GEMMs on the configs[4] shape: where do the wide predictor
cells (EpiLSTMw, 1024 hypothesis rows) spend their time -- prologue (row compaction), K loop, LDS reduction, epilogue?"""
import os, sys
os.environ["LASR_DBG_TIMING"] = "1"
os.environ["LASR_DBG_BEAM"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
from libreasr_amd import synth
from libreasr_amd.engine import Engine
cfg = synth.model_cfg("cfg5"); sd = synth.synth_state_dict(cfg); B = 128
eng = Engine(sd, cfg, max_streams=B, dtype="bf16", beam=8)
n = 16
pcm = torch.as_tensor(np.stack([synth.synth_pcm(1, n * 1280, seed=1234 + s)[0] for s in range(B)]).reshape(B, n, 1280).transpose(1, 0, 2).copy()).cuda()
slots = [eng.open() for _ in range(B)]
names = ["enc cell", "pred layer0", "pred layer1", "ppj", "logits"]
def dump(tag):
    buf = np.zeros(5 * 4096 * 16, dtype=np.uint64)
    eng._chk(eng.lib.lasr_debug_timing(eng.ctx, buf.ctypes.data_as(C.c_void_p)))
    buf = buf.reshape(5, 4096, 16).astype(np.float64)
    print(f"--- {tag}")
    for k in range(5):
        b = buf[k]
        ok = b[:, 0] > 0
        if not ok.any(): continue
        b = b[ok]
        wall0 = b[:, 5].min(); wall1 = b[:, 6].max()
        blk_wall = (b[:, 6] - b[:, 5]) / 100.0
        blk_cyc = b[:, 4] - b[:, 0]
        clk = np.median(blk_cyc / np.maximum(blk_wall, 1e-3))
        ph = np.stack([b[:, 1] - b[:, 0], b[:, 2] - b[:, 1], b[:, 3] - b[:, 2], b[:, 4] - b[:, 3]], 1) / max(clk, 1.0)
        busy = ph[:, 1] > 0.5                      # workgroups that ran a K loop
        print(f"{names[k]:12s} blocks {len(b):4d} (K loop in {int(busy.sum())}) span {(wall1-wall0)/100:6.2f} us  wg avg {blk_wall.mean():5.2f} us  start-skew avg {(b[:,5]-wall0).mean()/100:5.2f} max {(b[:,5]-wall0).max()/100:5.2f} us | "
              f"setup {ph[busy,0].mean() if busy.any() else 0:5.2f} | K-loop {ph[busy,1].mean() if busy.any() else 0:5.2f} (max {ph[:,1].max():5.2f}) | reduce {ph[busy,2].mean() if busy.any() else 0:5.2f} | epilogue {ph[busy,3].mean() if busy.any() else 0:5.2f} us; idle wgs: epilogue {ph[~busy,3].mean() if (~busy).any() else 0:5.2f} us")
for k in range(n):
    eng.push(slots, pcm[k]); ran = eng.step(slots)
    if ran: eng.fetch_many(slots, 8192)
    if k in (7, 11, 15): dump(f"after chunk {k}")
