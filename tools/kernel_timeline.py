"""In-kernel phase timelines (LASR_DBG_TIMING): where does a GEMM workgroup spend its cycles?"""
import os, sys
os.environ["LASR_DBG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from libreasr_amd import synth
from libreasr_amd.engine import Engine
cfg = synth.model_cfg("cfg2"); sd = synth.synth_state_dict(cfg); B = 64
eng = Engine(sd, cfg, max_streams=B)
n = 24
pcm = torch.as_tensor(np.stack([synth.synth_pcm(1, n * 1280, seed=1234 + s)[0] for s in range(B)]).reshape(B, n, 1280).transpose(1, 0, 2).copy()).cuda()
slots = [eng.open() for _ in range(B)]
names = ["enc cell", "pred layer0", "pred layer1", "ppj", "logits"]
grids = [256, 256, 256, 64, 512]
def dump(tag):
    buf = np.zeros(5 * 4096 * 16, dtype=np.uint64)
    eng._chk(eng.lib.lasr_debug_timing(eng.ctx, buf.ctypes.data_as(C.c_void_p)))
    buf = buf.reshape(5, 4096, 16).astype(np.float64)
    print(f"--- {tag}")
    for k in range(5):
        b = buf[k, :grids[k]]
        ok = b[:, 0] > 0
        if not ok.any(): continue
        b = b[ok]
        wall0 = b[:, 5].min(); wall1 = b[:, 6].max()
        blk_wall = (b[:, 6] - b[:, 5]) / 100.0              # us per workgroup (wall clock = 100 MHz)
        blk_cyc = b[:, 4] - b[:, 0]
        clk = np.median(blk_cyc / np.maximum(blk_wall, 1e-3))  # MHz-ish: s_memtime ticks per us
        ph = np.stack([b[:, 1] - b[:, 0], b[:, 2] - b[:, 1], b[:, 3] - b[:, 2], b[:, 4] - b[:, 3]], 1)
        if k == 0:
            wv = b[:, 8:16] - b[:, 1:2]            # per-wave K-loop end relative to setup end
            print("   per-wave K-loop cycles (mean over workgroups):", " ".join(f"{x:7.0f}" for x in wv.mean(0)),
                  "| slowest-wave histogram:", np.bincount(wv.argmax(1), minlength=8))
        print(f"{names[k]:12s} blocks {len(b):4d} span {(wall1-wall0)/100:6.2f} us  wg avg {blk_wall.mean():5.2f} us (ticks/us {clk:5.0f}) start-skew {(b[:,5]-wall0).mean()/100:5.2f}/{(b[:,5]-wall0).max()/100:5.2f} us | "
              f"setup {ph[:,0].mean():6.0f} | K-loop {ph[:,1].mean():7.0f} (max {ph[:,1].max():7.0f}) | reduce {ph[:,2].mean():5.0f} | epilogue {ph[:,3].mean():6.0f}")
for k in range(n):
    eng.push(slots, pcm[k]); ran = eng.step(slots)
    if ran: eng.fetch_many(slots, 64)
    if k in (9, 15, 21): dump(f"after chunk {k}")
us = eng.bench_cell(1, 50); dump(f"bench_cell layer1 {us:.2f} us")
