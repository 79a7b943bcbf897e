"""Interference probe (lasr_debug_fe_race): the streaming log-mel kernel back to back beside one decode-stream kernel at a time.
usage: fe_race.py <model> <beam> <streams> <iters> [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libreasr_amd import synth
from libreasr_amd.engine import Engine
name, W, B, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dtype = sys.argv[5] if len(sys.argv) > 5 else "bf16"
cfg = synth.model_cfg(name); sd = synth.synth_state_dict(cfg, seed=0)
pcm = np.stack([synth.synth_pcm(1, 8 * 1280, seed=1234 + s)[0] for s in range(B)])
eng = Engine(sd, cfg, max_streams=B, dtype=dtype, beam=W)
slots = [eng.open() for _ in range(B)]
for k in range(8):
    eng.push(slots, pcm[:, k * 1280:(k + 1) * 1280])
    if eng.step(slots): eng.fetch_many(slots, 8192 if W > 1 else 64)
print("engine's fe_lds_pad:", eng.config("fe_lds_pad"), flush=True)
for pad in (0, -1):
    for agg, nm in ((0, "nothing"), (1, "vocabulary GEMM"), (2, "predictor pass"), (3, "joint half"), (1, "vocabulary GEMM")):
        for per in (1, 4):
            bl, br = eng.debug_fe_race(iters, agg, per, pad)
            print(f"lds_pad {'none' if pad == 0 else 'engine'}: beside {nm:16s} x{per}: {bl} of {iters} launches differ ({br} rows)", flush=True)
eng.close()
