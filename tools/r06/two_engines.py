"""Would two independent decode loops per GPU pay?  Two engines in ONE process (own main + decode streams, own pump thread), half
the streams each, driven by two host threads, against one engine with all the streams.
usage: two_engines.py <model> <beam> <streams> <chunks> [dtype] [depth]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from libreasr_amd import synth
from libreasr_amd.engine import Engine
name, W, B, n_chunks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dtype = sys.argv[5] if len(sys.argv) > 5 else "bf16"
depth = int(sys.argv[6]) if len(sys.argv) > 6 else 6
cfg = synth.model_cfg(name); sd = synth.synth_state_dict(cfg, seed=0)
pcm = np.stack([synth.synth_pcm(1, 64 * 1280, seed=1234 + s)[0] for s in range(B)])
def drive(eng, slots, rows, n, barrier, out):
    dev = [torch.as_tensor(np.ascontiguousarray(pcm[rows, (k % 64) * 1280:(k % 64 + 1) * 1280])).cuda() for k in range(64)]
    cap = 8192 if W > 1 else 64
    def pump(nn):
        for k in range(nn):
            eng.push_submit(slots, dev[k % 64])
            while eng.pending() >= depth:
                if eng.wait(): eng.fetch_many(slots, cap)
        while eng.pending():
            if eng.wait(): eng.fetch_many(slots, cap)
    pump(16)
    barrier.wait()
    t0 = time.perf_counter()
    pump(n)
    torch.cuda.synchronize()
    out.append(time.perf_counter() - t0)
def run(n_eng):
    per = B // n_eng
    engs = [Engine(sd, cfg, max_streams=per, dtype=dtype, beam=W) for _ in range(n_eng)]
    slots = [[e.open() for _ in range(per)] for e in engs]
    barrier = threading.Barrier(n_eng); out = []
    th = [threading.Thread(target=drive, args=(engs[i], slots[i], list(range(i * per, (i + 1) * per)), n_chunks, barrier, out)) for i in range(n_eng)]
    for t in th: t.start()
    for t in th: t.join()
    for e in engs: e.close()
    return B * n_chunks * 0.08 / max(out)
for n_eng in (1, 2, 1, 2):
    print(f"{name} beam {W} {dtype}: {n_eng} engine(s) x {B // n_eng} streams: {run(n_eng):.0f} audio-s/s", flush=True)
