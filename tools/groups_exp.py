"""Experiment: G independent engine contexts (B/G streams each) on one GPU, one host thread + HIP
stream per context, so the latency-bound kernel chains of different groups overlap on the chip."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from libreasr_amd import synth
from libreasr_amd.engine import Engine
B = 64; G = int(sys.argv[1]) if len(sys.argv) > 1 else 2; K = int(sys.argv[2]) if len(sys.argv) > 2 else 100; W = 10
cfg = synth.model_cfg("cfg2"); sd = synth.synth_state_dict(cfg)
n = K + W + 2
pcm_host = np.stack([synth.synth_pcm(1, n * 1280, seed=1234 + s)[0] for s in range(B)])
per = B // G
engines, pcms, streams = [], [], []
for g in range(G):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        e = Engine(sd, cfg, max_streams=per)
        slots = [e.open() for _ in range(per)]
        p = torch.as_tensor(pcm_host[g*per:(g+1)*per].reshape(per, n, 1280).transpose(1, 0, 2).copy()).cuda()
    engines.append((e, slots)); pcms.append(p); streams.append(st)
torch.cuda.synchronize()
bar = threading.Barrier(G + 1)
lat = [[] for _ in range(G)]
def worker(g):
    e, slots = engines[g]
    with torch.cuda.stream(streams[g]):
        for k in range(W):
            e.push(slots, pcms[g][k]); e.step(slots); [e.fetch(s, 256) for s in slots] if k % 2 else None
        bar.wait()
        for k in range(W, W + K):
            t = time.perf_counter()
            e.push(slots, pcms[g][k]); ran = e.step(slots)
            if ran: [e.fetch(s, 256) for s in slots]
            if ran: lat[g].append(time.perf_counter() - t)
        bar.wait()
ths = [threading.Thread(target=worker, args=(g,)) for g in range(G)]
[t.start() for t in ths]
bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
[t.join() for t in ths]
al = np.concatenate([np.array(l) for l in lat])
print(f"groups={G} streams/group={per}: {K*B*0.08/dt:.0f} audio-s/s  step {1e3*dt/K:.3f} ms  p50 model-chunk latency {1e3*np.median(al):.3f} ms  p95 {1e3*np.percentile(al,95):.3f} ms")
