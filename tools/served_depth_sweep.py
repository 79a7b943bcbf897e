"""Served rate of the trunk form with the servicer's reset rule on every stream, by the number of model steps kept in flight
while a stream is held at the reset threshold (Scheduler(held_depth=...)).  64 streams of configs[1], 128 chunks each."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import __graft_entry__ as graft

graft.build()
from libreasr_amd import server as srv, synth
from libreasr_amd.lib.inference import load_stuff

conf, language, model, _, _ = load_stuff("en", config_path="/nonexistent.yaml", synthetic="cfg2", max_streams=64)
eng = model.engine
B = 64
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128          # chunks per stream
pcm = np.stack([synth.synth_pcm(1, n * 1280, seed=4321 + s)[0] for s in range(B)])
chunks = np.ascontiguousarray(pcm.reshape(B, n, 1280).transpose(1, 0, 2))
out = {}
ref = None
for hd in (12, 6, 4, 3, 2, 1, 4):
    sc = srv.Scheduler(eng, depth=12, held_depth=hd)
    sc.start()
    sts = [sc.open(text_of=language.denumericalize) for _ in range(B)]
    got = {st.slot: [] for st in sts}
    t0 = time.perf_counter()
    for k in range(n):
        sc.push_batch(sts, chunks[k])
    seen = 0
    while seen < B * ((n - 2) // 2):
        item = sc.batch_outq.get(timeout=120)
        assert not isinstance(item, Exception), item
        seen += len(item[0])
        for st, t in zip(*item):
            got[st.slot].append(t)
    dt = time.perf_counter() - t0
    toks = [got[st.slot] for st in sts]
    if ref is None:
        ref = toks
    assert toks == ref, "tokens depend on held_depth"
    for st in sts:
        sc.close(st)
    sc.shutdown()
    sc.join(timeout=30)
    out.setdefault(str(hd), []).append({"audio_sec_per_sec": round(B * n * 0.08 / dt, 1), "rows_per_step": round(float(np.mean(sc.step_rows)), 1)})
    print(hd, out[str(hd)][-1], flush=True)
print(json.dumps(out))
