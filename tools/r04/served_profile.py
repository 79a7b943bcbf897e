"""Where the scheduler thread's time goes in a faster-than-real-time replay with the servicer's reset rule on every stream
(tests/test_gpu_server.py (c) / (d)): cProfile of Scheduler.run + tick statistics.  argv: depth [stagger 0|1] [held_depth]"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft  # noqa: E402

graft.build()
from libreasr_amd import server as srv, synth  # noqa: E402
from libreasr_amd.lib.inference import load_stuff  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 12
stagger = int(sys.argv[2]) if len(sys.argv) > 2 else 0
held = int(sys.argv[3]) if len(sys.argv) > 3 else None
prof_on = os.environ.get("PROF", "1") != "0"
rule = os.environ.get("RULE", "1") != "0"          # RULE=0: the trunk form without the servicer's reset rule
conf, language, model, _, _ = load_stuff("en", config_path="/nonexistent.yaml", synthetic="cfg2", max_streams=64)
eng = model.engine
B, n2 = 64, 256
pcm2 = np.stack([synth.synth_pcm(1, n2 * 1280, seed=4321 + s)[0] for s in range(B)])
chunks2 = np.ascontiguousarray(pcm2.reshape(B, n2, 1280).transpose(1, 0, 2))


class Sched(srv.Scheduler):
    def run(self):
        if not prof_on:
            return super().run()
        self.prof = cProfile.Profile()
        self.prof.enable()
        try:
            super().run()
        finally:
            self.prof.disable()


reps = int(os.environ.get("REPS", "2"))
rates = []
for rep in range(reps):
    kw = {} if held is None else {"held_depth": held}
    sc = Sched(eng, depth=depth, **kw)
    sc.start()
    sts = [sc.open(text_of=language.denumericalize if rule else None) for _ in range(B)]
    if stagger:
        for i, st in enumerate(sts):
            sc.stp[st.slot] = (i * sc.reset_steps) // B
    t0 = time.perf_counter()
    for k in range(n2):
        sc.push_batch(sts, chunks2[k])
    seen, items = 0, 0
    while seen < B * ((n2 - 2) // 2):
        item = sc.batch_outq.get(timeout=120)
        assert not isinstance(item, Exception), item
        seen += len(item[0])
        items += 1
    dt = time.perf_counter() - t0
    for st in sts:
        sc.close(st)
    sc.shutdown()
    sc.join(timeout=30)
    rows = np.array(sc.step_rows)
    rates.append(B * n2 * 0.08 / dt)
    print(f"rep {rep}: rule {int(rule)} depth {depth} stagger {stagger} held_depth {sc.held_depth}: {B * n2 * 0.08 / dt:.0f} audio-s/s, {len(rows)} model steps of "
          f"{rows.mean():.1f} rows (p10 {np.percentile(rows, 10):.0f}), {dt / len(rows) * 1e6:.0f} us per model step, {items} result items")
if reps > 2:
    print(f"median of reps 1..{reps - 1}: {np.median(rates[1:]):.0f} audio-s/s (min {min(rates[1:]):.0f}, max {max(rates[1:]):.0f})")
if prof_on:
    s = io.StringIO()
    pstats.Stats(sc.prof, stream=s).sort_stats("tottime").print_stats(22)
    print(s.getvalue()[:6000])
