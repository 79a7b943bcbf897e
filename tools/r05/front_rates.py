"""Native front (lasr_bench_front: one native producer thread per stream): served rate of 64 streams of configs[1] with and without
the servicer's reset rule, by chunks per stream (the pipeline's fill / drain amortises over longer streams)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from libreasr_amd import synth                      # noqa: E402
from libreasr_amd.engine import Engine              # noqa: E402
from libreasr_amd.front import bench_native_producers   # noqa: E402

cfg = synth.model_cfg("cfg2")
sd = synth.synth_state_dict(cfg, seed=0)
B = 64
eng = Engine(sd, cfg, max_streams=B)
out = {}
base = np.stack([synth.synth_pcm(1, 256 * 1280, seed=1234 + s)[0] for s in range(B)])
for n_chunks in (256, 1024):
    pcm = np.concatenate([base] * (n_chunks // 256), axis=1)
    for rule in (0, 25):
        for depth in ((12,) if not rule else (12, 6)):
            best = None
            for rep in range(3):
                toks, sec, st = bench_native_producers(eng, pcm, depth=depth, reset_steps=rule, chunks_per_push=1, cap=8192)
                r = {"audio_sec_per_sec": round(B * n_chunks * 0.08 / sec, 1), "rows_per_model_step": round(st["rows"] / max(1, st["steps"]), 2),
                     "resets": st["resets"], "tokens": int(sum(len(t) for t in toks))}
                if best is None or r["audio_sec_per_sec"] > best["audio_sec_per_sec"]:
                    best = r
            out[f"chunks{n_chunks}_rule{rule}_depth{depth}"] = best
            print(f"chunks {n_chunks} rule {rule} depth {depth}: {best}", flush=True)
json.dump(out, open(os.path.join("gpurun_out", "front_rates.json"), "w"), indent=1)
eng.close()
