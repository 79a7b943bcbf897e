"""Repro driver: native front under irregular arrivals (tests/test_gpu_round5.py::test_native_front_irregular_arrivals...), variants."""
import os, random, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from libreasr_amd import synth
from libreasr_amd.engine import Engine
from libreasr_amd.front import RES_EOF, RES_RESET, NativeFront
from oracle import rnnt_oracle as O

cfg = synth.model_cfg("tiny"); sd = synth.synth_state_dict(cfg, seed=0)
m = O.OracleTransducer(sd, cfg)
eng = Engine(sd, cfg, max_streams=16)

def scenario(seed):
    rng = random.Random(seed)
    B = 12
    specs = [(seed * 100 + i, rng.choice([3.0, 5.5, 7.0, [("speech", 2.0), ("silence", 5.0), ("speech", 1.5)]])) for i in range(B)]
    pcm = [synth.servicer_pcm(s_, sp) for s_, sp in specs]
    chunks = [synth.stream_chunks(p, 1280, lead=1, tail=10) for p in pcm]
    delays = [rng.uniform(0.0, 0.004) for _ in range(B)]
    return specs, chunks, delays

def oracle(chunks, rule):
    want = []
    for ch in chunks:
        fe, dec = O.StreamFrontend(), m.stream_decoder()
        y, steps = [], 0
        per = []
        for c in ch:
            o = fe.push(c)
            if o is None: continue
            ys = dec.step(o); steps += 1; y += ys; per.append(list(ys))
            if rule and not ys and O.should_reset(steps):
                dec.reset(); steps = 0
        want.append((y, per))
    return want

def run(seed, chunks, delays, depth, rule, single_chunk=False, sleepy=True):
    B = len(chunks)
    front = NativeFront(eng, depth=depth, reset_steps=rule)
    got = [[] for _ in range(B)]; per = [[] for _ in range(B)]
    def one(i):
        r = random.Random(seed * 1000 + i)
        if sleepy: time.sleep(delays[i])
        sid = front.open(); k = 0
        while k < len(chunks[i]):
            n = 1 if single_chunk else min(r.choice([1, 1, 2, 3]), len(chunks[i]) - k)
            front.push(sid, np.concatenate(chunks[i][k:k + n])); k += n
            if sleepy and r.random() < 0.3: time.sleep(r.uniform(0.0, 0.0003))
        front.eof(sid)
        while True:
            toks, flags = front.next(sid)
            if flags & RES_EOF: break
            got[i] += toks; per[i].append(toks)
        front.close(sid)
    ths = [threading.Thread(target=one, args=(i,)) for i in range(B)]
    [t.start() for t in ths]; [t.join(timeout=120) for t in ths]
    st = front.stats(); front.destroy()
    return got, per, st

for seed in (2, 1, 3, 4, 5):
    specs, chunks, delays = scenario(seed)
    w_rule = oracle(chunks, True); w_norule = oracle(chunks, False)
    for name, depth, rule, single, sleepy in (("d8 early", 8, 25, False, True), ("d8 noearly", 8, -25, False, True), ("d8 norule", 8, 0, False, True),
                                              ("d1 rule", 1, 25, False, True), ("d8 early single", 8, 25, True, True), ("d8 early nosleep", 8, 25, False, False)):
        for rep in range(3):
            got, per, st = run(seed, chunks, delays, depth, rule, single, sleepy)
            want = w_rule if rule else w_norule
            bad = [i for i in range(len(chunks)) if got[i] != want[i][0]]
            msg = ""
            for i in bad[:1]:
                a, b = per[i], want[i][1]
                k = next((j for j in range(min(len(a), len(b))) if a[j] != b[j]), -1)
                msg = f" stream {i} {specs[i][1]} steps {len(a)} vs {len(b)} first differing step {k}: got {a[k] if k >= 0 else None} want {b[k] if k >= 0 else None}"
            print(f"seed {seed} {name} rep {rep}: {'ok' if not bad else 'BAD ' + str(bad)}{msg} {st}", flush=True)
eng.close()
