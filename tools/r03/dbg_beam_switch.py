import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libreasr_amd import synth
from libreasr_amd.engine import Engine
from oracle import rnnt_oracle as O
name = "tiny"
for W in (1, 2, 4):
    cfg = synth.model_cfg(name); sd = synth.synth_state_dict(cfg, seed=0)
    eng = Engine(sd, cfg, max_streams=8, beam=W)
    m = O.OracleTransducer(sd, cfg)
    pcm = synth.synth_pcm(2, 16000 * 3, seed=31)
    chunks = synth.stream_chunks(pcm[0], 1280, lead=1, tail=6) + synth.stream_chunks(pcm[1], 1280, lead=0, tail=2)[:8]
    for n_pipe in (0, 20, 21, 45):
        slot = eng.open()
        fe = O.StreamFrontend(); dec = O.StreamBeamDecoder(m, W) if W > 1 else m.stream_decoder()
        ref = []; got = []
        acc = []
        for k, ch in enumerate(chunks):
            o = fe.push(ch)
            if o is not None:
                if W > 1: ref.append(list(dec.step(o)[0]))
                else: dec.step(o); ref.append(list(dec.y))
            if k < n_pipe:
                eng.push_submit([slot], ch[None])
                if eng.pending() >= 3 and eng.wait():
                    t = eng.fetch(slot)[0]; acc = t if W > 1 else acc + t; got.append(list(acc))
            else:
                while eng.pending():
                    if eng.wait():
                        t = eng.fetch(slot)[0]; acc = t if W > 1 else acc + t; got.append(list(acc))
                eng.push([slot], ch[None])
                if eng.step([slot]):
                    t = eng.fetch(slot)[0]; acc = (t if t else acc) if W > 1 else acc + t; got.append(list(acc))
        while eng.pending():
            if eng.wait():
                t = eng.fetch(slot)[0]; acc = t if W > 1 else acc + t; got.append(list(acc))
        bad = next((j for j in range(min(len(ref), len(got))) if ref[j] != got[j]), None)
        print(f"W={W} n_pipe={n_pipe}: steps {len(ref)}/{len(got)} first mismatch {bad}", "" if bad is None else (ref[bad], got[bad]))
        eng.close_slot(slot)
    eng.close()
