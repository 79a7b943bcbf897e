"""Beam search (BASELINE configs[2]/[4]).  The reference has no beam search, so parity is
UNPINNED against it; the contract is the oracle's spec (oracle/rnnt_oracle.py:_beam_frame):
  * beam = 1 is the greedy decode token-for-token (the greedy engine is pinned to the reference goldens;
    here the oracle's W = 1 beam is checked against the oracle's greedy, on CPU in tests/test_oracle.py)
  * for W in {2, 4, 8}: best hypothesis and its score equal the oracle's beam on the same inputs
  * the best score of W = 8 is not below the greedy score (not guaranteed per width: pruned search)."""
import numpy as np
import pytest
import torch

from libreasr_amd import synth
from oracle import rnnt_oracle as O

pytestmark = pytest.mark.gpu

_ENGINES = {}


def engine(name, beam, dtype="f32", max_streams=8):
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    key = (name, beam, dtype)
    if key not in _ENGINES:
        graft.build()
        cfg = synth.model_cfg(name)
        sd = synth.synth_state_dict(cfg, seed=0)
        _ENGINES[key] = (Engine(sd, cfg, max_streams=max_streams, beam=beam, dtype=dtype),
                         O.OracleTransducer(sd, cfg, operand=dtype), cfg)
    return _ENGINES[key]


# (("tiny_soft", 3): an odd width in the 4-slot kernel; ("cfg2", 8): V = 2048 fills every register slot of k_beam_select_rw<8> --
#  the cases round 4 checked against the spread kernel that round 6 deleted)
@pytest.mark.parametrize("name,W", [("tiny", 2), ("tiny", 4), ("tiny_lstm", 4), ("tiny", 8), ("cfg2", 4), ("tiny_soft", 3), ("cfg2", 8)])
def test_beam_offline_matches_oracle(name, W):
    eng, m, cfg = engine(name, W)
    n = 3 if name != "cfg2" else 2
    secs = 3 if name != "cfg2" else 2
    pcm = synth.synth_pcm(n, 16000 * secs, seed=1234)
    slots = [eng.open() for _ in range(n)]
    eng.transcribe_pcm(slots, [pcm[i] for i in range(n)])
    for i, s in enumerate(slots):
        toks, neg_logp, _ = eng.fetch(s)
        f = O.features_offline(pcm[i])
        y, score, _ = m.decode_beam(f, W)
        g = m.decode_greedy(f)
        print(f"{name} W={W} utt {i}: {len(toks)} tokens score {-neg_logp:.4f} (oracle {score:.4f}, greedy {-g[1]:.4f})")
        assert toks == y, (toks, y)
        assert abs(-neg_logp - score) < 2e-3 * max(1.0, abs(score))
        if W == 8:
            assert -neg_logp >= -g[1] - 1e-3
        eng.close_slot(s)


def test_beam_streaming_matches_oracle_and_ragged():
    W = 4
    eng, m, cfg = engine("tiny_lstm", W)
    n = 3
    pcm = synth.synth_pcm(n, 16000 * 3, seed=1234)
    chunks = [synth.stream_chunks(pcm[i], 1280, lead=1, tail=4) for i in range(n)]
    slots = [eng.open() for _ in range(n)]
    got = [[] for _ in range(n)]
    fes = [O.StreamFrontend() for _ in range(n)]
    decs = [O.StreamBeamDecoder(m, W) for _ in range(n)]
    start = [0, 1, 3]                               # streams join at different chunks: ragged steps
    for k in range(len(chunks[0]) + max(start)):
        act = [i for i in range(n) if 0 <= k - start[i] < len(chunks[i])]
        if not act:
            continue
        eng.push([slots[i] for i in act], np.stack([chunks[i][k - start[i]] for i in act]))
        ran = eng.step([slots[i] for i in act])
        for i in act:
            o = fes[i].push(chunks[i][k - start[i]])
            if o is not None:
                decs[i].step(o)
        if ran:
            for i in act:
                t = eng.fetch(slots[i])[0]
                if t:
                    got[i] = t
    for i in range(n):
        y, score = decs[i].best()
        print(f"stream {i}: {len(got[i])} tokens; oracle {len(y)}")
        assert got[i] == y, (i, got[i], y)
    # a predictor reset freezes the best hypothesis and restarts the beam
    eng.reset(slots[0], 1 | 2 | 4)
    decs[0] = O.StreamBeamDecoder(m, W)
    fes0 = fes[0]
    frozen = list(got[0])
    for ch in synth.stream_chunks(pcm[1], 1280, lead=0, tail=2):
        eng.push([slots[0]], ch[None])
        if eng.step([slots[0]]):
            t = eng.fetch(slots[0])[0]
            if t:
                got[0] = t
        o = fes0.push(ch)
        if o is not None:
            decs[0].step(o)
    assert got[0] == frozen + decs[0].best()[0]
    for s in slots:
        eng.close_slot(s)


def test_beam_bf16_cfg2_runs_and_tracks_emulation():
    W = 4
    eng, m, cfg = engine("cfg2", W, dtype="bf16")
    pcm = synth.synth_pcm(2, 16000 * 2, seed=1234)
    slots = [eng.open() for _ in range(2)]
    eng.transcribe_pcm(slots, [pcm[0], pcm[1]])
    for i, s in enumerate(slots):
        toks, neg_logp, _ = eng.fetch(s)
        y, score, _ = m.decode_beam(O.features_offline(pcm[i]), W)
        print(f"bf16 W={W} utt {i}: {len(toks)} tokens score {-neg_logp:.4f} (emulation {len(y)} tokens, {score:.4f})")
        assert len(toks) > 0
        assert abs(-neg_logp - score) < 0.05 * max(1.0, abs(score))     # bf16 contract: tolerance, not bit parity
        eng.close_slot(s)


@pytest.mark.parametrize("name,W", [("tiny", 2), ("tiny", 4), ("tiny_lstm", 4), ("tiny", 8)])
def test_beam_on_the_pipelined_protocol_equals_the_oracle_per_model_step(name, W):
    """Round 3: beam search on lasr_push_submit / lasr_step_wait -- the selection loop runs across chunk boundaries, every stream
    on its own frame cursor.  After EVERY model step the best hypothesis must be the oracle's (StreamBeamDecoder), for streams
    that join at different chunks, with several steps in flight; then a predictor reset freezes the best hypothesis."""
    eng, m, cfg = engine(name, W)
    n = 3
    pcm = synth.synth_pcm(n, 16000 * 3, seed=31)
    chunks = [synth.stream_chunks(pcm[i], 1280, lead=1, tail=6) for i in range(n)]
    slots = [eng.open() for _ in range(n)]
    fes = [O.StreamFrontend() for _ in range(n)]
    decs = [O.StreamBeamDecoder(m, W) for _ in range(n)]
    ref_hist = [[] for _ in range(n)]               # oracle: best hypothesis after every model step
    got_hist = [[] for _ in range(n)]
    order = []                                      # per submitted model step: the streams that ran
    start = [0, 1, 3]

    def collect():
        rows = order.pop(0)
        assert eng.wait() == len(rows)
        for i in rows:
            t, neg_logp, _ = eng.fetch(slots[i])
            got_hist[i].append((t, -neg_logp))

    for k in range(len(chunks[0]) + max(start)):
        act = [i for i in range(n) if 0 <= k - start[i] < len(chunks[i])]
        if not act:
            continue
        before = eng.pending()
        eng.push_submit([slots[i] for i in act], np.stack([chunks[i][k - start[i]] for i in act]))
        ran = []
        for i in act:
            o = fes[i].push(chunks[i][k - start[i]])
            if o is not None:
                y, sc = decs[i].step(o)
                ref_hist[i].append((list(y), sc))
                ran.append(i)
        assert (eng.pending() > before) == bool(ran)
        if ran:
            order.append(ran)
        if eng.pending() >= 4:
            collect()
    while eng.pending():
        collect()
    n_tok = 0
    for i in range(n):
        assert len(got_hist[i]) == len(ref_hist[i]) > 5
        for j, ((t, sc), (y, rsc)) in enumerate(zip(got_hist[i], ref_hist[i])):
            assert t == y, (i, j, t, y)
            assert abs(sc - rsc) < 1e-3 * max(1.0, abs(rsc)), (i, j, sc, rsc)
        n_tok += len(ref_hist[i][-1][0])
    assert n_tok > 3
    # the synchronous protocol continues a stream the pipelined one started (same device state, same host trees): the
    # hypothesis after every further step must be what an all-synchronous and an all-pipelined run of the same chunks give
    extra = synth.stream_chunks(pcm[1], 1280, lead=0, tail=2)[:6]
    mixed = []
    for ch in extra:
        eng.push([slots[0]], ch[None])
        if eng.step([slots[0]]):
            mixed.append(eng.fetch(slots[0])[0])
    assert len(mixed) >= 2
    whole = chunks[0] + extra
    for proto in ("sync", "pipelined"):
        sl = eng.open()
        hist = []
        for ch in whole:
            if proto == "sync":
                eng.push([sl], ch[None])
                if eng.step([sl]):
                    hist.append(eng.fetch(sl)[0])
            else:
                eng.push_submit([sl], ch[None])
                if eng.pending() >= 3 and eng.wait():
                    hist.append(eng.fetch(sl)[0])
        while eng.pending():
            if eng.wait():
                hist.append(eng.fetch(sl)[0])
        assert hist[-len(mixed):] == mixed, proto
        assert hist[:len(got_hist[0])] == [t for t, _ in got_hist[0]], proto
        eng.close_slot(sl)
    for sl in slots:
        eng.close_slot(sl)


def test_beam_rejects_bad_width():
    from libreasr_amd.engine import Engine
    cfg = synth.model_cfg("tiny")
    sd = synth.synth_state_dict(cfg, seed=0)
    with pytest.raises(ValueError):
        Engine(sd, cfg, max_streams=8, beam=9)
    with pytest.raises(ValueError):
        Engine(sd, cfg, max_streams=512, beam=4)      # 512 x 4 decoder rows > 1024


@pytest.mark.parametrize("name,lm_name,W", [("tiny", "tiny_lm", 2), ("tiny", "tiny_lm", 4), ("tiny_lstm", "tiny_lm_untied", 4)])
def test_beam_with_lm_fusion_matches_the_oracle_on_every_protocol(name, lm_name, W):
    """Round 3: LM shallow fusion inside the beam (spec: oracle _beam_frame with an LM -- a hypothesis offers its blank and its best
    non-blank extension; the emitted token is the fuser's re-pick, lm.py:59-79; W = 1 is the reference's greedy + LM, checked on the
    CPU in tests/test_oracle.py).  Offline, synchronous streaming and pipelined streaming against the oracle."""
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    graft.build()
    cfg = synth.model_cfg(name)
    sd = synth.synth_state_dict(cfg, seed=0)
    lm_sd = synth.synth_lm_state_dict(lm_name)
    eng = Engine(sd, cfg, max_streams=8, beam=W)
    eng.attach_lm(lm_sd, int8=False)
    try:
        m = O.OracleTransducer(sd, cfg)
        m.lm = O.OracleLM(lm_sd)
        m0 = O.OracleTransducer(sd, cfg)                       # the same beam without an LM: the fusion must matter somewhere
        n = 3
        pcm = synth.synth_pcm(n, 16000 * 3, seed=1234)
        slots = [eng.open() for _ in range(n)]
        eng.transcribe_pcm(slots, [pcm[i] for i in range(n)])
        differs = 0
        for i, sl in enumerate(slots):
            toks, neg_logp, _ = eng.fetch(sl)
            f = O.features_offline(pcm[i])
            y, score, _ = m.decode_beam(f, W)
            assert toks == y, (i, toks, y)
            assert abs(-neg_logp - score) < 2e-3 * max(1.0, abs(score))
            differs += y != m0.decode_beam(f, W)[0]
        for proto in ("sync", "pipelined"):
            for sl in slots:
                eng.reset(sl, 15)
            fes = [O.StreamFrontend() for _ in range(n)]
            decs = [O.StreamBeamDecoder(m, W) for _ in range(n)]
            plain = [O.StreamBeamDecoder(m0, W) for _ in range(n)]
            ref = [[] for _ in range(n)]
            got = [[] for _ in range(n)]
            chunks = [synth.stream_chunks(pcm[i], 1280, lead=1, tail=6) for i in range(n)]
            for k in range(len(chunks[0])):
                batch = np.stack([chunks[i][k] for i in range(n)])
                for i in range(n):
                    o = fes[i].push(chunks[i][k])
                    if o is not None:
                        ref[i].append(list(decs[i].step(o)[0]))
                        plain[i].step(o)
                if proto == "sync":
                    eng.push(slots, batch)
                    if eng.step(slots):
                        for i in range(n):
                            got[i].append(eng.fetch(slots[i])[0])
                else:
                    eng.push_submit(slots, batch)
                    if eng.pending() >= 4 and eng.wait():
                        for i in range(n):
                            got[i].append(eng.fetch(slots[i])[0])
            while eng.pending():
                if eng.wait():
                    for i in range(n):
                        got[i].append(eng.fetch(slots[i])[0])
            for i in range(n):
                assert got[i] == ref[i], (proto, i)
                differs += decs[i].best()[0] != plain[i].best()[0]
        if lm_name == "tiny_lm":                            # (a peaked LM: it must have changed a hypothesis somewhere)
            assert differs > 0, "the LM never changed a hypothesis: the test does not exercise the fusion"
    finally:
        eng.close()
