"""LM shallow fusion (lm.py LM / LMFuser inside both greedy loops) through the C ABI, against
goldens produced by the reference with its own fp32 LM class attached (oracle/make_golden.py "lm") and
against the numpy oracle.  Tokens identical; offline -log p within 1e-2 * max(1, |.|)."""
import os

import numpy as np
import pytest

from libreasr_amd import synth
from oracle import rnnt_oracle as O

pytestmark = pytest.mark.gpu

CASES = [("tiny_soft", "tiny_lm", 3.0, 3), ("tiny_lstm", "tiny_lm_untied", 3.0, 2), ("cfg2", "lm768", 3.0, 1)]
_ENGINES = {}


def engine(name, lm_name, dtype="f32", int8=False):
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    key = (name, lm_name, dtype, int8)
    if key not in _ENGINES:
        graft.build()
        cfg = synth.model_cfg(name)
        sd = synth.synth_state_dict(cfg, seed=0)
        lsd = synth.synth_lm_state_dict(lm_name)
        eng = Engine(sd, cfg, max_streams=8, dtype=dtype)
        eng.attach_lm(lsd, int8=int8)
        m = O.OracleTransducer(sd, cfg, operand=dtype)
        m.lm = O.OracleLM(lsd, quantized=int8)
        _ENGINES[key] = (eng, m, cfg)
    return _ENGINES[key]


@pytest.mark.parametrize("name,lm_name,n_sec,n_streams", CASES)
def test_lm_fusion_offline_and_streaming_match_reference(name, lm_name, n_sec, n_streams, golden_dir):
    eng, m, cfg = engine(name, lm_name)
    g = np.load(os.path.join(golden_dir, f"model_{name}__{lm_name}.npz"))
    pcm = synth.synth_pcm(n_streams, int(16000 * n_sec), seed=1234)
    slots = [eng.open() for _ in range(n_streams)]
    eng.transcribe_pcm(slots, [pcm[i] for i in range(n_streams)])
    changed = 0
    for i, s in enumerate(slots):
        toks, neg_logp, _ = eng.fetch(s)
        assert toks == list(g[f"off_tokens_{i}"]), (i, toks, list(g[f"off_tokens_{i}"]))
        ref = float(g[f"off_neglogp_{i}"])
        assert abs(neg_logp - ref) < 1e-2 * max(1.0, abs(ref))
        changed += toks != list(g[f"off_tokens_nolm_{i}"])
    if name != "tiny_lstm":
        assert changed > 0
    # a second utterance on the same slots starts from a fresh LM state (new LMFuser per decode, models.py:401)
    eng.transcribe_pcm(slots, [pcm[i] for i in range(n_streams)])
    for i, s in enumerate(slots):
        assert eng.fetch(s)[0] == list(g[f"off_tokens_{i}"])
    # streaming, all streams batched, per-chunk token counts as the reference yields them
    for s in slots:
        eng.reset(s, 15)
    chunks = [synth.stream_chunks(pcm[i], 1280, lead=1, tail=10) for i in range(n_streams)]
    got = [[] for _ in range(n_streams)]
    counts = [[] for _ in range(n_streams)]
    for k in range(len(chunks[0])):
        eng.push(slots, np.stack([chunks[i][k] for i in range(n_streams)]))
        if eng.step(slots):
            for i, t in enumerate(eng.fetch_many(slots, cap=512)):
                got[i] += t
                counts[i].append(len(t))
    for i in range(n_streams):
        assert got[i] == list(g[f"st_tokens_{i}"]), (i, got[i][:20], list(g[f"st_tokens_{i}"])[:20])
        assert counts[i] == list(g[f"st_counts_{i}"])
    for s in slots:
        eng.close_slot(s)


def test_lm_fusion_pipelined_and_reset_bits():
    """submit/wait (continuous decode loop) with an LM attached == the oracle per chunk; reset bit 4
    (reset_lm, models.py:491-492) alone clears only the LM."""
    eng, m, cfg = engine("tiny_soft", "tiny_lm")
    n = 4
    pcm = synth.synth_pcm(n, 16000 * 2, seed=77)
    chunks = [synth.stream_chunks(pcm[i], 1280, lead=1, tail=6) for i in range(n)]
    slots = [eng.open() for _ in range(n)]
    got = [[] for _ in range(n)]
    for k in range(len(chunks[0])):
        eng.push(slots, np.stack([chunks[i][k] for i in range(n)]))
        eng.submit(slots)
        if eng.pending() >= 3 and eng.wait():
            for i, t in enumerate(eng.fetch_many(slots, cap=512)):
                got[i] += t
    while eng.pending():
        if eng.wait():
            for i, t in enumerate(eng.fetch_many(slots, cap=512)):
                got[i] += t
    decs = []
    for i in range(n):
        fe, dec = O.StreamFrontend(), m.stream_decoder()
        for ch in chunks[i]:
            o = fe.push(ch)
            if o is not None:
                dec.step(o)
        assert got[i] == dec.y, (i, got[i][:20], dec.y[:20])
        decs.append((fe, dec))
    # LM-only reset on stream 0, then two more seconds of audio
    eng.reset(slots[0], 4)
    fe, dec = decs[0]
    dec.fuser.reset()
    before = len(dec.y)
    extra = []
    for ch in synth.stream_chunks(pcm[1], 1280, lead=0, tail=4):
        eng.push([slots[0]], ch[None])
        if eng.step([slots[0]]):
            extra += eng.fetch(slots[0])[0]
        o = fe.push(ch)
        if o is not None:
            dec.step(o)
    assert extra == dec.y[before:]
    for s in slots:
        eng.close_slot(s)


def test_lm_bf16_tracks_emulation():
    import difflib
    eng, m, cfg = engine("cfg2", "lm768", dtype="bf16")
    # the LM itself runs with bf16 operands too: emulate (weights + GEMM-input activations rounded)
    lsd = synth.synth_lm_state_dict("lm768")
    q = O.bf16_round
    lsd_q = {k: (q(np.asarray(v)) if ("weight_hh" in k or (k.startswith("rnn.weight_ih") and not k.endswith("l0")) or k == "linear.weight") else v)
             for k, v in lsd.items()}
    m.lm = O.OracleLM(lsd_q)
    pcm = synth.synth_pcm(1, 16000 * 3, seed=1234)
    s = eng.open()
    eng.transcribe_pcm([s], [pcm[0]])
    toks = eng.fetch(s)[0]
    ref = m.decode_greedy(O.features_offline(pcm[0]))[0]
    sim = difflib.SequenceMatcher(None, toks, ref, autojunk=False).ratio() if (toks or ref) else 1.0
    print(f"bf16 + LM: {len(toks)} tokens, similarity to the bf16 emulation {sim:.3f}")
    assert len(toks) > 0 and sim >= 0.8
    eng.close_slot(s)


def test_lm_attach_errors():
    from libreasr_amd import _native as N
    from libreasr_amd.engine import Engine
    cfg = synth.model_cfg("tiny")
    sd = synth.synth_state_dict(cfg, seed=0)
    eng = Engine(sd, cfg, max_streams=4)
    bad = synth.synth_lm_state_dict(dict(vocab=128, embed=32, hidden=32, layers=1))
    with pytest.raises(N.LasrError):
        eng.attach_lm(bad)                          # vocabulary mismatch
    eng.attach_lm(synth.synth_lm_state_dict("tiny_lm"))
    with pytest.raises(N.LasrError):
        eng.attach_lm(synth.synth_lm_state_dict("tiny_lm"))   # already attached
    eng.close()
    eng = Engine(sd, cfg, max_streams=4, beam=2)
    with pytest.raises(N.LasrError):
        eng.attach_lm(synth.synth_lm_state_dict("tiny_lm"), int8=True)   # the int8-served form is greedy only
    eng.attach_lm(synth.synth_lm_state_dict("tiny_lm"), int8=False)      # the beam takes the fp32 / bf16 LM (round 3)
    eng.close()


def test_lm_through_the_facade(golden_dir):
    """LibreASR.load(..., synthetic_lm=...) == config.py:140-157 attaching the LM at load time."""
    from libreasr_amd.api import LibreASR
    g = np.load(os.path.join(golden_dir, "model_tiny_soft__tiny_lm.npz"))
    asr = LibreASR.load("en", synthetic="tiny_soft", synthetic_lm="tiny_lm", max_streams=4)
    pcm = synth.synth_pcm(3, 16000 * 3, seed=1234)
    ids = asr.transcribe([pcm[0], pcm[1]], return_ids=True)
    assert ids[0] == list(g["off_tokens_0"]) and ids[1] == list(g["off_tokens_1"])
    last = None
    for y in asr.stream(synth.stream_chunks(pcm[2], 1280, lead=1, tail=10), return_ids=True):
        last = y
    assert last == list(g["st_tokens_2"])
    asr.engine.close()


@pytest.mark.parametrize("name,lm_name,n_sec,n_streams", CASES)
def test_fp32_lm_against_the_int8_served_reference(name, lm_name, n_sec, n_streams, golden_dir):
    """The reference SERVES its LM int8-dynamically-quantised (load_lm, lm.py:97); the engine runs the LM in fp32.  Goldens
    from the reference's own maybe_quantize'd LM (oracle/ref_fixture.py:ref_lm_int8; the oracle's int8 emulation reproduces
    them exactly, tests/test_oracle.py).  The quantisation noise enters the decision through 0.1 x the standardised LM
    log-probs: on these fixtures it flips no fused decision, so the fp32 engine must give the int8 reference's tokens."""
    eng, m, cfg = engine(name, lm_name)
    g = np.load(os.path.join(golden_dir, f"model_{name}__{lm_name}_int8.npz"))
    pcm = synth.synth_pcm(n_streams, int(16000 * n_sec), seed=1234)
    slots = [eng.open() for _ in range(n_streams)]
    try:
        eng.transcribe_pcm(slots, [pcm[i] for i in range(n_streams)])
        for i, s in enumerate(slots):
            assert eng.fetch(s)[0] == list(g[f"off_tokens_{i}"]), i
        for s in slots:
            eng.reset(s, 15)
        chunks = [synth.stream_chunks(pcm[i], 1280, lead=1, tail=10) for i in range(n_streams)]
        got = [[] for _ in range(n_streams)]
        for k in range(len(chunks[0])):
            eng.push(slots, np.stack([chunks[i][k] for i in range(n_streams)]))
            if eng.step(slots):
                for i, t in enumerate(eng.fetch_many(slots, 64)):
                    got[i] += t
        for i in range(n_streams):
            assert got[i] == list(g[f"st_tokens_{i}"]), i
    finally:
        for s in slots:
            eng.close_slot(s)


@pytest.mark.parametrize("name,lm_name,n_sec,n_streams", CASES)
def test_int8_lm_engine_matches_the_int8_served_reference(name, lm_name, n_sec, n_streams, golden_dir):
    """lasr_attach_lm_int8: the LM quantised as load_lm does (quantize_dynamic qint8), integer arithmetic on the MFMA
    (integer-valued bf16 operands, exact).  Tokens == the goldens of the reference's own quantised LM, offline and streaming;
    pipelined protocol == the oracle's int8 emulation per stream."""
    eng, m, cfg = engine(name, lm_name, int8=True)
    g = np.load(os.path.join(golden_dir, f"model_{name}__{lm_name}_int8.npz"))
    pcm = synth.synth_pcm(n_streams, int(16000 * n_sec), seed=1234)
    slots = [eng.open() for _ in range(n_streams)]
    try:
        eng.transcribe_pcm(slots, [pcm[i] for i in range(n_streams)])
        for i, s in enumerate(slots):
            toks, neg_logp, _ = eng.fetch(s)
            assert toks == list(g[f"off_tokens_{i}"]), i
            ref = float(g[f"off_neglogp_{i}"])
            assert abs(neg_logp - ref) < 1e-2 * max(1.0, abs(ref))
        for s in slots:
            eng.reset(s, 15)
        chunks = [synth.stream_chunks(pcm[i], 1280, lead=1, tail=10) for i in range(n_streams)]
        got = [[] for _ in range(n_streams)]
        counts = [[] for _ in range(n_streams)]
        for k in range(len(chunks[0])):
            eng.push(slots, np.stack([chunks[i][k] for i in range(n_streams)]))
            if eng.step(slots):
                for i, t in enumerate(eng.fetch_many(slots, 512)):
                    got[i] += t
                    counts[i].append(len(t))
        for i in range(n_streams):
            assert got[i] == list(g[f"st_tokens_{i}"]), i
            assert counts[i] == list(g[f"st_counts_{i}"])
        # pipelined protocol on fresh audio against the oracle's emulation
        for s in slots:
            eng.reset(s, 15)
        pcm2 = synth.synth_pcm(n_streams, 16000 * 2, seed=99)
        chunks = [synth.stream_chunks(pcm2[i], 1280, lead=1, tail=6) for i in range(n_streams)]
        got = [[] for _ in range(n_streams)]
        for k in range(len(chunks[0])):
            eng.push(slots, np.stack([chunks[i][k] for i in range(n_streams)]))
            eng.submit(slots)
            if eng.pending() >= 3 and eng.wait():
                for i, t in enumerate(eng.fetch_many(slots, 512)):
                    got[i] += t
        while eng.pending():
            if eng.wait():
                for i, t in enumerate(eng.fetch_many(slots, 512)):
                    got[i] += t
        for i in range(n_streams):
            fe, dec = O.StreamFrontend(), m.stream_decoder()
            for ch in chunks[i]:
                o = fe.push(ch)
                if o is not None:
                    dec.step(o)
            assert got[i] == dec.y, (i, got[i][:20], dec.y[:20])
    finally:
        for s in slots:
            eng.close_slot(s)


def test_int8_lm_is_not_the_fp32_lm():
    """An utterance on which the quantisation noise DOES flip a fused re-pick (found by a seed search on the oracle): the int8
    engine must follow the oracle's int8 emulation, the fp32 engine the fp32 LM, and the two transcripts differ."""
    pcm = synth.synth_pcm(1, 16000 * 3, seed=328)[0]
    feats = O.features_offline(pcm)
    out = {}
    for int8 in (False, True):
        eng, m, cfg = engine("tiny_soft", "tiny_lm", int8=int8)
        s = eng.open()
        try:
            eng.transcribe_pcm([s], [pcm])
            out[int8] = eng.fetch(s)[0]
        finally:
            eng.close_slot(s)
        assert out[int8] == m.decode_greedy(feats)[0], int8
    assert out[False] != out[True]
