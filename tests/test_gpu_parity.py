"""Parity tests proper (need an MI355X): the HIP path, called through the C ABI, against the
numpy oracle on the same seeded inputs and against the golden vectors the reference produced.

Tolerances (north star): greedy tokens identical; joint logits within 1e-3 (fp32)."""
import os

import numpy as np
import pytest
import torch

from libreasr_amd import synth
from oracle import rnnt_oracle as O

pytestmark = pytest.mark.gpu

_ENGINES = {}


def engine(name, max_streams=16):
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    key = (name, max_streams)
    if key not in _ENGINES:
        graft.build()
        cfg = synth.model_cfg(name)
        sd = synth.synth_state_dict(cfg, seed=0)
        _ENGINES[key] = (Engine(sd, cfg, max_streams=max_streams), O.OracleTransducer(sd, cfg), cfg)
    return _ENGINES[key]


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def maxerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    i = np.unravel_index(np.argmax(np.abs(a - b)), a.shape)
    return float(np.abs(a - b).max()), i


# ------------------------------------------------------------------------------- front-end
def test_logmel_matches_oracle_and_reference_golden(golden_dir):
    eng, _, _ = engine("tiny")
    g = np.load(os.path.join(golden_dir, "frontend.npz"))
    pcm = synth.synth_pcm(2, 16000 + 937, seed=7)
    out = eng.logmel(dev(pcm)).cpu().numpy()
    for s in range(2):
        e, i = maxerr(out[s], O.logmel(pcm[s]))
        assert e < 2e-4, f"log-mel vs oracle: max err {e} at {i}"
        e, i = maxerr(out[s], g[f"logmel_{s}"])
        assert e < 3e-4, f"log-mel vs reference golden: max err {e} at {i}"
    z = eng.logmel(torch.zeros(1, 3840, device="cuda")).cpu().numpy()
    assert np.allclose(z, np.log(np.float32(1e-6)), atol=1e-6)      # silence -> log(1e-6) exactly


def test_stack_layout(golden_dir):
    eng, _, _ = engine("tiny")
    g = np.load(os.path.join(golden_dir, "frontend.npz"))
    pcm = synth.synth_pcm(2, 16000 + 937, seed=7)
    lm = eng.logmel(dev(pcm))
    st = eng.stack(lm).cpu().numpy()
    ref = np.stack([O.stack_downsample(lm[s].cpu().numpy()) for s in range(2)])
    assert np.array_equal(st, ref)                                   # pure data movement: bit exact
    e, i = maxerr(st[0], g["feats_0"])
    assert e < 3e-4, (e, i)
    assert eng.stack(lm[:, :9]).shape[1] == 0                        # fewer than n_stack frames -> no stacked frame


# ------------------------------------------------------------------------------- model pieces
@pytest.mark.parametrize("name", ["tiny", "tiny_lstm", "cfg2"])
def test_encoder(name, golden_dir):
    eng, m, cfg = engine(name)
    g = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
    n_sec = {"tiny": 3.0, "tiny_lstm": 3.0, "cfg2": 4.0}[name]
    pcm = synth.synth_pcm(3 if name == "tiny" else 2, int(16000 * n_sec), seed=1234)
    feats = np.stack([O.features_offline(p) for p in pcm])
    out, h, c = eng.encoder(dev(feats), return_state=True)
    ref, st = m.encoder(feats)
    e, i = maxerr(out.cpu().numpy(), ref)
    assert e < 5e-4, f"encoder out vs oracle: {e} at {i}"
    e, i = maxerr(h.cpu().numpy(), np.stack([a[0] for a in st]))
    assert e < 2e-4, f"h: {e} at {i}"
    e, i = maxerr(c.cpu().numpy(), np.stack([a[1] for a in st]))
    assert e < 5e-4, f"c: {e} at {i}"
    e, i = maxerr(out[0].cpu().numpy(), g["enc_out_0"])
    assert e < 5e-4, f"encoder out vs reference golden: {e} at {i}"


@pytest.mark.parametrize("name", ["tiny", "tiny_lstm", "cfg2", "cfg2_lstm"])
def test_predictor_and_joint(name, golden_dir):
    eng, m, cfg = engine(name)
    g = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
    toks = np.array([[2, 5, 7], [2, 9, 9], [3, 1, 4]], dtype=np.int32)
    hp = eng.predictor(toks).cpu().numpy()
    for r in range(3):
        st = None
        for t in toks[r]:
            x, st = m.predictor([t], st)
        e, i = maxerr(hp[r], x[0])
        assert e < 2e-4, f"predictor row {r}: {e} at {i}"
    e, _ = maxerr(eng.predictor(np.array([[2]], np.int32)).cpu().numpy()[0], g["pred_bos"])
    assert e < 2e-4
    e, _ = maxerr(eng.predictor(np.array([[2, 5]], np.int32)).cpu().numpy()[0], g["pred_bos_5"])
    assert e < 2e-4
    rng = np.random.default_rng(0)
    H = cfg["hidden"]
    a = rng.standard_normal((5, H)).astype(np.float32)
    b = rng.standard_normal((5, H)).astype(np.float32)
    logits, lp, am = eng.joint(dev(a), dev(b))
    ref_lp, ref_z = m.joint_logp(a, b)
    e, i = maxerr(logits.cpu().numpy(), ref_z)
    assert e < 1e-3, f"joint logits: {e} at {i}"
    assert list(am.cpu().numpy()) == list(ref_z.argmax(-1))
    e, _ = maxerr(lp.cpu().numpy(), ref_lp.max(-1))
    assert e < 1e-3
    # the reference's own joint output on (pred(BOS,5), enc frame 3)
    hp2 = eng.predictor(np.array([[2, 5]], np.int32))
    enc3 = dev(g["enc_out_0"][3][None])
    z, _, _ = eng.joint(hp2, enc3)
    e, i = maxerr(z.cpu().numpy()[0], g["joint_logits"])
    assert e < 1e-3, f"joint logits vs reference golden: {e} at {i}"


# ------------------------------------------------------------------------------- end to end
@pytest.mark.parametrize("name,n_sec,n_streams", [("tiny", 3.0, 3), ("tiny_lstm", 3.0, 2),
                                                   ("cfg2", 4.0, 2), ("cfg2_lstm", 2.0, 1),
                                                   ("ref6", 2.0, 1), ("cfg5", 2.0, 1)])
def test_offline_transcribe_matches_reference(name, n_sec, n_streams, golden_dir):
    eng, m, cfg = engine(name)
    g = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
    pcm = synth.synth_pcm(n_streams, int(16000 * n_sec), seed=1234)
    slots = [eng.open() for _ in range(n_streams)]
    try:
        eng.transcribe_pcm(slots, [dev(p) for p in pcm])
        for s, slot in enumerate(slots):
            toks, neg_logp, align = eng.fetch(slot)
            assert toks == list(g[f"off_tokens_{s}"]), f"stream {s}: tokens differ from the reference"
            assert abs(neg_logp - float(g[f"off_neglogp_{s}"])) < 2e-2
            assert abs(align - float(g[f"off_align_{s}"])) < 1e-9
        # same through the feature entry point (x_tfm output -> Transducer.transcribe), host buffers
        eng.transcribe_feats(slots, [O.features_offline(p) for p in pcm])
        for s, slot in enumerate(slots):
            assert eng.fetch(slot)[0] == list(g[f"off_tokens_{s}"])
    finally:
        for slot in slots:
            eng.close_slot(slot)


@pytest.mark.parametrize("name,n_sec,n_streams", [("tiny", 3.0, 3), ("tiny_lstm", 3.0, 2),
                                                   ("cfg2", 4.0, 2), ("cfg2_lstm", 2.0, 1),
                                                   ("ref6", 2.0, 1), ("cfg5", 2.0, 1)])
def test_streaming_matches_reference(name, n_sec, n_streams, golden_dir):
    eng, m, cfg = engine(name)
    g = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
    pcm = synth.synth_pcm(n_streams, int(16000 * n_sec), seed=1234)
    slots = [eng.open() for _ in range(n_streams)]
    try:
        chunks = [synth.stream_chunks(p, 1280, lead=1, tail=10) for p in pcm]
        got = [[] for _ in slots]
        counts = [[] for _ in slots]
        for k in range(len(chunks[0])):
            eng.push(slots, dev(np.stack([c[k] for c in chunks])))
            ran = eng.step(slots)
            for s, slot in enumerate(slots):
                t, _, _ = eng.fetch(slot)
                got[s] += t
                if ran:
                    counts[s].append(len(t))
        for s in range(n_streams):
            assert got[s] == list(g[f"st_tokens_{s}"]), f"stream {s}: streaming tokens differ from the reference"
            assert counts[s] == list(g[f"st_counts_{s}"])
    finally:
        for slot in slots:
            eng.close_slot(slot)


def test_ragged_batch_and_slot_isolation():
    """Utterances of different lengths in one batch, on non-contiguous slots, give the same tokens
    as each alone; a stream on another slot is not disturbed (row == slot, rows are masked)."""
    eng, m, cfg = engine("tiny")
    lens = [16000 * 2 + 311, 16000 * 3, 9000, 16000]
    pcm = [synth.synth_pcm(1, n, seed=50 + i)[0] for i, n in enumerate(lens)]
    ref = [m.decode_greedy(O.features_offline(p))[0] for p in pcm]
    slots = [eng.open() for _ in range(6)]
    try:
        use = [slots[5], slots[0], slots[3], slots[2]]
        # a live stream on slots[1] in the middle of its utterance
        live = slots[1]
        lp = synth.synth_pcm(1, 16000 * 2, seed=99)[0]
        ch = synth.stream_chunks(lp, 1280, lead=1, tail=10)
        fe, dec = O.StreamFrontend(), m.stream_decoder()
        got = []
        for k, c in enumerate(ch):
            eng.push([live], c[None])
            eng.step([live])
            got += eng.fetch(live)[0]
            o = fe.push(c)
            if o is not None:
                dec.step(o)
            if k == 12:
                eng.transcribe_pcm(use, pcm)
                for slot, r in zip(use, ref):
                    assert eng.fetch(slot)[0] == r
        assert got == dec.y
    finally:
        for slot in slots:
            eng.close_slot(slot)


def test_reset_semantics():
    """reset() of models.py:494-497 restores the learned initial state and re-runs the predictor on BOS."""
    eng, m, cfg = engine("tiny")
    pcm = synth.synth_pcm(1, 16000 * 2, seed=5)[0]
    slot = eng.open()
    try:
        ch = synth.stream_chunks(pcm, 1280, lead=1, tail=4)

        def run():
            out = []
            for c in ch:
                eng.push([slot], c[None])
                eng.step([slot])
                out += eng.fetch(slot)[0]
            return out

        a = run()
        eng.reset(slot, 15)
        b = run()
        assert a == b and len(a) > 0
    finally:
        eng.close_slot(slot)


def test_error_codes():
    from libreasr_amd._native import LasrError
    eng, _, _ = engine("tiny")
    with pytest.raises(LasrError):
        eng.step([15])                       # slot not open
    with pytest.raises(LasrError):
        eng.fetch(14)
    slot = eng.open()
    with pytest.raises(LasrError):
        eng.transcribe_pcm([slot], [np.zeros(100, np.float32)])   # shorter than the reflect padding
    eng.close_slot(slot)


@pytest.mark.parametrize("name,n_sec,n_streams", [("tiny", 3.0, 3), ("cfg2", 4.0, 2)])
def test_pipelined_submit_wait_matches_reference(name, n_sec, n_streams, golden_dir):
    """lasr_step_submit / lasr_step_wait (encoder of chunk k+1 overlapping the decode of chunk k on a
    second HIP stream) must give exactly the tokens and per-call counts of the reference."""
    from libreasr_amd._native import LasrError
    eng, m, cfg = engine(name)
    g = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
    pcm = synth.synth_pcm(n_streams, int(16000 * n_sec), seed=1234)
    slots = [eng.open() for _ in range(n_streams)]
    try:
        chunks = [synth.stream_chunks(p, 1280, lead=1, tail=10) for p in pcm]
        got = [[] for _ in slots]
        counts = [[] for _ in slots]

        def collect():
            if eng.wait():
                for s, t in enumerate(eng.fetch_many(slots, 64)):
                    got[s] += t
                    counts[s].append(len(t))

        for k in range(len(chunks[0])):
            eng.push(slots, dev(np.stack([c[k] for c in chunks])))
            eng.submit(slots)
            if eng.pending() >= 2:
                with pytest.raises(LasrError):          # state-changing calls are refused while steps are in flight
                    eng.reset(slots[0], 7)
                collect()
        while eng.pending():
            collect()
        for s in range(n_streams):
            assert got[s] == list(g[f"st_tokens_{s}"])
            assert counts[s] == list(g[f"st_counts_{s}"])
    finally:
        for slot in slots:
            eng.close_slot(slot)


def test_wide_batch_128_rows_and_staggered_streams():
    """128 slots (two 64-row groups in the predictor / joint kernels; config 4/5 batch shape) with
    streams that start at different chunks and drop out early, through the continuous submit/wait
    path and the synchronous path: every stream's tokens equal the oracle's for that stream alone."""
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    graft.build()
    cfg = synth.model_cfg("tiny")
    sd = synth.synth_state_dict(cfg, seed=0)
    m = O.OracleTransducer(sd, cfg)
    eng = Engine(sd, cfg, max_streams=128)
    try:
        rng = np.random.default_rng(3)
        n_streams = 100
        lens = rng.integers(6, 22, n_streams)                  # chunks per stream
        starts = rng.integers(0, 9, n_streams)                 # first global chunk index
        pcm = [synth.synth_pcm(1, int(l) * 1280, seed=200 + i)[0] for i, l in enumerate(lens)]
        ref = []
        for i in range(n_streams):
            fe, dec = O.StreamFrontend(), m.stream_decoder()
            for k in range(int(lens[i])):
                o = fe.push(pcm[i][k * 1280:(k + 1) * 1280])
                if o is not None:
                    dec.step(o)
            ref.append(dec.y)
        for mode in ("continuous", "sync"):
            slots = {}
            got = [[] for _ in range(n_streams)]
            total = int((starts + lens).max())

            def collect():
                if eng.wait():
                    live = sorted(slots)
                    for i, t in zip(live, eng.fetch_many([slots[i] for i in live], 64)):
                        got[i] += t

            for g in range(total + 1):
                for i in range(n_streams):                     # streams join ...
                    if starts[i] == g:
                        slots[i] = eng.open()
                active = [i for i in sorted(slots) if g - starts[i] < lens[i]]
                if active:
                    sl = [slots[i] for i in active]
                    chunk = np.stack([pcm[i][(g - starts[i]) * 1280:(g - starts[i] + 1) * 1280] for i in active])
                    eng.push(sl, dev(chunk))
                    if mode == "sync":
                        if eng.step(sl):
                            for i, t in zip(active, eng.fetch_many(sl, 64)):
                                got[i] += t
                    else:
                        eng.submit(sl)
                        if eng.pending() >= 3:
                            collect()
                done = [i for i in sorted(slots) if g - starts[i] >= lens[i] - 1]
                if done and mode == "continuous":              # ... and leave: a slot with a step in flight
                    while eng.pending():                       #     cannot be closed, so drain first
                        collect()
                for i in done:
                    eng.close_slot(slots.pop(i))
            while eng.pending():
                collect()
            for i in range(n_streams):
                assert got[i] == ref[i], f"{mode}: stream {i} (start {starts[i]}, {lens[i]} chunks)"
    finally:
        eng.close()


def test_config5_shape_against_oracle():
    """BASELINE configs[4] model shape (8 x LSTM(1536) encoder, 2-layer LSTM predictor, J = 1536) in fp32:
    offline and continuous-streaming tokens of streams the goldens do not cover, against the oracle."""
    eng, m, cfg = engine("cfg5")
    pcm = synth.synth_pcm(3, 16000 * 2, seed=77)
    slots = [eng.open() for _ in range(3)]
    try:
        eng.transcribe_pcm(slots, [dev(p) for p in pcm])
        for s, slot in enumerate(slots):
            ref = m.decode_greedy(O.features_offline(pcm[s]))
            toks, neg_logp, align = eng.fetch(slot)
            assert toks == ref[0]
            assert abs(neg_logp - ref[1]) < 2e-2 * max(1.0, abs(ref[1]))
        for slot in slots:
            eng.reset(slot, 15)
        chunks = [synth.stream_chunks(p, 1280, lead=1, tail=6) for p in pcm]
        got = [[] for _ in slots]
        refs = []
        for s in range(3):
            fe, dec = O.StreamFrontend(), m.stream_decoder()
            for c in chunks[s]:
                o = fe.push(c)
                if o is not None:
                    dec.step(o)
            refs.append(dec.y)

        def collect():
            if eng.wait():
                for s, t in enumerate(eng.fetch_many(slots, 64)):
                    got[s] += t

        for k in range(len(chunks[0])):
            eng.push(slots, dev(np.stack([c[k] for c in chunks])))
            eng.submit(slots)
            if eng.pending() >= 3:
                collect()
        while eng.pending():
            collect()
        assert got == refs
    finally:
        for slot in slots:
            eng.close_slot(slot)


@pytest.mark.parametrize("la", [2, 4])
def test_streaming_lookahead_is_invisible(la, monkeypatch, golden_dir):
    """Greedy lookahead (k_select evaluating frames t .. t+la-1 against one predictor state and consuming
    the run of blanks) must not change a single token or per-call count: synchronous steps with 2 and 5
    frames, the pipelined / continuous loop, and ragged streams (env knob read at engine creation)."""
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    graft.build()
    monkeypatch.setenv("LASR_LOOKAHEAD", str(la))
    name = "tiny_lstm"
    cfg = synth.model_cfg(name)
    sd = synth.synth_state_dict(cfg, seed=0)
    eng = Engine(sd, cfg, max_streams=8)
    m = O.OracleTransducer(sd, cfg)
    g = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
    n = 2
    pcm = synth.synth_pcm(n, 16000 * 3, seed=1234)
    chunks = [synth.stream_chunks(p, 1280, lead=1, tail=10) for p in pcm]
    for mode in ("sync", "pipelined"):
        slots = [eng.open() for _ in range(n)]
        got, counts = [[] for _ in range(n)], [[] for _ in range(n)]

        def take():
            for s, t in enumerate(eng.fetch_many(slots, 64)):
                got[s] += t
                counts[s].append(len(t))

        for k in range(len(chunks[0])):
            eng.push(slots, np.stack([c[k] for c in chunks]))
            if mode == "sync":
                if eng.step(slots):
                    take()
            else:
                eng.submit(slots)
                if eng.pending() >= 3 and eng.wait():
                    take()
        while eng.pending():
            if eng.wait():
                take()
        for s in range(n):
            assert got[s] == list(g[f"st_tokens_{s}"]), (mode, s)
            assert counts[s] == list(g[f"st_counts_{s}"]), (mode, s)
        for s in slots:
            eng.close_slot(s)
    # feature-level steps with 5 frames per call and different lengths of history per stream
    slots = [eng.open() for _ in range(n)]
    feats = [O.features_offline(p) for p in pcm]
    decs = [m.stream_decoder() for _ in range(n)]
    got = [[] for _ in range(n)]
    for t0 in range(0, 30, 5):
        eng.step_feats(slots, np.stack([f[t0:t0 + 5] for f in feats]))
        for s, t in enumerate(eng.fetch_many(slots, 256)):
            got[s] += t
        for s in range(n):
            decs[s].step(feats[s][t0:t0 + 5])
    for s in range(n):
        assert got[s] == decs[s].y
    eng.close()


def test_full_size_properties_config2():
    """BASELINE configs[1] at full size (64 streams, 4x1024 encoder): properties that do not need the
    oracle to run at that size.  (i) batch invariance: a stream's tokens do not depend on which other
    streams share the batch; (ii) protocol invariance: pipelined/continuous == synchronous, per chunk;
    (iii) chunking invariance: feeding the same stacked frames 2 or 6 at a time gives the same tokens;
    (iv) 3 of the streams against the oracle as an anchor."""
    eng, m, cfg = engine("cfg2", max_streams=64)
    n = 64
    n_chunks = 16                                   # 1.28 s of audio per stream
    pcm = np.stack([synth.synth_pcm(1, n_chunks * 1280, seed=4000 + s)[0] for s in range(n)])
    chunks = pcm.reshape(n, n_chunks, 1280)

    def run(idx, pipelined):
        slots = [eng.open() for _ in idx]
        out = [[] for _ in idx]
        per_call = [[] for _ in idx]

        def take():
            for i, t in enumerate(eng.fetch_many(slots, 128)):
                out[i] += t
                per_call[i].append(len(t))

        for k in range(n_chunks):
            eng.push(slots, dev(chunks[idx, k]))
            if pipelined:
                eng.submit(slots)
                if eng.pending() >= 4 and eng.wait():
                    take()
            elif eng.step(slots):
                take()
        while eng.pending():
            if eng.wait():
                take()
        for s in slots:
            eng.close_slot(s)
        return out, per_call

    full_sync, calls_sync = run(list(range(n)), False)
    full_pipe, calls_pipe = run(list(range(n)), True)
    assert full_sync == full_pipe and calls_sync == calls_pipe          # (ii)
    assert sum(len(t) for t in full_sync) > n                            # the workload emits tokens
    some = [3, 17, 42, 63]
    solo, _ = run(some, False)
    for i, s in enumerate(some):
        assert solo[i] == full_sync[s], s                                # (i)
    for s in some[:3]:                                                   # (iv)
        fe, dec = O.StreamFrontend(), m.stream_decoder()
        for k in range(n_chunks):
            o = fe.push(chunks[s, k])
            if o is not None:
                dec.step(o)
        assert dec.y == full_sync[s], s
    # (iii) same frames, different call granularity (encoder / predictor state carried across calls)
    feats = np.stack([O.features_offline(pcm[s]) for s in range(8)])[:, :12]
    res = []
    for step in (2, 6):
        slots = [eng.open() for _ in range(8)]
        out = [[] for _ in range(8)]
        for t0 in range(0, 12, step):
            eng.step_feats(slots, feats[:, t0:t0 + step])
            for i, t in enumerate(eng.fetch_many(slots, 256)):
                out[i] += t
        res.append(out)
        for s in slots:
            eng.close_slot(s)
    assert res[0] == res[1]


def test_odd_shapes_take_the_generic_paths():
    """Nothing about the reference shape is baked in: 64 mels x 12 stacked frames (feat 768, the runtime
    k_stack_ln), stride 6, a 3-frame Buffer, hidden 96 / joint 80 / vocab 48 / embed 48 (no static K
    schedule, partial waves in the K split), 3 encoder layers, 1-layer LSTM predictor, 5 slots (not a
    multiple of 16), a 4-chunk window of 960-sample chunks.  Offline + streaming against the oracle."""
    import __graft_entry__ as graft
    from libreasr_amd.engine import Engine
    graft.build()
    cfg = dict(feat=768, embed=48, vocab=48, hidden=96, joint=80, enc_layers=3, pred_layers=1,
               pred_cell="LSTM", blank_bias=6.0, out_scale=6.0)
    sd = synth.synth_state_dict(cfg, seed=3)
    fe = dict(n_mels=64, n_stack=12, stride=6, n_buffer=3, n_window=4, chunk=960, win=320, hop=120)
    eng = Engine(sd, cfg, max_streams=5, **fe)
    m = O.OracleTransducer(sd, cfg)
    mel = dict(n_mels=64, win=320, hop=120)
    n = 5
    pcm = synth.synth_pcm(n, 16000 * 2 + 777, seed=21)
    slots = [eng.open() for _ in range(n)]
    lens = [len(pcm[0]), 9000, 16000, 20011, 32000]
    eng.transcribe_pcm(slots, [pcm[i][:lens[i]] for i in range(n)])
    n_tok = 0
    for i, s in enumerate(slots):
        toks, neg_logp, align = eng.fetch(s)
        ref = m.decode_greedy(O.features_offline(pcm[i][:lens[i]], n_stack=12, downsample=6, **mel))
        assert toks == ref[0], (i, toks, ref[0])
        assert abs(neg_logp - ref[1]) < 1e-2 * max(1.0, abs(ref[1]))
        assert abs(align - ref[2]) < 1e-6
        n_tok += len(toks)
    assert n_tok > 0
    # streaming: 60 ms chunks, model every 3 calls of the transform
    for s in slots:
        eng.reset(s, 15)
    chunks = [synth.stream_chunks(p[:24000], 960, lead=1, tail=6) for p in pcm]
    got = [[] for _ in range(n)]
    for k in range(len(chunks[0])):
        eng.push(slots, np.stack([c[k] for c in chunks]))
        if eng.step(slots):
            for i, t in enumerate(eng.fetch_many(slots, 256)):
                got[i] += t
    for i in range(n):
        f, dec = O.StreamFrontend(n_stack=12, downsample=6, n_buffer=3, n_window=4, **mel), m.stream_decoder()
        for ch in chunks[i]:
            o = f.push(ch)
            if o is not None:
                dec.step(o)
        assert got[i] == dec.y, (i, got[i], dec.y)
    eng.close()


def test_maximum_stream_count_1024():
    """max_streams = 1024 (the ABI's maximum: 16 row groups in the predictor / joint kernels, the in-kernel
    compaction scan at its 1024-row limit, the command-block push path for M > 512): all slots open, 640
    of them streaming (ragged: the rest idle), synchronous and pipelined; spot-checked against the oracle."""
    import __graft_entry__ as graft
    from libreasr_amd._native import LasrError
    from libreasr_amd.engine import Engine
    graft.build()
    cfg = synth.model_cfg("tiny")
    sd = synth.synth_state_dict(cfg, seed=0)
    eng = Engine(sd, cfg, max_streams=1024)
    m = O.OracleTransducer(sd, cfg)
    slots = [eng.open() for _ in range(1024)]
    assert slots == list(range(1024))
    with pytest.raises(LasrError):
        eng.open()                                   # all slots taken
    act = list(range(0, 1024, 2))[:320] + list(range(1, 1024, 2))[:320]      # 640 active slots, interleaved
    base = synth.synth_pcm(8, 16 * 1280, seed=77)
    chunks = base.reshape(8, 16, 1280)
    which = np.array([s % 8 for s in act])
    for mode in ("sync", "pipelined"):
        got = {s: [] for s in act}
        for k in range(16):
            eng.push(act, chunks[which, k])
            if mode == "sync":
                if eng.step(act):
                    for s, t in zip(act, eng.fetch_many(act, 64)):
                        got[s] += t
            else:
                eng.submit(act)
                if eng.pending() >= 3 and eng.wait():
                    for s, t in zip(act, eng.fetch_many(act, 64)):
                        got[s] += t
        while eng.pending():
            if eng.wait():
                for s, t in zip(act, eng.fetch_many(act, 64)):
                    got[s] += t
        ref = []
        for i in range(8):
            fe, dec = O.StreamFrontend(), m.stream_decoder()
            for k in range(16):
                o = fe.push(chunks[i, k])
                if o is not None:
                    dec.step(o)
            ref.append(dec.y)
        assert sum(len(r) for r in ref) > 0
        for s in act:
            assert got[s] == ref[s % 8], (mode, s)
        for s in act:
            eng.reset(s, 15)
    eng.close()


def test_long_stream_wraps_every_ring():
    """40 s of audio per stream through the continuous loop: the per-row token ring (256), the step-mark
    ring (16), the pe frame ring (32), the flag ring (64 iterations) and the command-block ring (64) all wrap
    many times; tokens per chunk must still equal the oracle's."""
    eng, m, cfg = engine("tiny_lstm", max_streams=16)
    n, n_chunks = 4, 500
    pcm = np.stack([synth.synth_pcm(1, n_chunks * 1280, seed=900 + s)[0] for s in range(n)])
    chunks = pcm.reshape(n, n_chunks, 1280)
    slots = [eng.open() for _ in range(n)]
    got = [[] for _ in range(n)]
    for k in range(n_chunks):
        eng.push(slots, chunks[:, k])
        eng.submit(slots)
        if eng.pending() >= 5 and eng.wait():
            for i, t in enumerate(eng.fetch_many(slots, 128)):
                got[i] += t
    while eng.pending():
        if eng.wait():
            for i, t in enumerate(eng.fetch_many(slots, 128)):
                got[i] += t
    total = 0
    for i in range(n):
        fe, dec = O.StreamFrontend(), m.stream_decoder()
        for k in range(n_chunks):
            o = fe.push(chunks[i, k])
            if o is not None:
                dec.step(o)
        assert got[i] == dec.y, i
        total += len(dec.y)
    assert total > 4 * 256                      # enough tokens to wrap the token ring of a row
    for s in slots:
        eng.close_slot(s)
