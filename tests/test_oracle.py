"""Pins the numpy oracle (oracle/rnnt_oracle.py) against golden vectors produced by the
REFERENCE's own code (oracle/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from libreasr_amd import synth
from oracle import rnnt_oracle as O


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_logmel_matches_reference(golden_dir):
    g = load(golden_dir, "frontend.npz")
    pcm = synth.synth_pcm(2, 16000 + 937, seed=7)
    for s in range(2):
        lm = O.logmel(pcm[s])
        assert lm.shape == g[f"logmel_{s}"].shape
        np.testing.assert_allclose(lm, g[f"logmel_{s}"], atol=2e-4, rtol=0)
        feats = O.features_offline(pcm[s])
        np.testing.assert_allclose(feats, g[f"feats_{s}"], atol=2e-4, rtol=0)
    z = O.logmel(np.zeros(3840, np.float32))
    np.testing.assert_allclose(z, g["logmel_zero"], atol=1e-6)
    assert np.allclose(z, np.log(np.float32(1e-6)))


def test_stack_layout_is_mel_major_frame_minor():
    spec = np.arange(30 * 128, dtype=np.float32).reshape(30, 128)
    st = O.stack_downsample(spec)
    assert st.shape == (3, 1280)
    for t in range(3):
        for m in (0, 5, 127):
            for k in (0, 9):
                assert st[t, m * 10 + k] == spec[8 * t + k, m]


def test_stream_frontend_matches_reference(golden_dir):
    g = load(golden_dir, "frontend.npz")
    pcm = synth.synth_pcm(2, 16000 + 937, seed=7)
    fe = O.StreamFrontend()
    pattern, outs = [], []
    for c in synth.stream_chunks(pcm[0], 1280, lead=1, tail=2):
        o = fe.push(c)
        if not fe.called:           # window not yet full: the servicer does not call the pipeline
            continue
        pattern.append(0 if o is None else 1)
        if o is not None:
            outs.append(o)
    assert pattern == list(g["stream_pattern"])
    np.testing.assert_allclose(np.stack(outs), g["stream_feats"], atol=2e-4, rtol=0)


@pytest.mark.parametrize("name,n_sec,n_streams", [("tiny", 3.0, 3), ("tiny_lstm", 3.0, 2),
                                                   ("cfg2", 4.0, 2), ("cfg2_lstm", 2.0, 1),
                                                   ("ref6", 2.0, 1), ("cfg5", 2.0, 1)])
def test_model_matches_reference(golden_dir, name, n_sec, n_streams):
    g = load(golden_dir, f"model_{name}.npz")
    cfg = synth.model_cfg(name)
    m = O.OracleTransducer(synth.synth_state_dict(cfg, seed=0), cfg)
    pcm = synth.synth_pcm(n_streams, int(16000 * n_sec), seed=1234)
    for s in range(n_streams):
        feats = O.features_offline(pcm[s])
        if s == 0 and "enc_out_0" in g:
            enc, st = m.encoder(feats[None])
            np.testing.assert_allclose(enc[0], g["enc_out_0"], atol=2e-4)
            np.testing.assert_allclose(np.stack([a[0][0] for a in st]), g["enc_h_0"], atol=1e-4)
            np.testing.assert_allclose(np.stack([a[1][0] for a in st]), g["enc_c_0"], atol=2e-4)
            hp, ps = m.predictor([m.bos])
            np.testing.assert_allclose(hp[0], g["pred_bos"], atol=1e-4)
            hp2, _ = m.predictor([5], ps)
            np.testing.assert_allclose(hp2[0], g["pred_bos_5"], atol=1e-4)
            _, z = m.joint_logp(hp2, enc[0, 3][None])
            np.testing.assert_allclose(z[0], g["joint_logits"], atol=1e-3)
        toks, neg_logp, score, iters, outs = m.decode_greedy(feats, return_logits=True)
        assert toks == list(g[f"off_tokens_{s}"])
        assert iters == list(g[f"off_iters_{s}"])
        assert abs(neg_logp - float(g[f"off_neglogp_{s}"])) < 1e-2
        assert abs(score - float(g[f"off_align_{s}"])) < 1e-9
        # streaming: api-server window + Buffer + transcribe_stream
        fe, dec = O.StreamFrontend(), m.stream_decoder()
        counts = []
        for c in synth.stream_chunks(pcm[s], 1280, lead=1, tail=10):
            o = fe.push(c)
            if o is not None:
                counts.append(len(dec.step(o)))
        assert dec.y == list(g[f"st_tokens_{s}"])
        assert counts == list(g[f"st_counts_{s}"])


def test_should_reset_policy():
    # api-server.py:44-50: 10 ms * downsample 8 * n_buffer 2 * steps >= 4000 ms
    assert not O.should_reset(24) and O.should_reset(25)


@pytest.mark.parametrize("name", ["tiny", "tiny_lstm"])
def test_beam_width_1_is_the_reference_greedy(golden_dir, name):
    """The beam spec (absent from the reference: parity unpinned) degenerates to the reference's greedy
    decode at W = 1: tokens and log-prob of the goldens the reference produced."""
    g = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
    cfg = synth.model_cfg(name)
    m = O.OracleTransducer(synth.synth_state_dict(cfg, seed=0), cfg)
    pcm = synth.synth_pcm(2, 16000 * 3, seed=1234)
    for s in range(2):
        f = O.features_offline(pcm[s])
        y, score, _ = m.decode_beam(f, 1)
        assert y == list(g[f"off_tokens_{s}"])
        assert abs(-score - float(g[f"off_neglogp_{s}"])) < 1e-3 * max(1.0, abs(score))
        y8, s8, hyps = m.decode_beam(f, 8)
        assert len(hyps) == 8 and s8 >= score - 1e-6          # wider beam: not worse on these inputs
        # streaming form == offline form when fed the same frames with the offline iteration cap
        sb = O.StreamBeamDecoder(m, 4, max_iters=3)
        for t in range(0, f.shape[0], 2):
            sb.step(f[t:t + 2])
        assert sb.best()[0] == m.decode_beam(f, 4)[0]


LM_CASES = [("tiny_soft", "tiny_lm", 3.0, 3), ("tiny_lstm", "tiny_lm_untied", 3.0, 2), ("cfg2", "lm768", 3.0, 1)]


@pytest.mark.parametrize("name,lm_name,n_sec,n_streams", LM_CASES)
def test_lm_shallow_fusion_matches_reference(golden_dir, name, lm_name, n_sec, n_streams):
    """LMFuser (lm.py:43-83) inside both greedy loops, goldens from the reference with its own LM class
    attached (fp32; the int8 dynamic quantisation of load_lm is un-vendored numerics: parity unpinned)."""
    g = load(golden_dir, f"model_{name}__{lm_name}.npz")
    cfg = synth.model_cfg(name)
    m = O.OracleTransducer(synth.synth_state_dict(cfg, seed=0), cfg)
    m.lm = O.OracleLM(synth.synth_lm_state_dict(lm_name))
    pcm = synth.synth_pcm(n_streams, int(16000 * n_sec), seed=1234)
    changed = 0
    for s in range(n_streams):
        feats = O.features_offline(pcm[s])
        toks, neg_logp, score, iters = m.decode_greedy(feats)
        assert toks == list(g[f"off_tokens_{s}"])
        assert iters == list(g[f"off_iters_{s}"])
        assert abs(neg_logp - float(g[f"off_neglogp_{s}"])) < 1e-2
        changed += toks != list(g[f"off_tokens_nolm_{s}"])
        fe, dec = O.StreamFrontend(), m.stream_decoder()
        counts = []
        for c in synth.stream_chunks(pcm[s], 1280, lead=1, tail=10):
            o = fe.push(c)
            if o is not None:
                counts.append(len(dec.step(o)))
        assert dec.y == list(g[f"st_tokens_{s}"])
        assert counts == list(g[f"st_counts_{s}"])
    if name != "tiny_lstm":
        assert changed > 0        # the fixture is only worth something if the LM overrides tokens


@pytest.mark.parametrize("name,lm_name,n_sec,n_streams", LM_CASES)
def test_int8_lm_shallow_fusion_matches_reference(golden_dir, name, lm_name, n_sec, n_streams):
    """The LM as the reference SERVES it (load_lm, lm.py:97: maybe_quantize = quantize_dynamic({LSTM, Linear}, qint8)): goldens
    from the reference's own maybe_quantize on its LM class (oracle/ref_fixture.py:ref_lm_int8, installed torch 2.10, x86
    engine).  The oracle's emulation (dq_linear: 7-bit per-call activations, int8 per-tensor weights) must reproduce the
    quantised LM's raw outputs and every fused decision."""
    g = load(golden_dir, f"model_{name}__{lm_name}_int8.npz")
    cfg = synth.model_cfg(name)
    m = O.OracleTransducer(synth.synth_state_dict(cfg, seed=0), cfg)
    lm = O.OracleLM(synth.synth_lm_state_dict(lm_name), quantized=True)
    lp, st = lm.step(5, None)
    np.testing.assert_allclose(lp, g["lm_logp_tok5"], atol=2e-4, rtol=0)
    lp2, _ = lm.step(7, st)
    np.testing.assert_allclose(lp2, g["lm_logp_tok5_7"], atol=2e-4, rtol=0)
    m.lm = lm
    pcm = synth.synth_pcm(n_streams, int(16000 * n_sec), seed=1234)
    g32 = load(golden_dir, f"model_{name}__{lm_name}.npz")
    differs = 0
    for s in range(n_streams):
        feats = O.features_offline(pcm[s])
        toks, neg_logp, score, iters = m.decode_greedy(feats)
        assert toks == list(g[f"off_tokens_{s}"])
        assert iters == list(g[f"off_iters_{s}"])
        differs += list(g[f"off_tokens_{s}"]) != list(g32[f"off_tokens_{s}"])
        fe, dec = O.StreamFrontend(), m.stream_decoder()
        counts = []
        for c in synth.stream_chunks(pcm[s], 1280, lead=1, tail=10):
            o = fe.push(c)
            if o is not None:
                counts.append(len(dec.step(o)))
        assert dec.y == list(g[f"st_tokens_{s}"])
        assert counts == list(g[f"st_counts_{s}"])
    print(f"{name}/{lm_name}: int8 LM changes the offline transcript of {differs} of {n_streams} utterances vs the fp32 LM")


def test_resample_restatement_properties():
    """kaldi LinearResample as used by torchaudio 0.6.0 transforms.Resample (un-vendored: parity unpinned).
    Output length rule, unit gain, tone preservation, and linearity."""
    assert O.resample_num_out(48000, 48000, 16000) == 16000
    assert O.resample_num_out(48001, 48000, 16000) == 16001        # one more tick inside [0, N/sr)
    assert O.resample_num_out(3 * 3840, 48000, 16000) == 3840
    assert O.resample_num_out(44100, 44100, 16000) == 16000
    first, w, iu, ou = O.resample_filters(44100, 16000)
    assert (iu, ou) == (441, 160) and w.shape[0] == 160
    assert np.abs(w.sum(1) - 1.0).max() < 2e-3                     # DC gain of every phase
    rng = np.random.default_rng(0)
    a = rng.standard_normal(9000).astype(np.float32)
    b = rng.standard_normal(9000).astype(np.float32)
    lhs = O.resample(a + 2 * b, 44100)
    rhs = O.resample(a, 44100) + 2 * O.resample(b, 44100)
    assert np.abs(lhs - rhs).max() < 1e-5                          # linear
    t = np.arange(48000) / 48000.0
    y = O.resample(np.sin(2 * np.pi * 1000 * t).astype(np.float32), 48000)
    ref = np.sin(2 * np.pi * 1000 * np.arange(len(y)) / 16000.0)
    assert np.abs(y[100:-100] - ref[100:-100]).max() < 2e-3


@pytest.mark.parametrize("name,n_sec,n_streams", [("tiny", 3.0, 3), ("tiny_lstm", 3.0, 2), ("cfg2", 4.0, 2),
                                                   ("cfg2_lstm", 2.0, 1)])
def test_torch_cpu_reference_path_matches_reference(golden_dir, name, n_sec, n_streams):
    """oracle/torch_cpu.py (the reference's torch-CPU execution path restated on the installed torch: the
    timed CPU neighbour of bench.py) reproduces the goldens the reference's own code produced: offline and
    streaming token ids, per-call counts, -log p."""
    from oracle import torch_cpu as TC
    g = load(golden_dir, f"model_{name}.npz")
    cfg = synth.model_cfg(name)
    m = TC.TorchTransducer(synth.synth_state_dict(cfg, seed=0), cfg)
    pcm = synth.synth_pcm(n_streams, int(16000 * n_sec), seed=1234)
    for s in range(n_streams):
        y, neg_logp = m.decode_greedy(TC.TorchFrontend().offline(pcm[s]))
        assert y == list(g[f"off_tokens_{s}"])
        assert abs(neg_logp - float(g[f"off_neglogp_{s}"])) < 1e-2
        fe, dec = TC.TorchFrontend(), m.stream_decoder()
        counts = []
        for c in synth.stream_chunks(pcm[s], 1280, lead=1, tail=10):
            o = fe.push(c)
            if o is not None:
                counts.append(len(dec.step(o)))
        assert dec.y == list(g[f"st_tokens_{s}"])
        assert counts == list(g[f"st_counts_{s}"])


@pytest.mark.parametrize("name", ["tiny", "tiny_lstm"])
def test_torch_cpu_batched_encoder_leg_equals_the_batch_1_path(name):
    """bench.py's cpu_baseline.best_effort leg (encoder + front-end batched over the streams, every host
    core) must decode exactly what the reference-faithful batch-1 path decodes."""
    from oracle import torch_cpu as TC
    cfg = synth.model_cfg(name)
    sd = synth.synth_state_dict(cfg, seed=0)
    rows = [synth.synth_pcm(1, 36 * 1280, seed=1234 + s)[0] for s in range(5)]
    _, one = TC.time_stream_path(sd, cfg, rows, 36)
    _, bat = TC.time_stream_path_batched(sd, cfg, rows, 36, threads=4)
    assert bat == one
    assert sum(len(t) for t in one) > 0


@pytest.mark.parametrize("name,lm_name", [("tiny", "tiny_lm"), ("tiny_lstm", "tiny_lm_untied")])
def test_beam_width_1_with_lm_is_the_greedy_loop_with_shallow_fusion(name, lm_name):
    """The builder-authored spec of LM fusion inside the beam (oracle _beam_frame) reduces, at W = 1, to the reference's greedy loop
    with shallow fusion (models.py:405-443 + lm.py:43-83), which the oracle pins to the reference's own goldens
    (test_lm_shallow_fusion_matches_reference): tokens and scores, offline and streaming."""
    cfg = synth.model_cfg(name)
    m = O.OracleTransducer(synth.synth_state_dict(cfg, seed=0), cfg)
    m.lm = O.OracleLM(synth.synth_lm_state_dict(lm_name))
    pcm = synth.synth_pcm(3, 16000 * 3, seed=1234)
    n_tok = 0
    for i in range(3):
        f = O.features_offline(pcm[i])
        g = m.decode_greedy(f)
        y, score, _ = m.decode_beam(f, 1)
        assert y == g[0] and abs(-g[1] - score) < 1e-4
        fe, d1, dg = O.StreamFrontend(), O.StreamBeamDecoder(m, 1), m.stream_decoder()
        for ch in synth.stream_chunks(pcm[i], 1280, lead=1, tail=4):
            o = fe.push(ch)
            if o is not None:
                d1.step(o)
                dg.step(o)
        assert d1.best()[0] == dg.y
        n_tok += len(dg.y)
    assert n_tok > 10


@pytest.mark.parametrize("name,n_streams", [("cfg2", 3), ("ref6", 1), ("cfg5", 1)])
def test_long_utterances_match_reference(golden_dir, name, n_streams):
    """SURVEY 8d's workload length (cfg2: 330 400 samples = 20.65 s, T' = 258; ref6 / cfg5: 10 s): hundreds of recurrent steps
    pinned to the reference's own decode, offline and streaming (goldens: oracle/make_golden.py `long`)."""
    g = load(golden_dir, f"model_{name}_long.npz")
    assert int(g["n_streams"]) == n_streams
    cfg = synth.model_cfg(name)
    m = O.OracleTransducer(synth.synth_state_dict(cfg, seed=0), cfg)
    pcm = synth.synth_pcm(n_streams, int(g["n_samples"]), seed=1234)
    for s in range(n_streams):
        toks, neg_logp, score, iters, _ = m.decode_greedy(O.features_offline(pcm[s]), return_logits=True)
        assert toks == list(g[f"off_tokens_{s}"])
        assert iters == list(g[f"off_iters_{s}"])
        assert abs(neg_logp - float(g[f"off_neglogp_{s}"])) < 2e-2
        fe, dec = O.StreamFrontend(), m.stream_decoder()
        counts = []
        for c in synth.stream_chunks(pcm[s], 1280, lead=1, tail=10):
            o = fe.push(c)
            if o is not None:
                counts.append(len(dec.step(o)))
        assert dec.y == list(g[f"st_tokens_{s}"])
        assert counts == list(g[f"st_counts_{s}"])


def servicer_golden(golden_dir, name="tiny"):
    g = load(golden_dir, f"servicer_{name}.npz")
    out = []
    for i in range(int(g["n"])):
        n = int(g[f"n_msgs_{i}"])
        out.append(([str(v) for v in g[f"msgs_{i}"][:n]], [int(v) for v in g[f"resets_{i}"]], str(g[f"unary_{i}"])))
    return out


def test_servicer_restatement_matches_the_reference_servicer(golden_dir):
    """oracle.servicer_stream (window, diff, "same diff twice", 4 s reset) against the message sequences the reference's OWN
    ASRServicer.TranscribeStream produced (api-server.py:82-135 imported by oracle/ref_fixture.py:ref_servicer), including a
    stream with a > 4 s silent stretch, one that starts with silence and one that crosses the threshold three times."""
    from libreasr_amd.lib.language import IdLanguage
    gold = servicer_golden(golden_dir)
    assert len(gold) == len(synth.SERVICER_STREAMS)
    cfg = synth.model_cfg("tiny")
    m = O.OracleTransducer(synth.synth_state_dict(cfg, seed=0), cfg)
    lang = IdLanguage()
    n_resets = 0
    for (seed, spec), (msgs, resets, unary) in zip(synth.SERVICER_STREAMS, gold):
        pcm = synth.servicer_pcm(seed, spec)
        got, got_resets = O.servicer_stream(m, pcm, lang.denumericalize)
        assert got == msgs, (seed, spec)
        assert got_resets == resets, (seed, spec)
        assert lang.denumericalize(m.decode_greedy(O.features_offline(pcm))[0]) == unary
        n_resets += len(resets)
    assert n_resets >= 6
    # the same streams' servicer on 100 ms frames (the generic-client / per-window form), from the reference's own servicer too
    g = load(golden_dir, "servicer_tiny.npz")
    pcm = synth.servicer_pcm(*synth.SERVICER_STREAMS[4])
    got, got_resets = O.servicer_stream(m, pcm, lang.denumericalize, chunk=1600, tail=8)
    assert got == [str(v) for v in g["msgs100_4"][:int(g["n_msgs100_4"])]] and got_resets == [int(v) for v in g["resets100_4"]]
    assert len(got_resets) == 2
