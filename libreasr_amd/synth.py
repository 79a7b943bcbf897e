"""
Synthetic model weights and 16 kHz PCM for benchmarks and parity tests.

No pretrained weights, tokenizer or norm file ship with the reference (they come from GitHub
releases, docs/docs.md:139-140), and there is no network here, so every measurement and parity
test runs on *seeded synthetic* weights laid out exactly like the reference's `state_dict`
(key names from libreasr/lib/models.py:190-234 and
libreasr/lib/layers/custom_rnn.py:113-126,265-269).

numpy's PCG64 generator is used (not torch's) so that the same arrays are produced in the
authoring container (where the goldens are made with the real reference) and on the GPU box.

Recipe: default-ish init for the RNN / Linear layers; non-trivial BatchNorm running
statistics and affine terms; non-zero learned initial states (they are zeros at init in the
reference, custom_rnn.py:75-82, which would hide bugs); the joint output layer is scaled so argmax
margins are realistic and the blank logit is biased so that most decisions are blank.
"""
import math

import numpy as np

# named model shapes (BASELINE.json configs)
CONFIGS = {
    # tiny shape for fast CPU-side tests (all dims multiples of 16)
    "tiny": dict(feat=1280, embed=32, vocab=64, hidden=64, joint=64, enc_layers=2,
                 pred_layers=2, pred_cell="NBRC", blank_bias=10.8, out_scale=8.0),
    # flatter joint output: small argmax margins, so LM shallow fusion (alpha = 0.1) overrides tokens
    "tiny_soft": dict(feat=1280, embed=32, vocab=64, hidden=64, joint=64, enc_layers=2,
                      pred_layers=2, pred_cell="NBRC", blank_bias=2.6, out_scale=2.0),
    "tiny_lstm": dict(feat=1280, embed=32, vocab=64, hidden=64, joint=64, enc_layers=2,
                      pred_layers=2, pred_cell="LSTM", blank_bias=10.4, out_scale=8.0),
    # configs 2/3/4: 4x1024 LSTM encoder, reference predictor (2x NBRC), J=1024, V=2048
    "cfg2": dict(feat=1280, embed=512, vocab=2048, hidden=1024, joint=1024, enc_layers=4,
                 pred_layers=2, pred_cell="NBRC", blank_bias=14.9, out_scale=8.0),
    # north-star variant with an LSTM prediction network
    "cfg2_lstm": dict(feat=1280, embed=512, vocab=2048, hidden=1024, joint=1024, enc_layers=4,
                      pred_layers=2, pred_cell="LSTM", blank_bias=14.2, out_scale=8.0),
    # reference default 6-2-1024 (config/testing.yaml:202-229)
    "ref6": dict(feat=1280, embed=512, vocab=2048, hidden=1024, joint=1024, enc_layers=6,
                 pred_layers=2, pred_cell="NBRC", blank_bias=15.0, out_scale=8.0),
    # config 5: 8x1536 encoder, 2-layer LSTM predictor
    "cfg5": dict(feat=1280, embed=512, vocab=2048, hidden=1536, joint=1536, enc_layers=8,
                 pred_layers=2, pred_cell="LSTM", blank_bias=17.4, out_scale=8.0),
}


def model_cfg(name_or_cfg):
    if isinstance(name_or_cfg, str):
        return dict(CONFIGS[name_or_cfg])
    return dict(name_or_cfg)


def blank_row(cfg):
    return 0  # Transducer(blank=0) (models.py:203)


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _bn(rng, sd, prefix, n):
    sd[prefix + ".weight"] = rng.uniform(0.8, 1.2, n).astype(np.float32)
    sd[prefix + ".bias"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
    # running stats near the real statistics of an LSTM/GRU output (|h| < 1, var ~ 0.03) so the
    # normalised activations are O(1) and the stack stays input-dependent
    sd[prefix + ".running_mean"] = (0.02 * rng.standard_normal(n)).astype(np.float32)
    sd[prefix + ".running_var"] = rng.uniform(0.02, 0.06, n).astype(np.float32)


def _lstm(rng, sd, prefix, i_sz, h_sz):
    k = 1.0 / math.sqrt(h_sz)
    sd[prefix + ".weight_ih_l0"] = _uniform(rng, (4 * h_sz, i_sz), k)
    sd[prefix + ".weight_hh_l0"] = _uniform(rng, (4 * h_sz, h_sz), k)
    sd[prefix + ".bias_ih_l0"] = _uniform(rng, (4 * h_sz,), k)
    sd[prefix + ".bias_hh_l0"] = _uniform(rng, (4 * h_sz,), k)


def _nbrc(rng, sd, prefix, i_sz, h_sz):
    # haste layout: kernel [I,3H], recurrent_kernel [H,3H], gate order z,r,g (haste/nbrc.py:112-122)
    xav = math.sqrt(6.0 / (i_sz + h_sz))
    sd[prefix + ".kernel"] = _uniform(rng, (i_sz, 3 * h_sz), xav)
    sd[prefix + ".recurrent_kernel"] = (
        rng.standard_normal((h_sz, 3 * h_sz)) / math.sqrt(h_sz)).astype(np.float32)
    sd[prefix + ".bias"] = (0.05 * rng.standard_normal(3 * h_sz)).astype(np.float32)
    sd[prefix + ".recurrent_bias"] = (0.05 * rng.standard_normal(3 * h_sz)).astype(np.float32)


def synth_state_dict(cfg, seed=0):
    """Return {reference state_dict key: float32 ndarray} for the model shape `cfg`."""
    cfg = model_cfg(cfg)
    rng = np.random.Generator(np.random.PCG64(seed))
    F, E, V, H, J = cfg["feat"], cfg["embed"], cfg["vocab"], cfg["hidden"], cfg["joint"]
    sd = {}
    sd["encoder.input_norm.weight"] = rng.uniform(0.8, 1.2, F).astype(np.float32)
    sd["encoder.input_norm.bias"] = (0.1 * rng.standard_normal(F)).astype(np.float32)
    for i in range(cfg["enc_layers"]):
        sd[f"encoder.rnn_stack.hs.{i}"] = (0.1 * rng.standard_normal((2, 1, 1, H))).astype(np.float32)
        _bn(rng, sd, f"encoder.rnn_stack.bns.{i}", H)
        _lstm(rng, sd, f"encoder.rnn_stack.rnns.{i}", F if i == 0 else H, H)
    emb = rng.standard_normal((V, E)).astype(np.float32)
    emb[0] = 0.0  # padding_idx = blank = 0 (models.py:159)
    sd["predictor.embed.weight"] = emb
    if E != H:  # models.py:160-163
        k = 1.0 / math.sqrt(E)
        sd["predictor.ffn.weight"] = _uniform(rng, (H, E), k)
        sd["predictor.ffn.bias"] = _uniform(rng, (H,), k)
    n_state = 2 if cfg["pred_cell"] == "LSTM" else 1
    for i in range(cfg["pred_layers"]):
        sd[f"predictor.rnn_stack.hs.{i}"] = (
            0.1 * rng.standard_normal((n_state, 1, 1, H))).astype(np.float32)
        _bn(rng, sd, f"predictor.rnn_stack.bns.{i}", H)
        if cfg["pred_cell"] == "LSTM":
            _lstm(rng, sd, f"predictor.rnn_stack.rnns.{i}", H, H)
        else:
            _nbrc(rng, sd, f"predictor.rnn_stack.rnns.{i}", H, H)
    k = 1.0 / math.sqrt(2 * H)
    sd["joint.joint.0.weight"] = _uniform(rng, (J, 2 * H), k) * np.float32(cfg.get("j0_scale", 6.0))
    sd["joint.joint.0.bias"] = _uniform(rng, (J,), k)
    k = 1.0 / math.sqrt(J)
    w2 = _uniform(rng, (V, J), k) * np.float32(cfg.get("out_scale", 8.0))
    b = _uniform(rng, (V,), k)
    # blank: constant logit (zero weight row, fixed bias).  With random weights a *learned-looking*
    # blank row makes the decoder bistable (a stream either never emits or always runs into
    # max_iters); a constant blank logit set near the 70-80 % quantile of the best non-blank
    # logit gives ~0.2-0.4 tokens per frame with short bursts, like a trained model.
    w2[blank_row(cfg)] = 0.0
    b[blank_row(cfg)] = np.float32(cfg.get("blank_bias", 0.0))
    sd["joint.joint.2.weight"] = w2
    sd["joint.joint.2.bias"] = b
    return sd


# language models for shallow fusion (lm.py:20-40): Embedding -> LSTM stack -> Linear (weights tied to
# the embedding when embed == hidden, lm.py:27-29) -> log_softmax.  "lm768" is the shipped shape
# (config/testing.yaml:308-313: 4 x 768).
LM_CONFIGS = {
    "tiny_lm": dict(vocab=64, embed=32, hidden=32, layers=2, emb_scale=4.0),
    "tiny_lm_untied": dict(vocab=64, embed=16, hidden=32, layers=2, emb_scale=1.0, out_scale=40.0),
    "lm768": dict(vocab=2048, embed=768, hidden=768, layers=4, emb_scale=1.0),
}


def lm_cfg(name_or_cfg):
    return dict(LM_CONFIGS[name_or_cfg]) if isinstance(name_or_cfg, str) else dict(name_or_cfg)


def synth_lm_state_dict(cfg, seed=100):
    """{lm.py LM state_dict key: float32 ndarray}.  The output layer is scaled so that the log-softmax
    is peaked: after standardisation (lm.py:50-52) the fuser only sees its shape, and with alpha = 0.1
    only a peaked LM ever overrides the joint's choice (the tests need overrides to happen)."""
    cfg = lm_cfg(cfg)
    rng = np.random.Generator(np.random.PCG64(seed))
    V, E, H, L = cfg["vocab"], cfg["embed"], cfg["hidden"], cfg["layers"]
    sd = {}
    emb = (rng.standard_normal((V, E)) * cfg.get("emb_scale", 1.0)).astype(np.float32)
    emb[0] = 0.0                                              # padding_idx=0 (lm.py:23)
    sd["embed.weight"] = emb
    k = 1.0 / math.sqrt(H)
    for l in range(L):
        i_sz = E if l == 0 else H
        sd[f"rnn.weight_ih_l{l}"] = _uniform(rng, (4 * H, i_sz), k)
        sd[f"rnn.weight_hh_l{l}"] = _uniform(rng, (4 * H, H), k)
        sd[f"rnn.bias_ih_l{l}"] = _uniform(rng, (4 * H,), k)
        sd[f"rnn.bias_hh_l{l}"] = _uniform(rng, (4 * H,), k)
    sd["linear.weight"] = emb if E == H else _uniform(rng, (V, H), cfg.get("out_scale", 4.0) * k)
    sd["linear.bias"] = _uniform(rng, (V,), k)
    return sd


def synth_pcm(n_streams, n_samples, seed=1234, sr=16000):
    """[n_streams, n_samples] float32 in [-1,1].  Speech-like synthetic audio: a sequence of
    60-260 ms "syllables", each a sum of three sinusoids at random formant-like frequencies with a
    raised-cosine envelope and random level (some are silent), over 0.02*N(0,1) noise.  The
    spectral shape changes every few stacked frames, so the RNN-T emission pattern is
    input-dependent (a slow chirp gives an almost constant encoder output)."""
    out = np.empty((n_streams, n_samples), dtype=np.float32)
    for s in range(n_streams):
        rng = np.random.Generator(np.random.PCG64([seed, s]))
        x = 0.02 * rng.standard_normal(n_samples)
        pos = 0
        while pos < n_samples:
            n = int(rng.integers(int(0.06 * sr), int(0.26 * sr)))
            n = min(n, n_samples - pos)
            level = 0.0 if rng.random() < 0.2 else rng.uniform(0.05, 0.3)
            t = np.arange(n, dtype=np.float64) / sr
            seg = np.zeros(n)
            for lo, hi in ((200.0, 900.0), (900.0, 2500.0), (2500.0, 5000.0)):
                f = rng.uniform(lo, hi)
                seg += rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
            env = 0.5 - 0.5 * np.cos(2 * np.pi * (np.arange(n) + 0.5) / n)
            x[pos:pos + n] += level * env * seg / 3.0 * 2.0
            pos += n
        out[s] = np.clip(x, -1.0, 1.0).astype(np.float32)
    return out


def stream_chunks(pcm_row, chunk=1280, lead=1, tail=10):
    """Cut one utterance into client chunks exactly as api-client.py:32-47 does: one leading zero
    chunk, the signal in `chunk`-sample slices (trailing partial slice dropped, as `len//slice_sz`),
    then `tail` zero chunks."""
    n = len(pcm_row) // chunk
    z = np.zeros(chunk, dtype=np.float32)
    out = [z.copy() for _ in range(lead)]
    out += [np.ascontiguousarray(pcm_row[i * chunk:(i + 1) * chunk]) for i in range(n)]
    out += [z.copy() for _ in range(tail)]
    return out


# streams of the servicer goldens (tests/golden/servicer_tiny.npz, produced by the reference's own ASRServicer.TranscribeStream:
# oracle/make_golden.py golden_servicer): (seed, spec); spec = seconds of speech-like PCM or a list of (kind, seconds) segments
SERVICER_STREAMS = [
    (1234, 3.0), (1235, 3.0), (1236, 3.0),
    (77, 7.0),                                                      # runs past the 4 s threshold while emitting
    (99, [("speech", 2.5), ("silence", 5.5), ("speech", 3.0)]),     # > 4 s silent stretch: reset inside the silence
    (5, 12.5),                                                      # crosses the threshold more than once
    (6, [("silence", 4.6), ("speech", 2.0)]),                       # silence first: reset before any token
]


def servicer_pcm(seed, spec, sr=16000):
    if not isinstance(spec, list):
        return synth_pcm(1, int(sr * spec), seed=seed)[0]
    parts = []
    for i, (kind, sec) in enumerate(spec):
        n = int(sr * sec)
        parts.append(synth_pcm(1, n, seed=seed + 100 * i)[0] if kind == "speech" else np.zeros(n, np.float32))
    return np.concatenate(parts).astype(np.float32)
