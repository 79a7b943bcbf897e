"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  A CPU (numpy, float32) restatement of the reference's
streaming RNN-T inference path.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this; the product (libreasr_amd/) never does.

Parity status: PINNED for everything whose source is in /root/reference -- tests/test_oracle.py
checks this file against golden vectors produced by the reference's own code
(oracle/make_golden.py, via oracle/ref_fixture.py) -- and "parity unpinned" only for the one
un-vendored numeric dependency, torchaudio==0.6.0 `MelSpectrogram` (restated from its published
algorithm; call site transforms.py:290-296,310), and for beam search, which the reference does not
have (models.py:8 imports PriorityQueue and never uses it).

Each function cites the reference file:line it follows (paths relative to /root/reference).
"""
import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- front-end
def hann_periodic(n):
    # torch.hann_window(n) (periodic=True), used by torchaudio MelSpectrogram 0.6.0
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)).astype(F32)


def htk_filterbank(n_freqs=513, f_min=0.0, f_max=8000.0, n_mels=128, sr=16000):
    # torchaudio 0.6.0 functional.create_fb_matrix: HTK mel scale, triangular, no area norm.
    all_freqs = np.linspace(0, sr // 2, n_freqs)
    m_min = 2595.0 * np.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * np.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(F32)  # [n_freqs, n_mels]


def logmel(pcm, n_fft=1024, win=400, hop=160, n_mels=128, sr=16000):
    """TransformTime.encodes (transforms.py:306-323) with melkwargs {n_fft:1024,n_mels:128}
    (config/testing.yaml:133-135), win 0.025 s, hop 0.01 s (transforms.py:288-289), deltas 0:
    MelSpectrogram (center=True, reflect pad n_fft/2, periodic Hann(win) zero-padded centred to
    n_fft, one-sided, power 2) then log(x + 1e-6) (transforms.py:311-313), permuted to [T, n_mels].
    pcm: [N] float32 -> [T, n_mels], T = 1 + N // hop."""
    pcm = np.asarray(pcm, dtype=F32)
    n = pcm.shape[0]
    pad = n_fft // 2
    xp = np.pad(pcm, (pad, pad), mode="reflect")
    w = np.zeros(n_fft, dtype=F32)
    off = (n_fft - win) // 2
    w[off:off + win] = hann_periodic(win)
    T = 1 + n // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(T)[:, None]
    frames = xp[idx] * w[None, :]
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)
    power = (spec.real ** 2 + spec.imag ** 2).astype(F32)
    mel = power @ htk_filterbank(n_fft // 2 + 1, 0.0, sr / 2.0, n_mels, sr)
    return np.log(mel + F32(1e-6)).astype(F32)


def resample_num_out(n_in, sr_in, sr_out):
    """kaldi LinearResample::GetNumOutputSamples(flush=true) as in torchaudio 0.6.0
    compliance/kaldi.py::_get_num_LR_output_samples: output samples with time in [0, n_in / sr_in)."""
    import math
    tick = sr_in * sr_out // math.gcd(sr_in, sr_out)
    ticks_in, ticks_out = tick // sr_in, tick // sr_out
    length = n_in * ticks_in
    if length <= 0:
        return 0
    last = length // ticks_out
    if last * ticks_out == length:
        last -= 1
    return last + 1


def resample_filters(sr_in, sr_out, lowpass_filter_width=6):
    """_get_LR_indices_and_weights: one windowed-sinc filter per output phase (sr_out / gcd of them).
    cutoff = 0.99 * min(sr)/2; Hann window of half-width lowpass_filter_width / (2 cutoff); float32
    arithmetic like the torch code.  Returns (first_index[phase] int, weights[phase, taps] f32, in_unit, out_unit)."""
    import math
    base = math.gcd(sr_in, sr_out)
    in_unit, out_unit = sr_in // base, sr_out // base
    cutoff = 0.99 * 0.5 * min(sr_in, sr_out)
    width = lowpass_filter_width / (2.0 * cutoff)
    out_t = (np.arange(out_unit, dtype=F32) / F32(sr_out)).astype(F32)
    min_idx = np.ceil((out_t - F32(width)) * F32(sr_in)).astype(F32)
    max_idx = np.floor((out_t + F32(width)) * F32(sr_in)).astype(F32)
    taps = int((max_idx - min_idx + 1).max())
    j = np.arange(taps, dtype=F32)[None, :]
    idx = min_idx[:, None] + j
    dt = (idx / F32(sr_in) - out_t[:, None]).astype(F32)
    w = np.zeros_like(dt)
    inside = np.abs(dt) < F32(width)
    w[inside] = (0.5 * (1 + np.cos(F32(2 * math.pi * cutoff / lowpass_filter_width) * dt[inside]))).astype(F32)
    nz = dt != 0
    w[nz] *= (np.sin(F32(2 * math.pi * cutoff) * dt[nz]) / (F32(math.pi) * dt[nz])).astype(F32)
    w[~nz] *= F32(2 * cutoff)
    w = (w / F32(sr_in)).astype(F32)
    return min_idx.astype(np.int64), w, in_unit, out_unit


def resample(pcm, sr_in, sr_out=16000):
    """torchaudio.transforms.Resample(sr_in, sr_out) of torchaudio 0.6.0 = kaldi.resample_waveform
    (Resample.encodes, transforms.py:141-144; un-vendored -> PARITY UNPINNED, restated from the published
    algorithm): out[n] = sum_j w[n % U][j] * x[first[n % U] + (n // U) * in_unit + j], zero outside x."""
    pcm = np.asarray(pcm, dtype=F32)
    first, w, in_unit, out_unit = resample_filters(sr_in, sr_out)
    n_out = resample_num_out(len(pcm), sr_in, sr_out)
    out = np.zeros(n_out, dtype=F32)
    taps = w.shape[1]
    xp = np.concatenate([np.zeros(taps + int(max(0, -first.min())), F32), pcm, np.zeros(taps + in_unit, F32)])
    off = taps + int(max(0, -first.min()))
    for ph in range(out_unit):
        n = np.arange(ph, n_out, out_unit)
        if n.size == 0:
            continue
        start = first[ph] + (n // out_unit) * in_unit + off
        ok = start + taps <= len(xp)
        seg = np.zeros((n.size, taps), F32)
        idx = start[:, None] + np.arange(taps)[None, :]
        idx = np.minimum(idx, len(xp) - 1)
        seg = xp[idx] * (start[:, None] + np.arange(taps)[None, :] < len(xp))
        out[n] = (seg.astype(F32) * w[ph][None, :]).sum(axis=1, dtype=F32)
    return out


def stream_postprocess(spec, n_stack=10):
    """StreamPostprocess.encodes (transforms.py:335-342): l = T//3; keep frames [l+1, l+1+n_stack)."""
    a = spec.shape[0] // 3 + 1
    return spec[a:][:n_stack]


def stack_downsample(spec, n_stack=10, downsample=8):
    """StackDownsample.encodes (transforms.py:436-441): unfold(-2, n_stack, downsample) then view:
    feat[t', m*n_stack + k] = spec[downsample*t' + k, m]  (mel-major, frame-minor)."""
    T, M = spec.shape
    if T < n_stack:
        return np.zeros((0, M * n_stack), dtype=F32)
    Tp = (T - n_stack) // downsample + 1
    idx = downsample * np.arange(Tp)[:, None] + np.arange(n_stack)[None, :]   # [Tp, k]
    uf = spec[idx]                       # [Tp, k, M]
    return np.ascontiguousarray(uf.transpose(0, 2, 1)).reshape(Tp, M * n_stack).astype(F32)


def features_offline(pcm, n_stack=10, downsample=8, **mel):
    """x_tfm pipeline (config/testing.yaml:341-356; api-server.py:74-75): [N] -> [T', 1280].
    **mel: non-default TransformTime settings (win, hop, n_mels) for shape-generality tests."""
    return stack_downsample(logmel(pcm, **mel), n_stack, downsample)


class StreamFrontend:
    """Per-stream restatement of the servicer's 3-chunk window (api-server.py:83-115,
    BUFFER_N_FRAMES=3 :26) + x_tfm_stream (testing.yaml:358-374) incl. Buffer(n_buffer)
    (transforms.py:455-471).  push(chunk) -> None or [n_buffer*T', 1280]."""

    def __init__(self, n_stack=10, downsample=8, n_buffer=2, n_window=3, sr=16000, **mel):
        self.n_stack, self.downsample, self.n_buffer, self.n_window = n_stack, downsample, n_buffer, n_window
        self.sr = sr                             # client sample rate: Resample (order 2, transforms.py:135-144) of the WINDOW
        self.mel = mel
        self.frames, self.saved = [], []

    def push(self, chunk):
        self.frames.append(np.asarray(chunk, dtype=F32))
        self.called = False                      # did this push reach the transform pipeline?
        if len(self.frames) != self.n_window:
            return None
        self.called = True
        aud = np.concatenate(self.frames)
        del self.frames[0]
        if self.sr != 16000:
            aud = resample(aud, self.sr)
        spec = stream_postprocess(logmel(aud, **self.mel), self.n_stack)
        st = stack_downsample(spec, self.n_stack, self.downsample)
        self.saved.append(st)
        if len(self.saved) == self.n_buffer:
            cat = np.concatenate(self.saved, axis=0)
            self.saved = []
            return cat
        return None


# ----------------------------------------------------------------------------- model pieces
def bf16_round(x):
    """f32 -> nearest-even bf16 -> f32 (what the engine stores with dtype=bf16)."""
    u = np.ascontiguousarray(x, dtype=F32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(F32)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(F32)))).astype(F32)


def layer_norm(x, w, b, eps=1e-5):
    # nn.LayerNorm(feature_sz) (models.py:84,107): biased variance over the last dim
    x = x.astype(F32)
    mu = x.mean(-1, keepdims=True, dtype=F32)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=F32)
    return ((x - mu) / np.sqrt(var + F32(eps)) * w + b).astype(F32)


def bn_eval(x, p, eps=1e-5):
    # BatchNorm1d in eval mode over the channel (= hidden) dim (custom_rnn.py:122-126,210-213)
    return ((x - p["running_mean"]) / np.sqrt(p["running_var"] + F32(eps)) * p["weight"] + p["bias"]).astype(F32)


def lstm_step(x, h, c, p):
    """torch.nn.LSTM cell, gate order i,f,g,o (custom_rnn.py:36-42,172; in-tree statement
    haste/lstm.py:34-68 with the weight map haste/lstm.py:181-187).  x [B,I], h,c [B,H]."""
    g = x @ p["weight_ih_l0"].T + p["bias_ih_l0"] + h @ p["weight_hh_l0"].T + p["bias_hh_l0"]
    H = h.shape[-1]
    i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
    c2 = sigmoid(f) * c + sigmoid(i) * np.tanh(gg)
    h2 = sigmoid(o) * np.tanh(c2)
    return h2.astype(F32), c2.astype(F32)


def nbrc_step(x, h, p):
    """NBRCScript (haste/nbrc.py:30-64): Wx = x K + b; Rh = h R + rb; layout z,r,g;
    g = tanh(Wx_g + r * Rh_g); h' = z h + (1 - z) g.  zoneout = 0 => branch dead (:57-61)."""
    H = h.shape[-1]
    Wx = x @ p["kernel"] + p["bias"]
    Rh = h @ p["recurrent_kernel"] + p["recurrent_bias"]
    z = sigmoid(Wx[:, :H] + Rh[:, :H])
    r = sigmoid(Wx[:, H:2 * H] + Rh[:, H:2 * H])
    g = np.tanh(Wx[:, 2 * H:] + r * Rh[:, 2 * H:])
    return (z * h + (F32(1.0) - z) * g).astype(F32)


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: np.asarray(v, dtype=F32) for k, v in sd.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------- LM shallow fusion
LM_ALPHA, LM_THETA, LM_MIN_VAL = F32(0.1), F32(1.0), F32(-10.0)       # lm.py:13-15


def standardize(t, eps=1e-5):
    """utils.py:162-164: t -= t.mean(); t /= (t.std() + eps)   (torch .std(): unbiased)."""
    t = (t - t.mean(dtype=F32)).astype(F32)
    return (t / (t.std(ddof=1, dtype=F32) + F32(eps))).astype(F32)


def dq_choose_qparams(mn, mx, qmin=0, qmax=127):
    """fbgemm ChooseQuantizationParams as torch's dynamic quantisation calls it (x86 / fbgemm engines, reduce_range:
    7-bit activations): min / max widened to contain 0, scale in double -> float32, zero point nudged to an integer."""
    mn, mx = min(float(mn), 0.0), max(float(mx), 0.0)
    scale = (np.float64(mx) - np.float64(mn)) / (qmax - qmin)
    if scale == 0.0 or np.isinf(1.0 / scale):
        scale = 0.1
    zp_min, zp_max = qmin - mn / scale, qmax - mx / scale
    izp = zp_min if abs(qmin) + abs(mn / scale) < abs(qmax) + abs(mx / scale) else zp_max
    zp = qmin if izp < qmin else qmax if izp > qmax else int(np.rint(izp))
    return np.float32(scale), int(zp)


def dq_weight(w):
    """quantize_dynamic's weight side (default_dynamic_qconfig: MinMaxObserver qint8 per_tensor_symmetric):
    scale = max|w| / 127.5, zero point 0, q = clamp(rint(w * (1 / scale)), -128, 127)."""
    w = np.asarray(w, F32)
    sw = np.float32(max(float(np.abs(w).max()), 0.0) / 127.5)
    if sw < np.finfo(np.float32).eps:
        sw = np.float32(np.finfo(np.float32).eps)      # MinMaxObserver clamps the scale to eps
    q = np.clip(np.rint(w * (np.float32(1.0) / sw)), -128, 127).astype(np.int32)
    return q, sw


def dq_linear(x, qw, sw, b):
    """quantized::linear_dynamic(x, packed, reduce_range=True) for one row: per-call activation quantisation,
    int32 accumulation, dequantisation, float bias."""
    x = np.asarray(x, F32).reshape(-1)
    s, zp = dq_choose_qparams(x.min(), x.max())
    qx = np.clip(np.rint(x * (np.float32(1.0) / s)) + zp, 0, 127).astype(np.int64)
    acc = (qx - zp) @ qw.astype(np.int64).T
    return (acc.astype(F32) * np.float32(s * sw) + np.asarray(b, F32)).astype(F32)


class OracleLM:
    """LM.forward (lm.py:20-40) for one token per call: Embedding -> nn.LSTM stack (zero initial state) ->
    dropout(eval) -> Linear -> log_softmax.  fp32, or with `quantized=True` what load_lm serves (lm.py:97
    maybe_quantize -> utils.py:197-210 torch.quantization.quantize_dynamic({LSTM, Linear}, qint8)): every LSTM gate
    matmul and the output layer become dynamically quantised int8 GEMVs (dq_linear); the Embedding stays fp32.  fbgemm is
    un-vendored: the int8 numerics are those of the INSTALLED torch (2.10, x86 engine), checked against it in
    tests/test_oracle.py (torch 1.6's differ in details: parity unpinned against the pinned version)."""

    def __init__(self, sd, quantized=False):
        self.quantized = quantized
        self.embed = np.asarray(sd["embed.weight"], F32)
        self.layers = []
        l = 0
        while f"rnn.weight_ih_l{l}" in sd:
            self.layers.append({k: np.asarray(sd[f"rnn.{k[:-3]}_l{l}"], F32)
                                for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")})
            l += 1
        self.w = np.asarray(sd["linear.weight"] if "linear.weight" in sd else sd["embed.weight"], F32)
        self.b = np.asarray(sd["linear.bias"], F32)
        self.H = self.layers[0]["weight_hh_l0"].shape[1]
        if quantized:
            self.q = [(dq_weight(p["weight_ih_l0"]), dq_weight(p["weight_hh_l0"])) for p in self.layers]
            self.qo = dq_weight(self.w)

    def _qstep(self, x, h, c, l):
        p, ((qi, si), (qh, sh)) = self.layers[l], self.q[l]
        g = dq_linear(x[0], qi, si, p["bias_ih_l0"]) + dq_linear(h[0], qh, sh, p["bias_hh_l0"])
        H = self.H
        i, f, gg, o = sigmoid(g[:H]), sigmoid(g[H:2 * H]), np.tanh(g[2 * H:3 * H]).astype(F32), sigmoid(g[3 * H:])
        c2 = (f * c[0] + i * gg).astype(F32)
        h2 = (o * np.tanh(c2)).astype(F32)
        return h2[None], c2[None]

    def step(self, tok, state):
        x = self.embed[np.asarray([tok])]
        if state is None:
            state = [(np.zeros((1, self.H), F32), np.zeros((1, self.H), F32)) for _ in self.layers]
        new = []
        for l, (p, (h, c)) in enumerate(zip(self.layers, state)):
            h, c = self._qstep(x, h, c, l) if self.quantized else lstm_step(x, h, c, p)
            new.append((h, c))
            x = h
        z = dq_linear(x[0], self.qo[0], self.qo[1], self.b) if self.quantized else (x @ self.w.T + self.b).astype(F32)[0]
        m = z.max()
        lp = ((z - m) - np.log(np.exp(z - m).sum(dtype=F32))).astype(F32)
        return lp, new


class LMFuser:
    """lm.py:43-83.  advance(): LM step on the emitted token, standardise, [0] = MIN_VAL.
    fuse(): once the LM has logits, standardise the joint log-softmax, [0] = MIN_VAL, and re-pick the
    token as argmax(alpha * lm + theta * joint).  Only ever called for a non-blank decision
    (models.py:427-431): the blank/non-blank decision itself is never changed."""

    def __init__(self, lm):
        self.lm = lm
        self.reset()

    def reset(self):
        self.lm_logits, self.lm_state = None, None

    def advance(self, tok):
        if self.lm is not None:
            lp, self.lm_state = self.lm.step(tok, self.lm_state)
            lp = standardize(lp)
            lp[0] = LM_MIN_VAL
            self.lm_logits = lp

    def fuse(self, joint_lp, pred):
        if self.lm is None or self.lm_logits is None:
            return pred
        jo = standardize(joint_lp)
        jo[0] = LM_MIN_VAL
        fused = (LM_ALPHA * self.lm_logits).astype(F32) + (LM_THETA * jo).astype(F32)
        return int(fused.argmax())


class OracleTransducer:
    """Restatement of Encoder / Predictor / Joint / Transducer.decode_greedy / transcribe_stream
    (models.py:68-187, 369-455, 457-577) + CustomRNN (custom_rnn.py:140-232)."""

    def __init__(self, sd, cfg, blank=0, bos=2, operand="f32"):
        """operand="bf16" emulates the engine's dtype=bf16 contract (NOT a reference mode: the
        reference is fp32 only): GEMM weights and GEMM-input activations (LayerNorm output, h state,
        BN(h), joint activation) are rounded to bf16; accumulation, cell state, biases, the
        predictor's embed->ffn->layer-0 input projection, pe/pp and logits stay f32."""
        assert operand in ("f32", "bf16")
        self.q = bf16_round if operand == "bf16" else (lambda a: a)
        self.cfg = cfg
        self.blank, self.bos = blank, bos  # models.py:203,225,227
        self.Le, self.Lp, self.H = cfg["enc_layers"], cfg["pred_layers"], cfg["hidden"]
        self.pred_lstm = cfg["pred_cell"] == "LSTM"
        self.ln_w = np.asarray(sd["encoder.input_norm.weight"], F32)
        self.ln_b = np.asarray(sd["encoder.input_norm.bias"], F32)
        self.enc = [dict(rnn=_sub(sd, f"encoder.rnn_stack.rnns.{i}."),
                         bn=_sub(sd, f"encoder.rnn_stack.bns.{i}."),
                         hs=np.asarray(sd[f"encoder.rnn_stack.hs.{i}"], F32)) for i in range(self.Le)]
        self.pred = [dict(rnn=_sub(sd, f"predictor.rnn_stack.rnns.{i}."),
                          bn=_sub(sd, f"predictor.rnn_stack.bns.{i}."),
                          hs=np.asarray(sd[f"predictor.rnn_stack.hs.{i}"], F32)) for i in range(self.Lp)]
        self.embed = np.asarray(sd["predictor.embed.weight"], F32)
        self.ffn_w = np.asarray(sd["predictor.ffn.weight"], F32) if "predictor.ffn.weight" in sd else None
        self.ffn_b = np.asarray(sd["predictor.ffn.bias"], F32) if "predictor.ffn.bias" in sd else None
        self.j0_w = np.asarray(sd["joint.joint.0.weight"], F32)
        self.j0_b = np.asarray(sd["joint.joint.0.bias"], F32)
        self.j2_w = np.asarray(sd["joint.joint.2.weight"], F32)
        self.j2_b = np.asarray(sd["joint.joint.2.bias"], F32)
        self.lm = None                                   # OracleLM or None (config.py:143-157 attaches it)
        if operand == "bf16":
            for i, l in enumerate(self.enc + self.pred):
                first_pred = i == self.Le            # predictor layer 0: input side stays f32 (table)
                for k in ("weight_ih_l0", "kernel"):
                    if k in l["rnn"] and not first_pred:
                        l["rnn"][k] = bf16_round(l["rnn"][k])
                for k in ("weight_hh_l0", "recurrent_kernel"):
                    if k in l["rnn"]:
                        l["rnn"][k] = bf16_round(l["rnn"][k])
            self.j0_w, self.j2_w = bf16_round(self.j0_w), bf16_round(self.j2_w)

    # -- encoder -------------------------------------------------------------------------
    def enc_init_state(self, B):
        # learned initial state broadcast over the batch (custom_rnn.py:152-158)
        return [(self.q(np.repeat(l["hs"][0, 0], B, 0).copy()), np.repeat(l["hs"][1, 0], B, 0).copy())
                for l in self.enc]

    def encoder(self, x, state=None):
        """Encoder.forward (models.py:105-113).  x [B,T,F] -> ([B,T,H], state)."""
        B, T, _ = x.shape
        x = self.q(layer_norm(x.reshape(B, T, -1), self.ln_w, self.ln_b))   # models.py:107
        if state is None:
            state = self.enc_init_state(B)
        new_state = []
        for l, (h, c) in zip(self.enc, state):
            ys = np.empty((B, T, self.H), F32)
            for t in range(T):
                h, c = lstm_step(x[:, t], h, c, l["rnn"])
                ys[:, t] = h
                h = self.q(h)
            x = self.q(bn_eval(ys, l["bn"]))                         # custom_rnn.py:210-213
            new_state.append((h, c))
        return x, new_state                                          # dropout(eval)=id, linear=id

    # -- predictor -----------------------------------------------------------------------
    def pred_init_state(self, B=1):
        if self.pred_lstm:
            return [(self.q(np.repeat(l["hs"][0, 0], B, 0).copy()), np.repeat(l["hs"][1, 0], B, 0).copy())
                    for l in self.pred]
        return [self.q(np.repeat(l["hs"][0, 0], B, 0).copy()) for l in self.pred]

    def predictor(self, tok, state=None):
        """Predictor.forward (models.py:181-187) for one token per row.  tok [B] -> ([B,H], state)."""
        tok = np.asarray(tok).reshape(-1)
        x = self.embed[tok]                                          # models.py:182
        if self.ffn_w is not None:
            x = (x @ self.ffn_w.T + self.ffn_b).astype(F32)          # models.py:183
        if state is None:
            state = self.pred_init_state(len(tok))
        new_state = []
        for l, s in zip(self.pred, state):
            if self.pred_lstm:
                h, c = lstm_step(x, s[0], s[1], l["rnn"])
                new_state.append((self.q(h), c))
            else:
                h = nbrc_step(x, s, l["rnn"])
                new_state.append(self.q(h))
            x = self.q(bn_eval(h, l["bn"]))
        return x, new_state

    # -- joint ---------------------------------------------------------------------------
    def joint_logp(self, h_pred, h_enc):
        """Joint.forward 'concat' (models.py:132-140; cat order pred, enc :136) + log_softmax
        (models.py:418).  h_pred, h_enc [B,H] -> log-probs [B,V]."""
        x = self.q(np.concatenate([h_pred, h_enc], axis=-1))
        a = self.q(np.tanh(x @ self.j0_w.T + self.j0_b).astype(F32))
        z = (a @ self.j2_w.T + self.j2_b).astype(F32)
        m = z.max(-1, keepdims=True)
        lse = m + np.log(np.exp(z - m).sum(-1, keepdims=True, dtype=F32))
        return (z - lse).astype(F32), z

    # -- decoders ------------------------------------------------------------------------
    def decode_greedy(self, feats, max_iters=3, return_logits=False):
        """Transducer.decode_greedy (models.py:369-455).  feats [T',F] (one utterance).
        Returns (tokens, neg_log_p, alignment_score, iters[, logits_per_eval])."""
        enc, _ = self.encoder(feats[None])
        enc = enc[0]
        h_pred, pstate = self.predictor([self.bos])                  # models.py:397-398
        fuser = LMFuser(self.lm)                                     # models.py:401
        y, log_p, iters_all, outs = [], 0.0, [], []
        for t in range(enc.shape[0]):
            iters = 0
            while iters < max_iters:                                 # models.py:408
                iters += 1
                lp, z = self.joint_logp(h_pred, enc[t][None])
                if return_logits:
                    outs.append(z[0].copy())
                pred = int(lp[0].argmax())
                log_p += float(lp[0, pred])                          # models.py:420-422
                if pred == self.blank:
                    break
                pred = fuser.fuse(lp[0], pred)                       # models.py:431
                y.append(pred)
                h_pred, pstate = self.predictor([pred], pstate)      # models.py:437
                fuser.advance(pred)                                  # models.py:440
            iters_all.append(iters)
        align = np.array(iters_all)
        s = align.sum()
        ones = int((align == 1).sum())
        score = (s - ones) / (s + 1e-4)                              # models.py:447-453
        res = (y, -log_p, float(score), iters_all)
        return res + (outs,) if return_logits else res

    # -- beam search (ABSENT from the reference -> parity unpinned; this is the spec) --
    def _beam_frame(self, hyps, enc_t, W, max_iters, margins=None):
        """One encoder frame of the slot-synchronous beam.  hyps: list of dicts(score, y, h_pred, pstate),
        len <= W.  Round r <= max_iters: candidates = hyps already done with this frame (B, unchanged)
        + every (hyp in A) x (token v) with score + log p(v); the W best survive (score desc; ties:
        source slot asc, then "already in B" first, then v asc).  Blank extensions join B, non-blank
        ones are expanded again (predictor stepped) unless the round cap is reached -- exactly the
        greedy loop of models.py:405-443 when W == 1.  Scores are float64 sums of float32 log-probs
        (every decision incl. blanks, as log_p at models.py:420-422); no length norm, no merging.

        With an LM attached (self.lm; builder-authored like the beam itself -- the reference fuses only in its greedy loop,
        lm.py:56-83 / models.py:427-431, where the LM never changes WHETHER a token is emitted or its log p, only WHICH token):
        a hypothesis offers TWO candidates per round, its blank extension and its best non-blank extension (score + log p of the
        joint's best non-blank token, first maximum); when the latter is selected the emitted token is the fuser's re-pick
        (LMFuser.fuse on that hypothesis' joint log-softmax and LM state) and the hypothesis' LM advances on it.  W = 1 is then
        exactly the reference's greedy loop with shallow fusion."""
        A = [dict(h, inB=False) for h in hyps]
        for rnd in range(1, max_iters + 1):
            cands = []
            for b, h in enumerate(A):
                if h["inB"]:
                    cands.append((-h["score"], b, 0, -1, h, None))
                    continue
                lp, _ = self.joint_logp(h["h_pred"], enc_t[None])
                lp = lp[0]
                if self.lm is not None:
                    nb = lp.copy()
                    nb[self.blank] = -np.inf
                    top = [self.blank, int(nb.argmax())]
                else:
                    # only the W best tokens of a hyp can make the global top W
                    top = np.argsort(-lp, kind="stable")[:W]
                for v in top:
                    cands.append((-(h["score"] + float(lp[v])), b, 1, int(v), h, lp))
            cands.sort(key=lambda c: c[:4])
            if margins is not None and len(cands) > W:      # score gap at the selection boundary (W-th vs next candidate)
                margins.append(cands[W][0] - cands[W - 1][0])
            new = []
            for negs, b, kind, v, h, lp in cands[:W]:
                if kind == 0:
                    new.append(h)
                elif v == self.blank:
                    new.append(dict(h, score=-negs, inB=True))
                else:
                    fz = h.get("fuser")
                    if fz is not None:
                        v = fz.fuse(lp, v)                   # lm.py:59-79: the emitted token; the score keeps the joint's log p
                        nf = LMFuser(self.lm)
                        nf.lm_logits, nf.lm_state = fz.lm_logits, fz.lm_state
                        nf.advance(v)                        # lm.py:49-53
                    hp, ps = self.predictor([v], h["pstate"])
                    nh = dict(score=-negs, y=h["y"] + [v], h_pred=hp, pstate=ps, inB=(rnd == max_iters))
                    if fz is not None:
                        nh["fuser"] = nf
                    new.append(nh)
            A = new
            if all(h["inB"] for h in A):
                break
        keys = ("score", "y", "h_pred", "pstate") + (("fuser",) if self.lm is not None else ())
        return [{k: h[k] for k in keys} for h in A]

    def beam_init(self):
        h_pred, pstate = self.predictor([self.bos])
        h = dict(score=0.0, y=[], h_pred=h_pred, pstate=pstate)
        if self.lm is not None:
            h["fuser"] = LMFuser(self.lm)
        return [h]

    def decode_beam(self, feats, W, max_iters=3):
        """Offline beam search over one utterance: returns (best tokens, best score, all hyps)."""
        enc, _ = self.encoder(feats[None])
        hyps = self.beam_init()
        for t in range(enc.shape[1]):
            hyps = self._beam_frame(hyps, enc[0, t], W, max_iters)
        best = min(range(len(hyps)), key=lambda i: (-hyps[i]["score"], i))
        return hyps[best]["y"], hyps[best]["score"], hyps

    def stream_decoder(self, max_iters=10):
        """Transducer.transcribe_stream (models.py:457-577) as a push-style object: call
        .step(chunk[T,F]) per non-None chunk -> new tokens of that chunk; .reset() = reset()."""
        return _StreamDecoder(self, max_iters)


class StreamBeamDecoder:
    """transcribe_stream with the beam of OracleTransducer._beam_frame: .step(chunk) -> best tokens so far."""

    def __init__(self, m, W, max_iters=10):
        self.m, self.W, self.max_iters = m, W, max_iters
        self.enc_state = None
        self.hyps = m.beam_init()
        self.step_margin = []          # per model step: smallest score gap that decided something (selection
                                       # boundary of any round, or best vs second-best hypothesis at the end)

    def step(self, chunk):
        enc, self.enc_state = self.m.encoder(chunk[None], self.enc_state)
        margins = []
        for t in range(enc.shape[1]):
            self.hyps = self.m._beam_frame(self.hyps, enc[0, t], self.W, self.max_iters, margins)
        sc = sorted((h["score"] for h in self.hyps), reverse=True)
        if len(sc) > 1:
            margins.append(sc[0] - sc[1])
        self.step_margin.append(min(margins) if margins else float("inf"))
        return self.best()

    def best(self):
        i = min(range(len(self.hyps)), key=lambda i: (-self.hyps[i]["score"], i))
        return self.hyps[i]["y"], self.hyps[i]["score"]


class _StreamDecoder:
    def __init__(self, m, max_iters):
        self.m, self.max_iters = m, max_iters
        self.y = []
        self.decisions = []            # (tokens emitted before the decision, top-1 minus top-2 logit) of every joint evaluation
        self.fuser = LMFuser(m.lm)                                    # models.py:478
        self.reset()

    def reset(self):                                                  # models.py:480-500
        self.enc_state = None
        self.h_pred, self.pstate = self.m.predictor([self.m.bos])
        self.fuser.reset()                                            # models.py:491-492

    def step(self, chunk, return_logits=False):
        m = self.m
        enc, self.enc_state = m.encoder(chunk[None], self.enc_state)  # models.py:517-522
        enc = enc[0]
        y_seq, outs = [], []
        for t in range(enc.shape[0]):
            iters = 0
            while iters < self.max_iters:
                iters += 1
                lp, z = m.joint_logp(self.h_pred, enc[t][None])
                if return_logits:
                    outs.append(z[0].copy())
                pred = int(lp[0].argmax())
                top2 = np.partition(z[0], -2)[-2:]
                self.decisions.append((len(self.y) + len(y_seq), float(top2[1] - top2[0])))
                if pred == m.blank:
                    break
                pred = self.fuser.fuse(lp[0], pred)                   # models.py:558
                y_seq.append(pred)
                self.h_pred, self.pstate = m.predictor([pred], self.pstate)
                self.fuser.advance(pred)                              # models.py:569
        self.y = self.y + y_seq
        return (y_seq, outs) if return_logits else y_seq


# ----------------------------------------------------------------------------- servicer policy
def should_reset(steps, downsample=8, n_buffer=2, thresh=4000):
    """api-server.py:44-50."""
    return int(10.0 * downsample * n_buffer * steps) >= thresh


def servicer_stream(m, pcm, denumericalize, sr=16000, chunk=1280, lead=1, tail=10):
    """ASRServicer.TranscribeStream (api-server.py:82-135) on the oracle's per-call outputs: 3-frame window + stream Pipeline
    (StreamFrontend), Transducer.transcribe_stream (stream_decoder), then the servicer's loop -- char diff of the running text
    (:123-126), "same diff twice" bail-out (:127-129), reset once 4 s of model steps have passed and a step emits nothing
    (:131-134).  Pinned to the reference's own servicer by tests/golden/servicer_tiny.npz.  -> (messages, steps-at-reset)."""
    import itertools as it
    from libreasr_amd import synth
    fe, dec = StreamFrontend(sr=sr), m.stream_decoder()
    out, resets, y, last, last_diff, steps = [], [], [], "", "", 0
    for c in synth.stream_chunks(pcm, chunk, lead=lead, tail=tail):
        o = fe.push(c)
        if o is None:
            continue
        y_seq = dec.step(o)
        steps += 1
        y = y + y_seq
        if denumericalize(y_seq) != "":
            now = denumericalize(y)
            diff = "".join(b for a, b in it.zip_longest(last, now) if a != b)
            last = now
            if diff == last_diff:
                continue
            last_diff = diff
            out.append(diff)
        elif should_reset(steps):
            resets.append(steps)
            dec.reset()
            steps = 0
    return out, resets
