"""
ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by libreasr_amd/).

The reference's CPU EXECUTION PATH restated on the installed torch: the same torch operators the
reference runs on a CPU (`nn.LayerNorm`, one `nn.LSTM(batch_first=True)` per layer followed by
`nn.BatchNorm1d` in eval mode, the haste NBRC cell as a python loop of torch matmuls, `nn.Embedding`,
`nn.Linear`, `torch.cat`, `F.log_softmax`, `.max(-1)`, `torch.stft`), batch 1 per stream, one joint
evaluation per python loop iteration, `torch.set_num_threads(2)` as `libreasr/lib/inference.py:21` does.
It exists for one purpose: bench.py's `cpu_baseline` leg ("kind": "reference-path"), i.e. the number
the reference itself would achieve on the GPU box's host cores, where /root/reference is absent.
(oracle/rnnt_oracle.py is the numpy restatement used as the *checker*; this file is the *timed
neighbour*.)  tests/test_oracle.py pins it against the goldens the reference's own code produced.

Module / parameter names equal the reference's, so a reference `state_dict` loads with
`load_state_dict` (paths relative to /root/reference):
  Encoder          libreasr/lib/models.py:68-113
  Predictor        libreasr/lib/models.py:143-187
  Joint            libreasr/lib/models.py:116-140
  CustomRNN        libreasr/lib/layers/custom_rnn.py:86-232   (learned initial state, BN after every layer)
  NBRCScript       libreasr/lib/layers/haste/nbrc.py:30-64
  decode loops     libreasr/lib/models.py:369-455 (offline), :457-577 (stream)
  front-end        libreasr/lib/transforms.py:269-342,429-471 + api-server.py:83-115
                   (torchaudio 0.6.0 MelSpectrogram restated with torch.stft: un-vendored dependency)
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class NBRC(nn.Module):
    """haste NBRC layer (haste/nbrc.py:30-64, 112-122): kernel [I,3H], recurrent_kernel [H,3H], gates z,r,g."""

    def __init__(self, i_sz, h_sz):
        super().__init__()
        self.kernel = nn.Parameter(torch.zeros(i_sz, 3 * h_sz))
        self.recurrent_kernel = nn.Parameter(torch.zeros(h_sz, 3 * h_sz))
        self.bias = nn.Parameter(torch.zeros(3 * h_sz))
        self.recurrent_bias = nn.Parameter(torch.zeros(3 * h_sz))

    def forward(self, x, h0):                        # x [B,T,I] (batch_first), h0 [1,B,H]
        inp = x.permute(1, 0, 2)
        h = [h0[0]]
        Wx = inp @ self.kernel + self.bias
        for t in range(inp.shape[0]):
            Rh = h[t] @ self.recurrent_kernel + self.recurrent_bias
            vx = torch.chunk(Wx[t], 3, 1)
            vh = torch.chunk(Rh, 3, 1)
            z = torch.sigmoid(vx[0] + vh[0])
            r = torch.sigmoid(vx[1] + vh[1])
            g = torch.tanh(vx[2] + r * vh[2])
            h.append(z * h[t] + (1 - z) * g)
        out = torch.stack(h[1:]).permute(1, 0, 2)
        return out, h[-1][None]


class RNNStack(nn.Module):
    """CustomCPURNN (custom_rnn.py:86-232, 265-269): per layer rnn -> BatchNorm1d over the feature dim."""

    def __init__(self, i_sz, h_sz, n_layers, rnn_type):
        super().__init__()
        self.rnn_type, self.h_sz = rnn_type, h_sz
        n_state = 2 if rnn_type == "LSTM" else 1
        self.hs = nn.ParameterList([nn.Parameter(torch.zeros(n_state, 1, 1, h_sz)) for _ in range(n_layers)])
        self.bns = nn.ModuleList([nn.BatchNorm1d(h_sz) for _ in range(n_layers)])
        sizes = [i_sz] + [h_sz] * (n_layers - 1)
        if rnn_type == "LSTM":
            self.rnns = nn.ModuleList([nn.LSTM(i, h_sz, batch_first=True) for i in sizes])
        else:
            self.rnns = nn.ModuleList([NBRC(i, h_sz) for i in sizes])

    def forward(self, x, state=None):
        bs = x.size(0)
        new_states = []
        for i, rnn in enumerate(self.rnns):
            if state is None:                        # learned initial state (custom_rnn.py:152-158)
                if self.rnn_type == "LSTM":
                    s = tuple(h.expand(1, bs, self.h_sz).contiguous() for h in self.hs[i])
                else:
                    s = self.hs[i][0].expand(1, bs, self.h_sz).contiguous()
            else:
                s = state[i]
            x, ns = rnn(x, s)
            x = self.bns[i](x.permute(0, 2, 1)).permute(0, 2, 1)      # custom_rnn.py:210-213
            new_states.append(ns)
        return x, new_states


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.input_norm = nn.LayerNorm(cfg["feat"])
        self.rnn_stack = RNNStack(cfg["feat"], cfg["hidden"], cfg["enc_layers"], "LSTM")

    def forward(self, x, state=None):
        x = x.reshape((x.size(0), x.size(1), -1))
        return self.rnn_stack(self.input_norm(x), state)


class Predictor(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embed = nn.Embedding(cfg["vocab"], cfg["embed"], padding_idx=0)
        self.ffn = nn.Linear(cfg["embed"], cfg["hidden"]) if cfg["embed"] != cfg["hidden"] else nn.Sequential()
        self.rnn_stack = RNNStack(cfg["hidden"], cfg["hidden"], cfg["pred_layers"], cfg["pred_cell"])

    def forward(self, x, state=None):
        return self.rnn_stack(self.ffn(self.embed(x)), state)


class Joint(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.joint = nn.Sequential(nn.Linear(2 * cfg["hidden"], cfg["joint"]), nn.Tanh(),
                                   nn.Linear(cfg["joint"], cfg["vocab"]))

    def forward(self, h_pred, h_enc):
        return self.joint(torch.cat((h_pred, h_enc), dim=-1))         # models.py:136: (pred, enc)


class TorchTransducer(nn.Module):
    def __init__(self, sd, cfg, blank=0, bos=2):
        super().__init__()
        self.cfg, self.blank, self.bos = cfg, blank, bos
        self.encoder, self.predictor, self.joint = Encoder(cfg), Predictor(cfg), Joint(cfg)
        t = {k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}
        missing, unexpected = self.load_state_dict(t, strict=False)
        assert all(k.endswith("num_batches_tracked") for k in missing), missing
        assert not unexpected, unexpected
        self.eval()

    @torch.no_grad()
    def decode_greedy(self, feats, max_iters=3):
        """models.py:369-455.  feats [T', F] -> (tokens, -log p)."""
        x = torch.as_tensor(np.asarray(feats))[None]
        encoder_out = self.encoder(x)[0][0]
        y1 = torch.LongTensor([[self.bos]])
        h_t_pred, pred_state = self.predictor(y1)
        y_seq, log_p = [], 0.0
        for h_t_enc in encoder_out:
            iters = 0
            while iters < max_iters:
                iters += 1
                joint_out = F.log_softmax(self.joint(h_t_pred[None], h_t_enc[None, None, None]), dim=-1)
                prob, pred = joint_out.max(-1)
                pred = int(pred)
                log_p += float(prob)
                if pred == self.blank:
                    break
                y_seq.append(pred)
                y1[0][0] = pred
                h_t_pred, pred_state = self.predictor(y1, state=pred_state)
        return y_seq, -log_p

    def stream_decoder(self, max_iters=10):
        return _TorchStream(self, max_iters)


class _TorchStream:
    """The closure state of Transducer.transcribe_stream (models.py:466-500), one instance per stream."""

    def __init__(self, m, max_iters):
        self.m, self.max_iters, self.y = m, max_iters, []
        self.enc_state = None
        with torch.no_grad():
            self.y1 = torch.LongTensor([[m.bos]])
            self.h_t_pred, self.pred_state = m.predictor(self.y1)

    @torch.no_grad()
    def step(self, chunk):                           # chunk [T, F] (x_tfm_stream output without the W axis)
        m = self.m
        x = torch.as_tensor(np.asarray(chunk))[None]
        encoder_out, self.enc_state = m.encoder(x, state=self.enc_state)
        return self.decode_frames(encoder_out[0])

    @torch.no_grad()
    def decode_frames(self, h_t_enc):                # h_t_enc [T, H]: the greedy loop of models.py:526-575 on encoded frames
        m = self.m
        y_seq = []
        for i in range(h_t_enc.size(-2)):
            h_enc = h_t_enc[..., i, :]
            iters = 0
            while iters < self.max_iters:
                iters += 1
                joint_out = F.log_softmax(m.joint(self.h_t_pred[None], h_enc[None, None, None]), dim=-1)
                _, pred = joint_out.max(-1)
                pred = int(pred)
                if pred == m.blank:
                    break
                y_seq.append(pred)
                self.y1[0][0] = pred
                self.h_t_pred, self.pred_state = m.predictor(self.y1, state=self.pred_state)
        self.y = self.y + y_seq
        return y_seq


# ----------------------------------------------------------------------------- front-end
class TorchFrontend:
    """api-server.py:83-115 (3-chunk window) + x_tfm_stream: TransformTime (torchaudio 0.6.0 MelSpectrogram:
    torch.stft center/reflect, periodic Hann(400) zero-padded to 1024, power 2, HTK mel, log(x+1e-6)),
    StreamPostprocess (transforms.py:335-342), StackDownsample (:436-441), Buffer (:461-471)."""

    def __init__(self, n_fft=1024, win=400, hop=160, n_mels=128, sr=16000, n_stack=10, downsample=8, n_buffer=2,
                 n_window=3):
        from .rnnt_oracle import htk_filterbank
        self.n_fft, self.win, self.hop = n_fft, win, hop
        self.n_stack, self.down, self.n_buffer, self.n_window = n_stack, downsample, n_buffer, n_window
        self.window = torch.hann_window(win)
        self.fb = torch.as_tensor(htk_filterbank(n_fft // 2 + 1, 0.0, sr / 2.0, n_mels, sr))
        self.frames, self.saved = [], []

    def logmel(self, aud):                           # aud [1, N] -> [T, n_mels]
        S = torch.stft(aud, self.n_fft, self.hop, self.win, self.window, center=True, pad_mode="reflect",
                       normalized=False, onesided=True, return_complex=True)
        P = S.abs().pow(2.0)
        mel = torch.matmul(P.transpose(1, 2), self.fb).transpose(1, 2)
        return torch.log(mel + 1e-6)[0].permute(1, 0)

    def stack(self, spec):                           # [T, n_mels] -> [T', n_mels * n_stack]  (mel-major)
        x = spec.unfold(0, self.n_stack, self.down)  # [T', n_mels, n_stack]
        return x.reshape(x.size(0), -1)

    def offline(self, pcm):
        return self.stack(self.logmel(torch.as_tensor(np.asarray(pcm))[None]))

    def push(self, chunk):
        self.frames.append(torch.as_tensor(np.asarray(chunk))[None])
        if len(self.frames) != self.n_window:
            return None
        aud = torch.cat(self.frames, dim=1)
        del self.frames[0]
        spec = self.logmel(aud)
        a = spec.size(0) // 3 + 1
        spec = spec[a:a + self.n_stack]
        self.saved.append(self.stack(spec))
        if len(self.saved) == self.n_buffer:
            out = torch.cat(self.saved, dim=0)
            self.saved = []
            return out
        return None


def time_stream_path(sd, cfg, pcm_rows, n_chunks, chunk=1280, threads=2):
    """Runs `len(pcm_rows)` streams x n_chunks 80 ms chunks through the reference-faithful pipeline,
    streams one after the other, batch 1 (what ASRServicer.TranscribeStream does per client,
    api-server.py:82-134).  Returns (seconds, tokens per stream)."""
    import time
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)                   # inference.py:21
    try:
        m = TorchTransducer(sd, cfg)
        toks = []
        t0 = time.perf_counter()
        for row in pcm_rows:
            fe, dec = TorchFrontend(), m.stream_decoder()
            for k in range(n_chunks):
                o = fe.push(row[k * chunk:(k + 1) * chunk])
                if o is not None:
                    dec.step(o)
            toks.append(dec.y)
        return time.perf_counter() - t0, toks
    finally:
        torch.set_num_threads(prev)


def time_stream_path_batched(sd, cfg, pcm_rows, n_chunks, chunk=1280, threads=None):
    """SURVEY 8d (ii) "best-effort CPU": the same operators with the ENCODER (and the front-end) batched over all
    streams -- nn.LSTM on [B, 2, F] with carried state [1, B, H] per layer -- on `threads` cores; the greedy loop stays per
    stream, batch 1, as in the reference (models.py:526-575).  Streams advance in lockstep (every stream pushes chunk k).
    Returns (seconds, tokens per stream); the tokens equal time_stream_path's (tests/test_oracle.py)."""
    import time
    prev = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    try:
        m = TorchTransducer(sd, cfg)
        B = len(pcm_rows)
        fe = TorchFrontend()
        decs = [m.stream_decoder() for _ in range(B)]
        pcm = torch.as_tensor(np.stack([np.asarray(r[:n_chunks * chunk], dtype=np.float32) for r in pcm_rows]))
        frames, saved, enc_state = [], [], None
        t0 = time.perf_counter()
        with torch.no_grad():
            for k in range(n_chunks):
                frames.append(pcm[:, k * chunk:(k + 1) * chunk])
                if len(frames) != fe.n_window:
                    continue
                aud = torch.cat(frames, dim=1)                       # [B, 3 * chunk]
                del frames[0]
                S = torch.stft(aud, fe.n_fft, fe.hop, fe.win, fe.window, center=True, pad_mode="reflect",
                               normalized=False, onesided=True, return_complex=True)
                mel = torch.matmul(S.abs().pow(2.0).transpose(1, 2), fe.fb)          # [B, T, n_mels]
                spec = torch.log(mel + 1e-6)
                a = spec.size(1) // 3 + 1
                spec = spec[:, a:a + fe.n_stack]                     # [B, n_stack, n_mels]
                saved.append(spec.permute(0, 2, 1).reshape(B, 1, -1))                # mel-major, frame-minor (transforms.py:436-441)
                if len(saved) != fe.n_buffer:
                    continue
                x = torch.cat(saved, dim=1)                          # [B, n_buffer, F]
                saved = []
                enc_out, enc_state = m.encoder(x, state=enc_state)
                for b in range(B):
                    decs[b].decode_frames(enc_out[b])
        return time.perf_counter() - t0, [d.y for d in decs]
    finally:
        torch.set_num_threads(prev)
