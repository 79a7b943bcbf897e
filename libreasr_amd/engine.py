"""
Engine: thin Python owner of one lasr_ctx (one per process and GPU).

Everything numeric happens in liblasr_hip.so (hand-written gfx950 kernels); this class only
marshals arguments.  PyTorch is used for device memory and the HIP stream.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N
from .weights import flatten_lm_state_dict, flatten_state_dict, infer_cfg

FRONTEND_DEFAULTS = dict(n_fft=1024, win=400, hop=160, n_mels=128, n_stack=10, stride=8, n_buffer=2,
                         n_window=3, chunk=1280, sample_rate=16000)


def _ptr(x):
    """void* of a torch tensor / numpy array / None."""
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return C.c_void_p(x.data_ptr())
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(C.c_void_p)
    raise TypeError(type(x))


class Engine:
    def __init__(self, state_dict, cfg=None, max_streams=64, device=0, max_iters_offline=3,
                 max_iters_stream=10, blank=0, bos=2, dtype="f32", beam=1, **frontend):
        self.lib = N.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("libreasr_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                               "and there is no CPU fallback")
        cfg = dict(cfg) if cfg is not None else infer_cfg(state_dict)
        self.cfg = cfg
        fe = dict(FRONTEND_DEFAULTS)
        fe.update(frontend)
        d = N.ModelDesc()
        self.lib.lasr_default_desc(C.byref(d))
        d.feat, d.hidden, d.embed, d.joint, d.vocab = cfg["feat"], cfg["hidden"], cfg["embed"], cfg["joint"], cfg["vocab"]
        d.enc_layers, d.pred_layers = cfg["enc_layers"], cfg["pred_layers"]
        d.pred_cell = 1 if cfg["pred_cell"] == "LSTM" else 0
        d.blank, d.bos = blank, bos
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        d.dtype = 1 if dtype == "bf16" else 0      # bf16: weights + GEMM-input activations, f32 accumulate
        self.dtype = dtype
        d.beam = int(beam)                         # 1 = greedy; 2..8 = beam search (both protocols)
        self.beam = int(beam)
        for k, v in fe.items():
            setattr(d, k, int(v))
        d.max_streams = int(max_streams)
        d.max_iters_offline, d.max_iters_stream = int(max_iters_offline), int(max_iters_stream)
        self.desc = d
        self.device = torch.device("cuda", device)
        blob = flatten_state_dict(state_dict, cfg)
        want = self.lib.lasr_weight_count(C.byref(d))
        if want == 0:
            raise ValueError("model description rejected by liblasr_hip (dims must be multiples of 16; 32 for bf16)")
        if blob.size != want:
            raise ValueError(f"weight blob has {blob.size} floats, liblasr_hip expects {want}")
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
        ctx = C.c_void_p()
        rc = self.lib.lasr_create(device, C.byref(d), blob.ctypes.data_as(C.c_void_p), blob.size,
                                  C.c_void_p(stream), C.byref(ctx))
        self.ctx = ctx
        if rc != 0:
            msg = self.lib.lasr_last_error(ctx).decode() if ctx else "no ctx"
            if ctx:
                self.lib.lasr_destroy(ctx)
            self.ctx = None
            raise N.LasrError(rc, msg)
        self.max_streams = int(max_streams)

    lm_cfg = None          # set by attach_lm

    def attach_lm(self, lm_state_dict, alpha=0.1, theta=1.0, min_val=-10.0, int8=False):
        """LM shallow fusion (lm.py LM / LMFuser; constants lm.py:13-15).  int8=False: fp32 / bf16 operands like the
        model; int8=True: the LM as the reference serves it (load_lm -> quantize_dynamic qint8, lm.py:97)."""
        cfg, blob = flatten_lm_state_dict(lm_state_dict)
        d = N.LmDesc(cfg["vocab"], cfg["embed"], cfg["hidden"], cfg["layers"], alpha, theta, min_val)
        fn = self.lib.lasr_attach_lm_int8 if int8 else self.lib.lasr_attach_lm
        self._chk(fn(self.ctx, C.byref(d), blob.ctypes.data_as(C.c_void_p), blob.size))
        self.lm_cfg = cfg

    # ------------------------------------------------------------------ plumbing
    def _chk(self, rc):
        if rc != 0:
            raise N.LasrError(rc, self.lib.lasr_last_error(self.ctx).decode())

    def close(self):
        if getattr(self, "ctx", None):
            for fr in list(getattr(self, "_fronts", ())):      # a native front's threads use the context: they go first
                fr.destroy()
            self.lib.lasr_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _slots(slots):
        a = np.ascontiguousarray(np.asarray(slots, dtype=np.int32))
        return a, a.ctypes.data_as(C.c_void_p), int(a.size)

    # ------------------------------------------------------------------ slots
    def open(self):
        s = C.c_int(-1)
        self._chk(self.lib.lasr_stream_open(self.ctx, C.byref(s)))
        return s.value

    def reset(self, slot, what=7, if_decoded=False):
        """what: 1 encoder | 2 predictor | 4 LM | 8 front-end.  if_decoded: accept a slot whose submitted steps are uncollected but
        already decoded for it (peek says so); otherwise a slot with a step in flight is refused."""
        self._chk(self.lib.lasr_stream_reset(self.ctx, int(slot), int(what) | (N.LASR_RESET_IF_DECODED if if_decoded else 0)))

    def reset_many(self, slots, what=7, if_decoded=False):
        """reset(slot, what, if_decoded) for several distinct slots in ONE call (lasr_stream_reset_many): all or nothing."""
        a, p, n = self._slots(slots)
        if n:
            self._chk(self.lib.lasr_stream_reset_many(self.ctx, p, n, int(what) | (N.LASR_RESET_IF_DECODED if if_decoded else 0)))

    def close_slot(self, slot):
        self._chk(self.lib.lasr_stream_close(self.ctx, int(slot)))

    # ------------------------------------------------------------------ streaming
    def _pcm(self, pcm, n):
        if isinstance(pcm, torch.Tensor):
            pcm = pcm.contiguous()
            assert pcm.dtype == torch.float32 and pcm.numel() == n * self.desc.chunk
        else:
            pcm = np.ascontiguousarray(pcm, dtype=np.float32)
            assert pcm.size == n * self.desc.chunk
        return pcm

    def push(self, slots, pcm, pinned_nocopy=False):
        """pcm: [n, chunk] float32 torch (cuda or cpu) tensor or numpy array.  Host memory (pageable or pinned) is copied before
        the call returns; device memory is read in stream order.  pinned_nocopy=True (the tensor must be pinned host memory):
        nothing is copied, the GPU reads the buffer later -- keep it untouched until push_consumed(ticket) is True.
        Returns the push ticket (-1 for device memory)."""
        a, p, n = self._slots(slots)
        pcm = self._pcm(pcm, n)
        t = C.c_longlong(-1)
        self._chk(self.lib.lasr_push_pcm_ex(self.ctx, p, n, _ptr(pcm), N.LASR_PUSH_PINNED_NOCOPY if pinned_nocopy else 0, C.byref(t)))
        return t.value

    def push_submit(self, slots, pcm, pinned_nocopy=False, device_stable=False):
        """push(slots, pcm) + submit(slots) in one call (lasr_push_submit): the front-end launch of a model step takes the newest
        chunk from `pcm` itself.  Same memory rules and results as push + submit.  device_stable=True (a cuda tensor): the caller
        keeps the tensor unchanged until the model step the chunk belongs to is collected (LASR_PUSH_DEVICE_STABLE: a chunk that
        completes no model step is appended by the next call's launch).  Returns the push ticket."""
        a, p, n = self._slots(slots)
        pcm = self._pcm(pcm, n)
        t = C.c_longlong(-1)
        flags = (N.LASR_PUSH_PINNED_NOCOPY if pinned_nocopy else 0) | (N.LASR_PUSH_DEVICE_STABLE if device_stable else 0)
        self._chk(self.lib.lasr_push_submit(self.ctx, p, n, _ptr(pcm), flags, C.byref(t)))
        return t.value

    def push_submit_rows(self, slots, addrs):
        """push_submit with the chunk of slots[i] at host address addrs[i] (uint64 array; `chunk` float32 each, e.g.
        arr.ctypes.data + row * arr.strides[0]): lasr_push_submit_rows -- no gather into one array on this side.  The memory must
        stay alive until the call returns (it is copied before that)."""
        a, p, n = self._slots(slots)
        addrs = np.ascontiguousarray(addrs, dtype=np.uint64)
        assert addrs.size == n
        t = C.c_longlong(-1)
        self._chk(self.lib.lasr_push_submit_rows(self.ctx, p, n, addrs.ctypes.data_as(C.c_void_p), C.byref(t)))
        return t.value

    def push_consumed(self, ticket):
        """True once the source buffer of the push that returned `ticket` has been read by the GPU."""
        rc = self.lib.lasr_push_consumed(self.ctx, int(ticket))
        if rc < 0:
            self._chk(rc)
        return rc == 1

    def max_inflight(self):
        return int(self.lib.lasr_max_inflight(self.ctx))

    def step(self, slots):
        a, p, n = self._slots(slots)
        ran = C.c_int(0)
        self._chk(self.lib.lasr_step_stream(self.ctx, p, n, C.byref(ran)))
        return ran.value

    def step_window(self, slots, windows, sr=16000):
        """Generic clients (any chunk length / sample rate): windows = [n, N] float32 (the last 3 client frames of every
        listed slot, concatenated as the servicer does); returns the number of slots whose model ran (lasr_step_window)."""
        a, p, n = self._slots(slots)
        if isinstance(windows, torch.Tensor):
            windows = windows.contiguous()
            assert windows.dtype == torch.float32 and windows.shape[0] == n
        else:
            windows = np.ascontiguousarray(windows, dtype=np.float32)
            assert windows.shape[0] == n
        ran = C.c_int(0)
        self._chk(self.lib.lasr_step_window(self.ctx, p, n, _ptr(windows), int(windows.shape[1]), int(sr), C.byref(ran)))
        return ran.value

    def submit(self, slots):
        """Pipelined step: enqueue front-end + encoder of the chunk just pushed; returns immediately."""
        a, p, n = self._slots(slots)
        self._chk(self.lib.lasr_step_submit(self.ctx, p, n))

    def pending(self):
        return int(self.lib.lasr_step_pending(self.ctx))

    def wait(self):
        """Decode the oldest submitted model step; returns the number of slots it ran for (0 = none pending)."""
        ran = C.c_int(0)
        self._chk(self.lib.lasr_step_wait(self.ctx, C.byref(ran)))
        return ran.value

    def peek(self, slot, cap=1024, cap_steps=32):
        """-> (token lists of the slot's submitted model steps that are already decoded, oldest first; steps in flight).  Non-consuming
        (lasr_peek_slot): wait() / fetch hand the same tokens out later.  A slot whose steps in flight are all decoded may be reset."""
        buf = np.empty(cap, dtype=np.int32)
        cnt = np.zeros(cap_steps, dtype=np.int32)
        nd, nf = C.c_int(0), C.c_int(0)
        self._chk(self.lib.lasr_peek_slot(self.ctx, int(slot), buf.ctypes.data_as(C.c_void_p), cap, cnt.ctypes.data_as(C.c_void_p), cap_steps,
                                          C.byref(nd), C.byref(nf)))
        out, o = [], 0
        for k in range(nd.value):
            out.append(buf[o:o + cnt[k]].tolist())
            o += int(cnt[k])
        return out, nf.value

    _peek_bufs = None

    def peek_many(self, slots, skip, cap=256, cap_steps=16):
        """lasr_peek_many: for every listed slot the token lists of its decoded, uncollected model steps behind the first skip[i]
        ones, and its steps in flight -> (list of lists of token lists, n_decoded array, n_inflight array)."""
        a, p, n = self._slots(slots)
        sk = np.ascontiguousarray(np.asarray(skip, dtype=np.int32))
        b = self._peek_bufs
        if b is None or b[0].shape[0] < n or b[0].shape[1] != cap or b[1].shape[1] != cap_steps:
            b = self._peek_bufs = (np.empty((max(n, 64), cap), np.int32), np.zeros((max(n, 64), cap_steps), np.int32),
                                   np.zeros(max(n, 64), np.int32), np.zeros(max(n, 64), np.int32))
        tok, cnt, nd, nf = b
        self._chk(self.lib.lasr_peek_many(self.ctx, p, n, sk.ctypes.data_as(C.c_void_p), tok.ctypes.data_as(C.c_void_p), cap,
                                          cnt.ctypes.data_as(C.c_void_p), cap_steps, nd.ctypes.data_as(C.c_void_p), nf.ctypes.data_as(C.c_void_p)))
        out = []
        for i in range(n):
            k_new = int(nd[i]) - int(sk[i])
            if k_new <= 0:
                out.append([])
                continue
            o, steps = 0, []
            for k in range(k_new):
                steps.append(tok[i, o:o + cnt[i, k]].tolist())
                o += int(cnt[i, k])
            out.append(steps)
        return out, nd[:n].copy(), nf[:n].copy()

    def step_feats(self, slots, feats):
        """feats [n, T, feat] (torch cuda/cpu or numpy): one streaming model call with carried state."""
        a, p, n = self._slots(slots)
        if isinstance(feats, torch.Tensor):
            feats = feats.contiguous().float()
            T = feats.shape[1]
        else:
            feats = np.ascontiguousarray(feats, dtype=np.float32)
            T = feats.shape[1]
        assert feats.shape[0] == n and feats.shape[2] == self.desc.feat
        self._chk(self.lib.lasr_step_feats(self.ctx, p, n, _ptr(feats), int(T)))

    def fetch(self, slot, cap=65536):
        buf = np.empty(cap, dtype=np.int32)
        n = C.c_int(0)
        nl, al = C.c_double(0.0), C.c_double(0.0)
        self._chk(self.lib.lasr_fetch(self.ctx, int(slot), buf.ctypes.data_as(C.c_void_p), cap, C.byref(n),
                                      C.byref(nl), C.byref(al)))
        return [int(v) for v in buf[:n.value]], nl.value, al.value

    def fetch_many(self, slots, cap=256):
        """New tokens of every listed slot in one call -> list of lists."""
        a, p, n = self._slots(slots)
        buf = np.empty((n, cap), dtype=np.int32)
        cnt = np.zeros(n, dtype=np.int32)
        self._chk(self.lib.lasr_fetch_many(self.ctx, p, n, buf.ctypes.data_as(C.c_void_p), cap,
                                           cnt.ctypes.data_as(C.c_void_p)))
        cl = cnt.tolist()
        mx = max(cl) if n else 0
        if mx == 0:
            return [[] for _ in range(n)]
        rows = buf[:, :mx].tolist()          # (one conversion of the occupied columns instead of a slice + tolist per slot)
        return [r[:c] for r, c in zip(rows, cl)]

    # ------------------------------------------------------------------ offline
    def transcribe_pcm(self, slots, pcm_list):
        """pcm_list: list of 1-D float32 arrays/tensors (one utterance per slot)."""
        a, p, n = self._slots(slots)
        assert len(pcm_list) == n
        if all(isinstance(x, torch.Tensor) and x.is_cuda for x in pcm_list):
            cat = torch.cat([x.reshape(-1).float() for x in pcm_list]).contiguous()
        else:   # mixed / host inputs: gather on the host, one H2D copy inside the library
            cat = np.ascontiguousarray(np.concatenate([
                np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x, dtype=np.float32).reshape(-1)
                for x in pcm_list]))
        ns = np.ascontiguousarray(np.array([int(np.prod(x.shape)) for x in pcm_list], dtype=np.int64))
        self._chk(self.lib.lasr_transcribe_pcm(self.ctx, p, n, _ptr(cat), ns.ctypes.data_as(C.c_void_p)))

    def transcribe_feats(self, slots, feats_list):
        """feats_list: list of [T', feat] float32 arrays/tensors."""
        a, p, n = self._slots(slots)
        assert len(feats_list) == n
        F = self.desc.feat
        if all(isinstance(x, torch.Tensor) and x.is_cuda for x in feats_list):
            cat = torch.cat([x.reshape(-1, F).float() for x in feats_list]).contiguous()
            nf = [x.reshape(-1, F).shape[0] for x in feats_list]
        else:
            arrs = [np.asarray(x.cpu() if isinstance(x, torch.Tensor) else x, dtype=np.float32).reshape(-1, F) for x in feats_list]
            cat = np.ascontiguousarray(np.concatenate(arrs))
            nf = [a_.shape[0] for a_ in arrs]
        nf = np.ascontiguousarray(np.array(nf, dtype=np.int32))
        self._chk(self.lib.lasr_transcribe_feats(self.ctx, p, n, _ptr(cat), nf.ctypes.data_as(C.c_void_p)))

    # ------------------------------------------------------------------ op-level (tests, microbench)
    def logmel(self, pcm):
        """pcm [B, N] cuda float32 -> [B, T, n_mels]"""
        assert pcm.is_cuda and pcm.dtype == torch.float32 and pcm.dim() == 2
        pcm = pcm.contiguous()
        B, Ns = pcm.shape
        T = 1 + Ns // self.desc.hop
        out = torch.empty(B, T, self.desc.n_mels, device=pcm.device, dtype=torch.float32)
        self._chk(self.lib.lasr_logmel(self.ctx, _ptr(pcm), B, Ns, _ptr(out)))
        return out

    def resample(self, pcm, sr_in):
        """pcm [B, N] float32 cuda at `sr_in` Hz -> [B, N'] at the model's rate (Resample, transforms.py:135-144)."""
        assert pcm.is_cuda and pcm.dtype == torch.float32 and pcm.dim() == 2
        pcm = pcm.contiguous()
        B, N = pcm.shape
        n_out = C.c_int64(0)
        self._chk(self.lib.lasr_resample(self.ctx, None, B, N, int(sr_in), None, C.byref(n_out)))
        out = torch.empty(B, n_out.value, device=pcm.device, dtype=torch.float32)
        self._chk(self.lib.lasr_resample(self.ctx, _ptr(pcm), B, N, int(sr_in), _ptr(out), C.byref(n_out)))
        return out

    def stack(self, logmel):
        logmel = logmel.contiguous()
        B, T, _ = logmel.shape
        d = self.desc
        Tp = 0 if T < d.n_stack else (T - d.n_stack) // d.stride + 1
        out = torch.empty(B, Tp, d.feat, device=logmel.device, dtype=torch.float32)
        tp = C.c_int(0)
        self._chk(self.lib.lasr_stack(self.ctx, _ptr(logmel), B, T, _ptr(out), C.byref(tp)))
        assert tp.value == Tp
        return out

    def encoder(self, feats, return_state=False):
        """feats [B, T', feat] cuda -> [B, T', hidden] (fresh learned initial state; clobbers rows 0..B-1)."""
        feats = feats.contiguous()
        B, Tp, _ = feats.shape
        H, L = self.desc.hidden, self.desc.enc_layers
        out = torch.empty(B, Tp, H, device=feats.device, dtype=torch.float32)
        h = torch.empty(L, B, H, device=feats.device, dtype=torch.float32) if return_state else None
        c = torch.empty(L, B, H, device=feats.device, dtype=torch.float32) if return_state else None
        self._chk(self.lib.lasr_encoder(self.ctx, _ptr(feats), B, Tp, _ptr(out), _ptr(h), _ptr(c)))
        return (out, h, c) if return_state else out

    def predictor(self, tokens):
        """tokens [B, U] int (host) -> h_pred [B, hidden] after the last token, from the learned initial state."""
        tok = np.ascontiguousarray(np.asarray(tokens, dtype=np.int32))
        B, U = tok.shape
        out = torch.empty(B, self.desc.hidden, device=self.device, dtype=torch.float32)
        self._chk(self.lib.lasr_predictor(self.ctx, tok.ctypes.data_as(C.c_void_p), B, U, _ptr(out)))
        return out

    def joint(self, h_pred, h_enc):
        h_pred, h_enc = h_pred.contiguous(), h_enc.contiguous()
        B = h_pred.shape[0]
        logits = torch.empty(B, self.desc.vocab, device=h_pred.device, dtype=torch.float32)
        lp = torch.empty(B, device=h_pred.device, dtype=torch.float32)
        am = torch.empty(B, device=h_pred.device, dtype=torch.int32)
        self._chk(self.lib.lasr_joint(self.ctx, _ptr(h_pred), _ptr(h_enc), B, _ptr(logits), _ptr(lp), _ptr(am)))
        return logits, lp, am

    # ------------------------------------------------------------------ stats
    def set_profiling(self, on=True):
        self._chk(self.lib.lasr_set_profiling(self.ctx, 1 if on else 0))

    def stats(self):
        s = N.StepStats()
        self._chk(self.lib.lasr_get_stats(self.ctx, C.byref(s)))
        return {k: getattr(s, k) for k, _ in N.StepStats._fields_}

    def sync(self):
        self._chk(self.lib.lasr_sync(self.ctx))

    DEBUG_READS = dict(x0=0, enc_h=1, enc_c=2, pe=3, pp=4, pred_h=5, ring=6, pend=7, enc_out=8, ints=9, pendlog=10)

    def debug_read(self, what, index=0):
        """Resident state as a [rows, cols] float32 array (lasr_debug_read; tests and the soak tool)."""
        w = self.DEBUG_READS[what] if isinstance(what, str) else int(what)
        r, k = C.c_int(0), C.c_int(0)
        probe = np.empty(1, dtype=np.float32)
        rc = self.lib.lasr_debug_read(self.ctx, w, int(index), probe.ctypes.data_as(C.c_void_p), 0, C.byref(r), C.byref(k))
        if rc != N.LASR_EFULL:
            self._chk(rc)
        out = np.empty((r.value, k.value), dtype=np.float32)
        self._chk(self.lib.lasr_debug_read(self.ctx, w, int(index), out.ctypes.data_as(C.c_void_p), out.size, C.byref(r), C.byref(k)))
        return out

    def debug_enclog(self):
        """lasr_debug_enclog (LASR_DBG_ENCLOG=N): [steps, 32, M] uint32 checksums of the encoder's inputs and state behind each step."""
        n = C.c_int(0)
        rc = self.lib.lasr_debug_enclog(self.ctx, None, 0, C.byref(n))
        if rc != N.LASR_EFULL:
            self._chk(rc)
            return np.zeros((0, 32, self.config("M")), dtype=np.uint32)
        M = self.config("M")
        out = np.empty((n.value, 32, M), dtype=np.uint32)
        self._chk(self.lib.lasr_debug_enclog(self.ctx, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)))
        return out

    def debug_fe_race(self, iters, aggressor, per_iter=4, lds_pad=-1):
        """lasr_debug_fe_race: (launches, (launch, row) pairs) of the log-mel kernel whose output changed beside the aggressor."""
        a, b = C.c_int(0), C.c_int(0)
        self._chk(self.lib.lasr_debug_fe_race(self.ctx, int(iters), int(aggressor), int(per_iter), int(lds_pad), C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def config(self, key):
        """lasr_debug_config: the engine's resolved configuration (defaults + LASR_* switches), e.g. "enc_wave", "pump_G", "la_stream"."""
        v = C.c_int(0)
        self._chk(self.lib.lasr_debug_config(self.ctx, key.encode(), C.byref(v)))
        return int(v.value)

    def cell_prof(self, on=True):
        """In-job HIP-event timing of the encoder-cell launches (see lasr_cell_prof)."""
        self._chk(self.lib.lasr_cell_prof(self.ctx, int(on)))      # True / 1: events + in-kernel clocks; 2: clocks only

    def cell_prof_kernel(self):
        """-> (microseconds, launches, cells): the cell kernels' own durations since cell_prof(True) (in-kernel wall clock)
        and the LSTM cells they computed (a layer-wavefront launch holds several independent cells)."""
        us, n, k = C.c_double(0.0), C.c_longlong(0), C.c_longlong(0)
        self._chk(self.lib.lasr_cell_prof_kernel(self.ctx, C.byref(us), C.byref(n), C.byref(k)))
        return float(us.value), int(n.value), int(k.value)

    def trace(self, on=True):
        """Timestamped marks on the main / decode streams of the pipelined protocol (see lasr_trace)."""
        self._chk(self.lib.lasr_trace(self.ctx, 1 if on else 0))

    def trace_read(self, cap=8192):
        """-> [(tag, microseconds since trace(True)), ...] in record order."""
        us = (C.c_double * cap)()
        tags = (C.c_int * cap)()
        n = C.c_int(0)
        self._chk(self.lib.lasr_trace_read(self.ctx, us, tags, cap, C.byref(n)))
        return [(int(tags[i]), float(us[i])) for i in range(n.value)]

    def cell_prof_read(self):
        """-> (microseconds, cell launches) accumulated since cell_prof(True)."""
        us, n = C.c_double(0.0), C.c_longlong(0)
        self._chk(self.lib.lasr_cell_prof_read(self.ctx, C.byref(us), C.byref(n)))
        return us.value, n.value

    def bench_neighbour(self, kind, n_wg=0, ms=0):
        """Experiment: kind 1 (MFMA only) / 2 (HBM loads) / 3 (L2 hits) / 4 (Infinity Cache loads) starts a synthetic neighbour of n_wg one-wave workgroups for `ms` ms on a
        stream of its own and returns at once; kind 0 waits for it and returns what it achieved (TFLOP/s / GB/s)."""
        r = C.c_double(0.0)
        self._chk(self.lib.lasr_bench_neighbour(self.ctx, int(kind), int(n_wg), int(ms), C.byref(r)))
        return r.value

    def overlap_probe(self, delay_us=10000):
        """wall time / delay of two delay kernels, one per engine stream: ~1 = the streams overlap, ~2 = one hardware queue."""
        r = C.c_double(0.0)
        self._chk(self.lib.lasr_overlap_probe(self.ctx, int(delay_us), C.byref(r)))
        return r.value

    def bench_cell(self, layer=1, iters=200):
        us = C.c_double(0.0)
        self._chk(self.lib.lasr_bench_cell(self.ctx, int(layer), int(iters), C.byref(us)))
        return us.value
