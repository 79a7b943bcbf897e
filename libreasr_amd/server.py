"""gRPC front of the engine: the reference's `api-server.py` (ASRServicer, ports, reset policy,
transcript diffing) with a batching scheduler in place of its 4-thread / batch-1 model calls.

Reference behaviour kept (api-server.py:16-26, 44-50, 64-135):
  * `ASR.Transcribe(Audio) -> Transcript`, `ASR.TranscribeStream(stream Audio) -> stream Transcript`,
    ports en 50051 / de 50052 / fr 50053, `Audio.data` = raw little-endian float32 PCM.
  * streaming: the model runs once the 3-chunk window is full and the 2-frame Buffer is complete;
    after every model call: if the chunk produced text, re-denumericalize the whole hypothesis and
    send only the characters that changed (zip_longest diff), skipping a repeat of the previous
    diff; otherwise, after >= 4000 ms of empty output (10 ms * downsample * n_buffer * steps), reset().
  * clients with another sample rate or frame length (the browser sends 44.1 / 48 kHz, apps/web/src/App.js:68):
    the last 3 frames are concatenated and the WINDOW is resampled + transformed per call, exactly the
    servicer's sequence (api-server.py:83-115, transforms.py:141-144, 335-342) -> `lasr_step_window`;
    16 kHz / 80 ms clients take the fused path that keeps the window on the GPU.
What differs: the reference shares one model + one stateful `Buffer` transform between 4 worker
threads at batch 1 (a race).  Here RPC threads only queue PCM; ONE scheduler thread
owns the engine (a lasr_ctx is single-caller) and batches every stream that has a chunk ready into one
`lasr_push_submit`, with up to 12 model steps in flight (`lasr_step_wait` collects them): the pipelined protocol
that bench.py measures."""
import collections
import itertools as it
import queue
import threading
from concurrent import futures

import grpc
import numpy as np

from .interfaces import libreasr_pb2 as ap
from .interfaces import libreasr_pb2_grpc as apg
from .lib.utils import tensorize

WORKERS = 64
PORTS = {"en": "[::]:50051", "de": "[::]:50052", "fr": "[::]:50053"}
THRESH = 4000           # ms without output before reset (api-server.py:25)


def should_reset(steps, downsample, n_buffer):
    return int(10.0 * downsample * n_buffer * steps) >= THRESH          # api-server.py:44-50


EOF = object()          # end-of-stream marker on a stream's result queue


class _Stream:
    __slots__ = ("slot", "inq", "outq", "n_pend", "generic", "frames", "results", "text_of", "eof", "y_prev", "trunk", "col", "closed")

    def __init__(self, slot):
        self.slot, self.inq, self.outq = slot, collections.deque(), queue.SimpleQueue()
        self.generic = None                 # decided by the first frame: False = 16 kHz / 80 ms fused path, True = window form
        self.frames, self.n_pend = [], 0    # generic form: the last (up to) 3 client frames (api-server.py:85-99), windows in the Buffer
        # results leave in chunk order: one entry per accepted chunk, [value] once known (None = no model call for the chunk,
        # list = new token ids, Exception); a chunk whose model step is still in flight holds the queue behind it
        self.results = collections.deque()
        self.text_of = None                 # tokens -> text of the chunk (reset policy: "the chunk produced no text")
        self.eof = False
        self.y_prev = []                    # beam search: the best hypothesis as of the previous model step
        self.trunk, self.col = None, -1     # fed through push_batch: its trunk and its column there (results go to batch_outq)
        self.closed = False                 # frames that arrive after close (a reader thread still running) are dropped


class _Trunk:
    """Streams fed together through push_batch: the frames stay in the caller's [n, chunk] arrays, every stream has a cursor."""
    __slots__ = ("streams", "slots", "live", "batches", "base", "pushed", "lut", "addr", "addr_base", "addr_n")

    def __init__(self, streams, n_slots):
        self.streams = list(streams)
        self.slots = np.array([s.slot for s in streams], np.int64)
        self.live = self.slots.copy()       # slots of the streams that are still open
        self.lut = np.full(n_slots, -1, np.int64)              # slot -> column of the batches
        self.lut[self.slots] = np.arange(len(self.slots))
        self.batches, self.base, self.pushed = collections.deque(), 0, 0   # batches[k] has serial base + k
        self.addr = np.zeros(64, np.uint64)                    # host address of the batch with serial addr_base + k (push_submit_rows)
        self.addr_base, self.addr_n = 0, 0

    def note_addr(self, a):
        """Address of the batch just appended (serial addr_base + addr_n); entries of batches every stream has passed are dropped."""
        if self.addr_n == len(self.addr):
            drop = self.base - self.addr_base
            if drop >= len(self.addr) // 2:
                self.addr[:self.addr_n - drop] = self.addr[drop:self.addr_n]
                self.addr_base += drop
                self.addr_n -= drop
            else:
                self.addr = np.concatenate((self.addr, np.zeros(len(self.addr), np.uint64)))
        self.addr[self.addr_n] = a
        self.addr_n += 1


_FAR = 1 << 60


class Scheduler(threading.Thread):
    """Single owner of the engine (a lasr_ctx is single-caller).  RPC threads only queue PCM.  Every tick the scheduler
    batches ONE chunk of every stream that has one waiting into a single lasr_push_submit (front-end + encoder enqueued, the
    greedy loop running on its own stream), keeps up to `depth` model steps in flight and hands the tokens of every collected
    step (lasr_step_wait) to the streams' result queues -- the protocol bench.py measures.  When nothing else is waiting it
    collects at once, so a lone real-time stream sees the latency of the synchronous protocol.

    The reset policy of the servicer (api-server.py:44-50,131-134: after >= 4000 ms since the last reset, the first model step
    without text resets encoder / predictor / LM) is applied HERE, between two model steps of the stream as the reference does:
    a stream that has reached the threshold is not run ahead -- the chunk that would start its next model step waits until the
    step in flight has been judged (chunks that only fill the window / the Buffer go through: a reset does not touch those).
    The other streams keep going; a model step costs the GPU the same for 1 row or 64, so the tick keeps the streams in phase:
    when the streams whose chunk completes a model step are not the majority of a tick they wait for the next one (one loop
    iteration), where the others have caught up.
    Generic client frames (other rates / lengths) take the synchronous entry point.

    The per-stream bookkeeping (frames queued, frames to the next model step, steps in flight, steps since the last reset) lives
    in arrays indexed by slot: a tick classifies all streams with a dozen vector operations whatever their number."""

    def __init__(self, engine, depth=12, downsample=None, n_buffer=None, held_depth=None):
        super().__init__(daemon=True, name="lasr-scheduler")
        self.eng, self.cv = engine, threading.Condition()
        self.streams, self.ctl, self.stop_flag = {}, collections.deque(), False
        self.batches = []                   # sizes of the streaming batches (observability / tests)
        self.step_rows = []                 # rows per submitted model step (observability / tests)
        self.max_inflight_seen = 0
        self.depth = max(1, min(int(depth), engine.max_inflight()))
        # while a stream is held at the reset threshold its verdict is `steps in flight` model steps away and the steps submitted
        # meanwhile run without it: fewer steps in flight then (faster-than-real-time replays with the reset rule, 64 streams of
        # configs[1] on the GPU: 13.9 k audio-s/s at 12, 20.1 k at 6, 21.4 k at 3, 20.3 k at 1, tools/served_depth_sweep.py; a
        # real-time stream is never run ahead at all)
        self.beam = engine.beam
        # Early verdicts (round 4): the engine lets the scheduler LOOK at a slot's decoded-but-uncollected steps (lasr_peek_slot) and
        # reset a slot whose steps in flight are all decoded, so a held stream is judged when its row is decoded -- a few hundred
        # microseconds after the submit -- instead of `steps in flight` collections later, and the other streams keep the full depth
        # meanwhile.  (Greedy decode; with beam > 1, or an engine without peek, the round-3 behaviour: held_depth 3.)
        self.can_peek = self.beam == 1 and hasattr(engine, "peek_many")
        # the frames of a tick go to the engine as ROW ADDRESSES (lasr_push_submit_rows): no gather into one matrix on this side
        # when the streams of a trunk stand at different batches or only some of them run (the replay with the reset rule spent
        # 70 us per tick there); several resets of a tick are one engine call (lasr_stream_reset_many)
        self.can_rows = hasattr(engine, "push_submit_rows")
        self.can_reset_many = hasattr(engine, "reset_many")
        self._keep = []                      # arrays whose rows the tick's addresses point into (alive until the submit has returned)
        if held_depth is None:               # (tools/r04/served_profile.py, early verdicts + row addresses + stored BOS state, 64 streams in
            # phase / clocks out of phase, medians of 7 replays: 6 -> 34.6 / 30.6 k, 9 -> 33.3 / 29.7 k; single runs: 3 -> 31.3 / 25.9 k,
            # 12 -> 34.5 / 31.2 k -- run-to-run spread +-4 %)
            held_depth = 6 if self.can_peek else 3
        self.held_depth = max(1, min(int(held_depth), self.depth))
        self.any_held = False
        self._beam_cap = 1024               # tokens per stream fetch_many makes room for with beam > 1 (grows on LASR_EFULL)
        # early verdicts: every step in flight of a slot decoded = depth x n_buffer x max_iters_stream tokens (20 per step with
        # the reference's front-end: 300 at depth 15)
        self._peek_cap = max(256, 32 * (self.depth + 1))
        self.inflight = collections.deque() # per submitted model step: (slots of its streams, {slot: result cell} | None)
        self.batch_outq = queue.SimpleQueue()   # trunk interface (push_batch): one item per collected model step
        self.downsample = downsample or engine.desc.stride
        self.n_buffer = n_buffer or engine.desc.n_buffer
        k = 1                               # should_reset is monotone in the step count: the first count at which it holds
        while not should_reset(k, self.downsample, self.n_buffer) and k < (1 << 20):
            k += 1
        self.reset_steps = k
        N = int(getattr(engine, "max_streams", 0) or getattr(engine.desc, "max_streams", 0) or 1024)
        z = lambda v=0: np.full(N, v, np.int64)
        self.qn = z()                       # frames waiting (written under cv by the producers)
        self.eofp = np.zeros(N, bool)       # an EOF marker waits behind the frames
        self.n_wait = 0                     # frames + EOF markers waiting, all streams
        self.fast = np.zeros(N, bool)       # fused path decided (16 kHz / 80 ms frames)
        self.phase = z()                    # frames up to and including the one that completes the next model step (1: the next frame does)
        self.infl = z()                     # model steps submitted and not collected
        self.stp = z()                      # model steps since the last reset (api-server.py:117,133)
        self.judged = z()                   # steps in flight that have already been judged (and counted in stp) through peek
        self.rat = z(_FAR)                  # steps since the last reset from which the reset rule can fire (_FAR: no text function)
        self.cur = z()                      # trunk streams: serial of the next batch to take
        self.is_trunk = np.zeros(N, bool)
        self.trunks = []
        self.error = None                   # the exception that ended the scheduler thread, if any
        # counters that let a tick skip whole groups of vector operations
        self.n_slow = 0                     # open streams whose form is undecided or generic
        self.n_ruled = 0                    # open streams with a text function (the reset rule applies)
        self.n_eof = 0                      # EOF markers waiting
        self.n_blocked = 0                  # frames waiting in streams that the last tick found held at the reset threshold
        self._rows = np.zeros((N, engine.desc.chunk), np.float32)          # per-stream form: the frames of a tick, row by row
        self._take_mask = np.zeros(N, bool)
        self._arange = np.arange(N, dtype=np.int64)

    # ---- called from RPC threads -----------------------------------------------------------
    def _call(self, fn):
        done = queue.SimpleQueue()
        with self.cv:
            if self.stop_flag:
                raise RuntimeError("scheduler shut down" if self.error is None else f"scheduler stopped by {type(self.error).__name__}: {self.error}")
            self.ctl.append((fn, done))
            self.cv.notify()
        res = done.get()
        if isinstance(res, Exception):
            raise res
        return res

    def open(self, text_of=None):
        return self._call(lambda: self._open(text_of))

    def close(self, st):
        self._call(lambda: self._close(st))

    def reset(self, st):
        self._call(lambda: self._reset(st.slot))

    def transcribe(self, pcm, sr=16000):
        return self._call(lambda: self._offline(pcm, sr))

    def push_nowait(self, st, chunk, sr=16000):
        """Queue one client frame; its result (None / token list / Exception) appears on st.outq, in frame order."""
        with self.cv:
            if self.stop_flag:
                raise RuntimeError("scheduler shut down" if self.error is None else f"scheduler stopped by {type(self.error).__name__}: {self.error}")
            if st.closed:
                return
            if st.trunk is not None:
                raise ValueError("this stream is fed through push_batch")
            st.inq.append((chunk, int(sr) or 16000))
            self.qn[st.slot] += 1
            self.n_wait += 1
            self.cv.notify()

    def push(self, st, chunk, sr=16000):
        """Queue one frame and wait for its result: None = no model call for this chunk, else the new token ids."""
        self.push_nowait(st, chunk, sr)
        return st.outq.get()

    def push_batch(self, streams, chunks):
        """Trunk interface (one producer feeding many streams, e.g. a bridge that demultiplexes one connection): ONE call for a
        whole batch -- chunks [n, chunk] float32 at the model rate, chunks[i] belongs to streams[i]; the first call fixes the
        trunk's stream list, every later call passes the same list.  The array is not copied here: hand over a fresh one per
        call.  Every collected model step arrives as ONE item (streams_that_ran, token_lists) on self.batch_outq: a stream's
        steps arrive in order, but one item need not hold every stream of the trunk (a stream at the reset threshold waits for
        its previous step, see the class docstring).  Same engine calls, same reset rule as the per-stream form; a stream uses
        either this form or push / push_nowait."""
        with self.cv:
            if self.stop_flag:
                raise RuntimeError("scheduler shut down" if self.error is None else f"scheduler stopped by {type(self.error).__name__}: {self.error}")
            T = streams[0].trunk
            if T is None:
                if any(s.trunk is not None or s.inq or s.generic is not None for s in streams):
                    raise ValueError("push_batch: a trunk is made of fresh streams")
                T = _Trunk(streams, len(self.qn))
                for i, s in enumerate(streams):
                    s.trunk, s.col, s.generic = T, i, False
                sl = T.slots
                self.n_slow -= int(np.count_nonzero(~self.fast[sl]))
                self.is_trunk[sl] = True
                self.fast[sl] = True
                self.cur[sl] = 0
                self.trunks.append(T)
            elif len(streams) != len(T.streams) or streams[-1] is not T.streams[-1]:
                raise ValueError("push_batch: the stream list of a trunk is fixed by its first call")
            if chunks.shape[0] != len(T.streams) or chunks.shape[1] != self.eng.desc.chunk:
                raise ValueError(f"push_batch: chunks must be [{len(T.streams)}, {self.eng.desc.chunk}]")
            if self.can_rows:
                if chunks.dtype != np.float32 or not chunks.flags.c_contiguous:
                    chunks = np.ascontiguousarray(chunks, np.float32)
                T.note_addr(chunks.ctypes.data)
            T.batches.append(chunks)
            T.pushed += 1
            self.qn[T.live] += 1
            self.n_wait += len(T.live)
            self.cv.notify()

    def push_eof(self, st):
        """After the last frame: EOF is put on st.outq behind the result of the last frame."""
        with self.cv:
            if st.closed or self.eofp[st.slot]:
                return
            self.eofp[st.slot] = True
            self.n_wait += 1
            self.n_eof += 1
            self.cv.notify()

    def shutdown(self):
        with self.cv:
            self.stop_flag = True
            self.cv.notify()

    # ---- scheduler thread ------------------------------------------------------------------
    def _open(self, text_of=None):
        st = _Stream(self.eng.open())
        st.text_of = text_of
        i, d = st.slot, self.eng.desc
        with self.cv:
            self.streams[i] = st
            self.qn[i], self.eofp[i], self.fast[i], self.is_trunk[i] = 0, False, False, False
            self.n_slow += 1
            self.n_ruled += text_of is not None
        self.infl[i], self.stp[i], self.cur[i], self.judged[i] = 0, 0, 0, 0
        self.rat[i] = self.reset_steps if text_of is not None else _FAR
        self.phase[i] = d.n_window + d.n_buffer - 1      # window full at frame n_window, then every n_buffer-th frame completes a step
        return st

    def _close(self, st):
        i = st.slot
        with self.cv:
            self.streams.pop(i, None)
            self.n_wait -= int(self.qn[i]) + int(self.eofp[i])
            self.n_eof -= int(self.eofp[i])
            self.n_slow -= not self.fast[i]
            self.n_ruled -= st.text_of is not None
            self.qn[i], self.eofp[i], self.fast[i] = 0, False, False
            st.inq.clear()
            st.closed = True
            if st.trunk is not None:
                T = st.trunk
                T.live = T.live[T.live != i]
                self.is_trunk[i] = False
                if not len(T.live):
                    self.trunks.remove(T)
        self.eng.close_slot(i)

    def _reset(self, i, if_decoded=False):
        if if_decoded:                       # (early verdict: the slot's steps in flight are decoded, not collected)
            self.eng.reset(i, 1 | 2 | 4, if_decoded=True)
        else:
            self.eng.reset(i, 1 | 2 | 4)                                   # models.py:494-497
        self.stp[i] = 0

    def _reset_all(self, slots, if_decoded=False):
        """The resets a tick has decided on (models.py:494-497 per stream), in one engine call when it offers that."""
        if not slots:
            return
        if len(slots) == 1 or not self.can_reset_many:
            for i in slots:
                self._reset(i, if_decoded)
            return
        sl = np.array(slots, np.int64)
        if if_decoded:
            self.eng.reset_many(sl, 1 | 2 | 4, if_decoded=True)
        else:
            self.eng.reset_many(sl, 1 | 2 | 4)
        self.stp[sl] = 0

    def _offline(self, pcm, sr=16000):
        slot = self.eng.open()
        try:
            if sr != self.eng.desc.sample_rate:      # Resample (transforms.py:135-144) of the whole utterance, on the GPU
                import torch
                pcm = self.eng.resample(torch.as_tensor(pcm[None]).to(self.eng.device), sr)[0]
            self.eng.transcribe_pcm([slot], [pcm])
            return self.eng.fetch(slot)
        finally:
            self.eng.close_slot(slot)

    def _flush(self, st):
        while st.results and st.results[0]:
            st.outq.put(st.results.popleft()[0])
        if st.eof and not st.results:
            st.outq.put(EOF)
            st.eof = False

    def _judge(self, st, tokens):
        """A model step of `st` (already counted in stp) has been decoded: the servicer's reset rule (api-server.py:131-134)."""
        new = tokens
        if self.beam > 1:                   # the engine hands out the whole best hypothesis: this chunk's part follows the common prefix
            n = 0
            while n < min(len(st.y_prev), len(tokens)) and st.y_prev[n] == tokens[n]:
                n += 1
            st.y_prev, new = list(tokens), tokens[n:]
        # (True: reset wanted -- the stream has nothing in flight, see _take; the caller resets all such streams of the step together)
        return bool(self.stp[st.slot] >= self.rat[st.slot] and (not new or st.text_of(new) == ""))

    def _collect(self):
        """Tokens of the oldest model step in flight -> its streams."""
        sl, cells = self.inflight.popleft()
        ran = self.eng.wait()
        if ran != len(sl):                   # the host mirror of the window / Buffer bookkeeping and the engine disagree
            raise RuntimeError(f"scheduler: expected a model step of {len(sl)} streams, the engine ran {ran}")
        toks = self._fetch(sl)
        self.infl[sl] -= 1
        pre = self.judged[sl] > 0            # judged (and counted) when the row was decoded: see _early_verdicts
        if pre.any():
            self.judged[sl[pre]] -= 1
            self.stp[sl[~pre]] += 1
        else:
            self.stp[sl] += 1
        streams = self.streams
        rows = [streams[i] for i in sl.tolist()]
        if self.beam > 1:                    # (the hypothesis of the previous step is needed for every later judgement)
            self._reset_all([s.slot for s, t in zip(rows, toks) if s.text_of is not None and self._judge(s, t)])
        elif self.n_ruled:
            hot = np.flatnonzero((self.stp[sl] >= self.rat[sl]) & ~pre)
            if len(hot):                     # only the streams within reach of the reset rule
                self._reset_all([rows[k].slot for k in hot.tolist() if self._judge(rows[k], toks[k])])
        self.n_blocked = 0                   # (a judged step may have released a held stream: the next tick finds out)
        if cells is None:                    # a step of trunk streams only
            self.batch_outq.put((rows, toks))
            return
        t_streams, t_toks = [], []
        for s, t in zip(rows, toks):
            cell = cells.get(s.slot)
            if cell is None:
                t_streams.append(s)
                t_toks.append(t)
            else:
                cell.append(t)
                self._flush(s)
        if t_streams:
            self.batch_outq.put((t_streams, t_toks))

    def _fetch(self, sl):
        """New tokens of the slots of a collected step.  Beam search hands out the whole best hypothesis, which grows with the
        stream: the buffer grows with it (LASR_EFULL leaves the engine's queues untouched) instead of ending the scheduler."""
        if self.beam == 1:
            return self.eng.fetch_many(sl, cap=256)
        from ._native import LASR_EFULL, LasrError
        while True:
            try:
                return self.eng.fetch_many(sl, cap=self._beam_cap)
            except LasrError as e:
                if e.code != LASR_EFULL or self._beam_cap >= (1 << 22):
                    raise
                self._beam_cap *= 4

    def _early_verdicts(self):
        """Streams whose next model step waits for the verdict on a step in flight (see _take): look at the engine's decoded,
        uncollected steps of the slot and judge them now, in order -- the servicer's rule, api-server.py:131-134, between the
        step and the stream's next one exactly as if the step had been collected."""
        unj = self.infl - self.judged
        cand = np.flatnonzero((unj > 0) & (self.stp + unj + 1 >= self.rat))
        if not len(cand):
            return
        # worst case per slot: every step in flight decoded, n_buffer frames x max_iters_stream tokens each (the fixed
        # cap of 256 tokens / 16 steps overflowed at depth 13-15 and the LASR_EFULL ended the scheduler for every stream)
        from ._native import LASR_EFULL, LasrError
        steps_cap = max(16, int(self.depth) + 1)
        try:
            steps_of, n_dec, n_inflight = self.eng.peek_many(cand, self.judged[cand], cap=self._peek_cap, cap_steps=steps_cap)
        except LasrError as e:
            if e.code != LASR_EFULL:
                raise
            self._peek_cap = max(256, 4 * self._peek_cap)      # (a model with a larger n_buffer x max_iters_stream than the reference's)
            return                               # the verdicts of this tick fall back to collect time
        resets = []
        for q, i in enumerate(cand.tolist()):
            st = self.streams.get(i)
            if st is None or st.text_of is None:
                continue
            for new in steps_of[q]:
                self.judged[i] += 1
                self.stp[i] += 1
                if self.stp[i] >= self.rat[i] and (not new or st.text_of(new) == ""):
                    # (past the threshold a stream has ONE step in flight at a time: this was its last, and it is decoded)
                    if int(self.judged[i]) != int(n_inflight[q]):
                        raise RuntimeError(f"scheduler: slot {i} ran ahead of the reset threshold ({int(n_inflight[q])} steps in flight)")
                    resets.append(i)
                    self.stp[i] = 0
        self._reset_all(resets, if_decoded=True)

    def _drain(self):
        while self.inflight:
            self._collect()

    def run(self):
        cause = None
        try:
            self._run()
        except BaseException as e:           # an engine error ends the scheduler: the waiters are told why
            cause = e
            raise
        finally:                             # whoever still waits for a result or a call gets an error, not a hang
            with self.cv:
                self.stop_flag = True
                err = RuntimeError("scheduler shut down" if cause is None else f"scheduler stopped by {type(cause).__name__}: {cause}")
                self.error = cause
                for fn, done in self.ctl:
                    done.put(err)
                self.ctl.clear()
                trunk_waits = bool(self.inflight) or any(self.qn[T.live].any() for T in self.trunks)
                for st in self.streams.values():
                    st.inq.clear()
                    st.outq.put(err)
                self.qn[:] = 0
                self.eofp[:] = False
                self.n_wait = self.n_eof = 0
                if trunk_waits:
                    self.batch_outq.put(err)                   # a trunk consumer waiting for the steps it has queued

    def _take(self):
        """(cv held) One frame of every stream that can go now.  Returns (slots whose frame only fills the window / Buffer, slots
        whose frame completes a model step, [(stream, frame)] of the generic form); the frames of the fused path are taken by
        _gather."""
        d = self.eng.desc
        qn, fast = self.qn, self.fast
        has = qn > 0
        gen = []
        if self.n_slow:                      # first frame of a stream, or the generic form
            for i in np.flatnonzero(has & ~fast).tolist():
                s = self.streams[i]
                if s.generic is None:        # the stream's first frame decides its form
                    pcm, sr = s.inq[0]
                    s.generic = not (sr == d.sample_rate and pcm.shape[0] == d.chunk)
                    if not s.generic:
                        fast[i] = True
                        self.n_slow -= 1
                        continue
                gen.append((s, s.inq.popleft()))
                qn[i] -= 1
                self.n_wait -= 1
                has[i] = qn[i] > 0
        ready = has & fast
        step_f = ready & (self.phase == 1)
        if self.n_ruled:                     # at the reset threshold: the step in flight is judged before the next one starts
            infl = self.infl
            unj = infl - self.judged         # steps in flight whose verdict is still out
            held = step_f & (unj > 0) & (self.stp + unj + 1 >= self.rat)
            self.n_blocked = int(qn[held].sum()) + (int(np.count_nonzero(self.eofp & held)) if self.n_eof else 0)
            self.any_held = self.n_blocked > 0
            steps_m = step_f & ~held
        else:
            steps_m = step_f
            self.any_held = False
        fills_m = ready & ~step_f
        n_st, n_fl = int(np.count_nonzero(steps_m)), int(np.count_nonzero(fills_m))
        # a model step costs the GPU the same for one row or all of them: step frames that are not the majority of the tick wait
        # for the next one, where the streams that only fill their window / Buffer now have their own step frame up
        if n_fl and n_st <= n_fl:
            n_st = 0
        fills = np.flatnonzero(fills_m) if n_fl else _EMPTY
        steps = np.flatnonzero(steps_m) if n_st else _EMPTY
        if self.n_eof:
            for i in np.flatnonzero(self.eofp & (qn == 0)).tolist():
                self.eofp[i] = False
                self.n_wait -= 1
                self.n_eof -= 1
                s = self.streams.get(i)
                if s is not None:
                    s.eof = True
                    self._flush(s)
        return fills, steps, gen

    def _gather(self, take):
        """(cv held) The frames of the fused-path slots `take` -> (slots in row order, [rows, chunk] matrix or list of rows,
        {slot: cell} of the per-stream form)."""
        d = self.eng.desc
        CH = d.chunk
        self.qn[take] -= 1
        self.n_wait -= len(take)
        mats, order, cells = [], [], None
        n_trunk = 0
        rows_mode = self.can_rows            # addresses of the rows instead of a gathered matrix (see __init__)
        if self.trunks:
            whole = len(self.trunks) == 1 and len(take) == len(self.trunks[0].live) and len(take) == len(self.streams)
            for T in self.trunks:
                if len(T.live):
                    lo = int(self.cur[T.live].min())
                    while T.base < lo:       # batches every live stream has passed -- dropped a tick late: the previous tick's row
                        T.batches.popleft()  # addresses pointed into them until its submit returned
                        T.base += 1
                sl = T.live                  # (whole: the tick took exactly the trunk's streams)
                if not whole:
                    sl = sl[self._mask(take)[sl]]
                    if not len(sl):
                        continue
                n_trunk += len(sl)
                cur = self.cur[sl]
                c0 = int(cur[0])
                cols = T.lut[sl]             # row of every taken slot in the trunk's arrays (slot order != row order in general:
                                             # a trunk built from re-opened slots)
                same = bool((cur == c0).all())
                if same and len(sl) == len(T.slots) and (cols == self._arange[:len(sl)]).all():
                    mats.append(T.batches[c0 - T.base])      # the common case: every stream of the trunk, the same batch, in row order
                    order.append(sl)
                elif rows_mode:
                    mats.append(T.addr[cur - T.addr_base] + (cols * (4 * CH)).astype(np.uint64))
                    order.append(sl)
                elif same:
                    mats.append(T.batches[c0 - T.base][cols])
                    order.append(sl)
                else:
                    for c in np.unique(cur).tolist():
                        m = cur == c
                        mats.append(T.batches[c - T.base][cols[m]])
                        order.append(sl[m])
                self.cur[sl] += 1
        if n_trunk != len(take):
            cells = {}
            sl, k = [], 0
            buf = self._rows                 # (push_submit copies host memory before it returns: one buffer serves every tick)
            for i in take[~self.is_trunk[take]].tolist():
                s = self.streams[i]
                pcm, sr = s.inq.popleft()
                cell = []
                s.results.append(cell)
                if sr != d.sample_rate or pcm.shape[0] > CH:
                    cell.append(ValueError(f"stream opened with {CH}-sample {d.sample_rate} Hz frames: got {pcm.shape[0]} samples at {sr} Hz"))
                    continue                 # (flushed by the caller, outside the lock)
                if pcm.shape[0] < CH:                           # api-client.py:40-41 pads the last slice with zeros
                    buf[k, :pcm.shape[0]] = pcm
                    buf[k, pcm.shape[0]:] = 0.0
                else:
                    buf[k] = pcm
                k += 1
                cells[i] = cell
                sl.append(i)
            if k:
                mats.append(buf[:k])
                order.append(np.array(sl, np.int64))
        if not mats:
            return _EMPTY, None, cells
        if len(mats) == 1:
            return order[0], mats[0], cells
        if rows_mode:                        # several sources: everything as row addresses
            mats = [m if m.ndim == 1 else self._row_addrs(m) for m in mats]
        return np.concatenate(order), np.concatenate(mats), cells

    def _mask(self, take):
        """Boolean mask over the slots for the tick's `take` (one buffer, cleared after use by the next call)."""
        m = self._take_mask
        m[:] = False
        m[take] = True
        return m

    @staticmethod
    def _row_addrs(m):
        """Host addresses of the rows of a C-contiguous float32 matrix."""
        return np.uint64(m.ctypes.data) + np.arange(m.shape[0], dtype=np.uint64) * np.uint64(m.strides[0])

    def _work_waiting(self):
        # frames are waiting in streams that can go (n_blocked: frames and EOF markers behind a stream held at the reset threshold,
        # as of the last tick; an estimate on the safe side costs one empty tick)
        return bool(self.ctl) or self.n_wait > self.n_blocked

    def _run(self):
        d = self.eng.desc
        while True:
            if self.can_peek and self.n_ruled and self.inflight:
                self._early_verdicts()
            with self.cv:
                while not (self.stop_flag or self.ctl or self.inflight or self.n_wait > 0):
                    self.cv.wait()
                if self.stop_flag:
                    return
                ctl = list(self.ctl)
                self.ctl.clear()
                if not ctl:
                    fills, steps, gen = self._take()
                    take = fills if not len(steps) else steps if not len(fills) else np.concatenate((fills, steps))
                    order, mat, cells = self._gather(take) if len(take) else (_EMPTY, None, None)
            if ctl:                          # open / close / reset / offline calls: before any frame of this tick is taken
                self._drain()                # (state-changing calls need an idle engine)
                for fn, done in ctl:
                    try:
                        done.put(fn())
                    except Exception as e:   # surfaced in the calling RPC thread
                        done.put(e)
                continue
            if not len(take) and not gen:    # nothing can move (streams held at the reset threshold, or only markers handled):
                if self.inflight:            # the way forward is the oldest step's verdict
                    self._collect()
                else:
                    with self.cv:
                        if not (self.stop_flag or self.ctl) and self.n_wait > 0:
                            self.cv.wait(0.01)                  # (defensive: never spin on frames that cannot go)
                continue
            if len(take):
                self._submit_fast(fills, steps, take, order, mat, cells)
            if gen:
                self._submit_generic(gen, d)
            # collect: at the depth limit, or as soon as no further chunk is waiting (latency of a lightly loaded server)
            while self.inflight and (len(self.inflight) >= (self.held_depth if self.any_held else self.depth) or not self._work_waiting()):
                self._collect()

    def _fail(self, s, cell, e):
        if not cell:
            cell.append(e)
        self._flush(s)

    def _submit_fast(self, fills, steps, take, order, mat, cells):
        """One frame of each taken stream -> ONE lasr_push_submit; `steps` (the streams whose frame completes a model step)
        share the model step."""
        nb = self.eng.desc.n_buffer
        if cells is not None and len(order) != len(take):       # refused frames (wrong size / rate) left the batch
            ok = np.zeros(len(self.qn), bool)
            ok[order] = True
            for i in take[~ok[take]].tolist():
                self._flush(self.streams[i])
            fills, steps = fills[ok[fills]], steps[ok[steps]]
            if not len(order):
                return
        try:
            self.batches.append(len(order))
            if mat.ndim == 1:                                   # row addresses (see _gather)
                self.eng.push_submit_rows(order, mat)
            else:
                self.eng.push_submit(order, mat)
        except Exception as e:
            if cells is None or len(cells) != len(order):
                self.batch_outq.put(e)
            for i, cell in (cells or {}).items():
                self._fail(self.streams[i], cell, e)
            return
        self.phase[order] -= 1                                  # host mirror of the engine's window / Buffer bookkeeping
        if cells is not None:
            for i in fills.tolist():
                cell = cells.pop(i, None)
                if cell is not None:
                    cell.append(None)                           # no model call for this frame
                    self._flush(self.streams[i])
        if len(steps):
            self.phase[steps] = nb
            self.infl[steps] += 1
            self.step_rows.append(len(steps))
            self.inflight.append((steps, cells if cells else None))
            if len(self.inflight) > self.max_inflight_seen:
                self.max_inflight_seen = len(self.inflight)

    def _submit_generic(self, gen, d):
        """api-server.py:85-99: window of the last 3 frames, resampled + transformed per call (lasr_step_window, synchronous)."""
        groups = collections.OrderedDict()
        for s, (pcm, sr) in gen:
            cell = []
            s.results.append(cell)
            s.frames.append(pcm)
            if len(s.frames) != d.n_window:
                cell.append(None)
                self._flush(s)
                continue
            win = np.concatenate(s.frames)
            del s.frames[0]
            groups.setdefault((win.shape[0], sr), []).append((s, cell, win))
        if groups:
            self._drain()
        for (N, sr), group in groups.items():                   # same window length and rate: one batched call
            try:
                slots = [s.slot for s, _, _ in group]
                self.eng.step_window(slots, np.stack([w for _, _, w in group]), sr)
                toks = self._fetch(slots)
                for (s, cell, _), t in zip(group, toks):
                    s.n_pend += 1
                    if s.n_pend == d.n_buffer:
                        s.n_pend = 0
                        self.stp[s.slot] += 1
                        # (the servicer's reset rule for these streams too -- api-server.py:131-134 does not look at the client's
                        #  rate; round 5: the verdict used to be computed and dropped.  Nothing of the slot is in flight here.)
                        if s.text_of is not None and self._judge(s, t):
                            self._reset(s.slot)
                        cell.append(t)
                    else:
                        cell.append(None)
                    self._flush(s)
            except Exception as e:
                for s, cell, _ in group:
                    self._fail(s, cell, e)


_EMPTY = np.zeros(0, np.int64)


class ASRServicer(apg.ASRServicer):
    def __init__(self, lang, scheduler, language, conf=None):
        self.lang_name, self.sched, self.lang = lang, scheduler, language
        eng = scheduler.eng
        self.downsample, self.n_buffer, self.chunk = eng.desc.stride, eng.desc.n_buffer, eng.desc.chunk
        self.beam = eng.beam

    @staticmethod
    def _guard(context, fn):
        """All stream slots taken (LASR_EFULL) is RESOURCE_EXHAUSTED for the client, not UNKNOWN."""
        from ._native import LASR_EFULL, LasrError
        try:
            return fn()
        except LasrError as e:
            if e.code == LASR_EFULL:
                context.abort(grpc.StatusCode.RESOURCE_EXHAUSTED, "all stream slots are in use")
            raise

    def Transcribe(self, request, context):                                # api-server.py:64-80
        aud = tensorize(request.data)[0].numpy()
        tokens, _, _ = self._guard(context, lambda: self.sched.transcribe(aud, request.sr or 16000))
        return ap.Transcript(data=self.lang.denumericalize(tokens))

    def TranscribeStream(self, request_iterator, context):                 # api-server.py:82-134
        st = self._guard(context, lambda: self.sched.open(text_of=self.lang.denumericalize))

        def reader():                       # frames are queued as they arrive; results come back in frame order on st.outq
            try:
                for frame in request_iterator:
                    self.sched.push_nowait(st, tensorize(frame.data)[0].numpy(), frame.sr or 16000)
            except Exception as e:          # client gone / scheduler shut down
                st.outq.put(e)
            finally:
                try:
                    self.sched.push_eof(st)
                except Exception:
                    st.outq.put(EOF)

        threading.Thread(target=reader, daemon=True, name=f"lasr-reader-{st.slot}").start()
        try:
            y, last, last_diff, steps = [], "", "", 0
            while True:
                res = st.outq.get()
                if res is EOF:
                    break
                if isinstance(res, ValueError):
                    context.abort(grpc.StatusCode.INVALID_ARGUMENT, str(res))
                if isinstance(res, Exception):
                    raise res
                if res is None:
                    continue                                               # no model call for this chunk
                steps += 1
                if self.beam > 1:
                    # beam search: the engine hands out the WHOLE current best hypothesis after every model step (it may
                    # change retroactively); this chunk's text is what follows the common prefix (lib/models.py does the same)
                    n = 0
                    while n < min(len(y), len(res)) and y[n] == res[n]:
                        n += 1
                    y, res = list(res), res[n:]
                else:
                    y = y + res
                y_one = self.lang.denumericalize(res)
                if y_one != "":
                    now = self.lang.denumericalize(y)
                    diff = "".join(b for a, b in it.zip_longest(last, now) if a != b)
                    last = now
                    if diff == last_diff:
                        continue
                    last_diff = diff
                    yield ap.Transcript(data=diff)
                # (the reset rule of api-server.py:131-134 is applied by the scheduler, between two model steps of the stream)
        finally:
            self.sched.close(st)


class NativeASRServicer(apg.ASRServicer):
    """The same RPC surface on the NATIVE front (libreasr_amd/front.py, include/lasr.h lasr_front_*): the RPC threads push
    frames into per-stream rings and block in C for their results; batching, steps in flight and the reset rule
    (api-server.py:131-134) run in the library's front thread -- no Python on the tick path.  16 kHz / 80 ms clients take this
    path; other rates / frame lengths run the servicer's per-window sequence (api-server.py:83-115 -> lasr_step_window) with the
    front paused for the call, as the Python scheduler runs them synchronously."""

    def __init__(self, lang, front, language, conf=None):
        self.lang_name, self.front, self.lang = lang, front, language
        eng = front.eng
        self.eng = eng
        self.downsample, self.n_buffer, self.chunk, self.n_window = eng.desc.stride, eng.desc.n_buffer, eng.desc.chunk, eng.desc.n_window
        self.sr = eng.desc.sample_rate

    _guard = staticmethod(ASRServicer._guard)

    def Transcribe(self, request, context):                                # api-server.py:64-80
        aud = tensorize(request.data)[0].numpy()
        sr = request.sr or 16000

        def run():
            with self.front.paused() as eng:
                slot = eng.open()
                try:
                    pcm = aud
                    if sr != self.sr:
                        import torch
                        pcm = eng.resample(torch.as_tensor(aud[None]).to(eng.device), sr)[0]
                    eng.transcribe_pcm([slot], [pcm])
                    return eng.fetch(slot)[0]
                finally:
                    eng.close_slot(slot)

        return ap.Transcript(data=self.lang.denumericalize(self._guard(context, run)))

    def _diffs(self, results):
        """api-server.py:116-130 over an iterator of per-model-step token lists: the servicer's diff + "same diff twice" rule."""
        y, last, last_diff = [], "", ""
        for res in results:
            y = y + res
            if self.lang.denumericalize(res) != "":
                now = self.lang.denumericalize(y)
                diff = "".join(b for a, b in it.zip_longest(last, now) if a != b)
                last = now
                if diff == last_diff:
                    continue
                last_diff = diff
                yield ap.Transcript(data=diff)

    def TranscribeStream(self, request_iterator, context):                 # api-server.py:82-134
        from .front import RES_EOF
        frames = iter(request_iterator)
        try:
            first = next(frames)
        except StopIteration:
            return
        sr0 = first.sr or 16000
        if sr0 != self.sr or len(first.data) != 4 * self.chunk:
            yield from self._stream_generic(tensorize(first.data)[0].numpy(), sr0, frames, context)
            return
        fr = self.front
        sid = self._guard(context, fr.open)
        err = []

        def reader():
            try:
                fr.push(sid, first.data)          # (the wire bytes go straight into the stream's ring: tensorize without the tensor)
                for frame in frames:
                    if (frame.sr or 16000) != self.sr or len(frame.data) != 4 * self.chunk:
                        raise ValueError("a stream keeps the sample rate and frame length of its first frame")
                    fr.push(sid, frame.data)
            except Exception as e:          # client gone / front stopped / bad frame
                err.append(e)
            finally:
                try:
                    fr.eof(sid)
                except Exception as e:
                    err.append(e)

        rt = threading.Thread(target=reader, daemon=True, name=f"lasr-reader-{sid}")
        rt.start()

        def results():
            while True:
                r = fr.next(sid, timeout_ms=500)
                if err and isinstance(err[0], ValueError):
                    context.abort(grpc.StatusCode.INVALID_ARGUMENT, str(err[0]))
                if r is None:
                    continue
                toks, flags = r
                if flags & RES_EOF:
                    if err:
                        raise err[0]
                    return
                yield toks

        try:
            yield from self._diffs(results())
        finally:
            # close releases a reader blocked on a full ring (LASR_ESTATE) and waits until it is out of the library; the id carries
            # the stream's generation, so a reader that wakes up later (client gone, iterator raising) touches nothing of the slot's
            # next stream
            fr.close(sid)
            rt.join(timeout=2.0)

    def _stream_generic(self, pcm0, sr0, frames, context):
        """Other client rates / frame lengths: the servicer's own sequence per window, synchronous, the front paused for the call."""
        eng = self.eng
        with self.front.paused():
            slot = self._guard(context, eng.open)
        buf, steps = [], [0]

        def results():
            for pcm, sr in it.chain([(pcm0, sr0)], ((tensorize(f.data)[0].numpy(), f.sr or 16000) for f in frames)):
                if sr != sr0 or len(pcm) != len(pcm0):
                    context.abort(grpc.StatusCode.INVALID_ARGUMENT, "a stream keeps the sample rate and frame length of its first frame")
                buf.append(pcm)
                if len(buf) != self.n_window:
                    continue
                win = np.concatenate(buf)
                del buf[0]
                try:
                    with self.front.paused():
                        ran = eng.step_window([slot], win[None], sr)       # (the engine keeps the stream's Buffer: ran = the model ran)
                        toks = eng.fetch(slot)[0] if ran else []
                except Exception as e:
                    from ._native import LASR_EINVAL, LasrError
                    if isinstance(e, ValueError) or (isinstance(e, LasrError) and e.code == LASR_EINVAL):
                        context.abort(grpc.StatusCode.INVALID_ARGUMENT, str(e))    # (e.g. windows too short for n_stack frames)
                    raise
                if not ran:
                    continue
                steps[0] += 1
                yield toks
                if self.lang.denumericalize(toks) == "" and should_reset(steps[0], self.downsample, self.n_buffer):
                    with self.front.paused():
                        eng.reset(slot, 1 | 2 | 4)                         # api-server.py:131-134
                    steps[0] = 0

        try:
            yield from self._diffs(results())
        finally:
            with self.front.paused():
                eng.close_slot(slot)


def serve(lang="en", port=None, block=True, depth=None, front=None, **load_kw):
    """Start the gRPC server (api-server.py:138-145).  Returns (server, scheduler, port).  front="native": the library's front
    thread instead of the Python scheduler (the returned object is the NativeFront; `shutdown()` works on both).  depth / front:
    None = the YAML's `engine:` section, else 12 / "python" (lib/config.py:engine_settings)."""
    from .lib.config import engine_settings
    from .lib.inference import load_stuff
    conf, language, model, _, _ = load_stuff(lang, **load_kw)
    es = engine_settings(conf, depth=depth, front=front)
    depth, front = int(es["depth"]), es["front"]
    server = grpc.server(futures.ThreadPoolExecutor(max_workers=WORKERS))
    if front == "native":
        from .front import NativeFront
        k = 1
        while not should_reset(k, model.engine.desc.stride, model.engine.desc.n_buffer) and k < (1 << 20):
            k += 1
        # the rule tests the step's TEXT (api-server.py:124): the ids this tokenizer decodes to "" go to the front at start-up
        sched = NativeFront(model.engine, depth=depth, reset_steps=k,
                            empty_tokens=NativeFront.empty_token_ids(language, model.engine.desc.vocab))
        apg.add_ASRServicer_to_server(NativeASRServicer(lang, sched, language, conf), server)
    else:
        sched = Scheduler(model.engine, depth=depth)
        sched.start()
        apg.add_ASRServicer_to_server(ASRServicer(lang, sched, language, conf), server)
    bound = server.add_insecure_port(port or PORTS[lang])
    server.start()
    print("[api-server] gRPC server running on", port or PORTS[lang], "language", lang)
    if block:
        server.wait_for_termination()
    return server, sched, bound


if __name__ == "__main__":
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument("lang", help="language to serve")
    p.add_argument("--synthetic", default=None, help="seeded synthetic weights of this shape (libreasr_amd.synth.CONFIGS)")
    a = p.parse_args()
    serve(a.lang, synthetic=a.synthetic, max_streams=64)
