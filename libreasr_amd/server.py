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
    def __init__(self, slot):
        self.slot, self.inq, self.outq = slot, collections.deque(), queue.Queue()
        self.n_chunks, self.n_pend = 0, 0
        self.generic = None                 # decided by the first frame: False = 16 kHz / 80 ms fused path, True = window form
        self.frames = []                    # generic form: the last (up to) 3 client frames (api-server.py:85-99)
        # results leave in chunk order: one entry per accepted chunk, [value] once known (None = no model call for the chunk,
        # list = new token ids, Exception); a chunk whose model step is still in flight holds the queue behind it
        self.results = collections.deque()
        self.inflight = 0                   # model steps submitted and not collected
        self.steps = 0                      # model steps since the last reset (api-server.py:117,133)
        self.text_of = None                 # tokens -> text of the chunk (reset policy: "the chunk produced no text")
        self.eof = False
        self.y_prev = []                    # beam search: the best hypothesis as of the previous model step


class Scheduler(threading.Thread):
    """Single owner of the engine (a lasr_ctx is single-caller).  RPC threads only queue PCM.  Every tick the scheduler
    batches ONE chunk of every stream that has one waiting into a single lasr_push_submit (front-end + encoder enqueued, the
    greedy loop running on its own stream), keeps up to `depth` model steps in flight and hands the tokens of every collected
    step (lasr_step_wait) to the streams' result queues -- the protocol bench.py measures.  When nothing else is waiting it
    collects at once, so a lone real-time stream sees the latency of the synchronous protocol.

    The reset policy of the servicer (api-server.py:44-50,131-134: after >= 4000 ms since the last reset, the first model step
    without text resets encoder / predictor / LM) is applied HERE, between two model steps of the stream as the reference does:
    a stream that has reached the threshold is not run ahead (its next chunk waits until the step in flight has been judged).
    Beam search (beam > 1) and generic client frames take the synchronous entry points."""

    def __init__(self, engine, depth=12, downsample=None, n_buffer=None):
        super().__init__(daemon=True, name="lasr-scheduler")
        self.eng, self.cv = engine, threading.Condition()
        self.streams, self.ctl, self.stop_flag = {}, collections.deque(), False
        self.batches = []                   # sizes of the streaming batches (observability / tests)
        self.max_inflight_seen = 0
        self.depth = max(1, min(int(depth), engine.max_inflight()))
        self.beam = engine.beam
        self.inflight = collections.deque() # per submitted model step: the streams whose model ran, in slot-list order
        self.batchq, self.batch_outq = collections.deque(), queue.Queue()     # trunk interface (push_batch)
        self.downsample = downsample or engine.desc.stride
        self.n_buffer = n_buffer or engine.desc.n_buffer

    # ---- called from RPC threads -----------------------------------------------------------
    def _call(self, fn):
        done = queue.Queue()
        with self.cv:
            if self.stop_flag:
                raise RuntimeError("scheduler shut down")
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
        self._call(lambda: self._reset(st))

    def transcribe(self, pcm, sr=16000):
        return self._call(lambda: self._offline(pcm, sr))

    def push_nowait(self, st, chunk, sr=16000):
        """Queue one client frame; its result (None / token list / Exception) appears on st.outq, in frame order."""
        with self.cv:
            if self.stop_flag:
                raise RuntimeError("scheduler shut down")
            st.inq.append((chunk, int(sr) or 16000))
            self.cv.notify()

    def push(self, st, chunk, sr=16000):
        """Queue one frame and wait for its result: None = no model call for this chunk, else the new token ids."""
        self.push_nowait(st, chunk, sr)
        return st.outq.get()

    def push_batch(self, streams, chunks):
        """Trunk interface (one producer feeding many streams, e.g. a bridge that demultiplexes one connection): ONE queue entry
        for a whole batch -- chunks [n, chunk] float32 at the model rate, chunks[i] belongs to streams[i].  Every collected model
        step arrives as ONE item (streams_that_ran, token_lists) on self.batch_outq.  Same engine calls, same reset rule as the
        per-stream form, O(1) queue traffic per batch instead of O(streams); a stream uses either this form or push / push_nowait."""
        with self.cv:
            if self.stop_flag:
                raise RuntimeError("scheduler shut down")
            self.batchq.append((list(streams), chunks))
            self.cv.notify()

    def push_eof(self, st):
        """After the last frame: EOF is put on st.outq behind the result of the last frame."""
        with self.cv:
            st.inq.append(EOF)
            self.cv.notify()

    def shutdown(self):
        with self.cv:
            self.stop_flag = True
            self.cv.notify()

    # ---- scheduler thread ------------------------------------------------------------------
    def _open(self, text_of=None):
        st = _Stream(self.eng.open())
        st.text_of = text_of
        with self.cv:
            self.streams[st.slot] = st
        return st

    def _close(self, st):
        with self.cv:
            self.streams.pop(st.slot, None)
        self.eng.close_slot(st.slot)

    def _reset(self, st):
        self.eng.reset(st.slot, 1 | 2 | 4)                                 # models.py:494-497
        st.steps = 0

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

    def _deliver(self, st, cell, tokens):
        """A model step of `st` has been decoded: its tokens, and the servicer's reset rule (api-server.py:131-134)."""
        st.steps += 1
        cell.append(tokens)
        if st.text_of is not None and isinstance(tokens, list):
            new = tokens
            if self.beam > 1:               # the engine hands out the whole best hypothesis: this chunk's part follows the common prefix
                n = 0
                while n < min(len(st.y_prev), len(tokens)) and st.y_prev[n] == tokens[n]:
                    n += 1
                st.y_prev, new = list(tokens), tokens[n:]
            if st.text_of(new) == "" and should_reset(st.steps, self.downsample, self.n_buffer):
                self._reset(st)             # (the stream has nothing in flight: see _may_run_ahead)
        self._flush(st)

    def _may_run_ahead(self, st):
        # once the reset threshold is within reach of the steps in flight, every further step must see the decision of the one before
        return st.inflight == 0 or st.text_of is None or not should_reset(st.steps + st.inflight + 1, self.downsample, self.n_buffer)

    def _collect(self):
        """Tokens of the oldest model step in flight -> its streams."""
        rows = self.inflight.popleft()
        self.eng.wait()
        toks = self.eng.fetch_many([s.slot for s, _ in rows], cap=256 if self.beam == 1 else 8192)
        if rows and rows[0][1] is None:      # a push_batch step: one item for the whole step
            for (s, _), t in zip(rows, toks):
                s.inflight -= 1
                s.steps += 1
                new = t
                if self.beam > 1:
                    n = 0
                    while n < min(len(s.y_prev), len(t)) and s.y_prev[n] == t[n]:
                        n += 1
                    s.y_prev, new = list(t), t[n:]
                if s.text_of is not None and s.text_of(new) == "" and should_reset(s.steps, self.downsample, self.n_buffer):
                    self._reset(s)
            self.batch_outq.put(([s for s, _ in rows], toks))
            return
        for (s, cell), t in zip(rows, toks):
            s.inflight -= 1
            self._deliver(s, cell, t)

    def _drain(self):
        while self.inflight:
            self._collect()

    def run(self):
        try:
            self._run()
        finally:                             # whoever still waits for a result or a call gets an error, not a hang
            with self.cv:
                self.stop_flag = True
                err = RuntimeError("scheduler shut down")
                for fn, done in self.ctl:
                    done.put(err)
                self.ctl.clear()
                for st in self.streams.values():
                    st.inq.clear()
                    st.outq.put(err)

    def _run(self):
        d = self.eng.desc
        while True:
            with self.cv:
                while (not self.stop_flag and not self.ctl and not self.inflight and not self.batchq
                       and not any(s.inq and (s.inq[0] is EOF or self._may_run_ahead(s)) for s in self.streams.values())):
                    self.cv.wait()
                if self.stop_flag:
                    return
                ctl = list(self.ctl)
                self.ctl.clear()
                batch = None
                if self.batchq and all(self._may_run_ahead(s) for s in self.batchq[0][0]):
                    batch = self.batchq.popleft()
                ready, chunks = [], []
                for s in self.streams.values():
                    if not s.inq:
                        continue
                    if s.inq[0] is EOF:
                        s.inq.popleft()
                        s.eof = True
                        self._flush(s)
                        continue
                    if self._may_run_ahead(s):
                        ready.append(s)
                        chunks.append(s.inq.popleft())
            if ctl:
                self._drain()                # state-changing calls need an idle engine
                for fn, done in ctl:
                    try:
                        done.put(fn())
                    except Exception as e:   # surfaced in the calling RPC thread
                        done.put(e)
            if batch is not None:
                self._submit_batch(*batch)
            if not ready:
                if self.inflight and (batch is None or len(self.inflight) >= self.depth or not self._work_waiting()):
                    self._collect()
                continue
            fast, generic = [], collections.OrderedDict()
            for s, (pcm, sr) in zip(ready, chunks):
                cell = []
                s.results.append(cell)
                if s.generic is None:
                    s.generic = not (sr == d.sample_rate and pcm.shape[0] == d.chunk)
                if not s.generic:
                    if sr != d.sample_rate or pcm.shape[0] > d.chunk:
                        cell.append(ValueError(f"stream opened with {d.chunk}-sample {d.sample_rate} Hz frames: got {pcm.shape[0]} samples at {sr} Hz"))
                        self._flush(s)
                        continue
                    if pcm.shape[0] < d.chunk:                  # api-client.py:40-41 pads the last slice with zeros
                        pcm = np.concatenate([pcm, np.zeros(d.chunk - pcm.shape[0], np.float32)])
                    fast.append((s, cell, pcm))
                else:                                           # api-server.py:85-99: window of the last 3 frames
                    s.frames.append(pcm)
                    if len(s.frames) != d.n_window:
                        cell.append(None)
                        self._flush(s)
                        continue
                    win = np.concatenate(s.frames)
                    del s.frames[0]
                    generic.setdefault((win.shape[0], sr), []).append((s, cell, win))
            try:
                if fast:
                    slots = [s.slot for s, _, _ in fast]
                    self.batches.append(len(fast))
                    ran = []
                    for s, cell, _ in fast:                     # host mirror of the engine's window / Buffer bookkeeping
                        s.n_chunks += 1
                        r = False
                        if s.n_chunks >= d.n_window:            # window full -> one more frame in the Buffer
                            s.n_pend += 1
                            if s.n_pend == d.n_buffer:
                                s.n_pend, r = 0, True
                        ran.append(r)
                    before = self.eng.pending()
                    self.eng.push_submit(slots, np.stack([p for _, _, p in fast]))
                    rows = [(s, cell) for (s, cell, _), r in zip(fast, ran) if r]
                    assert (self.eng.pending() > before) == bool(rows)
                    for (s, cell, _), r in zip(fast, ran):
                        if not r:
                            cell.append(None)
                            self._flush(s)
                    if rows:
                        for s, _ in rows:
                            s.inflight += 1
                        self.inflight.append(rows)
                        self.max_inflight_seen = max(self.max_inflight_seen, len(self.inflight))
            except Exception as e:
                for s, cell, _ in fast:
                    if not cell:
                        cell.append(e)
                    self._flush(s)
            if generic:
                self._drain()                                   # lasr_step_window is a synchronous entry point
            for (N, sr), group in generic.items():              # same window length and rate: one batched call
                try:
                    slots = [s.slot for s, _, _ in group]
                    self.eng.step_window(slots, np.stack([w for _, _, w in group]), sr)
                    toks = self.eng.fetch_many(slots, cap=8192 if self.eng.beam > 1 else 256)
                    for (s, cell, _), t in zip(group, toks):
                        s.n_pend += 1
                        if s.n_pend == d.n_buffer:
                            s.n_pend = 0
                            self._deliver(s, cell, t)
                        else:
                            cell.append(None)
                            self._flush(s)
                except Exception as e:
                    for s, cell, _ in group:
                        if not cell:
                            cell.append(e)
                        self._flush(s)
            # collect: at the depth limit, or as soon as no further chunk is waiting (latency of a lightly loaded server)
            while self.inflight and (len(self.inflight) >= self.depth or not self._work_waiting()):
                self._collect()

    def _submit_batch(self, streams, chunks):
        d = self.eng.desc
        try:
            rows = []
            for s in streams:                                   # host mirror of the engine's window / Buffer bookkeeping
                s.n_chunks += 1
                if s.n_chunks >= d.n_window:
                    s.n_pend += 1
                    if s.n_pend == d.n_buffer:
                        s.n_pend = 0
                        s.inflight += 1
                        rows.append((s, None))
            self.batches.append(len(streams))
            self.eng.push_submit([s.slot for s in streams], chunks)
            if rows:
                self.inflight.append(rows)
                self.max_inflight_seen = max(self.max_inflight_seen, len(self.inflight))
        except Exception as e:
            self.batch_outq.put(e)

    def _work_waiting(self):
        with self.cv:
            if self.batchq and all(self._may_run_ahead(s) for s in self.batchq[0][0]):
                return True
            return bool(self.ctl) or any(s.inq and s.inq[0] is not EOF and self._may_run_ahead(s) for s in self.streams.values())


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


def serve(lang="en", port=None, block=True, depth=12, **load_kw):
    """Start the gRPC server (api-server.py:138-145).  Returns (server, scheduler, port)."""
    from .lib.inference import load_stuff
    conf, language, model, _, _ = load_stuff(lang, **load_kw)
    sched = Scheduler(model.engine, depth=depth)
    sched.start()
    server = grpc.server(futures.ThreadPoolExecutor(max_workers=WORKERS))
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
