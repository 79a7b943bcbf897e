"""Native serving front (include/lasr.h: lasr_front_*): per-stream producers without Python on the tick path.

A `NativeFront` owns the engine's streaming entry points while it exists: RPC / producer threads call `push` (one foreign call
per chunk or per run of chunks, the GIL released inside) and `next` (blocks in C until the stream's next model step has been
decoded); one native thread batches the streams, keeps model steps in flight and applies the servicer's reset rule
(api-server.py:44-50,131-134).  `libreasr_amd.server.serve(front="native")` puts it behind the gRPC servicer."""
import contextlib
import ctypes as C
import threading

import numpy as np

from . import _native as N

RES_STEP, RES_RESET, RES_EOF = 1, 2, 4


class NativeFront:
    def __init__(self, engine, depth=12, reset_steps=0, empty_tokens=None):
        self.eng, self.lib = engine, engine.lib
        self.chunk = int(engine.desc.chunk)
        h = C.c_void_p()
        engine._chk(self.lib.lasr_front_create(engine.ctx, int(depth), int(reset_steps), C.byref(h)))
        self.h = h
        self._cap = max(64, int(engine.desc.n_buffer) * int(engine.desc.max_iters_stream) + 4)
        if not hasattr(engine, "_fronts"):
            engine._fronts = []
        engine._fronts.append(self)          # Engine.close() destroys the front before the context its threads use
        # calls in progress on other threads: destroy() frees the handle only when none is left (see shutdown)
        self._lk = threading.Condition()
        self._inside = 0
        self._stopped = False
        if empty_tokens:
            self.set_empty_tokens(empty_tokens)

    @contextlib.contextmanager
    def _call(self):
        with self._lk:
            if self.h is None or self._stopped:
                raise N.LasrError(N.LASR_ESTATE, "the front has been stopped")
            self._inside += 1
        try:
            yield self.h
        finally:
            with self._lk:
                self._inside -= 1
                if self._inside == 0:
                    self._lk.notify_all()

    def _chk(self, rc):
        if rc < 0:
            msg = self.lib.lasr_front_error(self.h) or self.lib.lasr_last_error(self.eng.ctx) or b""
            raise N.LasrError(rc, msg.decode(errors="replace"))
        return rc

    def set_empty_tokens(self, ids):
        """Token ids whose piece decodes to "" in the servicer's tokenizer: the reset rule then judges a step by its TEXT
        (api-server.py:124 `y_one != ""`), not by its token count.  Call before streams are opened."""
        a = np.ascontiguousarray(sorted(set(int(i) for i in ids)), dtype=np.int32)
        with self._call() as h:
            self._chk(self.lib.lasr_front_set_empty_tokens(h, a.ctypes.data if a.size else None, int(a.size)))

    @staticmethod
    def empty_token_ids(language, vocab):
        """ids of `language` (denumericalize: list[int] -> str) that decode to the empty string on their own."""
        out = []
        for i in range(int(vocab)):
            try:
                if language.denumericalize([i]) == "":
                    out.append(i)
            except Exception:
                pass
        return out

    def open(self):
        s = C.c_int(-1)
        with self._call() as h:
            self._chk(self.lib.lasr_front_open(h, C.byref(s)))
        return int(s.value)

    def push(self, stream, pcm):
        """pcm: k >= 1 whole client chunks as a float32 host array OR as the raw little-endian float32 bytes of the wire format
        (libreasr.proto Audio.data: no numpy / torch object on the RPC thread's path); copied before the call returns."""
        if isinstance(pcm, (bytes, bytearray, memoryview)):
            n = len(pcm) // 4
            if n == 0 or len(pcm) % 4 or n % self.chunk:
                raise ValueError(f"push takes whole chunks of {self.chunk} float32 samples")
            with self._call() as h:
                self._chk(self.lib.lasr_front_push(h, int(stream), bytes(pcm) if not isinstance(pcm, bytes) else pcm, n // self.chunk))
            return
        a = np.ascontiguousarray(pcm, dtype=np.float32).reshape(-1)
        if a.size == 0 or a.size % self.chunk:
            raise ValueError(f"push takes whole chunks of {self.chunk} samples")
        with self._call() as h:
            self._chk(self.lib.lasr_front_push(h, int(stream), a.ctypes.data, a.size // self.chunk))

    def eof(self, stream):
        with self._call() as h:
            self._chk(self.lib.lasr_front_eof(h, int(stream)))

    def next(self, stream, timeout_ms=-1):
        """-> (tokens, flags) of the stream's next model step, or None on time-out.  flags: RES_STEP | RES_RESET | RES_EOF."""
        buf = (C.c_int32 * self._cap)()
        n, fl = C.c_int(0), C.c_int(0)
        with self._call() as h:
            rc = self._chk(self.lib.lasr_front_next(h, int(stream), buf, self._cap, C.byref(n), C.byref(fl), int(timeout_ms)))
        if rc == 1:
            return None
        return list(buf[:n.value]), int(fl.value)

    def close(self, stream):
        with self._call() as h:
            self._chk(self.lib.lasr_front_close(h, int(stream)))

    @contextlib.contextmanager
    def paused(self):
        """The engine for the caller (unary Transcribe): every step in flight is collected first, the front thread stays out."""
        with self._call() as h:
            self._chk(self.lib.lasr_front_pause(h))
            try:
                yield self.eng
            finally:
                self.lib.lasr_front_resume(h)

    def stats(self):
        v = [C.c_longlong(0) for _ in range(4)]
        with self._call() as h:
            self._chk(self.lib.lasr_front_stats(h, *[C.byref(x) for x in v]))
        return dict(zip(("ticks", "steps", "rows", "resets"), (int(x.value) for x in v)))

    def stop(self):
        """Releases every thread blocked in push / next (they raise) and stops the front thread; the handle stays valid: join
        those threads, then destroy()."""
        with self._lk:
            if self.h is None or self._stopped:
                return
            self._stopped = True              # (no new call gets in; the ones inside return LASR_ESTATE)
        self.lib.lasr_front_stop(self.h)

    def destroy(self, timeout=10.0):
        """stop(), wait until no other thread is inside a call any more, then free the front (lasr_front_destroy: collects what
        is in flight, closes its streams, joins its threads)."""
        self.stop()
        with self._lk:
            if self.h is None:
                return
            if not self._lk.wait_for(lambda: self._inside == 0, timeout):
                raise RuntimeError(f"NativeFront.destroy: {self._inside} call(s) still inside the front after {timeout} s")
            h, self.h = self.h, None
        self.lib.lasr_front_destroy(h)
        if self in getattr(self.eng, "_fronts", ()):
            self.eng._fronts.remove(self)

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    shutdown = destroy                      # (same name as the Python scheduler's)


def bench_native_producers(engine, pcm, depth=12, reset_steps=0, chunks_per_push=1, cap=2048):
    """lasr_bench_front: one NATIVE producer thread per stream (no Python on either side of the front).  pcm [n_streams, n_chunks *
    chunk] float32 -> (token lists, seconds, stats)."""
    a = np.ascontiguousarray(pcm, dtype=np.float32)
    B, n = a.shape[0], a.shape[1] // int(engine.desc.chunk)
    tok = np.zeros((B, cap), np.int32)
    cnt = np.zeros(B, np.int32)
    sec = C.c_double(0.0)
    st = (C.c_longlong * 4)()
    engine._chk(engine.lib.lasr_bench_front(engine.ctx, int(depth), int(reset_steps), B, a.ctypes.data, n, int(chunks_per_push),
                                            tok.ctypes.data, cap, cnt.ctypes.data, C.byref(sec), st))
    return [tok[i, :cnt[i]].tolist() for i in range(B)], float(sec.value), dict(zip(("ticks", "steps", "rows", "resets"), (int(v) for v in st)))
