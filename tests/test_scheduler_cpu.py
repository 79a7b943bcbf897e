"""CPU tests of the gRPC scheduler's logic (libreasr_amd/server.py) against a FAKE engine: result order per stream, the
servicer's reset rule applied between two model steps of a stream (api-server.py:44-50,131-134), steps in flight, the trunk
interface, end-of-stream, and shutdown (callers must get an error, not hang).  The engine calls themselves are covered
on the GPU (tests/test_gpu_server.py)."""
import collections
import threading
import time
import types

import numpy as np
import pytest

from libreasr_amd import server as srv


class FakeEngine:
    """Front-end bookkeeping of the real engine (3-chunk window, 2-frame Buffer) with 'tokens' that are a pure function of the
    stream's model-step history since its last reset, so any mis-ordering or misplaced reset shows in the output."""

    def __init__(self, max_streams=8, beam=1, inflight=15, silent=(), silent_after=None):
        self.desc = types.SimpleNamespace(sample_rate=16000, chunk=4, n_window=3, n_buffer=2, stride=8)
        self.beam, self._max, self._inflight = beam, max_streams, inflight
        self.open_, self.n_chunks, self.n_pend, self.acc = set(), {}, {}, {}
        self.since_reset = {}
        self.steps, self.queue = collections.deque(), {}
        self.resets = []                    # (slot, model steps the slot had run when it was reset)
        self.total_steps = {}
        self.silent = set(silent)           # slots whose steps produce no tokens
        self.silent_after = silent_after    # beam > 1: model steps since the last reset after which the hypothesis stops growing
        self.hyp = {}                       # beam > 1: the whole best hypothesis since the last reset (what lasr_fetch hands out)
        self.max_pending = 0
        self.lock_owner = None
        self.peek_lag = 0
        self.clock = 0                      # engine calls so far

    def _own(self):                         # a lasr_ctx is single-caller: every call must come from ONE thread
        me = threading.get_ident()
        assert self.lock_owner in (None, me), "engine called from two threads"
        self.lock_owner = me

    def max_inflight(self):
        return self._inflight

    def open(self):
        self._own()
        s = min(set(range(self._max)) - self.open_)
        self.open_.add(s)
        self.n_chunks[s] = self.n_pend[s] = self.since_reset[s] = self.total_steps[s] = 0
        self.acc[s] = 0.0
        self.queue[s] = []
        self.hyp[s] = []
        return s

    def close_slot(self, s):
        self._own()
        assert not any(s in rows for rows in self.steps), "close with a step in flight"
        self.open_.discard(s)

    def peek(self, s):
        """The real engine's lasr_peek_slot: (token lists of the slot's decoded, uncollected steps, steps in flight).  A step counts as
        decoded `peek_lag` engine calls after its submit (0: at once): time passes with every call."""
        self._own()
        assert self.beam == 1
        self.clock += 1
        mine = [out for out in self.steps if s in out]
        done = []
        for out in mine:                    # decoded steps are a prefix
            if self.clock - out["_t"] < self.peek_lag:
                break
            done.append(list(out[s]))
        return done, len(mine)

    def peek_many(self, slots, skip, cap=256, cap_steps=16):
        res, nd, nf = [], [], []
        for s, sk in zip(np.asarray(slots).tolist(), np.asarray(skip).tolist()):
            done, n_in = self.peek(s)
            new = done[sk:]
            self.peek_caps_seen = (cap, cap_steps)
            if len(new) > cap_steps or sum(len(t) for t in new) > cap:      # what lasr_peek_many answers when the caller's buffers are short
                from libreasr_amd._native import LASR_EFULL, LasrError
                self.peek_efull = getattr(self, "peek_efull", 0) + 1
                raise LasrError(LASR_EFULL, "peek buffers too small")
            res.append(new)
            nd.append(len(done))
            nf.append(n_in)
        return res, np.array(nd), np.array(nf)

    def reset(self, s, what, if_decoded=False):
        self._own()
        if if_decoded:
            assert all(self.clock - out["_t"] >= self.peek_lag for out in self.steps if s in out), "reset with an undecoded step in flight"
        else:
            assert not any(s in out for out in self.steps), "reset with a step in flight"
        self.resets.append((s, self.total_steps[s]))
        self.since_reset[s] = 0
        self.hyp[s] = []

    ROWS = False                            # set by the `engine_form` fixture: the engine also offers push_submit_rows / reset_many

    def __getattr__(self, name):            # (hasattr() of the optional entry points follows ROWS)
        if name in ("push_submit_rows", "reset_many") and type(self).ROWS:
            return getattr(self, "_opt_" + name)
        raise AttributeError(name)

    def _opt_push_submit_rows(self, slots, addrs):
        """lasr_push_submit_rows: the chunk of slots[i] is read from host address addrs[i]."""
        import ctypes
        addrs = np.asarray(addrs)
        assert addrs.dtype == np.uint64 and addrs.ndim == 1 and len(addrs) == len(slots)
        self.n_row_pushes = getattr(self, "n_row_pushes", 0) + 1
        ch = self.desc.chunk
        pcm = np.stack([np.ctypeslib.as_array((ctypes.c_float * ch).from_address(int(a))).copy() for a in addrs])
        self.push_submit(slots, pcm)

    def _opt_reset_many(self, slots, what, if_decoded=False):
        slots = np.asarray(slots).tolist()
        assert len(set(slots)) == len(slots)
        self.n_reset_many = getattr(self, "n_reset_many", 0) + 1
        for s in slots:                     # all or nothing: checked before anything changes
            if if_decoded:
                assert all(self.clock - out["_t"] >= self.peek_lag for out in self.steps if s in out), "reset with an undecoded step in flight"
            else:
                assert not any(s in out for out in self.steps), "reset with a step in flight"
        for s in slots:
            self.reset(s, what, if_decoded)

    def push_submit(self, slots, pcm):
        self._own()
        self.clock += 1
        assert len(self.steps) < self._inflight
        rows = []
        for s, ch in zip(slots, np.asarray(pcm)):
            self.n_chunks[s] += 1
            self.acc[s] += float(ch.sum())
            if self.n_chunks[s] >= self.desc.n_window:
                self.n_pend[s] += 1
                if self.n_pend[s] == self.desc.n_buffer:
                    self.n_pend[s] = 0
                    rows.append(s)
        if rows:
            out = {}
            for s in rows:
                self.total_steps[s] += 1
                self.since_reset[s] += 1
                out[s] = [] if s in self.silent else [int(self.acc[s]) % 97, self.since_reset[s]]
                if self.beam > 1:           # the engine hands out the WHOLE hypothesis after every model step
                    if self.silent_after is None or self.since_reset[s] <= self.silent_after:
                        self.hyp[s] = self.hyp[s] + out[s]
                    out[s] = list(self.hyp[s])
            out["_t"] = self.clock
            self.steps.append(out)
            self.max_pending = max(self.max_pending, len(self.steps))

    def pending(self):
        return len(self.steps)

    def wait(self):
        self._own()
        self.clock += 1
        out = self.steps.popleft()
        out.pop("_t", None)
        for s, t in out.items():
            self.queue[s] = t
        return len(out)

    def fetch_many(self, slots, cap=256):
        self._own()
        r = [self.queue[s] for s in slots]
        for s in slots:
            self.queue[s] = []
        return r


def run_stream(sched, st, chunks, out):
    for c in chunks:
        sched.push_nowait(st, c)
    sched.push_eof(st)
    while True:
        r = st.outq.get(timeout=20)
        if r is srv.EOF:
            return
        out.append(r)


def expected(chunks, slot_silent=False, text_rule=True, thresh_steps=25):
    """What the reference servicer's loop would see for one stream of the fake engine (incl. its reset rule)."""
    out, n, pend, acc, since = [], 0, 0, 0.0, 0
    steps = 0
    for c in chunks:
        n += 1
        acc += float(c.sum())
        ran = False
        if n >= 3:
            pend += 1
            if pend == 2:
                pend, ran = 0, True
        if not ran:
            out.append(None)
            continue
        since += 1
        steps += 1
        toks = [] if slot_silent else [int(acc) % 97, since]
        out.append(toks)
        if text_rule and not toks and srv.should_reset(steps, 8, 2):
            since, steps = 0, 0
    return out


@pytest.fixture(autouse=True, params=["matrix", "rows"])
def engine_form(request):
    """Every test runs against an engine with only lasr_push_submit / lasr_stream_reset and against one that also offers
    lasr_push_submit_rows / lasr_stream_reset_many (the scheduler then hands over row addresses and batches a tick's resets)."""
    FakeEngine.ROWS = request.param == "rows"
    yield request.param
    FakeEngine.ROWS = False


def test_results_in_order_steps_in_flight_and_eof():
    eng = FakeEngine()
    sched = srv.Scheduler(eng, depth=4)
    sched.start()
    try:
        rng = np.random.default_rng(0)
        data = [[rng.integers(0, 9, 4).astype(np.float32) for _ in range(41 + 3 * i)] for i in range(5)]
        sts = [sched.open(text_of=lambda t: "x" if t else "") for _ in range(5)]
        outs = [[] for _ in range(5)]
        gate = threading.Event()              # hold the scheduler thread while the producers fill their queues: the ticks behind it
        holder = threading.Thread(target=lambda: sched._call(lambda: gate.wait(10)))       # then see several streams ready at once
        holder.start()
        ths = [threading.Thread(target=run_stream, args=(sched, sts[i], data[i], outs[i])) for i in range(5)]
        [t.start() for t in ths]
        time.sleep(0.2)
        gate.set()
        holder.join(timeout=10)
        [t.join(timeout=30) for t in ths]
        for i in range(5):
            assert outs[i] == expected(data[i]), i
        assert eng.max_pending > 1, "model steps were never in flight together"
        assert max(sched.batches) > 1
        for st in sts:
            sched.close(st)
    finally:
        sched.shutdown()


def test_reset_rule_is_applied_between_two_steps_of_the_stream_not_later():
    """A silent stream crosses the 4 s threshold (25 model steps of 160 ms): the reset must land right after the 25th step
    although the producer has queued every chunk up front and other streams keep the pipeline full."""
    eng = FakeEngine(silent={0})
    sched = srv.Scheduler(eng, depth=8)
    sched.start()
    try:
        rng = np.random.default_rng(1)
        data = [[rng.integers(0, 9, 4).astype(np.float32) for _ in range(140)] for _ in range(3)]
        sts = [sched.open(text_of=lambda t: "x" if t else "") for _ in range(3)]
        assert sts[0].slot == 0
        outs = [[] for _ in range(3)]
        ths = [threading.Thread(target=run_stream, args=(sched, sts[i], data[i], outs[i])) for i in range(3)]
        [t.start() for t in ths]
        [t.join(timeout=30) for t in ths]
        assert outs[0] == expected(data[0], slot_silent=True)
        assert [r for r in eng.resets if r[0] == 0] == [(0, 25), (0, 50)]           # 69 model steps: resets after the 25th and 50th
        assert outs[1] == expected(data[1]) and outs[2] == expected(data[2])
        assert not [r for r in eng.resets if r[0] != 0]
    finally:
        sched.shutdown()


def test_trunk_interface_and_blocking_push():
    eng = FakeEngine(max_streams=16)
    sched = srv.Scheduler(eng, depth=6)
    sched.start()
    try:
        rng = np.random.default_rng(2)
        B, n = 12, 30
        chunks = rng.integers(0, 9, (n, B, 4)).astype(np.float32)
        sts = [sched.open() for _ in range(B)]
        for k in range(n):
            sched.push_batch(sts, chunks[k])
        got = {st.slot: [] for st in sts}
        while sum(len(g) for g in got.values()) < B * ((n - 2) // 2):      # one item per collected model step
            rows, toks = sched.batch_outq.get(timeout=20)
            for st, t in zip(rows, toks):
                got[st.slot].append(t)
        assert sched.step_rows == [B] * ((n - 2) // 2)          # nothing held anybody back: every step ran all the streams
        for i, st in enumerate(sts):
            exp = [t for t in expected([chunks[k, i] for k in range(n)], text_rule=False) if t is not None]
            assert got[st.slot] == exp
        # the blocking form returns each chunk's own result
        st = sched.open()
        res = [sched.push(st, chunks[k, 0]) for k in range(9)]
        assert res == expected([chunks[k, 0] for k in range(9)], text_rule=False)
    finally:
        sched.shutdown()


def test_trunk_streams_at_the_reset_threshold_hold_only_themselves():
    """Trunk form with the servicer's reset rule: silent streams reach the threshold after 25 model steps and are then taken one
    step at a time (each step judged before the next starts) while the other streams keep the pipeline full.  Every stream's
    results and the resets are exactly the per-stream sequence."""
    eng = FakeEngine(max_streams=8, silent={1, 4})
    sched = srv.Scheduler(eng, depth=6)
    sched.start()
    try:
        rng = np.random.default_rng(5)
        B, n = 6, 150
        chunks = rng.integers(0, 9, (n, B, 4)).astype(np.float32)
        sts = [sched.open(text_of=lambda t: "x" if t else "") for _ in range(B)]
        for k in range(n):
            sched.push_batch(sts, chunks[k])
        got = {st.slot: [] for st in sts}
        while sum(len(g) for g in got.values()) < B * ((n - 2) // 2):
            item = sched.batch_outq.get(timeout=20)
            assert not isinstance(item, Exception), item
            for st, t in zip(*item):
                got[st.slot].append(t)
        for i, st in enumerate(sts):
            exp = [t for t in expected([chunks[k, i] for k in range(n)], slot_silent=st.slot in eng.silent) if t is not None]
            assert got[st.slot] == exp, i
        assert sorted(eng.resets) == [(1, 25), (1, 50), (4, 25), (4, 50)]
        assert eng.max_pending > 1
    finally:
        sched.shutdown()


def test_streams_that_start_out_of_phase_end_up_sharing_their_model_steps():
    """A model step costs the GPU the same for one row or all of them.  Half of the streams are one frame ahead of the others
    (their model steps would fall on the other half's fill frames): the step frames of a minority wait one tick, after which
    all streams complete their model steps in the same tick."""
    eng = FakeEngine(max_streams=8)
    sched = srv.Scheduler(eng, depth=4)
    sched.start()
    try:
        rng = np.random.default_rng(6)
        n = 61
        data = [[rng.integers(0, 9, 4).astype(np.float32) for _ in range(n + (i % 2))] for i in range(8)]
        sts = [sched.open() for _ in range(8)]
        outs = [[] for _ in range(8)]
        for i in range(1, 8, 2):                                # the odd streams have had one frame already
            assert sched.push(sts[i], data[i][0]) is None
            outs[i].append(None)
        gate = threading.Event()
        holder = threading.Thread(target=lambda: sched._call(lambda: gate.wait(10)))
        holder.start()
        ths = [threading.Thread(target=run_stream, args=(sched, sts[i], data[i][(i % 2):], outs[i])) for i in range(8)]
        [t.start() for t in ths]
        time.sleep(0.2)
        gate.set()
        holder.join(timeout=10)
        [t.join(timeout=30) for t in ths]
        for i in range(8):
            assert outs[i] == expected(data[i], text_rule=False), i
        assert sched.step_rows.count(8) >= len(sched.step_rows) - 2, sched.step_rows
    finally:
        sched.shutdown()


def test_frames_after_close_are_dropped():
    eng = FakeEngine()
    sched = srv.Scheduler(eng, depth=2)
    sched.start()
    try:
        st = sched.open()
        sched.push_nowait(st, np.ones(4, np.float32))
        sched.close(st)
        sched.push_nowait(st, np.ones(4, np.float32))            # a reader thread that has not noticed yet
        sched.push_eof(st)
        st2 = sched.open()                                       # takes the slot over
        assert st2.slot == st.slot
        res = [sched.push(st2, np.full(4, k, np.float32)) for k in range(6)]
        assert res == expected([np.full(4, k, np.float32) for k in range(6)], text_rule=False)
    finally:
        sched.shutdown()


def test_beam_engine_reset_rule_looks_at_what_the_step_added_to_the_hypothesis():
    """beam > 1: every fetch is the whole best hypothesis; "the chunk produced no text" (api-server.py:131-134) is judged on what
    follows the common prefix with the previous step's hypothesis."""
    eng = FakeEngine(beam=4, silent_after=10)
    sched = srv.Scheduler(eng, depth=5)
    sched.start()
    try:
        rng = np.random.default_rng(7)
        data = [rng.integers(0, 9, 4).astype(np.float32) for _ in range(2 + 2 * 60)]
        st = sched.open(text_of=lambda t: "x" if t else "")
        out = []
        run_stream(sched, st, data, out)
        hyps = [o for o in out if o is not None]
        assert len(hyps) == 60
        # the hypothesis grows for 10 steps, then stands still: the 25th step since the last reset adds nothing -> reset; the same again
        assert eng.resets == [(0, 25), (0, 50)]
        assert len(hyps[9]) == 20 and hyps[24] == hyps[9] and len(hyps[25]) == 2 and hyps[49] == hyps[34] and len(hyps[34]) == 20
    finally:
        sched.shutdown()


@pytest.mark.filterwarnings("ignore::pytest.PytestUnhandledThreadExceptionWarning")      # (the thread re-raises: traceback in the log)
def test_an_engine_error_ends_the_scheduler_and_reaches_every_waiter():
    class Broken(FakeEngine):
        def wait(self):
            raise OSError("device lost")

    eng = Broken()
    sched = srv.Scheduler(eng, depth=2)
    sched.start()
    st = sched.open()
    for k in range(6):
        sched.push_nowait(st, np.full(4, k, np.float32))
    got = []
    while True:
        r = st.outq.get(timeout=20)
        got.append(r)
        if isinstance(r, Exception):
            break
    assert isinstance(got[-1], RuntimeError) and "device lost" in str(got[-1])
    sched.join(timeout=10)
    assert not sched.is_alive() and isinstance(sched.error, OSError)
    with pytest.raises(RuntimeError, match="device lost"):
        sched.open()


def test_random_stream_lifetimes_keep_the_bookkeeping_consistent():
    """Streams of random length opened, fed (per-stream form, with and without the reset rule, some silent), reset and closed from
    several client threads while others run: every stream's results are its own sequence, slots are reused, and the scheduler ends
    with nothing waiting and nothing in flight."""
    eng = FakeEngine(max_streams=6, silent={1, 4})
    sched = srv.Scheduler(eng, depth=5)
    sched.start()
    errors = []

    def client(seed):
        rng = np.random.default_rng(seed)
        try:
            for _ in range(6):
                ruled = bool(rng.integers(0, 2))
                st = sched.open(text_of=(lambda t: "x" if t else "") if ruled else None)
                n = int(rng.integers(1, 90))
                data = [rng.integers(0, 9, 4).astype(np.float32) for _ in range(n)]
                out = []
                if rng.integers(0, 2):
                    run_stream(sched, st, data, out)
                else:                       # the blocking form, one frame at a time
                    out = [sched.push(st, c) for c in data]
                exp = expected(data, slot_silent=st.slot in eng.silent, text_rule=ruled)
                if out != exp:
                    errors.append((seed, st.slot, n, ruled))
                if rng.integers(0, 3) == 0:
                    sched.reset(st)
                sched.close(st)
        except Exception as e:              # pragma: no cover
            errors.append((seed, repr(e)))

    try:
        ths = [threading.Thread(target=client, args=(100 + i,)) for i in range(5)]
        [t.start() for t in ths]
        [t.join(timeout=60) for t in ths]
        assert not any(t.is_alive() for t in ths)
        assert not errors, errors
        assert sched.n_wait == 0 and not sched.inflight and not sched.streams
        assert int(sched.qn.sum()) == 0 and not sched.eofp.any() and sched.n_eof == 0 and sched.n_slow == 0 and sched.n_ruled == 0
        assert int(sched.infl.sum()) == 0
    finally:
        sched.shutdown()


@pytest.mark.parametrize("ruled,close_at", [(False, None), (True, 17), (True, None)])
def test_a_trunk_beside_per_stream_clients_with_a_member_closed_midway(ruled, close_at):
    """A trunk of three streams fed by one producer, two per-stream clients beside it in the same scheduler, one trunk member
    closed while batches keep coming: everybody still gets exactly their own sequence."""
    eng = FakeEngine(max_streams=8, silent={2})
    sched = srv.Scheduler(eng, depth=4)
    sched.start()
    errors = []
    tf = (lambda t: "x" if t else "") if ruled else None
    try:
        rng0 = np.random.default_rng(3)
        B, n = 3, 90
        sts = [sched.open(text_of=tf) for _ in range(B)]
        chunks = rng0.integers(0, 9, (n, B, 4)).astype(np.float32)

        def trunk():
            for k in range(n):
                if close_at is not None and k == close_at:
                    sched.close(sts[1])
                sched.push_batch(sts, chunks[k])

        def client(seed):
            rng = np.random.default_rng(seed)
            for _ in range(3):
                st = sched.open(text_of=tf)
                data = [rng.integers(0, 9, 4).astype(np.float32) for _ in range(int(rng.integers(1, 80)))]
                out = []
                run_stream(sched, st, data, out)
                if out != expected(data, slot_silent=st.slot in eng.silent, text_rule=ruled):
                    errors.append((seed, st.slot))
                sched.close(st)

        ths = [threading.Thread(target=trunk)] + [threading.Thread(target=client, args=(50 + i,)) for i in range(2)]
        [t.start() for t in ths]
        [t.join(timeout=60) for t in ths]
        assert not any(t.is_alive() for t in ths) and not errors, errors
        live = [i for i in range(B) if not (close_at is not None and i == 1)]
        want = {sts[i].slot: [t for t in expected([chunks[k, i] for k in range(n)], slot_silent=sts[i].slot in eng.silent,
                                                  text_rule=ruled) if t is not None] for i in live}
        got = {sts[i].slot: [] for i in range(B)}
        while any(len(got[sl]) < len(want[sl]) for sl in want):
            item = sched.batch_outq.get(timeout=20)
            assert not isinstance(item, Exception), item
            for st, t in zip(*item):
                got[st.slot].append(t)
        for sl in want:
            assert got[sl] == want[sl], sl
        assert sched.n_wait == 0 and not sched.inflight
    finally:
        sched.shutdown()


def test_shutdown_unblocks_everyone():
    eng = FakeEngine()
    sched = srv.Scheduler(eng, depth=2)
    sched.start()
    st = sched.open()
    got = []

    def blocked():
        try:
            got.append(st.outq.get(timeout=20))
        except Exception as e:              # pragma: no cover
            got.append(e)

    t = threading.Thread(target=blocked)
    t.start()
    time.sleep(0.05)
    sched.shutdown()
    t.join(timeout=10)
    assert got and isinstance(got[0], RuntimeError)
    sched.join(timeout=10)
    assert not sched.is_alive()
    with pytest.raises(RuntimeError):
        sched.open()
    with pytest.raises(RuntimeError):
        sched.push_nowait(st, np.zeros(4, np.float32))


def test_a_trunk_whose_slots_are_not_ascending_keeps_every_stream_on_its_own_audio():
    """slot order != row order of the trunk's arrays as soon as a lower slot was freed before a later open().
    Streams in slots [3, 1] (and a per-stream client beside them, so that the tick is not 'exactly the trunk'): every stream's
    tokens are those of ITS chunks."""
    eng = FakeEngine(max_streams=8)
    sched = srv.Scheduler(eng, depth=4)
    sched.start()
    try:
        rng = np.random.default_rng(11)
        n = 24
        sts = [sched.open() for _ in range(4)]                  # slots 0..3
        sched.close(sts[1])
        re = sched.open()                                       # slot 1 again
        assert re.slot == 1
        trunk = [sts[3], re]                                    # slots [3, 1]: descending
        chunks = rng.integers(0, 9, (n, 2, 4)).astype(np.float32)
        solo = [rng.integers(0, 9, 4).astype(np.float32) for _ in range(n)]
        out_solo = []
        t = threading.Thread(target=run_stream, args=(sched, sts[0], solo, out_solo))
        t.start()
        for k in range(n):
            sched.push_batch(trunk, chunks[k])
        got = {s.slot: [] for s in trunk}
        while sum(len(g) for g in got.values()) < 2 * ((n - 2) // 2):
            rows, toks = sched.batch_outq.get(timeout=20)
            for st, tk in zip(rows, toks):
                got[st.slot].append(tk)
        t.join(timeout=20)
        for i, st in enumerate(trunk):
            exp = [x for x in expected([chunks[k, i] for k in range(n)], text_rule=False) if x is not None]
            assert got[st.slot] == exp, f"trunk stream {i} (slot {st.slot}) got another stream's audio"
        assert out_solo == expected(solo, text_rule=False)
        # ... and the all-of-the-trunk tick (no client beside it) with descending slots
        eng2 = FakeEngine(max_streams=8)
        sched2 = srv.Scheduler(eng2, depth=4)
        sched2.start()
        try:
            a = [sched2.open() for _ in range(3)]
            sched2.close(a[0])
            b = sched2.open()                                   # slot 0, opened last
            tr = [a[2], a[1], b]                                # slots [2, 1, 0]
            ch = rng.integers(0, 9, (n, 3, 4)).astype(np.float32)
            for k in range(n):
                sched2.push_batch(tr, ch[k])
            got = {s.slot: [] for s in tr}
            while sum(len(g) for g in got.values()) < 3 * ((n - 2) // 2):
                rows, toks = sched2.batch_outq.get(timeout=20)
                for st, tk in zip(rows, toks):
                    got[st.slot].append(tk)
            for i, st in enumerate(tr):
                exp = [x for x in expected([ch[k, i] for k in range(n)], text_rule=False) if x is not None]
                assert got[st.slot] == exp
        finally:
            sched2.shutdown()
    finally:
        sched.shutdown()


def test_a_long_beam_hypothesis_grows_the_fetch_buffer_instead_of_ending_the_scheduler():
    """with beam > 1 every fetch is the whole best hypothesis; when it outgrows the buffer the engine reports
    LASR_EFULL (and keeps its queues): the scheduler retries with a larger buffer."""
    from libreasr_amd._native import LASR_EFULL, LasrError

    class Strict(FakeEngine):
        def fetch_many(self, slots, cap=256):
            if any(len(self.queue[s]) > cap for s in slots):
                raise LasrError(LASR_EFULL, "token buffer too small")
            return super().fetch_many(slots, cap)

    eng = Strict(beam=4)
    sched = srv.Scheduler(eng, depth=3)
    sched._beam_cap = 8                                         # (as if the stream had been running for hours)
    sched.start()
    try:
        rng = np.random.default_rng(5)
        data = [rng.integers(0, 9, 4).astype(np.float32) for _ in range(2 + 2 * 30)]
        st = sched.open()
        out = []
        run_stream(sched, st, data, out)
        hyps = [o for o in out if o is not None]
        assert len(hyps) == 30 and len(hyps[-1]) == 60
        assert sched._beam_cap >= 60 and sched.error is None
    finally:
        sched.shutdown()


@pytest.mark.parametrize("lag", [0, 2])
def test_early_verdicts_keep_held_streams_in_the_batch_and_place_resets_where_the_reference_does(lag):
    """Round 4: with lasr_peek_slot the verdict on a held stream's step arrives when its row is decoded (here: `lag` submits after
    its own), not `depth` collections later.  Same tokens, same resets as the reference's rule (expected()); and the streams at
    the threshold miss far fewer model steps than without peek."""
    rng = np.random.default_rng(9)
    B, n = 8, 170
    chunks = rng.integers(0, 9, (n, B, 4)).astype(np.float32)

    def run(can_peek):
        eng = FakeEngine(max_streams=8, silent={1, 4, 6})
        eng.peek_lag = lag
        sched = srv.Scheduler(eng, depth=8)
        if not can_peek:
            sched.can_peek, sched.held_depth = False, 3
        sched.start()
        try:
            sts = [sched.open(text_of=lambda t: "x" if t else "") for _ in range(B)]
            for k in range(n):
                sched.push_batch(sts, chunks[k])
            got = {st.slot: [] for st in sts}
            while sum(len(g) for g in got.values()) < B * ((n - 2) // 2):
                item = sched.batch_outq.get(timeout=20)
                assert not isinstance(item, Exception), item
                for st, t in zip(*item):
                    got[st.slot].append(t)
            for i, st in enumerate(sts):
                exp = [t for t in expected([chunks[k, i] for k in range(n)], slot_silent=st.slot in eng.silent) if t is not None]
                assert got[st.slot] == exp, (can_peek, i)
            assert sorted(eng.resets) == sorted((s, k) for s in (1, 4, 6) for k in (25, 50, 75))
            assert sched.error is None
            return float(np.mean(sched.step_rows)), len(sched.step_rows)
        finally:
            sched.shutdown()

    rows_peek, steps_peek = run(True)
    rows_old, steps_old = run(False)
    assert steps_peek <= steps_old and rows_peek >= rows_old
    assert rows_peek > 0.9 * B, (rows_peek, rows_old)         # held streams rejoin within a step or two


def test_row_addresses_and_batched_resets_give_what_the_matrix_form_gives():
    """The same replay (trunk of 6 streams, two silent ones that hit the reset threshold in the same model step and are then held
    while the others run on, so the streams of the trunk stand at different batches) against an engine with and without
    lasr_push_submit_rows / lasr_stream_reset_many: same tokens, same resets; the address form was really used, and the two
    silent streams' first reset went out as ONE call."""
    res = {}
    for form in (False, True):
        FakeEngine.ROWS = form
        eng = FakeEngine(max_streams=8, silent={1, 4})
        eng.peek_lag = 9                     # a verdict takes a few ticks: the other streams run ahead meanwhile
        sched = srv.Scheduler(eng, depth=6)
        sched.start()
        try:
            rng = np.random.default_rng(5)
            B, n = 6, 150
            chunks = rng.integers(0, 9, (n, B, 4)).astype(np.float32)
            sts = [sched.open(text_of=lambda t: "x" if t else "") for _ in range(B)]
            for k in range(n):
                sched.push_batch(sts, chunks[k].copy())          # a fresh array per call, as the interface asks
            got = {st.slot: [] for st in sts}
            while sum(len(g) for g in got.values()) < B * ((n - 2) // 2):
                rows, toks = sched.batch_outq.get(timeout=20)
                for st, t in zip(rows, toks):
                    got[st.slot].append(t)
            res[form] = (got, sorted(eng.resets))
            if form:
                assert getattr(eng, "n_row_pushes", 0) > 10
                assert getattr(eng, "n_reset_many", 0) >= 1
            else:
                assert not hasattr(eng, "n_row_pushes")
        finally:
            sched.shutdown()
            sched.join(timeout=10)
    assert res[False] == res[True]
    assert len(res[True][1]) >= 4


@pytest.mark.parametrize("lag", [0, 5])
def test_two_trunks_fed_at_different_paces_with_the_reset_rule(lag):
    """Two trunks (3 + 4 streams; the second starts 7 batches later and one of its streams is opened into a freed lower slot, so its
    rows are not in slot order) with silent members that reach the reset threshold: every stream sees exactly what the per-stream
    rule gives on its own audio -- with a contiguous matrix per tick or row addresses, verdicts at once or a few engine calls late."""
    eng = FakeEngine(max_streams=10, silent={1, 5})
    eng.peek_lag = lag
    sched = srv.Scheduler(eng, depth=5)
    sched.start()
    try:
        rng = np.random.default_rng(11)
        n = 140
        text = lambda t: "x" if t else ""
        a = [sched.open(text_of=text) for _ in range(3)]                  # slots 0, 1, 2
        filler = sched.open()                                            # slot 3
        b_hi = [sched.open(text_of=text) for _ in range(3)]              # slots 4, 5, 6
        sched.close(filler)
        b = [b_hi[2], sched.open(text_of=text), b_hi[0], b_hi[1]]        # slots 6, 3, 4, 5: not ascending
        ca = rng.integers(0, 9, (n, 3, 4)).astype(np.float32)
        cb = rng.integers(0, 9, (n, 4, 4)).astype(np.float32)
        for k in range(n + 7):
            if k < n:
                sched.push_batch(a, ca[k].copy())
            if k >= 7:
                sched.push_batch(b, cb[k - 7].copy())
        got = {st.slot: [] for st in a + b}
        want = (len(a) + len(b)) * ((n - 2) // 2)
        while sum(len(g) for g in got.values()) < want:
            item = sched.batch_outq.get(timeout=20)
            assert not isinstance(item, Exception), item
            for st, t in zip(*item):
                got[st.slot].append(t)
        for i, st in enumerate(a):
            exp = [t for t in expected([ca[k, i] for k in range(n)], slot_silent=st.slot in eng.silent) if t is not None]
            assert got[st.slot] == exp, ("a", i)
        for i, st in enumerate(b):
            exp = [t for t in expected([cb[k, i] for k in range(n)], slot_silent=st.slot in eng.silent) if t is not None]
            assert got[st.slot] == exp, ("b", i)
        assert sorted(s for s, _ in eng.resets) == [1, 1, 5, 5]
    finally:
        sched.shutdown()
        sched.join(timeout=10)


def test_early_verdict_buffers_follow_the_depth_and_a_short_buffer_is_not_fatal():
    """at depth 13-15 a held slot can have more decoded tokens than the fixed 256-token / 16-step peek buffers held;
    the LASR_EFULL then ended the scheduler for every stream.  The buffers are sized from the depth now, and a short buffer
    (forced here) costs the tick its early verdicts -- they fall back to collect time -- not the scheduler."""
    eng = FakeEngine(max_streams=8, silent={1, 4})
    eng.peek_lag = 0
    sched = srv.Scheduler(eng, depth=15)
    assert sched._peek_cap >= 15 * 20 and sched._peek_cap >= 256
    sched._peek_cap = 0                      # force the overflow path on the first peek
    sched.start()
    try:
        rng = np.random.default_rng(3)
        B, n = 6, 120
        chunks = rng.integers(0, 9, (n, B, 4)).astype(np.float32)
        sts = [sched.open(text_of=lambda t: "x" if t else "") for _ in range(B)]
        for k in range(n):
            sched.push_batch(sts, chunks[k].copy())
        got = {st.slot: [] for st in sts}
        while sum(len(g) for g in got.values()) < B * ((n - 2) // 2):
            item = sched.batch_outq.get(timeout=20)
            assert not isinstance(item, Exception), item
            for st, t in zip(*item):
                got[st.slot].append(t)
        for i, st in enumerate(sts):
            exp = [t for t in expected([chunks[k, i] for k in range(n)], slot_silent=st.slot in eng.silent) if t is not None]
            assert got[st.slot] == exp, i
        assert sched.error is None
        assert getattr(eng, "peek_efull", 0) >= 1 and sched._peek_cap > 0     # the overflow happened and the buffer grew
    finally:
        sched.shutdown()
