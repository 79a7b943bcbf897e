// lasr_front.hip.h -- native serving front: the per-stream producer side of a server without
// Python on the tick path.  Part of the engine unit lasr_engine.hip, included LAST: it is written against the public
// C ABI of include/lasr.h only (lasr_push_submit_rows, lasr_step_wait, lasr_fetch_many, lasr_stream_*), like an application.
//
// What it replaces: the reference servicer runs each TranscribeStream RPC on one of 4 Python worker threads, batch 1
// (api-server.py:82-135,139); round 3-4's libreasr_amd.server.Scheduler batches the streams in ONE Python thread and reaches
// 41 k audio-s/s through the trunk form but 9 k through per-stream producer threads (a queue operation per 80 ms chunk and stream
// under the GIL).  Here every stream has a single-producer ring of client chunks in host memory; ONE native thread owns the
// engine: every tick it takes one chunk of every stream that has one waiting, hands their ring addresses to
// lasr_push_submit_rows, keeps up to `depth` model steps in flight and delivers the tokens of every collected step to the
// streams' result queues.  A Python (or any other) producer pays one foreign call per push -- several chunks per call when it
// has them -- and one blocking call per result, both outside the GIL.
//
// Reset rule of the servicer (api-server.py:44-50,131-134), applied between two model steps of the stream as the reference does:
// after `reset_steps` model steps since the last reset, the first step that emits no token resets encoder / predictor / LM.
// A stream whose step in flight may trigger a reset is not run ahead: the chunk that would start its next model step waits for
// that verdict (chunks that only fill the window / Buffer go through); all other streams keep going.  Streams whose
// step-completing chunk falls on a different tick than the majority's wait one tick, so the streams share their model steps.
#pragma once

struct lasr_front {
    struct Stream {
        int slot = -1;
        std::atomic<bool> open{false};
        std::atomic<bool> closing{false};     // set by lasr_front_close: the front thread takes nothing more, a blocked producer leaves
        // stream id = slot | generation << 16: a stale id (a reader thread of the slot's PREVIOUS stream) matches no open stream
        std::atomic<int> gen{0};
        std::atomic<int> inside{0};           // producer calls (push / eof) between their id check and their return: close waits for 0
        std::atomic<long long> last_push_ns{0};   // steady clock of the last chunk pushed (thin-batch rule: who counts as "about to push")
        // input: single-producer / single-consumer ring of client chunks (host memory, `chunk` floats each)
        std::vector<float> ring;
        std::atomic<long long> head{0}, tail{0};     // chunks produced / consumed
        std::atomic<bool> eof_in{false};
        bool eof_out = false;
        // mirrors of the engine's window / Buffer bookkeeping, and the rule's counters (front thread only)
        long long n_chunks = 0; int n_pend = 0;
        int infl = 0; long long stp = 0;
        int judged = 0;                       // steps in flight already judged (and counted in stp) through an early verdict
        std::deque<char> judged_reset;        // ... one flag per judged step, oldest first: that step ended in a reset (reported with its result)
        // results: one entry per collected model step
        std::mutex rm; std::condition_variable rcv;
        struct Res { std::vector<int32_t> tok; int flags; };
        std::deque<Res> res;
        std::condition_variable pcv;          // producer waiting for ring space / closer waiting for infl == 0 (under rm)
    };
    lasr_ctx* c = nullptr;
    int depth = 12, reset_steps = 0, ring_chunks = 64, chunk = 0, n_window = 0, n_buffer = 0, max_tok = 0;
    std::vector<std::unique_ptr<Stream>> st;  // by engine slot
    std::mutex em;                            // the engine is single-caller: front thread vs open / close / pause
    std::atomic<int> em_waiters{0};           // callers waiting for em: the front thread, which re-takes em every tick, lets them in
    std::thread th;
    std::atomic<bool> stop{false};
    std::atomic<long long> work{0};           // bumped by every push / eof / close: the sleeping front thread re-checks
    std::mutex fm; std::condition_variable fcv; std::atomic<bool> sleeping{false};
    std::deque<std::vector<int>> inflight;    // rows (slots) of every submitted, uncollected model step
    // results leave through a second thread: waking 64 blocked consumers (one futex wake each) costs the thread that does it
    // 60-190 us per model step -- on the front thread that made the HOST the limit (33 k audio-s/s with full 64-row steps)
    struct Out { std::vector<int> rows; std::vector<int32_t> tok; std::vector<int> cnt, flags; int cap = 0; };
    std::deque<Out> outq; std::mutex om; std::condition_variable ocv; std::thread oth;
    bool out_busy = false;                    // the delivery thread holds a record it has taken off outq (under om)
    std::atomic<int> rc{0}; std::string err;  // first engine error: the front stops, every call returns it (err is written before rc: under em)
    std::atomic<long long> n_ticks{0}, n_steps{0}, n_rows{0}, n_resets{0};
    bool early = true;
    // reset rule on TEXT (api-server.py:124-133: `y_one != ""`): ids whose pieces decode to the empty string (a lone word-boundary
    // piece of a BPE vocabulary); a step whose tokens are all in the set counts as empty.  Empty set = "no token" (IdLanguage)
    std::vector<unsigned char> empty_tok;
    int held_last = 0;                        // streams the last tick found held at the reset threshold (early verdicts are tried for them)
    std::vector<int> ev_slots, ev_skip, ev_cnt, ev_ndec, ev_nfl; std::vector<int32_t> ev_tok;
    // scratch of the front thread
    std::vector<int> slots, step_rows; std::vector<const float*> rows; std::vector<int32_t> tokbuf; std::vector<int> cnt;
};

namespace {

// em for a caller thread (open / close / pause).  A plain std::mutex is not fair: the front thread unlocks and re-locks it within
// nanoseconds while it has work, and a sleeping waiter could starve for as long as the replay lasts; the waiter announces itself
// and the front thread stays out until it has been served (front_main).
struct FrontCallerLock {
    lasr_front* f;
    explicit FrontCallerLock(lasr_front* f_) : f(f_) { f->em_waiters.fetch_add(1, std::memory_order_acq_rel); f->em.lock(); f->em_waiters.fetch_sub(1, std::memory_order_acq_rel); }
    ~FrontCallerLock() { f->em.unlock(); }
};

constexpr int FRONT_RES_STEP = 1, FRONT_RES_RESET = 2, FRONT_RES_EOF = 4;

inline long long front_now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline int front_id(const lasr_front::Stream& s) { return s.slot | (s.gen.load(std::memory_order_relaxed) << 16); }
// the open stream an id names, or nullptr (unknown slot, closed, or the id of an earlier stream of the slot)
lasr_front::Stream* front_stream(lasr_front* f, int id) {
    if (!f || id < 0) return nullptr;
    const int slot = id & 0xffff, gen = id >> 16;
    if (slot >= (int)f->st.size() || !f->st[slot]) return nullptr;
    lasr_front::Stream* s = f->st[slot].get();
    if (s->gen.load(std::memory_order_acquire) != gen || !s->open.load(std::memory_order_acquire)) return nullptr;
    return s;
}
// does the step's text decode to ""?  (no token, or only pieces of the tokenizer's empty set)
inline bool front_step_empty(const lasr_front* f, const int32_t* tok, int n) {
    for (int i = 0; i < n; ++i)
        if (tok[i] < 0 || tok[i] >= (int)f->empty_tok.size() || !f->empty_tok[tok[i]]) return false;
    return true;
}

int front_fail(lasr_front* f, int rc) {
    if (!f->rc.load(std::memory_order_acquire)) { f->err = lasr_last_error(f->c); f->rc.store(rc, std::memory_order_release); }
    for (auto& s : f->st)                   // wake every waiter: they return the error
        if (s) { std::lock_guard<std::mutex> lk(s->rm); s->rcv.notify_all(); s->pcv.notify_all(); }
    return rc;
}

void front_deliver(lasr_front::Stream& s, const int32_t* tok, int n, int flags) {
    std::lock_guard<std::mutex> lk(s.rm);
    s.res.push_back({std::vector<int32_t>(tok, tok + n), flags});
    s.rcv.notify_one();
}

void front_out(lasr_front* f, lasr_front::Out&& o) {
    { std::lock_guard<std::mutex> lk(f->om); f->outq.push_back(std::move(o)); }
    f->ocv.notify_one();
}
// delivery thread: the results of collected steps (and end-of-stream marks, in order behind them) -> the streams' queues
void front_out_main(lasr_front* f) {
    for (;;) {
        lasr_front::Out o;
        {
            std::unique_lock<std::mutex> lk(f->om);
            f->ocv.wait(lk, [&] { return !f->outq.empty() || f->stop.load(); });
            if (f->outq.empty()) return;
            o = std::move(f->outq.front());
            f->outq.pop_front();
            f->out_busy = true;
        }
        for (size_t i = 0; i < o.rows.size(); ++i)
            front_deliver(*f->st[o.rows[i]], o.tok.data() + i * (size_t)o.cap, o.cnt[i], o.flags[i]);
        { std::lock_guard<std::mutex> lk(f->om); f->out_busy = false; }
    }
}

// tokens of the oldest model step in flight -> its streams; the reset rule.  em held.
int front_collect(lasr_front* f) {
    if (f->inflight.empty()) return LASR_OK;
    int ran = 0;
    int rc = lasr_step_wait(f->c, &ran);
    if (rc) return front_fail(f, rc);
    std::vector<int> rows = std::move(f->inflight.front());
    f->inflight.pop_front();
    if (ran != (int)rows.size()) {
        fail(f->c, LASR_ESTATE, "front: expected a model step of %d streams, the engine ran %d", (int)rows.size(), ran);
        return front_fail(f, LASR_ESTATE);
    }
    const int cap = f->max_tok;
    f->tokbuf.resize((size_t)rows.size() * cap);
    f->cnt.resize(rows.size());
    rc = lasr_fetch_many(f->c, rows.data(), (int)rows.size(), f->tokbuf.data(), cap, f->cnt.data());
    if (rc) return front_fail(f, rc);
    lasr_front::Out o;
    o.cap = cap; o.flags.assign(rows.size(), FRONT_RES_STEP);
    for (size_t i = 0; i < rows.size(); ++i) {
        lasr_front::Stream& s = *f->st[rows[i]];
        s.infl--;
        if (s.judged > 0) {                  // judged (and counted, and reset if the rule said so) when its row was decoded
            s.judged--;
            if (!s.judged_reset.empty()) { if (s.judged_reset.front()) o.flags[i] |= FRONT_RES_RESET; s.judged_reset.pop_front(); }
            continue;
        }
        s.stp++;
        if (f->reset_steps > 0 && s.stp >= f->reset_steps && front_step_empty(f, f->tokbuf.data() + i * (size_t)cap, f->cnt[i])) {
            // (the stream has nothing else in flight: see the hold rule in front_tick)
            rc = lasr_stream_reset(f->c, s.slot, 1 | 2 | 4);            // models.py:494-497
            if (rc) return front_fail(f, rc);
            s.stp = 0;
            o.flags[i] |= FRONT_RES_RESET;
            f->n_resets.fetch_add(1, std::memory_order_relaxed);
        }
    }
    o.rows = std::move(rows); o.tok.swap(f->tokbuf); o.cnt.swap(f->cnt);
    front_out(f, std::move(o));
    return LASR_OK;
}

// Early verdicts: a stream held at the reset threshold is judged as soon as its row is decoded (lasr_peek_many), not when its
// step is collected `steps in flight` later; a reset the rule asks for is applied at once (LASR_RESET_IF_DECODED: the slot's
// steps are decoded, not collected) and the stream's next step can start.  em held.
int front_early_verdicts(lasr_front* f) {
    f->ev_slots.clear(); f->ev_skip.clear();
    for (auto& sp : f->st) {
        if (!sp || !sp->open.load(std::memory_order_acquire) || sp->closing.load(std::memory_order_acquire)) continue;
        const int unj = sp->infl - sp->judged;
        if (unj > 0 && sp->stp + unj >= f->reset_steps) { f->ev_slots.push_back(sp->slot); f->ev_skip.push_back(sp->judged); }
    }
    const int n = (int)f->ev_slots.size();
    if (!n) return LASR_OK;
    const int cap = f->max_tok * (f->depth + 1), cap_steps = f->depth + 1;
    f->ev_tok.resize((size_t)n * cap); f->ev_cnt.assign((size_t)n * cap_steps, 0); f->ev_ndec.assign(n, 0); f->ev_nfl.assign(n, 0);
    int rc = lasr_peek_many(f->c, f->ev_slots.data(), n, f->ev_skip.data(), f->ev_tok.data(), cap, f->ev_cnt.data(), cap_steps,
                            f->ev_ndec.data(), f->ev_nfl.data());
    if (rc == LASR_EFULL) return LASR_OK;                 // (verdicts fall back to collect time)
    if (rc) return front_fail(f, rc);
    for (int q = 0; q < n; ++q) {
        lasr_front::Stream& s = *f->st[f->ev_slots[q]];
        const int k_new = f->ev_ndec[q] - f->ev_skip[q];
        const int32_t* tq = f->ev_tok.data() + (size_t)q * cap;      // the decoded steps' tokens, step after step
        for (int k = 0; k < k_new; ++k) {
            const int nk = f->ev_cnt[(size_t)q * cap_steps + k];
            s.judged++;
            s.stp++;
            s.judged_reset.push_back(0);
            const bool empty = front_step_empty(f, tq, nk);
            tq += nk;
            if (s.stp >= f->reset_steps && empty) {
                // past the threshold a stream has ONE unjudged step in flight at a time: this was its last, and it is decoded
                if (s.judged != f->ev_nfl[q]) {
                    fail(f->c, LASR_ESTATE, "front: slot %d ran ahead of the reset threshold (%d steps in flight, %d judged)", s.slot, f->ev_nfl[q], s.judged);
                    return front_fail(f, LASR_ESTATE);
                }
                rc = lasr_stream_reset(f->c, s.slot, 1 | 2 | 4 | LASR_RESET_IF_DECODED);
                if (rc) return front_fail(f, rc);
                s.stp = 0;
                s.judged_reset.back() = 1;
                f->n_resets.fetch_add(1, std::memory_order_relaxed);
            }
        }
    }
    return LASR_OK;
}

// one tick: one chunk of every stream that has one waiting and may run.  Returns 1 when something was done.  em held.
int front_tick(lasr_front* f, bool* did) {
    *did = false;
    lasr_ctx* c = f->c;
    while ((int)f->inflight.size() >= f->depth) { int rc = front_collect(f); if (rc) return rc; *did = true; }
    f->slots.clear(); f->rows.clear(); f->step_rows.clear();
    int n_step = 0, n_fill = 0, n_absent = 0, n_held = 0;
    const long long now_ns = front_now_ns();
    if (f->reset_steps > 0 && f->early && f->held_last > 0) { int rc = front_early_verdicts(f); if (rc) return rc; }
    // pass 1: classify
    struct Cand { int slot; bool stepper; };
    static thread_local std::vector<Cand> cand;
    cand.clear();
    for (auto& sp : f->st) {
        if (!sp || !sp->open.load(std::memory_order_acquire) || sp->closing.load(std::memory_order_acquire)) continue;
        lasr_front::Stream& s = *sp;
        if (s.head.load(std::memory_order_acquire) == s.tail.load(std::memory_order_relaxed)) {
            if (!s.eof_in.load(std::memory_order_acquire)) {
                // a live stream with nothing waiting counts as "about to push" only if it pushed within the last 2 ms (a producer that
                // is behind by a turn on the GIL): an idle or real-time client must not hold the others at 2 steps in flight
                if (now_ns - s.last_push_ns.load(std::memory_order_relaxed) < 2000000LL) n_absent++;
                continue;
            }
            if (!s.eof_out && s.infl == 0) {                                                  // everything delivered
                s.eof_out = true;
                lasr_front::Out o;                 // (behind the stream's last step result, through the same queue)
                o.rows = {s.slot}; o.cnt = {0}; o.flags = {FRONT_RES_EOF}; o.cap = 1; o.tok.assign(1, 0);
                front_out(f, std::move(o));
                *did = true;
            }
            continue;
        }
        const bool stepper = s.n_chunks + 1 >= f->n_window && s.n_pend + 1 == f->n_buffer;
        // the step in flight may end in a reset (its ordinal since the last reset is >= reset_steps): its verdict first
        if (stepper && f->reset_steps > 0 && s.infl - s.judged > 0 && s.stp + (s.infl - s.judged) >= f->reset_steps) { n_held++; continue; }
        cand.push_back({s.slot, stepper});
        (stepper ? n_step : n_fill)++;
    }
    f->held_last = n_held;
    if (cand.empty()) {
        if (!f->inflight.empty()) { int rc = front_collect(f); if (rc) return rc; *did = true; }   // nothing waiting: results at once
        return LASR_OK;
    }
    // A model step costs the GPU the same for 1 row or 64.  While the GPU has work queued (>= 2 steps in flight) and some open
    // stream has no chunk waiting yet, the tick collects the oldest step instead of submitting a thin batch: the front then runs
    // at the GPU's pace and the producers that are behind (Python threads taking turns on the GIL) fill their rings meanwhile.
    // With fewer than 2 steps in flight the batch goes out as it is (a lone real-time stream sees the synchronous latency).
    if (n_absent > 0 && (int)f->inflight.size() >= 2) {
        int rc = front_collect(f);
        if (rc) return rc;
        *did = true;
        return LASR_OK;
    }
    // streams out of phase share their model steps from the second one on: a minority of step-completing chunks waits a tick
    const bool hold_steppers = n_step > 0 && n_fill > 0 && n_step <= n_fill;
    for (const Cand& k : cand) {
        if (k.stepper && hold_steppers) continue;
        lasr_front::Stream& s = *f->st[k.slot];
        f->slots.push_back(k.slot);
        f->rows.push_back(s.ring.data() + (size_t)(s.tail.load(std::memory_order_relaxed) % f->ring_chunks) * f->chunk);
        if (k.stepper) f->step_rows.push_back(k.slot);
    }
    if (f->slots.empty()) return LASR_OK;
    const int before = lasr_step_pending(c);
    int rc = lasr_push_submit_rows(c, f->slots.data(), (int)f->slots.size(), f->rows.data(), nullptr);
    if (rc) return front_fail(f, rc);
    *did = true;
    f->n_ticks.fetch_add(1, std::memory_order_relaxed);
    for (int slot : f->slots) {                 // (the chunks have been copied into the engine's staging ring: the ring entries are free)
        lasr_front::Stream& s = *f->st[slot];
        s.n_chunks++;
        if (s.n_chunks >= f->n_window && ++s.n_pend == f->n_buffer) s.n_pend = 0;
        const long long t1 = s.tail.fetch_add(1, std::memory_order_release) + 1;
        // a producer blocked on a full ring is woken when the ring is half empty, not after every chunk taken (a futex wake per
        // stream and tick on this thread otherwise: 64 system calls per tick while a fast producer is throttled)
        if (s.head.load(std::memory_order_acquire) - t1 == f->ring_chunks / 2) { std::lock_guard<std::mutex> lk(s.rm); s.pcv.notify_all(); }
    }
    const int after = lasr_step_pending(c);
    if ((after > before) != !f->step_rows.empty()) {
        fail(c, LASR_ESTATE, "front: the engine %s a model step where the front's window / Buffer mirror expected %s", after > before ? "ran" : "did not run",
             f->step_rows.empty() ? "none" : "one");
        return front_fail(f, LASR_ESTATE);
    }
    if (!f->step_rows.empty()) {
        for (int slot : f->step_rows) f->st[slot]->infl++;
        f->n_steps.fetch_add(1, std::memory_order_relaxed);
        f->n_rows.fetch_add((long long)f->step_rows.size(), std::memory_order_relaxed);
        f->inflight.push_back(f->step_rows);
    }
    return LASR_OK;
}

void front_main(lasr_front* f) {
    (void)hipSetDevice(f->c->device);
    long long seen = -1;
    while (!f->stop.load(std::memory_order_acquire)) {
        bool did = false;
        for (int spins = 0; f->em_waiters.load(std::memory_order_acquire) > 0 && spins < 100000; ++spins) std::this_thread::yield();
        if (!f->rc) {
            std::lock_guard<std::mutex> lk(f->em);
            (void)front_tick(f, &did);
        }
        if (did) continue;
        // nothing to do: spin briefly (chunks of a saturated replay arrive every few microseconds), then sleep until the next push
        const long long w = f->work.load(std::memory_order_acquire);
        if (w != seen) { seen = w; continue; }
        bool woke = false;
        for (int i = 0; i < 2000; ++i) {
            __builtin_ia32_pause();
            if (f->work.load(std::memory_order_acquire) != seen || f->stop.load(std::memory_order_relaxed)) { woke = true; break; }
        }
        if (woke) continue;
        std::unique_lock<std::mutex> lk(f->fm);
        f->sleeping.store(true, std::memory_order_seq_cst);
        f->fcv.wait_for(lk, std::chrono::milliseconds(50), [&] { return f->work.load() != seen || f->stop.load(); });
        f->sleeping.store(false, std::memory_order_seq_cst);
    }
}

void front_kick(lasr_front* f) {
    f->work.fetch_add(1, std::memory_order_seq_cst);
    if (f->sleeping.load(std::memory_order_seq_cst)) { std::lock_guard<std::mutex> lk(f->fm); f->fcv.notify_one(); }
}

}  // namespace

extern "C" {

int lasr_front_create(lasr_ctx* c, int depth, int reset_steps, lasr_front** out) {
    if (!c || !out) return LASR_EINVAL;
    *out = nullptr;
    if (c->W > 1) return fail(c, LASR_ESTATE, "lasr_front: greedy decode only (beam %d)", c->W);
    if (!c->fe_fused) return fail(c, LASR_ESTATE, "lasr_front: needs the fused streaming front-end (<= 512 streams, the reference's frame geometry)");
    lasr_front* f = new lasr_front();
    f->c = c;
    f->depth = std::max(1, std::min(depth > 0 ? depth : 12, lasr_max_inflight(c)));
    f->reset_steps = std::abs(reset_steps);
    f->early = reset_steps > 0;               // (reset_steps < 0: the same rule without early verdicts -- every verdict at collect time: A/B aid)
    f->chunk = c->d.chunk; f->n_window = c->d.n_window; f->n_buffer = c->d.n_buffer;
    f->max_tok = std::max(16, c->d.n_buffer * c->d.max_iters_stream + 4);
    f->st.resize(c->d.max_streams);
    f->th = std::thread(front_main, f);
    f->oth = std::thread(front_out_main, f);
    *out = f;
    return LASR_OK;
}

// Stops the front thread and releases every producer blocked on a full ring and every consumer blocked in lasr_front_next (they
// return LASR_ESTATE).  The handle stays valid: join those threads, then lasr_front_destroy.  Idempotent.
int lasr_front_stop(lasr_front* f) {
    if (!f) return LASR_EINVAL;
    f->stop.store(true, std::memory_order_release);
    front_kick(f);
    { std::lock_guard<std::mutex> lk(f->fm); f->fcv.notify_all(); }
    for (auto& s : f->st)
        if (s) { std::lock_guard<std::mutex> lk(s->rm); s->rcv.notify_all(); s->pcv.notify_all(); }
    return LASR_OK;
}

void lasr_front_destroy(lasr_front* f) {
    if (!f) return;
    (void)lasr_front_stop(f);
    f->th.join();
    { std::lock_guard<std::mutex> lk(f->om); }
    f->ocv.notify_all();
    f->oth.join();
    {   // leave the engine idle: collect what is in flight, close the streams
        std::lock_guard<std::mutex> lk(f->em);
        while (!f->inflight.empty() && !f->rc) (void)front_collect(f);
        for (auto& s : f->st)
            if (s && s->open.load()) { (void)lasr_stream_close(f->c, s->slot); s->open.store(false); }
    }
    for (auto& s : f->st)
        if (s) { std::lock_guard<std::mutex> lk(s->rm); s->rcv.notify_all(); s->pcv.notify_all(); }
    delete f;
}

int lasr_front_open(lasr_front* f, int* stream) {
    if (!f || !stream) return LASR_EINVAL;
    if (f->rc) return f->rc;
    FrontCallerLock lk(f);
    int slot = -1;
    int rc = lasr_stream_open(f->c, &slot);
    if (rc) return rc;
    if (!f->st[slot]) f->st[slot].reset(new lasr_front::Stream());
    lasr_front::Stream& s = *f->st[slot];
    // (no producer of the slot's previous stream is inside a call any more: lasr_front_close waited for that)
    s.slot = slot; s.closing.store(false); s.eof_out = false;
    s.gen.store((s.gen.load() % 0x7fff) + 1, std::memory_order_release);          // 1 .. 32767: a stale id never matches
    s.ring.assign((size_t)f->ring_chunks * f->chunk, 0.f);
    s.head.store(0); s.tail.store(0); s.eof_in.store(false); s.last_push_ns.store(0);
    s.n_chunks = 0; s.n_pend = 0; s.infl = 0; s.stp = 0; s.judged = 0; s.judged_reset.clear();
    { std::lock_guard<std::mutex> rl(s.rm); s.res.clear(); }
    s.open.store(true, std::memory_order_release);
    *stream = front_id(s);
    return LASR_OK;
}

// a producer call on stream `id`: counted in s.inside BEFORE the id is checked, so lasr_front_close (closing = true, then wait for
// inside == 0) either sees the call or the call sees `closing` -- nobody writes a ring that close has handed to the next stream
struct FrontProducer {
    lasr_front::Stream* s = nullptr;
    FrontProducer(lasr_front* f, int id) {
        if (!f || id < 0) return;
        const int slot = id & 0xffff;
        if (slot >= (int)f->st.size() || !f->st[slot]) return;
        lasr_front::Stream* q = f->st[slot].get();
        q->inside.fetch_add(1, std::memory_order_seq_cst);
        if (q->gen.load(std::memory_order_seq_cst) != (id >> 16) || !q->open.load(std::memory_order_seq_cst) || q->closing.load(std::memory_order_seq_cst)) {
            q->inside.fetch_sub(1, std::memory_order_seq_cst);
            return;
        }
        s = q;
    }
    ~FrontProducer() { if (s) s->inside.fetch_sub(1, std::memory_order_seq_cst); }
};

// n_chunks client chunks (chunk floats each, host memory) of `stream`, copied into its ring; blocks while the ring is full.
// LASR_ESTATE: the id names no open stream (closed meanwhile, or an earlier stream of the slot), or the stream is being closed.
int lasr_front_push(lasr_front* f, int stream, const float* pcm, int n_chunks) {
    if (!f || stream < 0 || (n_chunks > 0 && !pcm)) return LASR_EINVAL;
    FrontProducer P(f, stream);
    if (!P.s) return LASR_ESTATE;
    lasr_front::Stream& s = *P.s;
    if (s.eof_in.load()) return LASR_ESTATE;
    for (int k = 0; k < n_chunks; ++k) {
        if (f->rc) return f->rc;
        if (s.closing.load(std::memory_order_acquire)) return LASR_ESTATE;
        const long long h = s.head.load(std::memory_order_relaxed);
        if (h - s.tail.load(std::memory_order_acquire) >= f->ring_chunks) {          // back-pressure
            std::unique_lock<std::mutex> lk(s.rm);
            s.pcv.wait_for(lk, std::chrono::milliseconds(20), [&] {
                return h - s.tail.load(std::memory_order_acquire) < f->ring_chunks || f->rc || f->stop.load() || s.closing.load(std::memory_order_acquire); });
            if (f->stop.load()) return LASR_ESTATE;
            --k;
            continue;
        }
        memcpy(s.ring.data() + (size_t)(h % f->ring_chunks) * f->chunk, pcm + (size_t)k * f->chunk, sizeof(float) * f->chunk);
        s.last_push_ns.store(front_now_ns(), std::memory_order_relaxed);
        s.head.store(h + 1, std::memory_order_release);
        front_kick(f);
    }
    return LASR_OK;
}

int lasr_front_eof(lasr_front* f, int stream) {
    if (!f || stream < 0) return LASR_EINVAL;
    FrontProducer P(f, stream);
    if (!P.s) return LASR_ESTATE;
    P.s->eof_in.store(true, std::memory_order_release);
    front_kick(f);
    return LASR_OK;
}

// Next result of `stream`, in model-step order: *flags bit 1 = a model step (n_tokens new token ids, possibly 0), bit 2 = the
// reset rule fired after this step, bit 4 = end of stream (every step of the pushed chunks has been delivered; after
// lasr_front_eof).  Blocks up to timeout_ms (< 0: for ever); returns 1 on time-out.
int lasr_front_next(lasr_front* f, int stream, int32_t* tokens, int cap, int* n_tokens, int* flags, int timeout_ms) {
    if (!f || stream < 0 || !n_tokens || !flags) return LASR_EINVAL;
    lasr_front::Stream* sp = front_stream(f, stream);
    if (!sp) return LASR_ESTATE;
    lasr_front::Stream& s = *sp;
    const int gen = stream >> 16;
    std::unique_lock<std::mutex> lk(s.rm);
    auto ready = [&] { return !s.res.empty() || f->rc != 0 || f->stop.load() || s.gen.load() != gen || !s.open.load(); };
    if (timeout_ms < 0) s.rcv.wait(lk, ready);
    else if (!s.rcv.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready)) return 1;
    if (s.gen.load() != gen || !s.open.load()) return LASR_ESTATE;           // closed under the consumer
    if (s.res.empty()) return f->rc.load() ? f->rc.load() : LASR_ESTATE;
    lasr_front::Stream::Res& r = s.res.front();
    if ((int)r.tok.size() > cap) return LASR_EFULL;
    if (!r.tok.empty()) memcpy(tokens, r.tok.data(), sizeof(int32_t) * r.tok.size());
    *n_tokens = (int)r.tok.size();
    *flags = r.flags;
    s.res.pop_front();
    return LASR_OK;
}

int lasr_front_close(lasr_front* f, int stream) {
    if (!f || stream < 0) return LASR_EINVAL;
    lasr_front::Stream* sp = front_stream(f, stream);
    if (!sp) return LASR_ESTATE;
    lasr_front::Stream& s = *sp;
    // no more input: the front thread takes nothing more from the ring, a producer blocked on it (or about to write it) leaves
    s.closing.store(true, std::memory_order_seq_cst);
    { std::lock_guard<std::mutex> rl(s.rm); s.pcv.notify_all(); }
    while (s.inside.load(std::memory_order_seq_cst) > 0) std::this_thread::yield();
    for (;;) {      // its steps in flight are collected by the front thread; then the slot can be closed
        FrontCallerLock lk(f);
        if (f->rc) return f->rc;
        if (s.infl == 0) {
            for (;;) {              // its last results have left the delivery queue (a late one must not reach the slot's next stream)
                { std::lock_guard<std::mutex> ol(f->om); if (f->outq.empty() && !f->out_busy) break; }
                // (after lasr_front_stop the delivery thread exits once its queue is empty: a record queued behind that stays --
                //  nothing will deliver it, and the stream is going away: drop the queue instead of spinning for ever)
                if (f->stop.load(std::memory_order_acquire) || f->rc.load()) {
                    std::lock_guard<std::mutex> ol(f->om);
                    if (!f->out_busy) { f->outq.clear(); break; }
                }
                std::this_thread::yield();
            }
            int rc = lasr_stream_close(f->c, s.slot);
            s.open.store(false, std::memory_order_release);
            std::lock_guard<std::mutex> rl(s.rm);
            s.res.clear();
            s.rcv.notify_all();
            return rc;
        }
        int rc = front_collect(f);
        if (rc) return rc;
    }
}

// Reset rule on text (api-server.py:124-133 resets on `y_one == ""`, the DECODED step): ids[0..n) are the token ids whose piece
// decodes to the empty string in the servicer's tokenizer; a step whose tokens are all among them counts as empty.  n = 0: only
// "no token" is empty (what the rule is with an id-per-character vocabulary).  Call before streams are opened.
int lasr_front_set_empty_tokens(lasr_front* f, const int32_t* ids, int n) {
    if (!f || n < 0 || (n > 0 && !ids)) return LASR_EINVAL;
    FrontCallerLock lk(f);
    f->empty_tok.assign((size_t)f->c->d.vocab, 0);
    for (int i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= f->c->d.vocab) return fail(f->c, LASR_EINVAL, "lasr_front_set_empty_tokens: id %d outside the vocabulary", ids[i]);
        f->empty_tok[ids[i]] = 1;
    }
    return LASR_OK;
}

// The engine for the caller (a unary Transcribe RPC, an offline utterance): every step in flight is collected and the front
// thread stays out until lasr_front_resume, which the SAME thread calls.  Does not nest.
int lasr_front_pause(lasr_front* f) {
    if (!f) return LASR_EINVAL;
    f->em_waiters.fetch_add(1, std::memory_order_acq_rel);
    f->em.lock();
    f->em_waiters.fetch_sub(1, std::memory_order_acq_rel);
    while (!f->inflight.empty() && !f->rc) (void)front_collect(f);
    if (f->rc) { f->em.unlock(); return f->rc; }
    return LASR_OK;
}
int lasr_front_resume(lasr_front* f) {
    if (!f) return LASR_EINVAL;
    f->em.unlock();
    front_kick(f);
    return LASR_OK;
}

// counters: ticks (lasr_push_submit_rows calls), model steps submitted, rows of those steps, resets by the rule
int lasr_front_stats(lasr_front* f, long long* ticks, long long* steps, long long* rows, long long* resets) {
    if (!f) return LASR_EINVAL;
    if (ticks) *ticks = f->n_ticks.load();
    if (steps) *steps = f->n_steps.load();
    if (rows) *rows = f->n_rows.load();
    if (resets) *resets = f->n_resets.load();
    return f->rc;
}

const char* lasr_front_error(const lasr_front* f) {
    if (!f) return "null front";
    return f->rc.load(std::memory_order_acquire) ? f->err.c_str() : "";
}

// include/lasr_debug.h: capacity of the front with NATIVE per-stream producers (one std::thread per stream pushing
// chunks_per_push chunks per call, then reading its results to the end): what the per-stream form can carry when the producers
// are not Python threads taking turns on the GIL.  pcm [n_streams][n_chunks * chunk] host; tokens [n_streams][cap], n_tok [n_streams].
int lasr_bench_front(lasr_ctx* c, int depth, int reset_steps, int n_streams, const float* pcm, int n_chunks, int chunks_per_push,
                     int32_t* tokens, int cap, int* n_tok, double* seconds, long long* stats4) {
    if (!c || !pcm || !tokens || !n_tok || !seconds || n_streams < 1 || n_chunks < 1 || chunks_per_push < 1) return LASR_EINVAL;
    lasr_front* f = nullptr;
    RC(lasr_front_create(c, depth, reset_steps, &f));
    std::vector<int> sid(n_streams, -1);
    int rc = LASR_OK;
    for (int i = 0; i < n_streams && !rc; ++i) rc = lasr_front_open(f, &sid[i]);
    std::atomic<int> go{0}, bad{0};
    std::vector<std::thread> th;
    const size_t per = (size_t)n_chunks * c->d.chunk;
    for (int i = 0; i < n_streams && !rc; ++i)
        th.emplace_back([&, i] {
            while (!go.load(std::memory_order_acquire)) __builtin_ia32_pause();
            int n = 0;
            for (int k = 0; k < n_chunks; k += chunks_per_push)
                if (lasr_front_push(f, sid[i], pcm + i * per + (size_t)k * c->d.chunk, std::min(chunks_per_push, n_chunks - k))) { bad.fetch_add(1); break; }
            (void)lasr_front_eof(f, sid[i]);
            for (;;) {
                int32_t buf[64]; int nt = 0, fl = 0;
                const int r = lasr_front_next(f, sid[i], buf, 64, &nt, &fl, 20000);
                if (r != LASR_OK) { bad.fetch_add(1); break; }
                if (fl & FRONT_RES_EOF) break;
                for (int q = 0; q < nt && n < cap; ++q) tokens[(size_t)i * cap + n++] = buf[q];
            }
            n_tok[i] = n;
        });
    const auto t0 = std::chrono::steady_clock::now();
    go.store(1, std::memory_order_release);
    for (auto& t : th) t.join();
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats4) (void)lasr_front_stats(f, &stats4[0], &stats4[1], &stats4[2], &stats4[3]);
    if (!rc && bad.load()) rc = f->rc.load() ? f->rc.load() : LASR_ESTATE;
    for (int i = 0; i < n_streams; ++i)
        if (sid[i] >= 0) (void)lasr_front_close(f, sid[i]);
    lasr_front_destroy(f);
    return rc;
}

}  // extern "C"
