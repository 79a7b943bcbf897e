// front_standin.cpp -- the native serving front (libreasr_amd/csrc/lasr_front.hip.h) on a STAND-IN engine, host only: the
// sanitizer target SURVEY §5 promised.  GPU AddressSanitizer is not available on this pool; the front is the
// part of the library that is threads, rings and lifetimes rather than kernels, and it is written against the public C ABI, so
// it compiles with g++ against a mock of the dozen engine calls it makes.  Built twice by tests/test_host.py:
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -pthread
//     g++ -std=c++17 -O1 -g -fsanitize=thread -pthread
// and run: producers / consumers per stream, closes under a blocked producer, stale stream ids, the reset rule on text with
// reset_steps <= depth, stop with threads inside.  Exit code 0 = every check passed and no sanitizer report.
//
// The stand-in model: a chunk of stream s carries tag(s, k) in every sample (so a chunk that reaches the wrong slot, or a ring
// entry read after its stream was closed, is seen); model step j of a stream emits tokens that are a function of (tag, j) only.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lasr.h"

// ---------------------------------------------------------------------------- the stand-in engine
struct lasr_ctx {
    lasr_model_desc d;
    int W = 1, device = 0;
    bool fe_fused = true;
    std::string err;
    struct Slot {
        bool open = false;
        long long n_chunks = 0; int n_pend = 0;
        float tag = 0.f;                  // taken from the stream's first chunk
        long long steps = 0;              // model steps run since open
        std::deque<std::vector<int32_t>> decoded;      // tokens per submitted, uncollected step (oldest first)
        std::vector<int32_t> fetchable;   // tokens of collected steps not yet fetched
        int resets = 0;
    };
    std::vector<Slot> slot;
    std::deque<std::vector<int>> pending;              // rows of every submitted, uncollected model step
    std::atomic<int> bad_data{0};
    std::atomic<int> api_depth{0};                      // the engine is single-caller: concurrent entry = a bug of the front
};

static int fail(lasr_ctx* c, int code, const char* fmt, ...) {
    char buf[256];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (c) c->err = buf;
    return code;
}
#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
static int hipSetDevice(int) { return 0; }

struct Single {      // checks the single-caller rule of the engine ABI
    lasr_ctx* c;
    explicit Single(lasr_ctx* c_) : c(c_) { if (c->api_depth.fetch_add(1) != 0) { fprintf(stderr, "two threads inside the engine\n"); abort(); } }
    ~Single() { c->api_depth.fetch_sub(1); }
};

static std::vector<int32_t> model_tokens(float tag, long long step) {
    // steps 4..6 of every 9 are silent; the others emit 1-2 tokens, some of them only the "empty" piece 7
    std::vector<int32_t> t;
    const int ph = (int)(step % 9);
    if (ph >= 4 && ph <= 6) return t;
    const int base = ((int)tag * 7 + (int)step * 3) % 40 + 8;
    if (ph == 7 || ph == 8) { t.push_back(7); return t; }          // text-empty step with a token
    t.push_back(base);
    if (step % 2) t.push_back(base + 1);
    return t;
}

extern "C" {
const char* lasr_last_error(const lasr_ctx* c) { return c ? c->err.c_str() : "null"; }
int lasr_stream_open(lasr_ctx* c, int* slot) {
    Single g(c);
    for (int s = 0; s < (int)c->slot.size(); ++s)
        if (!c->slot[s].open) { c->slot[s] = lasr_ctx::Slot(); c->slot[s].open = true; *slot = s; return LASR_OK; }
    return fail(c, LASR_EFULL, "no free slot");
}
int lasr_stream_close(lasr_ctx* c, int s) {
    Single g(c);
    if (s < 0 || s >= (int)c->slot.size() || !c->slot[s].open) return fail(c, LASR_ESTATE, "slot %d not open", s);
    if (!c->slot[s].decoded.empty()) return fail(c, LASR_ESTATE, "slot %d closed with a step in flight", s);
    c->slot[s].open = false;
    return LASR_OK;
}
int lasr_stream_reset(lasr_ctx* c, int s, int what) {
    Single g(c);
    if (s < 0 || s >= (int)c->slot.size() || !c->slot[s].open) return fail(c, LASR_ESTATE, "slot %d not open", s);
    if (!(what & LASR_RESET_IF_DECODED) && !c->slot[s].decoded.empty()) return fail(c, LASR_ESTATE, "slot %d has a step in flight", s);
    c->slot[s].resets++;
    return LASR_OK;
}
int lasr_step_pending(lasr_ctx* c) { return (int)c->pending.size(); }
int lasr_max_inflight(const lasr_ctx*) { return 25; }
int lasr_push_submit_rows(lasr_ctx* c, const int* slots, int n, const float* const* rows, long long* ticket) {
    Single g(c);
    if (ticket) *ticket = -1;
    std::vector<int> step_rows;
    for (int i = 0; i < n; ++i) {
        lasr_ctx::Slot& S = c->slot[slots[i]];
        if (!S.open) return fail(c, LASR_ESTATE, "slot %d not open", slots[i]);
        const float* p = rows[i];
        float lo = p[0], hi = p[0];
        for (int q = 1; q < c->d.chunk; ++q) { lo = std::min(lo, p[q]); hi = std::max(hi, p[q]); }      // reads the whole ring entry (ASan)
        if (S.n_chunks == 0) S.tag = p[0];
        if (lo != hi || p[0] != S.tag) c->bad_data.fetch_add(1);               // a chunk of another stream, or a torn one
        S.n_chunks++;
        if (S.n_chunks >= c->d.n_window && ++S.n_pend == c->d.n_buffer) { S.n_pend = 0; step_rows.push_back(slots[i]); }
    }
    if (!step_rows.empty()) {
        if ((int)c->pending.size() >= 25) return fail(c, LASR_EFULL, "too many steps in flight");
        for (int s : step_rows) { lasr_ctx::Slot& S = c->slot[s]; S.decoded.push_back(model_tokens(S.tag, S.steps++)); }
        c->pending.push_back(step_rows);
    }
    return LASR_OK;
}
int lasr_step_wait(lasr_ctx* c, int* n_ran) {
    Single g(c);
    if (n_ran) *n_ran = 0;
    if (c->pending.empty()) return LASR_OK;
    std::this_thread::sleep_for(std::chrono::microseconds(30));                // "GPU time"
    for (int s : c->pending.front()) {
        lasr_ctx::Slot& S = c->slot[s];
        S.fetchable.insert(S.fetchable.end(), S.decoded.front().begin(), S.decoded.front().end());
        S.decoded.pop_front();
    }
    if (n_ran) *n_ran = (int)c->pending.front().size();
    c->pending.pop_front();
    return LASR_OK;
}
int lasr_fetch_many(lasr_ctx* c, const int* slots, int n, int32_t* tokens, int cap, int* n_new) {
    Single g(c);
    for (int i = 0; i < n; ++i) {
        lasr_ctx::Slot& S = c->slot[slots[i]];
        if ((int)S.fetchable.size() > cap) return fail(c, LASR_EFULL, "cap");
        std::copy(S.fetchable.begin(), S.fetchable.end(), tokens + (size_t)i * cap);
        n_new[i] = (int)S.fetchable.size();
        S.fetchable.clear();
    }
    return LASR_OK;
}
int lasr_peek_many(lasr_ctx* c, const int* slots, int n, const int* skip, int32_t* tokens, int cap, int32_t* counts, int cap_steps,
                   int* n_decoded, int* n_inflight) {
    Single g(c);
    for (int i = 0; i < n; ++i) {
        lasr_ctx::Slot& S = c->slot[slots[i]];
        n_decoded[i] = n_inflight[i] = (int)S.decoded.size();                  // (the stand-in decodes at submit)
        int used = 0, kept = 0;
        for (int k = skip ? skip[i] : 0; k < (int)S.decoded.size(); ++k) {
            if (kept >= cap_steps || used + (int)S.decoded[k].size() > cap) return fail(c, LASR_EFULL, "peek");
            counts[(size_t)i * cap_steps + kept++] = (int)S.decoded[k].size();
            for (int32_t t : S.decoded[k]) tokens[(size_t)i * cap + used++] = t;
        }
    }
    return LASR_OK;
}
}  // extern "C"

#include "../../libreasr_amd/csrc/lasr_front.hip.h"

// ---------------------------------------------------------------------------- the checks
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); exit(1); } } while (0)

static std::vector<float> chunk_of(float tag, int chunk) { return std::vector<float>((size_t)chunk, tag); }

// what the reset rule must do to a stream that runs `steps` model steps (text rule: token 7 decodes to "")
static int expected_resets(float tag, long long steps, int reset_steps, std::vector<char>* flags) {
    int n = 0; long long since = 0;
    for (long long j = 0; j < steps; ++j) {
        const std::vector<int32_t> t = model_tokens(tag, j);
        const bool empty = t.empty() || (t.size() == 1 && t[0] == 7);
        since++;
        const bool fire = reset_steps > 0 && empty && since >= reset_steps;
        if (flags) flags->push_back(fire ? 1 : 0);
        if (fire) { since = 0; n++; }
    }
    return n;
}

static void run_streams(lasr_ctx* c, int depth, int reset_steps, int n_streams, int n_chunks, unsigned seed) {
    lasr_front* f = nullptr;
    CHECK(lasr_front_create(c, depth, reset_steps, &f) == LASR_OK);
    const int32_t empty_ids[1] = {7};
    CHECK(lasr_front_set_empty_tokens(f, empty_ids, 1) == LASR_OK);
    std::vector<std::thread> th;
    std::atomic<int> failures{0};
    for (int i = 0; i < n_streams; ++i)
        th.emplace_back([&, i] {
            unsigned r = seed * 7919u + (unsigned)i * 104729u;
            auto rnd = [&] { r = r * 1664525u + 1013904223u; return r >> 8; };
            for (int round = 0; round < 3; ++round) {                          // every thread runs three streams one after the other
                int sid = -1;
                if (lasr_front_open(f, &sid) != LASR_OK) { failures++; return; }
                const float tag = (float)(1 + i * 3 + round);
                const int nch = n_chunks - (int)(rnd() % 7);
                std::thread prod([&] {
                    for (int k = 0; k < nch;) {
                        const int run = std::min(nch - k, 1 + (int)(rnd() % 3));
                        std::vector<float> buf;
                        for (int q = 0; q < run; ++q) { auto ch = chunk_of(tag, c->d.chunk); buf.insert(buf.end(), ch.begin(), ch.end()); }
                        if (lasr_front_push(f, sid, buf.data(), run) != LASR_OK) { failures++; return; }
                        k += run;
                        if (rnd() % 5 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rnd() % 200));
                    }
                    if (lasr_front_eof(f, sid) != LASR_OK) failures++;
                });
                long long steps = 0; int resets = 0;
                std::vector<char> got_flags;
                for (;;) {
                    int32_t tok[64]; int nt = 0, fl = 0;
                    const int rc = lasr_front_next(f, sid, tok, 64, &nt, &fl, 20000);
                    if (rc != LASR_OK) { failures++; break; }
                    if (fl & FRONT_RES_EOF) break;
                    const std::vector<int32_t> want = model_tokens(tag, steps);
                    if (std::vector<int32_t>(tok, tok + nt) != want) failures++;
                    got_flags.push_back((fl & FRONT_RES_RESET) ? 1 : 0);
                    resets += (fl & FRONT_RES_RESET) ? 1 : 0;
                    steps++;
                }
                prod.join();
                const long long want_steps = nch >= c->d.n_window ? (nch - c->d.n_window + 1) / c->d.n_buffer : 0;
                std::vector<char> want_flags;
                const int want_resets = expected_resets(tag, want_steps, reset_steps, &want_flags);
                if (steps != want_steps || resets != want_resets || got_flags != want_flags) {
                    fprintf(stderr, "stream %d.%d: steps %lld (want %lld) resets %d (want %d)\n", i, round, steps, want_steps, resets, want_resets);
                    failures++;
                }
                if (lasr_front_close(f, sid) != LASR_OK) failures++;
                // the id is stale now: nothing it names
                float z = 0.f; int nt = 0, fl = 0; int32_t tk[4];
                if (lasr_front_push(f, sid, &z, 0) != LASR_ESTATE || lasr_front_eof(f, sid) != LASR_ESTATE ||
                    lasr_front_next(f, sid, tk, 4, &nt, &fl, 0) != LASR_ESTATE || lasr_front_close(f, sid) != LASR_ESTATE) failures++;
            }
        });
    for (auto& t : th) t.join();
    CHECK(failures.load() == 0);
    CHECK(c->bad_data.load() == 0);
    lasr_front_destroy(f);
}

// a producer blocked on a full ring while the front is paused; close releases it; the slot's next stream is clean
static void close_under_a_blocked_producer(lasr_ctx* c) {
    lasr_front* f = nullptr;
    CHECK(lasr_front_create(c, 4, 0, &f) == LASR_OK);
    int old_id = -1;
    CHECK(lasr_front_open(f, &old_id) == LASR_OK);
    CHECK(lasr_front_pause(f) == LASR_OK);
    std::atomic<int> prc{12345};
    std::thread prod([&] {
        auto ch = chunk_of(99.f, c->d.chunk);
        int rc = LASR_OK;
        for (int k = 0; k < 300 && rc == LASR_OK; ++k) rc = lasr_front_push(f, old_id, ch.data(), 1);
        prc.store(rc);
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
    CHECK(prc.load() == 12345);                                            // blocked on the full ring (64 chunks)
    std::thread closer([&] { CHECK(lasr_front_close(f, old_id) == LASR_OK); });
    prod.join();                                                           // released by close before close could take the engine
    CHECK(prc.load() == LASR_ESTATE);
    CHECK(lasr_front_resume(f) == LASR_OK);
    closer.join();
    int new_id = -1;
    CHECK(lasr_front_open(f, &new_id) == LASR_OK);
    CHECK((new_id & 0xffff) == (old_id & 0xffff) && new_id != old_id);
    auto ch = chunk_of(5.f, c->d.chunk);
    CHECK(lasr_front_push(f, old_id, ch.data(), 1) == LASR_ESTATE);        // the stale reader cannot write the new stream's ring ...
    CHECK(lasr_front_eof(f, old_id) == LASR_ESTATE);                       // ... nor end it
    for (int k = 0; k < 9; ++k) CHECK(lasr_front_push(f, new_id, ch.data(), 1) == LASR_OK);
    CHECK(lasr_front_eof(f, new_id) == LASR_OK);
    long long steps = 0;
    for (;;) {
        int32_t tok[64]; int nt = 0, fl = 0;
        CHECK(lasr_front_next(f, new_id, tok, 64, &nt, &fl, 20000) == LASR_OK);
        if (fl & FRONT_RES_EOF) break;
        CHECK(std::vector<int32_t>(tok, tok + nt) == model_tokens(5.f, steps));
        steps++;
    }
    CHECK(steps == (9 - c->d.n_window + 1) / c->d.n_buffer);
    CHECK(c->bad_data.load() == 0);
    CHECK(lasr_front_close(f, new_id) == LASR_OK);
    lasr_front_destroy(f);
}

// stop with a consumer blocked in next and a close racing it: nobody spins for ever (close after stop)
static void stop_with_threads_inside(lasr_ctx* c) {
    lasr_front* f = nullptr;
    CHECK(lasr_front_create(c, 4, 0, &f) == LASR_OK);
    int a = -1, b = -1;
    CHECK(lasr_front_open(f, &a) == LASR_OK && lasr_front_open(f, &b) == LASR_OK);
    auto ch = chunk_of(3.f, c->d.chunk);
    for (int k = 0; k < 20; ++k) CHECK(lasr_front_push(f, b, ch.data(), 1) == LASR_OK);
    std::atomic<int> crc{12345};
    std::thread cons([&] { int32_t t[64]; int nt, fl; crc.store(lasr_front_next(f, a, t, 64, &nt, &fl, -1)); });
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    CHECK(lasr_front_stop(f) == LASR_OK);
    cons.join();
    CHECK(crc.load() == LASR_ESTATE);
    std::thread closer([&] { (void)lasr_front_close(f, b); });             // steps of b may still be in flight: must return
    closer.join();
    lasr_front_destroy(f);
}

int main() {
    lasr_ctx c;
    lasr_default_desc(&c.d);
    c.d.chunk = 64;                      // (small chunks: the test is about threads and lifetimes, not bytes)
    c.d.max_streams = 16;
    c.slot.resize(16);
    close_under_a_blocked_producer(&c);
    run_streams(&c, 8, 0, 12, 60, 1);    // no rule
    run_streams(&c, 8, 25, 12, 160, 2);  // the reference's threshold (4 s)
    run_streams(&c, 8, 3, 12, 90, 3);    // reset_steps <= depth: several judged steps per stream in flight
    run_streams(&c, 1, 3, 6, 40, 4);     // depth 1: every verdict at collect time
    stop_with_threads_inside(&c);
    for (auto& s : c.slot) CHECK(!s.open);
    printf("front stand-in: ok\n");
    return 0;
}

// lasr_default_desc of the real library (the stand-in links nothing of it)
extern "C" void lasr_default_desc(lasr_model_desc* d) {
    memset(d, 0, sizeof(*d));
    d->feat = 1280; d->hidden = 1024; d->enc_layers = 4; d->pred_layers = 2; d->embed = 512; d->joint = 1024; d->vocab = 2048;
    d->n_fft = 1024; d->win = 400; d->hop = 160; d->n_mels = 128; d->n_stack = 10; d->stride = 8; d->n_buffer = 2; d->n_window = 3;
    d->chunk = 1280; d->sample_rate = 16000; d->max_streams = 64; d->max_iters_offline = 3; d->max_iters_stream = 10; d->beam = 1;
}
