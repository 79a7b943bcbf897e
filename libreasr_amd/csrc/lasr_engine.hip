// lasr_engine.hip -- host side of liblasr_hip.so: weight packing, per-stream device state, the
// per-chunk step (front-end -> encoder -> greedy decode loop) and the C ABI of include/lasr.h.
// gfx950 only.  No CPU fallback: every numeric result comes from the kernels in lasr_kernels.hip.h.
#include "lasr_kernels.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <string>
#include <vector>

#include "../../include/lasr.h"

using namespace lasr;

namespace {

constexpr int NW = 8;          // waves per GEMM workgroup (K split)
constexpr int NCMD = 64;       // ring of host->device command blocks
constexpr int MTA = 4;         // m-tiles per workgroup in the "A" tiling (64 rows)

struct Cell {                  // one recurrent layer (+ its BatchNorm fold and learned initial state)
    int I = 0;                 // input width
    void *WxA = nullptr, *WhA = nullptr;  // packed (element-typed), tiling "A" (4 units x gates per tile): predictor
    void *WxC = nullptr, *WhC = nullptr;  // packed, tiling "C" (8 units x 2 gates per tile, 2 tiles per group): encoder
    float *bias = nullptr, *rbias = nullptr;
    float *bn_s = nullptr, *bn_t = nullptr;
    float *h0 = nullptr, *c0 = nullptr;
    float *tab = nullptr;      // predictor layer 0: per-token input projection table
};

}  // namespace

struct lasr_ctx {
    lasr_model_desc d;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void*> dev_allocs;
    std::vector<void*> host_allocs;

    int M = 0, MT = 0;         // padded rows (stream slots), m-tiles
    int W = 1;                 // beam width (hypothesis slots per stream); 1 = greedy
    int Md = 0, MTd = 0;       // decoder rows = M * W (row = stream * W + slot), m-tiles
    static constexpr int LA_MAX = 4;
    int la = 1;                // greedy lookahead: frames evaluated per row and iteration (1 with an LM or a beam)
    int la_stream = 1, la_offline = 3;   // measured on configs[1]: streaming steps have 2 frames and the per-iteration
                               // cost of a wider logits GEMM cancels the saved iterations; offline (258 frames) gains 10 %
    int MTj = 0;               // m-tiles of the ja / logits row space: max(Md, LA_MAX * M) / 16
    int bf = 0;                // 1: bf16 operands (weights + GEMM-input activations), f32 accumulate / state / logits
    int kch = 16;              // k per MFMA chunk (16 f32, 32 bf16)
    size_t esz = 4;            // bytes per operand element
    int G_pred = 0;            // gates of the predictor cell (3 NBRC / 4 LSTM)

    // front-end constants
    float* window = nullptr; float2* tw512 = nullptr; float2* tw1024 = nullptr;
    int* fb_start = nullptr; int* fb_off = nullptr; float* fb_w = nullptr; int fb_nnz = 0;
    float *ln_w = nullptr, *ln_b = nullptr;

    std::vector<Cell> enc, pred;
    void *W1p = nullptr, *W1e = nullptr, *W2 = nullptr;   // packed, element-typed
    float *b1 = nullptr, *b2 = nullptr;

    // recurrent state (row == slot)
    std::vector<void*> enc_h[2], pred_h[2], pred_y;      // element-typed (A operands)
    std::vector<float*> enc_c, pred_c;
    // beam search: c, BN(h) and pp ping-pong like h (every slot may be re-parented each round):
    // parity 0 = pred_c / pred_y / pp, parity 1 = the *1 buffers; all follow pred_par
    std::vector<float*> pred_c1;
    std::vector<void*> pred_y1;
    float* pp1 = nullptr;
    double* b_score = nullptr; int *b_alive = nullptr, *b_inB = nullptr, *b_parent = nullptr, *b_trellis = nullptr;
    std::vector<std::vector<std::vector<int32_t>>> hyp;   // host: token history of every hypothesis slot [M][W]
    std::vector<std::vector<int32_t>> committed;          // host: best hypothesis at the last predictor reset(s)
    std::vector<double> committed_score;
    std::vector<std::vector<int32_t>> best_full;          // host: committed + current best hypothesis
    int* trellis_host = nullptr; size_t trellis_host_ints = 0;
    int enc_par = 0;
    int pred_par = 0;               // predictor h ping-pong parity (row-major [M][H] buffers)
    void *cvt_a = nullptr, *cvt_b = nullptr;   // [M][H] element-typed staging of f32 op-level inputs
    bool dbg_gate = true;           // decode kernels record timestamps only in the first iteration of a step
    unsigned long long* dbg = nullptr;   // LASR_DBG_TIMING: [5 kinds][4096 blocks][16] phase timestamps
    std::vector<unsigned long long> tile_masks;   // per step t: m-tiles with an active row (from the host's T_row)
    float *pp = nullptr, *logits = nullptr;
    void* ja = nullptr;             // joint activation, fragment-major, element-typed
    DecState ds{};
    int n_iter_slots = 0;
    int* T_row_dev = nullptr;       // [M] current step's frames per row: points INTO the step's device command block
    int* zero_rows = nullptr;       // [M] zeros (reset passes: "no row is decoding")
    int* T_row_dec = nullptr;       // what the decode kernels read (T_row_dev; frames-available counters when continuous)
    int* dec_t_idx = nullptr;       // frame cursor array the decode kernels use (ds.t_idx, or c_cur when continuous)
    int pe_ring_R = 1 << 30;        // pe frame t lives at slot t % pe_ring_R
    // continuous decode (lasr_step_submit / lasr_step_wait): front-end + encoder of later chunks run on
    // the main stream while ONE greedy loop keeps running on stream_dec across chunk boundaries: a row
    // that finished chunk k moves on to chunk k+1's frames while a bursty row is still on chunk k.
    hipStream_t stream_dec = nullptr;
    static constexpr int NFLY = 8;  // steps in flight (ring of events / T_row snapshots)
    static constexpr int RING = 32; // pe ring, frames per row
    static constexpr int TOKRING = 256, ENDSLOTS = 16;
    hipEvent_t ev_enc[NFLY] = {};
    hipEvent_t ev_misc = nullptr;
    int* T_row_ring[NFLY] = {};
    float* pe_ring = nullptr;
    int *c_cur = nullptr, *c_avail = nullptr, *c_iters = nullptr, *c_target = nullptr, *c_ntotal = nullptr;
    int *c_ntok_end = nullptr, *c_tok_ring = nullptr, *c_behind = nullptr, *c_enc_frames = nullptr;
    int *c_done = nullptr, *c_flag_dev = nullptr;   // workgroups finished per iteration; device view of cont_host[0]
    int* c_iter = nullptr;                          // device-side iteration counter of the continuous loop
    std::map<std::tuple<int, int, int>, hipGraphExec_t> cgraphs;   // (iterations, predictor parity, LM parity) -> group
    int* cont_host = nullptr;       // pinned: [0] flag, [4..] target staging (NFLY blocks), then ntok_end + token ring
    struct PendingStep { std::vector<int> rows; int Tm; int idx; bool admitted; bool target_set; std::vector<int> target; const int* T_row_ptr; long long serial; };
    std::vector<PendingStep> pending;
    std::vector<long long> h_frames_sub, h_fetched;
    long long model_steps = 0, cont_iters = 0;
    bool group_inflight = false;    // a decode group has been launched and its flag not yet consumed
    long long inflight_for = -1;    // serial of the pending step the in-flight group's flag refers to
    long long done_serial = -1;     // serial of a pending step already known to be fully decoded
    int kick_iters = 0;
    int kick_n = 3, wait_n = 1;     // iterations per group: kicked from submit / launched while waiting (swept on configs[1])
    // hipGraph cache of streaming decode groups: key = (first iteration, iterations, pe/T_row buffer,
    // predictor parity at group start, frames)
    std::map<std::tuple<int, int, int, int, int>, hipGraphExec_t> graphs;
    bool use_graphs = true;

    // LM shallow fusion (lasr_attach_lm): Embedding -> LSTM stack -> Linear -> log_softmax, stepped once
    // per emitted token for the rows that emitted (same compacted cell kernels as the predictor)
    struct LM {
        bool on = false;
        int E = 0, H = 0, L = 0;
        float alpha = 0.1f, theta = 1.0f, min_val = -10.0f;
        std::vector<Cell> cells;        // tiling "A"; layer 0 input side = per-token table
        void* Wout = nullptr; float* bout = nullptr;
        float *ones = nullptr, *zeros = nullptr;       // "BatchNorm fold" of a plain LSTM: y = h
        std::vector<void*> h[2], y;     // row-major [M][H], element-typed; h ping-pongs
        std::vector<float*> cst;        // [H][M]
        int par = 0;
        float *raw = nullptr, *lmz = nullptr;          // [M][V] output-layer logits / standardised log-probs
        int* valid = nullptr;
    } lm;

    // resampling filters per client sample rate (lasr_resample)
    struct Resampler { int U = 0, taps = 0, in_unit = 0; int* first = nullptr; float* w = nullptr; };
    std::map<int, Resampler> resamplers;

    // time-series buffers (capacity Tcap frames)
    int Tcap = 0;
    void *x0 = nullptr, *ybuf[2] = {nullptr, nullptr};   // element-typed, fragment-major
    float *pe = nullptr, *pe_sync = nullptr;
    int tok_cap_alloc = 0;

    // front-end buffers
    float* win = nullptr; int* ring_pos = nullptr;
    float* pend = nullptr;          // [M][n_buffer*n_stack][n_mels]
    float* stage_pcm = nullptr; size_t stage_pcm_floats = 0;
    // streaming pushes from host memory: ring of device staging rows + one event per entry, so a push
    // never has to drain the stream (the copy of chunk k+1 overlaps the kernels of chunk k)
    static constexpr int NSTAGE = 16;
    float* push_stage = nullptr; hipEvent_t push_ev[NSTAGE] = {}; bool push_used[NSTAGE] = {}; int push_next = 0;
    float* push_stage_host = nullptr;     // pinned mirror of the ring: caller's (pageable) buffer -> memcpy -> async DMA
    hipStream_t stream_copy = nullptr;    // the DMA of chunk k+1 runs under the kernels of chunk k; the push kernel waits for it
    hipEvent_t push_copied[NSTAGE] = {};
    float* lm_buf = nullptr; size_t lm_floats = 0;       // offline log-mel
    float* feat_stage = nullptr; size_t feat_stage_floats = 0;

    // command blocks (pinned host ring + device ring)
    struct Cmd {
        int* T_row; int* what; int* src_idx; int* feat_sel; int* row_frames; int* token; int* emit;
        long long* row_N; long long* row_src_off; long long* row_feat_off;
    };
    char* cmd_host = nullptr; char* cmd_dev = nullptr; size_t cmd_bytes = 0; int cmd_next = 0; int cmd_inflight = 0;
    Cmd hc{}, dc{};

    // host results
    int* res_host = nullptr;        // pinned: unfinished flag + ntok + tokens + metrics
    size_t res_bytes = 0;

    // host mirrors
    std::vector<char> open_;
    std::vector<int> n_chunks, n_pend;
    std::vector<std::vector<int32_t>> queue;
    std::vector<double> neg_logp, align;

    // stats
    bool profiling = false;
    hipEvent_t ev[8];
    bool ev_ok = false;
    lasr_step_stats stats{};
};

namespace {

int fail(lasr_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(c, LASR_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <class T>
int dalloc(lasr_ctx* c, T** p, size_t n) {
    void* q = nullptr;
    if (n == 0) n = 1;
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess) return fail(c, LASR_ENOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
    c->dev_allocs.push_back(q);
    *p = (T*)q;
    return LASR_OK;
}
void dfree(lasr_ctx* c, void* p) {
    if (!p) return;
    auto it = std::find(c->dev_allocs.begin(), c->dev_allocs.end(), p);
    if (it != c->dev_allocs.end()) c->dev_allocs.erase(it);
    (void)hipFree(p);
}
template <class T>
int upload(lasr_ctx* c, T** p, const T* src, size_t n) {
    int rc = dalloc(c, p, n);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(*p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return LASR_OK;
}
#define RC(x)                \
    do {                     \
        int rc_ = (x);       \
        if (rc_) return rc_; \
    } while (0)

unsigned short host_bf16(float x) {            // round to nearest even (same as the device f32_to_bf16)
    unsigned u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
struct Packed {                                // host image of a packed operand
    std::vector<char> bytes;
    int bf = 0;
    void resize(size_t n_elems, int bf_) { bf = bf_; bytes.assign(n_elems * (bf_ ? 2 : 4), 0); }
    void set(size_t i, float v) {
        if (bf) ((unsigned short*)bytes.data())[i] = host_bf16(v);
        else ((float*)bytes.data())[i] = v;
    }
};
// Weight tile [16 columns][K] -> fragments: elem((tile*KC + c)*64 + lane, e) = get(tile, col = lane&15, k),
// k = KCH*c + EPL*(lane>>4) + e   (KCH = 16, EPL = 4 for f32;  32, 8 for bf16)
template <class F>
void pack_tiles(Packed& dst, int bf, int n_tiles, int K, F get) {
    const int KCH = bf ? 32 : 16, EPL = bf ? 8 : 4, KC = K / KCH;
    dst.resize((size_t)n_tiles * KC * 64 * EPL, bf);
    for (int t = 0; t < n_tiles; ++t)
        for (int c = 0; c < KC; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < EPL; ++e)
                    dst.set((((size_t)t * KC + c) * 64 + lane) * EPL + e, get(t, lane & 15, KCH * c + EPL * (lane >> 4) + e));
}
// "A" tiling of a pseudo-gated (NBRC) phase: tile = 4 units x {3 live gates}; fragment = [g][12 live cols][EPL]
// a = live column (gate = a/4, unit = a%4)
template <class F>
void pack_tiles12(Packed& dst, int bf, int n_tiles, int K, F get) {
    const int KCH = bf ? 32 : 16, EPL = bf ? 8 : 4, KC = K / KCH;
    dst.resize((size_t)n_tiles * KC * 48 * EPL, bf);
    for (int t = 0; t < n_tiles; ++t)
        for (int c = 0; c < KC; ++c)
            for (int g = 0; g < 4; ++g)
                for (int a = 0; a < 12; ++a)
                    for (int e = 0; e < EPL; ++e)
                        dst.set((((size_t)t * KC + c) * 48 + g * 12 + a) * EPL + e, get(t, a, KCH * c + EPL * g + e));
}
int upload_packed(lasr_ctx* c, void** p, const Packed& pk) {
    char* q = nullptr;
    int rc = dalloc(c, &q, pk.bytes.size());
    if (rc) return rc;
    hipError_t e = hipMemcpy(q, pk.bytes.data(), pk.bytes.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(c, LASR_EHIP, "hipMemcpy failed: %s", hipGetErrorString(e));
    *p = q;
    return LASR_OK;
}

bool valid_desc(const lasr_model_desc* d) {
    if (!d) return false;
    auto m16 = [](int v) { return v > 0 && v % 16 == 0; };
    if (!m16(d->feat) || !m16(d->hidden) || !m16(d->embed) || !m16(d->joint) || !m16(d->vocab)) return false;
    if (d->enc_layers < 1 || d->enc_layers > 16 || d->pred_layers < 1 || d->pred_layers > 8) return false;
    if (d->pred_cell != 0 && d->pred_cell != 1) return false;
    if (d->n_fft != 1024 || d->win <= 0 || d->win > d->n_fft || d->hop <= 0) return false;
    if (d->n_mels <= 0 || d->n_stack <= 0 || d->stride <= 0 || d->feat != d->n_mels * d->n_stack) return false;
    if (d->feat > 64 * 32) return false;
    if (d->n_buffer < 1 || d->n_window < 1 || d->chunk <= 0) return false;
    if (d->max_streams < 1 || d->max_streams > 1024) return false;
    if (d->max_iters_offline < 1 || d->max_iters_stream < 1) return false;
    if (d->blank < 0 || d->blank >= d->vocab || d->bos < 0 || d->bos >= d->vocab) return false;
    if ((d->dtype != 0 && d->dtype != 1) || d->beam < 1 || d->beam > 8) return false;
    if ((d->max_streams + 63) / 64 * 64 * d->beam > 1024) return false;      // decoder rows (streams x beam slots)
    if (d->beam > 1 && d->vocab > 4096) return false;                         // k_beam_select keeps a stream's logits in registers
    if (d->dtype == 1) {   // bf16 operands: 32-wide K chunks
        auto m32 = [](int v) { return v % 32 == 0; };
        if (!m32(d->feat) || !m32(d->hidden) || !m32(d->joint)) return false;
    }
    return true;
}

// k_stack_ln: the reference shape (1280 = 128 mels x 10 frames) has a fully static instantiation
#define LAUNCH_STACK_LN(grid, block, shmem, stream, args)                                              \
    do {                                                                                               \
        if ((args).F == 1280 && (args).n_stack == 10)                                                  \
            hipLaunchKernelGGL((k_stack_ln<20, 10>), grid, block, shmem, stream, args);                \
        else                                                                                           \
            hipLaunchKernelGGL((k_stack_ln<32, 0>), grid, block, shmem, stream, args);                 \
    } while (0)

// ---------------------------------------------------------------------------- launch helpers
template <class Ops, class Epi, int MT, bool AROW, int D = 3>
void launch_gemm(lasr_ctx* c, int n_groups, int m_groups, const GemmArgs& g, const typename Epi::Args& ea) {
    hipLaunchKernelGGL((k_gemm<Ops, Epi, MT, NW, AROW, D>), dim3(n_groups, m_groups), dim3(NW * 64), 0, c->stream, g, ea);
}

int grid1(size_t n, int b = 256) { return (int)((n + b - 1) / b); }

// encoder LSTM cell (layer l, step t): x from `xsrc` (fragment-major, K = I); tiling "C"
template <class Ops>
void launch_enc_cell_t(lasr_ctx* c, int l, int t, const void* xsrc, int x_mt_total, void* ydst, int y_mt_total) {
    const Cell& L = c->enc[l];
    const int H = c->d.hidden;
    GemmArgs g{};
    g.A[0] = xsrc; g.a_mt_total[0] = x_mt_total; g.a_mt_off[0] = t * c->MT; g.KC[0] = L.I / Ops::KCH; g.W[0] = L.WxC;
    g.A[1] = c->enc_h[c->enc_par][l]; g.a_mt_total[1] = c->MT; g.a_mt_off[1] = 0; g.KC[1] = H / Ops::KCH; g.W[1] = L.WhC;
    g.M = c->M; g.dbg = c->dbg;
    using E = EpiLSTM<Ops, false, false, 8>;
    typename E::Args ea{};
    ea.bias = L.bias; ea.flag = c->T_row_dev; ea.t = t; ea.tile_mask = c->tile_masks.empty() ? ~0ull : c->tile_masks[t];
    ea.c = c->enc_c[l]; ea.h_in = c->enc_h[c->enc_par][l]; ea.h_out = c->enc_h[c->enc_par ^ 1][l];
    ea.y = ydst; ea.y_mt_total = y_mt_total; ea.y_mt_off = t * c->MT;
    ea.bn_s = L.bn_s; ea.bn_t = L.bn_t; ea.H = H; ea.M = c->M; ea.MT = c->MT;
    launch_gemm<Ops, E, 2, false>(c, H / 8, c->M / 32, g, ea);
}
void launch_enc_cell(lasr_ctx* c, int l, int t, const void* xsrc, int x_mt_total, void* ydst, int y_mt_total) {
    if (c->bf) launch_enc_cell_t<OpsBF16>(c, l, t, xsrc, x_mt_total, ydst, y_mt_total);
    else launch_enc_cell_t<OpsF32>(c, l, t, xsrc, x_mt_total, ydst, y_mt_total);
}

// one predictor pass (all layers) for rows with emit != 0 (compacted inside the kernels); predictor
// state is row-major [M][H]; toggles pred_par
template <class Ops>
void launch_predictor_t(lasr_ctx* c, bool beam) {
    const int H = c->d.hidden;
    const int mgroups = c->Md / (16 * MTA);
    const int p = c->pred_par;
    for (int l = 0; l < c->d.pred_layers; ++l) {
        const Cell& L = c->pred[l];
        GemmArgs g{};
        // beam: parity p holds the current state; everything is written to parity p ^ 1
        void* y_out = (beam && !p) ? c->pred_y1[l] : c->pred_y[l];
        const void* y_in = (beam && p) ? c->pred_y1[l] : c->pred_y[l];
        if (l > 0) {
            g.A[0] = (beam && !p) ? c->pred_y1[l - 1] : c->pred_y[l - 1];   // what layer l-1 just wrote
            g.a_mt_total[0] = H; g.a_mt_off[0] = 0; g.KC[0] = H / Ops::KCH; g.W[0] = L.WxA;
        }
        g.A[1] = c->pred_h[p][l]; g.a_mt_total[1] = H; g.a_mt_off[1] = 0; g.KC[1] = H / Ops::KCH; g.W[1] = L.WhA;
        if (beam) { g.parent = c->b_parent; g.beam_w = c->W; }
        g.compact = c->ds.emit; g.M = c->Md; g.dbg = (c->dbg && c->dbg_gate) ? c->dbg + (size_t)(1 + std::min(l, 1)) * 4096 * 16 : nullptr;
        if (c->d.pred_cell == 1) {
            typename EpiLSTM<Ops, true, true, 4>::Args ea{};
            ea.bias = L.bias; ea.tab = L.tab; ea.token = c->ds.token; ea.flag = c->ds.emit; ea.t = 0;
            ea.c = (beam && !p) ? c->pred_c1[l] : c->pred_c[l]; ea.h_in = c->pred_h[p][l]; ea.h_out = c->pred_h[p ^ 1][l];
            ea.y = y_out; ea.y_mt_total = 0; ea.y_mt_off = 0;
            ea.bn_s = L.bn_s; ea.bn_t = L.bn_t; ea.H = H; ea.M = c->Md; ea.MT = c->MTd;
            if (beam) { ea.parent = c->b_parent; ea.W = c->W; ea.c_in = p ? c->pred_c1[l] : c->pred_c[l]; ea.y_in = y_in; }
            if (l == 0) {
                launch_gemm<Ops, EpiLSTM<Ops, true, true, 4>, MTA, true, -1>(c, H / 4, mgroups, g, ea);
            } else {
                typename EpiLSTM<Ops, true, false, 4>::Args eb{};
                static_assert(sizeof(eb) == sizeof(ea), "same Args layout");
                memcpy(&eb, &ea, sizeof(eb));
                launch_gemm<Ops, EpiLSTM<Ops, true, false, 4>, MTA, true, -1>(c, H / 4, mgroups, g, eb);
            }
        } else {
            typename EpiNBRC<Ops, true>::Args ea{};
            ea.bias = L.bias; ea.rbias = L.rbias; ea.tab = L.tab; ea.token = c->ds.token; ea.emit = c->ds.emit;
            ea.h_in = c->pred_h[p][l]; ea.h_out = c->pred_h[p ^ 1][l]; ea.y = y_out;
            ea.bn_s = L.bn_s; ea.bn_t = L.bn_t; ea.H = H; ea.M = c->Md;
            if (beam) { ea.parent = c->b_parent; ea.W = c->W; ea.y_in = y_in; }
            if (l == 0) {
                launch_gemm<Ops, EpiNBRC<Ops, true>, MTA, true, -1>(c, H / 4, mgroups, g, ea);
            } else {
                typename EpiNBRC<Ops, false>::Args eb{};
                static_assert(sizeof(eb) == sizeof(ea), "same Args layout");
                memcpy(&eb, &ea, sizeof(eb));
                launch_gemm<Ops, EpiNBRC<Ops, false>, MTA, true, -1>(c, H / 4, mgroups, g, eb);
            }
        }
    }
    if (!beam) c->pred_par ^= 1;      // beam: launch_ppj (same pass, same parities) toggles
}
void launch_predictor(lasr_ctx* c, bool beam = false) {
    if (c->bf) launch_predictor_t<OpsBF16>(c, beam);
    else launch_predictor_t<OpsF32>(c, beam);
}

// pp (for emitting rows) and the joint activation ja = tanh(pe[t_idx] + pp) for all rows still decoding
template <class Ops>
void launch_ppj_t(lasr_ctx* c, bool beam) {
    const int H = c->d.hidden, J = c->d.joint, L = c->d.pred_layers, p = c->pred_par;
    GemmArgs g{};
    g.A[0] = (beam && !p) ? c->pred_y1[L - 1] : c->pred_y[L - 1];      // what the predictor pass just wrote
    g.a_mt_total[0] = H; g.a_mt_off[0] = 0; g.KC[0] = H / Ops::KCH; g.W[0] = c->W1p;
    g.compact = c->ds.emit; g.M = c->Md; g.dbg = (c->dbg && c->dbg_gate) ? c->dbg + (size_t)3 * 4096 * 16 : nullptr;
    typename EpiPPJ<Ops>::Args ea{};
    ea.b1 = c->b1; ea.pp = (beam && !p) ? c->pp1 : c->pp; ea.pe = c->pe; ea.t_idx = c->dec_t_idx; ea.T_row = c->T_row_dec; ea.emit = c->ds.emit;
    ea.ja = c->ja; ea.J = J; ea.M = c->Md; ea.MT = c->MTj; ea.ring = c->pe_ring_R; ea.la = beam ? 1 : c->la;
    if (beam) { ea.parent = c->b_parent; ea.W = c->W; ea.M_enc = c->M; ea.pp_in = p ? c->pp1 : c->pp; }
    launch_gemm<Ops, EpiPPJ<Ops>, 1, true, -1>(c, J / 16, c->MTd, g, ea);
    if (beam) c->pred_par ^= 1;
}
void launch_ppj(lasr_ctx* c, bool beam = false) {
    if (c->bf) launch_ppj_t<OpsBF16>(c, beam);
    else launch_ppj_t<OpsF32>(c, beam);
}
float* cur_pp(lasr_ctx* c) { return (c->W > 1 && c->pred_par) ? c->pp1 : c->pp; }

template <bool AROW, int D>
void launch_linear(lasr_ctx* c, int n_groups, int m_groups, GemmArgs g, int K, const EpiLinear::Args& ea);

// LMFuser.advance (lm.py:49-53) for the rows with emit != 0: LM step on the token just emitted, then
// log_softmax + standardise + [0] = MIN_VAL into lmz (read by the next k_select of that row)
template <class Ops>
void launch_lm_t(lasr_ctx* c) {
    lasr_ctx::LM& m = c->lm;
    const int H = m.H, M = c->M, V = c->d.vocab, p = m.par;
    for (int l = 0; l < m.L; ++l) {
        const Cell& L = m.cells[l];
        GemmArgs g{};
        if (l > 0) { g.A[0] = m.y[l - 1]; g.a_mt_total[0] = H; g.a_mt_off[0] = 0; g.KC[0] = H / Ops::KCH; g.W[0] = L.WxA; }
        g.A[1] = m.h[p][l]; g.a_mt_total[1] = H; g.a_mt_off[1] = 0; g.KC[1] = H / Ops::KCH; g.W[1] = L.WhA;
        g.compact = c->ds.emit; g.M = M;
        typename EpiLSTM<Ops, true, true, 4>::Args ea{};
        ea.bias = L.bias; ea.tab = L.tab; ea.token = c->ds.token; ea.flag = c->ds.emit; ea.t = 0;
        ea.c = m.cst[l]; ea.h_in = m.h[p][l]; ea.h_out = m.h[p ^ 1][l]; ea.y = m.y[l];
        ea.bn_s = m.ones; ea.bn_t = m.zeros; ea.H = H; ea.M = M; ea.MT = c->MT;
        if (l == 0) {
            launch_gemm<Ops, EpiLSTM<Ops, true, true, 4>, MTA, true, -1>(c, H / 4, M / (16 * MTA), g, ea);
        } else {
            typename EpiLSTM<Ops, true, false, 4>::Args eb{};
            memcpy(&eb, &ea, sizeof(eb));
            launch_gemm<Ops, EpiLSTM<Ops, true, false, 4>, MTA, true, -1>(c, H / 4, M / (16 * MTA), g, eb);
        }
    }
    m.par ^= 1;
    GemmArgs g{};
    g.A[0] = m.y[m.L - 1]; g.a_mt_total[0] = H; g.a_mt_off[0] = 0; g.W[0] = m.Wout; g.a_rows = M;
    EpiLinear::Args ea{};
    ea.bias = m.bout; ea.out = m.raw; ea.ldo = V; ea.n_rows = M; ea.t_idx = nullptr; ea.T_row = nullptr; ea.M = M;
    launch_linear<true, -1>(c, V / 16, M / 16, g, H, ea);
    hipLaunchKernelGGL(k_lm_post, dim3(M), dim3(256), 0, c->stream, (const float*)m.raw, (const int*)c->ds.emit, m.lmz, m.valid, V, m.min_val);
}
void launch_lm(lasr_ctx* c) {
    if (!c->lm.on) return;
    if (c->bf) launch_lm_t<OpsBF16>(c);
    else launch_lm_t<OpsF32>(c);
}

// plain linear over element-typed A (fragment-major, or row-major when AROW); f32 row-major output
template <bool AROW, int D>
void launch_linear(lasr_ctx* c, int n_groups, int m_groups, GemmArgs g, int K, const EpiLinear::Args& ea) {
    g.KC[0] = K / c->kch;
    if (c->bf) launch_gemm<OpsBF16, EpiLinear, 1, AROW, D>(c, n_groups, m_groups, g, ea);
    else launch_gemm<OpsF32, EpiLinear, 1, AROW, D>(c, n_groups, m_groups, g, ea);
}

void launch_logits(lasr_ctx* c, float* out, int n_rows, bool gated) {
    const int J = c->d.joint, V = c->d.vocab;
    GemmArgs g{};
    g.A[0] = c->ja; g.a_mt_total[0] = c->MTj; g.a_mt_off[0] = 0; g.W[0] = c->W2; g.M = c->Md;
    g.dbg = (c->dbg && c->dbg_gate) ? c->dbg + (size_t)4 * 4096 * 16 : nullptr;
    EpiLinear::Args ea{};
    ea.bias = c->b2; ea.out = out; ea.ldo = V; ea.n_rows = n_rows;
    ea.t_idx = gated ? c->dec_t_idx : nullptr; ea.T_row = c->T_row_dec; ea.M = c->M; ea.W = c->W;
    launch_linear<false, -1>(c, V / 16, (n_rows + 15) / 16, g, J, ea);
}

// ---------------------------------------------------------------------------- command blocks
size_t cmd_layout(lasr_ctx::Cmd& k, char* base, int M) {
    size_t o = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += (bytes + 15) & ~size_t(15); return p; };
    k.T_row = (int*)take(sizeof(int) * M); k.what = (int*)take(sizeof(int) * M);
    k.src_idx = (int*)take(sizeof(int) * M); k.feat_sel = (int*)take(sizeof(int) * M);
    k.row_frames = (int*)take(sizeof(int) * M); k.token = (int*)take(sizeof(int) * M);
    k.emit = (int*)take(sizeof(int) * M);
    k.row_N = (long long*)take(sizeof(long long) * M); k.row_src_off = (long long*)take(sizeof(long long) * M);
    k.row_feat_off = (long long*)take(sizeof(long long) * M);
    return o;
}

// next command block: c->hc (host views) / c->dc (device views); zero-initialised
int cmd_begin(lasr_ctx* c) {
    if (c->cmd_inflight >= NCMD - 1) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->cmd_inflight = 0;
    }
    const int i = c->cmd_next;
    c->cmd_next = (i + 1) % NCMD;
    c->cmd_inflight++;
    cmd_layout(c->hc, c->cmd_host + (size_t)i * c->cmd_bytes, c->M);
    cmd_layout(c->dc, c->cmd_dev + (size_t)i * c->cmd_bytes, c->M);
    memset(c->cmd_host + (size_t)i * c->cmd_bytes, 0, c->cmd_bytes);
    return LASR_OK;
}
int cmd_commit(lasr_ctx* c) {
    HIPCHK(c, hipMemcpyAsync((char*)c->dc.T_row, (char*)c->hc.T_row, c->cmd_bytes, hipMemcpyHostToDevice, c->stream));
    return LASR_OK;
}

// device copy of the step's T_row (from the committed command block) + host-side per-step masks of
// the m-tiles that contain an active row (passed by value to the encoder cell kernels)
int commit_T_rows(lasr_ctx* c, int T_max) {
    c->T_row_dev = c->dc.T_row;             // the command ring (NCMD blocks) outlives every step in flight
    c->T_row_dec = c->T_row_dev;
    c->tile_masks.assign(std::max(T_max, 1), 0ull);
    for (int t = 0; t < T_max; ++t) {
        unsigned long long m = 0;
        for (int r = 0; r < c->M; ++r)
            if (t < c->hc.T_row[r]) m |= 1ull << (r >> 4);
        c->tile_masks[t] = m;
    }
    return LASR_OK;
}

// ---------------------------------------------------------------------------- buffers that grow
int ensure_T(lasr_ctx* c, int T) {
    if (T <= c->Tcap) return LASR_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second);   // captured pointers become stale
    c->graphs.clear();
    for (auto& kv : c->cgraphs) (void)hipGraphExecDestroy(kv.second);
    c->cgraphs.clear();
    const int M = c->M, H = c->d.hidden, F = c->d.feat, J = c->d.joint;
    int cap = std::max(T, std::max(2 * c->Tcap, c->d.n_buffer));
    dfree(c, c->x0); dfree(c, c->ybuf[0]); dfree(c, c->ybuf[1]); dfree(c, c->pe_sync);
    c->pe_sync = nullptr;
    dfree(c, c->ds.step_ntok); dfree(c, c->ds.unfinished);
    c->x0 = c->ybuf[0] = c->ybuf[1] = c->pe = nullptr; c->ds.step_ntok = nullptr; c->ds.step_tok = nullptr; c->ds.unfinished = nullptr;
    RC(dalloc(c, (char**)&c->x0, (size_t)cap * M * F * c->esz));
    RC(dalloc(c, (char**)&c->ybuf[0], (size_t)cap * M * H * c->esz));
    RC(dalloc(c, (char**)&c->ybuf[1], (size_t)cap * M * H * c->esz));
    RC(dalloc(c, &c->pe_sync, (size_t)cap * M * J));
    c->pe = c->pe_sync;
    const int mi = std::max(c->d.max_iters_offline, c->d.max_iters_stream);
    c->tok_cap_alloc = cap * mi;
    // [ntok M][tokens M x tok_cap]: one contiguous block so a group's results reach the host in one copy
    RC(dalloc(c, &c->ds.step_ntok, (size_t)M + (size_t)M * c->tok_cap_alloc));
    HIPCHK(c, hipMemset(c->ds.step_ntok, 0, sizeof(int) * M));
    c->ds.step_tok = c->ds.step_ntok + M;
    c->n_iter_slots = cap * mi + 8;
    RC(dalloc(c, &c->ds.unfinished, (size_t)c->n_iter_slots));
    if (c->W > 1) {
        dfree(c, c->b_trellis); c->b_trellis = nullptr;
        RC(dalloc(c, &c->b_trellis, (size_t)c->n_iter_slots * c->Md));
        if (c->trellis_host) (void)hipHostFree(c->trellis_host);
        c->trellis_host_ints = (size_t)c->n_iter_slots * c->Md + 4 * (size_t)c->Md + 16;
        HIPCHK(c, hipHostMalloc((void**)&c->trellis_host, sizeof(int) * c->trellis_host_ints));
    }
    HIPCHK(c, hipMemset(c->ybuf[0], 0, (size_t)cap * M * H * c->esz));
    HIPCHK(c, hipMemset(c->ybuf[1], 0, (size_t)cap * M * H * c->esz));
    HIPCHK(c, hipMemset(c->x0, 0, (size_t)cap * M * F * c->esz));
    // pinned result block: [0] unfinished, then ntok[M], sum_iters[M], n_ones[M], logp[M] (double), tokens
    if (c->res_host) (void)hipHostFree(c->res_host);
    c->res_bytes = sizeof(int) * (8 + 3 * (size_t)M) + sizeof(double) * M + sizeof(int) * (size_t)M * c->tok_cap_alloc + 64;
    HIPCHK(c, hipHostMalloc((void**)&c->res_host, c->res_bytes));
    c->Tcap = cap;
    return LASR_OK;
}

template <class T>
int ensure_buf(lasr_ctx* c, T** p, size_t* have, size_t need) {
    if (need <= *have) return LASR_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dfree(c, *p);
    *p = nullptr;
    need = need + need / 4;
    RC(dalloc(c, p, need));
    *have = need;
    return LASR_OK;
}

// ---------------------------------------------------------------------------- reset
// applies c->dc.what (already committed) to the state; runs the predictor on BOS for rows with bit 2
// plain_rows: op-level entry points address predictor rows directly (row = batch index, greedy kernels)
int apply_reset(lasr_ctx* c, bool any_pred, int mask = 3, bool plain_rows = false) {
    const bool beam = c->W > 1 && !plain_rows;
    ResetArgs a{};
    a.what = c->dc.what; a.mask = mask; a.M = c->M; a.MT = c->MT; a.H = c->d.hidden; a.Le = c->d.enc_layers; a.Lp = c->d.pred_layers;
    a.pred_lstm = c->d.pred_cell; a.bos = c->d.bos; a.bf = c->bf;
    a.W = beam ? c->W : 1; a.Md = c->Md; a.score = c->b_score; a.alive = c->b_alive; a.inB = c->b_inB; a.parent = c->b_parent;
    for (int l = 0; l < a.Le; ++l) {
        a.enc_h[l] = c->enc_h[c->enc_par][l]; a.enc_c[l] = c->enc_c[l];
        a.enc_h0[l] = c->enc[l].h0; a.enc_c0[l] = c->enc[l].c0;
    }
    for (int l = 0; l < a.Lp; ++l) {
        a.pred_h[l] = c->pred_h[c->pred_par][l];
        a.pred_c[l] = c->d.pred_cell ? ((beam && c->pred_par) ? c->pred_c1[l] : c->pred_c[l]) : nullptr;
        a.pred_h0[l] = c->pred[l].h0; a.pred_c0[l] = c->pred[l].c0;
    }
    a.token = c->ds.token; a.emit = c->ds.emit;
    hipLaunchKernelGGL(k_reset_rows, dim3(grid1((size_t)c->M * c->d.hidden)), dim3(256), 0, c->stream, a);
    if (c->lm.on && (mask & 2)) {      // LM state lives on the decode side, like the predictor's
        LmResetArgs la{};
        la.what = c->dc.what; la.M = c->M; la.H = c->lm.H; la.L = c->lm.L; la.bf = c->bf; la.lm_valid = c->lm.valid;
        for (int l = 0; l < c->lm.L; ++l) { la.h[l] = c->lm.h[c->lm.par][l]; la.c[l] = c->lm.cst[l]; }
        hipLaunchKernelGGL(k_lm_reset, dim3(grid1((size_t)c->M * c->lm.H)), dim3(256), 0, c->stream, la);
    }
    if (any_pred) {
        // T_row = 0 for every row: EpiPPJ then only refreshes pp (models.py:489: predictor(BOS))
        int* keep_dec = c->T_row_dec;
        c->T_row_dec = c->zero_rows;
        HIPCHK(c, hipMemsetAsync(c->ds.t_idx, 0, sizeof(int) * c->M, c->stream));
        launch_predictor(c, beam);
        launch_ppj(c, beam);
        c->T_row_dec = keep_dec;
    }
    return LASR_OK;
}

// ---------------------------------------------------------------------------- encoder + decode
// Encoder over T_max frames for rows with T_row > 0 (x0 already holds LayerNorm'ed features).
void run_encoder(lasr_ctx* c, int T_max) {
    const int L = c->d.enc_layers;
    const int mt_total = c->Tcap * c->MT;
    const int par0 = c->enc_par;
    // layer-major order: every layer starts from parity par0 and toggles T_max times (enc_h[par][l] is
    // indexed by the parity at launch time, so all layers end on par0 ^ (T_max & 1))
    for (int l = 0; l < L; ++l) {
        c->enc_par = par0;
        const void* xsrc = (l == 0) ? c->x0 : c->ybuf[(l - 1) & 1];
        void* ydst = c->ybuf[l & 1];
        for (int t = 0; t < T_max; ++t) {
            launch_enc_cell(c, l, t, xsrc, mt_total, ydst, mt_total);
            c->enc_par ^= 1;
        }
    }
    // encoder half of the joint for all frames: pe[t][r] = W1e * enc[t][r]
    const int H = c->d.hidden, J = c->d.joint;
    GemmArgs g{};
    g.A[0] = c->ybuf[(L - 1) & 1]; g.a_mt_total[0] = mt_total; g.a_mt_off[0] = 0; g.W[0] = c->W1e;
    EpiLinear::Args ea{};
    ea.bias = nullptr; ea.out = c->pe; ea.ldo = J; ea.n_rows = T_max * c->M; ea.t_idx = nullptr; ea.T_row = nullptr; ea.M = c->M;
    if (c->pe == c->pe_ring) { ea.ring_base = c->c_enc_frames; ea.ring = lasr_ctx::RING; }   // continuous mode: per-row frame ring
    launch_linear<false, 3>(c, J / 16, T_max * c->MT, g, H, ea);
}

// Greedy decode of the current step (T_row_dev, pe ready).  Blocks until done; fills host queues.
int run_decode_beam(lasr_ctx* c, int T_max, int max_iters, bool offline, const std::vector<int>& rows);

int run_decode(lasr_ctx* c, int T_max, int max_iters, bool offline, const std::vector<int>& rows) {
    if (c->W > 1) return run_decode_beam(c, T_max, max_iters, offline, rows);
    const int M = c->M, J = c->d.joint, V = c->d.vocab;
    c->la = offline ? c->la_offline : c->la_stream;
    DecState s = c->ds;
    s.tok_cap = T_max * max_iters;
    const int total_cap = T_max * max_iters;
    int iter = 0;
    // iterations are launched in even-sized groups (the predictor ping-pong parity then returns to
    // its start); after each group the "rows still decoding" counter and the step's tokens so far
    // come back in the same round trip.  In streaming mode every group is a cached hipGraph: one
    // launch instead of 4 kernels per iteration, so the GPU is not fed at host launch speed.
    // (with lookahead a row consumes up to `la` blank frames per iteration: fewer iterations up front)
    int group = offline ? std::min(total_cap, ((T_max + c->la - 1) / c->la + 16) & ~1) : std::min(total_cap, (T_max + 4) & ~1);
    const int next_group = offline ? 32 : 4;
    int* res = c->res_host;
    int* ntok = res + 4;
    int* toks = ntok + M;                      // contiguous with ntok, as on the device
    int* sum_iters = toks + (size_t)M * s.tok_cap;
    int* n_ones = sum_iters + M;
    double* logp = (double*)(((uintptr_t)(n_ones + M) + 15) & ~uintptr_t(15));
    // (the legacy NULL stream cannot be captured: graphs then only serve the pipelined path, whose
    //  decode loop runs on the ctx-owned stream_dec)
    const bool graphs = c->use_graphs && !offline && !c->profiling && !c->dbg && c->stream != nullptr;
    const int buf_idx = 0;
    auto enqueue_group = [&](int first, int n) -> int {
        if (first == 0) {
            hipLaunchKernelGGL(k_step_begin, dim3(grid1(std::max(M, c->n_iter_slots))), dim3(256), 0, c->stream, s, M,
                               c->n_iter_slots, offline ? 1 : 0);
            hipLaunchKernelGGL(k_ja, dim3(grid1((size_t)M * J)), dim3(256), 0, c->stream, c->pe, c->pp, c->dec_t_idx,
                               c->T_row_dec, c->ja, J, M, c->MTj, c->pe_ring_R, c->bf, 1, M, c->la);
        }
        for (int q = 0; q < n; ++q) {
            const int it = first + q;
            c->dbg_gate = (it == 0);
            launch_logits(c, c->logits, c->la * M, true);
            hipLaunchKernelGGL((k_select<false>), dim3(M), dim3(256), 0, c->stream, c->logits, V, c->d.blank, max_iters,
                               c->T_row_dec, s, it, (float*)nullptr, (int*)nullptr, c->la, M);
            launch_predictor(c);
            launch_ppj(c);
            launch_lm(c);
        }
        // payload first, the "rows still decoding" word last: the host spins on that word
        HIPCHK(c, hipMemcpyAsync(ntok, c->ds.step_ntok, sizeof(int) * ((size_t)M + (size_t)M * s.tok_cap), hipMemcpyDeviceToHost, c->stream));
        if (offline) {
            HIPCHK(c, hipMemcpyAsync(sum_iters, c->ds.sum_iters, sizeof(int) * M, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(n_ones, c->ds.n_ones, sizeof(int) * M, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(logp, c->ds.logp_sum, sizeof(double) * M, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipMemcpyAsync(res, c->ds.unfinished + (first + n - 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
        return LASR_OK;
    };
    while (iter < total_cap) {
        const int n = std::min(group, total_cap - iter);
        bool launched = false;
        if (graphs && (n % 2) == 0) {
            const auto key = std::make_tuple(iter, n, buf_idx, c->pred_par + 2 * c->lm.par, T_max * 1024 + max_iters);
            auto it = c->graphs.find(key);
            if (it == c->graphs.end()) {
                hipGraph_t gr = nullptr;
                hipGraphExec_t ex = nullptr;
                HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                int rc = enqueue_group(iter, n);
                hipError_t e = hipStreamEndCapture(c->stream, &gr);
                if (rc) return rc;
                if (e != hipSuccess || !gr) return fail(c, LASR_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
                e = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
                (void)hipGraphDestroy(gr);
                if (e != hipSuccess) return fail(c, LASR_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
                it = c->graphs.emplace(key, ex).first;
            }
            HIPCHK(c, hipGraphLaunch(it->second, c->stream));
            launched = true;
        }
        __atomic_store_n(&res[0], -1, __ATOMIC_RELEASE);      // sentinel, overwritten by the last copy of the group
        if (!launched) RC(enqueue_group(iter, n));
        iter += n;
        // spin on the pinned word instead of hipStreamSynchronize (interrupt wake-up costs ~10-20 us per
        // round trip, and there are 2-4 per step); fall back to a real sync if nothing arrives
        {
            unsigned long long spins = 0;
            while (__atomic_load_n((volatile int*)&res[0], __ATOMIC_ACQUIRE) == -1) {
                __builtin_ia32_pause();
                if (++spins > (1ull << 27)) { HIPCHK(c, hipStreamSynchronize(c->stream)); break; }
            }
        }
        if (res[0] == 0) break;
        group = next_group;
    }
    c->stats.decode_iters = iter;
    for (int r : rows) {
        const int n = std::min(ntok[r], s.tok_cap);
        for (int q = 0; q < n; ++q) c->queue[r].push_back(toks[(size_t)r * s.tok_cap + q]);
        if (offline) {
            c->neg_logp[r] = -logp[r];
            // alignment_score = (sum(iters) - #frames with 1 iter) / (sum(iters) + 1e-4)  (models.py:447-453)
            c->align[r] = ((double)sum_iters[r] - (double)n_ones[r]) / ((double)sum_iters[r] + 1e-4);
        }
    }
    return LASR_OK;
}

// Beam search over the current step (W > 1): one selection round per iteration for every stream that
// still has frames; blocks until done.  The per-round (parent, token) records come back in one copy and
// are replayed on the host into the token history of every hypothesis slot.
int run_decode_beam(lasr_ctx* c, int T_max, int max_iters, bool offline, const std::vector<int>& rows) {
    const int M = c->M, Md = c->Md, W = c->W, J = c->d.joint;
    BeamState b{};
    b.W = W; b.V = c->d.vocab; b.blank = c->d.blank; b.max_iters = max_iters; b.Md = Md;
    b.t_idx = c->ds.t_idx; b.iters = c->ds.iters; b.T_row = c->T_row_dec;
    b.score = c->b_score; b.alive = c->b_alive; b.inB = c->b_inB; b.token = c->ds.token; b.emit = c->ds.emit;
    b.parent = c->b_parent; b.trellis = c->b_trellis; b.unfinished = c->ds.unfinished;
    b.dbg = c->dbg ? c->dbg + (size_t)4 * 4096 * 16 : nullptr;      // reuses the "logits" slot of the debug buffer
    const int total_cap = T_max * max_iters;
    if (total_cap + 1 > c->n_iter_slots) return fail(c, LASR_EINVAL, "decode iteration budget exceeds the trellis");
    int* res = c->res_host;
    hipLaunchKernelGGL(k_beam_begin, dim3(grid1(std::max(Md, c->n_iter_slots))), dim3(256), 0, c->stream, b, M, c->n_iter_slots);
    hipLaunchKernelGGL(k_ja, dim3(grid1((size_t)Md * J)), dim3(256), 0, c->stream, (const float*)c->pe, (const float*)cur_pp(c),
                       (const int*)c->dec_t_idx, (const int*)c->T_row_dec, c->ja, J, Md, c->MTj, c->pe_ring_R, c->bf, W, M, 1);
    int iter = 0;
    int group = offline ? std::min(total_cap, T_max + 16) : std::min(total_cap, T_max + 4);
    const int next_group = offline ? 32 : 4;
    c->dbg_gate = false;
    while (iter < total_cap) {
        const int n = std::min(group, total_cap - iter);
        for (int q = 0; q < n; ++q) {
            launch_logits(c, c->logits, Md, true);
            if (W <= 2) hipLaunchKernelGGL((k_beam_select<2>), dim3(M), dim3(1024), 0, c->stream, (const float*)c->logits, b, iter + q);
            else if (W <= 4) hipLaunchKernelGGL((k_beam_select<4>), dim3(M), dim3(1024), 0, c->stream, (const float*)c->logits, b, iter + q);
            else hipLaunchKernelGGL((k_beam_select<8>), dim3(M), dim3(1024), 0, c->stream, (const float*)c->logits, b, iter + q);
            launch_predictor(c, true);
            launch_ppj(c, true);
        }
        iter += n;
        __atomic_store_n(&res[0], -1, __ATOMIC_RELEASE);
        HIPCHK(c, hipMemcpyAsync(res, c->ds.unfinished + (iter - 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
        unsigned long long spins = 0;
        while (__atomic_load_n((volatile int*)&res[0], __ATOMIC_ACQUIRE) == -1) {
            __builtin_ia32_pause();
            if (++spins > (1ull << 27)) { HIPCHK(c, hipStreamSynchronize(c->stream)); break; }
        }
        if (res[0] == 0) break;
        group = next_group;
    }
    c->stats.decode_iters = iter;
    // results: the rounds' records + final scores
    int* tre = c->trellis_host;
    double* sc = (double*)(tre + (((size_t)iter * Md + 1) & ~size_t(1)));
    int* alive = (int*)(sc + Md);
    HIPCHK(c, hipMemcpyAsync(tre, c->b_trellis, sizeof(int) * (size_t)iter * Md, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(sc, c->b_score, sizeof(double) * Md, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(alive, c->b_alive, sizeof(int) * Md, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<std::vector<int32_t>> nh(W);
    for (int r : rows) {
        auto& H = c->hyp[r];
        for (int it = 0; it < iter; ++it) {
            const int* e = tre + (size_t)it * Md + (size_t)r * W;
            if (e[0] == -1) continue;                          // stream idle in this round
            for (int j = 0; j < W; ++j) {
                if (e[j] < 0) { nh[j].clear(); continue; }     // dead slot
                nh[j] = H[e[j] >> 16];
                const int tok = e[j] & 0xffff;
                if (tok) nh[j].push_back(tok - 1);
            }
            for (int j = 0; j < W; ++j) H[j].swap(nh[j]);
        }
        int best = -1;
        for (int j = 0; j < W; ++j)
            if (alive[(size_t)r * W + j] && (best < 0 || sc[(size_t)r * W + j] > sc[(size_t)r * W + best])) best = j;
        auto& q = c->best_full[r];
        q = c->committed[r];                                   // what earlier predictor resets froze
        double score = c->committed_score[r];
        if (best >= 0) { q.insert(q.end(), H[best].begin(), H[best].end()); score += sc[(size_t)r * W + best]; }
        c->queue[r] = q;                                       // beam mode: lasr_fetch hands out the whole best hypothesis
        c->neg_logp[r] = -score;
        c->align[r] = 0.0;                                     // alignment_score is a greedy-loop metric
    }
    return LASR_OK;
}

// host side of a predictor reset in beam mode: the best hypothesis so far is frozen, the beam restarts
void beam_host_reset(lasr_ctx* c, int slot, bool forget) {
    if (c->W <= 1) return;
    auto& H = c->hyp[slot];
    if (forget) { c->committed[slot].clear(); c->committed_score[slot] = 0.0; c->best_full[slot].clear(); }
    else { c->committed[slot] = c->best_full[slot]; c->committed_score[slot] = -c->neg_logp[slot]; }
    for (auto& h : H) h.clear();
}

void rec(lasr_ctx* c, int i) {
    if (c->profiling && c->ev_ok) (void)hipEventRecord(c->ev[i], c->stream);
}
void collect_stats(lasr_ctx* c, int T) {
    c->stats.frames = T;
    if (!(c->profiling && c->ev_ok)) return;
    float a = 0, b = 0, d = 0;
    (void)hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&b, c->ev[1], c->ev[2]);
    (void)hipEventElapsedTime(&d, c->ev[2], c->ev[3]);
    c->stats.frontend_ms = a; c->stats.encoder_ms = b; c->stats.decode_ms = d;
    c->stats.cell_ms = b; c->stats.cell_launches = T * c->d.enc_layers;
}

int check_slots(lasr_ctx* c, const int* slots, int n, bool need_open) {
    if (!slots || n < 0 || n > c->d.max_streams) return fail(c, LASR_EINVAL, "bad slot list (n=%d)", n);
    std::vector<char> seen(c->M, 0);
    for (int i = 0; i < n; ++i) {
        const int s = slots[i];
        if (s < 0 || s >= c->d.max_streams) return fail(c, LASR_EINVAL, "slot %d out of range", s);
        if (seen[s]) return fail(c, LASR_EINVAL, "slot %d listed twice", s);
        seen[s] = 1;
        if (need_open && !c->open_[s]) return fail(c, LASR_ESTATE, "slot %d is not open", s);
    }
    return LASR_OK;
}

int require_idle(lasr_ctx* c) {
    if (!c->pending.empty()) return fail(c, LASR_ESTATE, "%d submitted step(s) not collected: call lasr_step_wait first", (int)c->pending.size());
    return LASR_OK;
}

bool is_device_ptr(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeDevice;
}

// HTK mel filterbank (torchaudio 0.6.0 create_fb_matrix semantics), sparse, bin-ascending
void build_fb(const lasr_model_desc& d, std::vector<int>& start, std::vector<int>& off, std::vector<float>& w) {
    const int nf = d.n_fft / 2 + 1, nm = d.n_mels;
    const double fmax = d.sample_rate / 2;
    auto mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
    std::vector<double> fpts(nm + 2);
    const double m0 = mel(0.0), m1 = mel(fmax);
    for (int i = 0; i < nm + 2; ++i) {
        const double m = m0 + (m1 - m0) * i / (nm + 1);
        fpts[i] = 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
    }
    start.assign(nm, 0); off.assign(nm + 1, 0); w.clear();
    for (int m = 0; m < nm; ++m) {
        int first = -1;
        std::vector<float> vals;
        for (int k = 0; k < nf; ++k) {
            const double f = fmax * k / (nf - 1);
            const double down = (f - fpts[m]) / (fpts[m + 1] - fpts[m]);
            const double up = (fpts[m + 2] - f) / (fpts[m + 2] - fpts[m + 1]);
            const double v = std::max(0.0, std::min(down, up));
            if (v > 0.0) {
                if (first < 0) first = k;
                while ((int)vals.size() < k - first) vals.push_back(0.f);
                vals.push_back((float)v);
            }
        }
        start[m] = first < 0 ? 0 : first;
        off[m] = (int)w.size();
        w.insert(w.end(), vals.begin(), vals.end());
    }
    off[nm] = (int)w.size();
}

// ---------------------------------------------------------------------------- weight loading
struct Reader {
    const float* p; size_t left;
    const float* take(size_t n) {
        if (n > left) return nullptr;
        const float* q = p; p += n; left -= n; return q;
    }
};

int fold_bn(lasr_ctx* c, Reader& rd, int H, float** s_dev, float** t_dev) {
    const float* w = rd.take(H); const float* b = rd.take(H); const float* mean = rd.take(H); const float* var = rd.take(H);
    if (!var) return fail(c, LASR_EINVAL, "weight blob too short (bn)");
    std::vector<float> s(H), t(H);
    for (int i = 0; i < H; ++i) {
        const double sc = (double)w[i] / std::sqrt((double)var[i] + 1e-5);   // BatchNorm1d eps
        s[i] = (float)sc;
        t[i] = (float)((double)b[i] - (double)mean[i] * sc);
    }
    RC(upload(c, s_dev, s.data(), H));
    RC(upload(c, t_dev, t.data(), H));
    return LASR_OK;
}

// LSTM layer in torch layout: W_ih [4H,I], W_hh [4H,H].
//   tiling C (encoder): tile t = (jb = t/2, nt = t%2): column col -> gate 2*nt + col/8, unit 8*jb + col%8
//   tiling A (predictor): tile jb = 4 units x 4 gates (col = gate*4 + unit)
int load_lstm(lasr_ctx* c, Reader& rd, Cell& L, int I, int H, bool tiling_a, std::vector<float>* keep_wih,
              std::vector<float>* keep_bias) {
    L.I = I;
    const float* wih = rd.take((size_t)4 * H * I); const float* whh = rd.take((size_t)4 * H * H);
    const float* bih = rd.take(4 * H); const float* bhh = rd.take(4 * H);
    if (!bhh) return fail(c, LASR_EINVAL, "weight blob too short (lstm)");
    Packed pk;
    if (!tiling_a) {
        pack_tiles(pk, c->bf, (H / 8) * 2, I, [&](int t, int col, int k) { return wih[((size_t)(2 * (t & 1) + (col >> 3)) * H + 8 * (t >> 1) + (col & 7)) * I + k]; });
        RC(upload_packed(c, &L.WxC, pk));
        pack_tiles(pk, c->bf, (H / 8) * 2, H, [&](int t, int col, int k) { return whh[((size_t)(2 * (t & 1) + (col >> 3)) * H + 8 * (t >> 1) + (col & 7)) * H + k]; });
        RC(upload_packed(c, &L.WhC, pk));
    } else {
        pack_tiles(pk, c->bf, H / 4, I, [&](int t, int col, int k) { return wih[((size_t)(col >> 2) * H + 4 * t + (col & 3)) * I + k]; });
        RC(upload_packed(c, &L.WxA, pk));
        pack_tiles(pk, c->bf, H / 4, H, [&](int t, int col, int k) { return whh[((size_t)(col >> 2) * H + 4 * t + (col & 3)) * H + k]; });
        RC(upload_packed(c, &L.WhA, pk));
    }
    std::vector<float> bias(4 * H);
    for (int i = 0; i < 4 * H; ++i) bias[i] = bih[i] + bhh[i];
    RC(upload(c, &L.bias, bias.data(), bias.size()));
    if (keep_wih) keep_wih->assign(wih, wih + (size_t)4 * H * I);
    if (keep_bias) *keep_bias = bias;
    return LASR_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

void lasr_default_desc(lasr_model_desc* d) {
    if (!d) return;
    memset(d, 0, sizeof(*d));
    d->feat = 1280; d->hidden = 1024; d->enc_layers = 4; d->pred_layers = 2; d->pred_cell = 0;
    d->embed = 512; d->joint = 1024; d->vocab = 2048; d->blank = 0; d->bos = 2;
    d->n_fft = 1024; d->win = 400; d->hop = 160; d->n_mels = 128; d->n_stack = 10; d->stride = 8;
    d->n_buffer = 2; d->n_window = 3; d->chunk = 1280; d->sample_rate = 16000; d->dtype = 0;
    d->max_streams = 64; d->max_iters_offline = 3; d->max_iters_stream = 10; d->beam = 1;
}

size_t lasr_weight_count(const lasr_model_desc* d) {
    if (!valid_desc(d)) return 0;
    const size_t F = d->feat, H = d->hidden, E = d->embed, V = d->vocab, J = d->joint;
    size_t n = 2 * F;
    for (int l = 0; l < d->enc_layers; ++l) {
        const size_t I = l == 0 ? F : H;
        n += 2 * H + 4 * H + 4 * H * I + 4 * H * H + 8 * H;
    }
    n += V * E;
    if (E != H) n += H * E + H;
    for (int l = 0; l < d->pred_layers; ++l) {
        if (d->pred_cell == 1) n += 2 * H + 4 * H + 4 * H * H + 4 * H * H + 8 * H;
        else n += H + 4 * H + 3 * H * H + 3 * H * H + 6 * H;
    }
    n += J * 2 * H + J + V * J + V;
    return n;
}

void lasr_destroy(lasr_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (void* p : c->dev_allocs) (void)hipFree(p);
    if (c->cmd_host) (void)hipHostFree(c->cmd_host);
    if (c->res_host) (void)hipHostFree(c->res_host);
    if (c->cont_host) (void)hipHostFree(c->cont_host);
    if (c->trellis_host) (void)hipHostFree(c->trellis_host);
    if (c->push_stage_host) (void)hipHostFree(c->push_stage_host);
    if (c->stream_copy) { (void)hipStreamSynchronize(c->stream_copy); (void)hipStreamDestroy(c->stream_copy); }
    for (auto& e : c->push_copied)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->push_ev)
        if (e) (void)hipEventDestroy(e);
    for (void* p : c->host_allocs) (void)hipHostFree(p);
    if (c->ev_ok)
        for (auto& e : c->ev) (void)hipEventDestroy(e);
    for (auto& e : c->ev_enc)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_misc) (void)hipEventDestroy(c->ev_misc);
    for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second);
    for (auto& kv : c->cgraphs) (void)hipGraphExecDestroy(kv.second);
    if (c->stream_dec) { (void)hipStreamSynchronize(c->stream_dec); (void)hipStreamDestroy(c->stream_dec); }
    delete c;
}

const char* lasr_last_error(const lasr_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

static int create_impl(lasr_ctx* c, const float* weights, size_t n_weights) {
    const lasr_model_desc& d = c->d;
    const int F = d.feat, H = d.hidden, E = d.embed, V = d.vocab, J = d.joint;
    c->M = (d.max_streams + 16 * MTA - 1) / (16 * MTA) * (16 * MTA);   // whole "A"-tiling row groups
    c->MT = c->M / 16;
    const int M = c->M;
    c->W = d.beam; c->Md = M * c->W; c->MTd = c->Md / 16;
    const int Md = c->Md;
    c->MTj = std::max(c->Md, lasr_ctx::LA_MAX * M) / 16;
    {   // frames evaluated per row and iteration in the greedy loop (see k_select); measured best on configs[1]
        const char* e = getenv("LASR_LOOKAHEAD");
        if (e) c->la_stream = c->la_offline = std::min(lasr_ctx::LA_MAX, std::max(1, atoi(e)));
        if (c->W > 1) c->la_stream = c->la_offline = 1;
        c->la = c->la_stream;
        if (getenv("LASR_KICK")) c->kick_n = std::max(1, atoi(getenv("LASR_KICK")));
        if (getenv("LASR_GROUP")) c->wait_n = std::max(1, atoi(getenv("LASR_GROUP")));
    }
    const size_t Mj = (size_t)c->MTj * 16;
    c->G_pred = d.pred_cell ? 4 : 3;
    c->bf = d.dtype == 1; c->kch = c->bf ? 32 : 16; c->esz = c->bf ? 2 : 4;
    Reader rd{weights, n_weights};
    {
        if (getenv("LASR_NO_GRAPH")) c->use_graphs = false;
        if (getenv("LASR_DBG_TIMING")) {
            RC(dalloc(c, &c->dbg, (size_t)5 * 4096 * 16));
            HIPCHK(c, hipMemset(c->dbg, 0, sizeof(unsigned long long) * 5 * 4096 * 16));
        }
    }

    // ---- front-end constants
    {
        std::vector<float> win(d.n_fft, 0.f);
        const int off = (d.n_fft - d.win) / 2;
        for (int i = 0; i < d.win; ++i) win[off + i] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * i / d.win));  // periodic Hann
        RC(upload(c, &c->window, win.data(), win.size()));
        std::vector<float2> t5(512), t10(513);
        for (int i = 0; i < 512; ++i) t5[i] = float2{(float)std::cos(2.0 * M_PI * i / 512), (float)-std::sin(2.0 * M_PI * i / 512)};
        for (int i = 0; i <= 512; ++i) t10[i] = float2{(float)std::cos(2.0 * M_PI * i / 1024), (float)-std::sin(2.0 * M_PI * i / 1024)};
        RC(upload(c, &c->tw512, t5.data(), t5.size()));
        RC(upload(c, &c->tw1024, t10.data(), t10.size()));
        std::vector<int> st, of; std::vector<float> w;
        build_fb(d, st, of, w);
        RC(upload(c, &c->fb_start, st.data(), st.size()));
        RC(upload(c, &c->fb_off, of.data(), of.size()));
        RC(upload(c, &c->fb_w, w.data(), w.size()));
        c->fb_nnz = (int)w.size();
        if (c->fb_nnz > 1536 || d.n_mels > 128) return fail(c, LASR_EINVAL, "mel filterbank too large for the LDS staging (%d weights)", c->fb_nnz);
    }
    // ---- encoder
    {
        const float* lw = rd.take(F); const float* lb = rd.take(F);
        if (!lb) return fail(c, LASR_EINVAL, "weight blob too short");
        RC(upload(c, &c->ln_w, lw, F)); RC(upload(c, &c->ln_b, lb, F));
    }
    c->enc.resize(d.enc_layers);
    for (int l = 0; l < d.enc_layers; ++l) {
        Cell& L = c->enc[l];
        const float* hs = rd.take(2 * H);
        if (!hs) return fail(c, LASR_EINVAL, "weight blob too short");
        RC(upload(c, &L.h0, hs, H)); RC(upload(c, &L.c0, hs + H, H));
        RC(fold_bn(c, rd, H, &L.bn_s, &L.bn_t));
        RC(load_lstm(c, rd, L, l == 0 ? F : H, H, false, nullptr, nullptr));
    }
    // ---- predictor
    const float* embed = rd.take((size_t)V * E);
    if (!embed) return fail(c, LASR_EINVAL, "weight blob too short");
    const float *ffn_w = nullptr, *ffn_b = nullptr;
    if (E != H) {
        ffn_w = rd.take((size_t)H * E); ffn_b = rd.take(H);
        if (!ffn_b) return fail(c, LASR_EINVAL, "weight blob too short");
    }
    c->pred.resize(d.pred_layers);
    std::vector<float> in0_w;      // layer-0 input weights as [G*H][H] (n-major) for the table build
    std::vector<float> in0_b;
    for (int l = 0; l < d.pred_layers; ++l) {
        Cell& L = c->pred[l];
        L.I = H;
        const int S = d.pred_cell ? 2 : 1;
        const float* hs = rd.take((size_t)S * H);
        if (!hs) return fail(c, LASR_EINVAL, "weight blob too short");
        RC(upload(c, &L.h0, hs, H));
        if (S == 2) RC(upload(c, &L.c0, hs + H, H));
        RC(fold_bn(c, rd, H, &L.bn_s, &L.bn_t));
        if (d.pred_cell == 1) {
            RC(load_lstm(c, rd, L, H, H, true, l == 0 ? &in0_w : nullptr, l == 0 ? &in0_b : nullptr));
        } else {
            const float* kx = rd.take((size_t)H * 3 * H); const float* kh = rd.take((size_t)H * 3 * H);
            const float* b = rd.take(3 * H); const float* rb = rd.take(3 * H);
            if (!rb) return fail(c, LASR_EINVAL, "weight blob too short (nbrc)");
            Packed pk;
            // haste layout [K][3H], gates z,r,g; tiling A: tile = 4 units, live column a -> (gate a/4, unit a%4)
            pack_tiles12(pk, c->bf, H / 4, H, [&](int t, int a, int k) { return kx[(size_t)k * 3 * H + (size_t)(a >> 2) * H + 4 * t + (a & 3)]; });
            RC(upload_packed(c, &L.WxA, pk));
            pack_tiles12(pk, c->bf, H / 4, H, [&](int t, int a, int k) { return kh[(size_t)k * 3 * H + (size_t)(a >> 2) * H + 4 * t + (a & 3)]; });
            RC(upload_packed(c, &L.WhA, pk));
            RC(upload(c, &L.bias, b, 3 * H)); RC(upload(c, &L.rbias, rb, 3 * H));
            if (l == 0) {
                in0_w.resize((size_t)3 * H * H);
                for (int n = 0; n < 3 * H; ++n)
                    for (int k = 0; k < H; ++k) in0_w[(size_t)n * H + k] = kx[(size_t)k * 3 * H + n];
                in0_b.assign(b, b + 3 * H);
            }
        }
    }
    // ---- joint
    {
        const float* w0 = rd.take((size_t)J * 2 * H); const float* b0 = rd.take(J);
        const float* w2 = rd.take((size_t)V * J); const float* b2 = rd.take(V);
        if (!b2) return fail(c, LASR_EINVAL, "weight blob too short (joint)");
        if (rd.left != 0) return fail(c, LASR_EINVAL, "weight blob has %zu extra floats", rd.left);
        Packed pk;
        pack_tiles(pk, c->bf, J / 16, H, [&](int t, int ui, int k) { return w0[(size_t)(16 * t + ui) * 2 * H + k]; });          // pred half (cat order pred, enc: models.py:136)
        RC(upload_packed(c, &c->W1p, pk));
        pack_tiles(pk, c->bf, J / 16, H, [&](int t, int ui, int k) { return w0[(size_t)(16 * t + ui) * 2 * H + H + k]; });
        RC(upload_packed(c, &c->W1e, pk));
        pack_tiles(pk, c->bf, V / 16, J, [&](int t, int ui, int k) { return w2[(size_t)(16 * t + ui) * J + k]; });
        RC(upload_packed(c, &c->W2, pk));
        RC(upload(c, &c->b1, b0, J)); RC(upload(c, &c->b2, b2, V));
    }

    // ---- state + work buffers
    for (int p = 0; p < 2; ++p) { c->enc_h[p].resize(d.enc_layers); c->pred_h[p].resize(d.pred_layers); }
    c->enc_c.resize(d.enc_layers); c->pred_c.assign(d.pred_layers, nullptr); c->pred_y.resize(d.pred_layers);
    for (int l = 0; l < d.enc_layers; ++l) {
        for (int p = 0; p < 2; ++p) { RC(dalloc(c, (char**)&c->enc_h[p][l], (size_t)M * H * c->esz)); HIPCHK(c, hipMemset(c->enc_h[p][l], 0, (size_t)M * H * c->esz)); }
        RC(dalloc(c, &c->enc_c[l], (size_t)M * H)); HIPCHK(c, hipMemset(c->enc_c[l], 0, (size_t)M * H * 4));
    }
    for (int l = 0; l < d.pred_layers; ++l) {
        for (int p = 0; p < 2; ++p) { RC(dalloc(c, (char**)&c->pred_h[p][l], (size_t)Md * H * c->esz)); HIPCHK(c, hipMemset(c->pred_h[p][l], 0, (size_t)Md * H * c->esz)); }
        if (d.pred_cell) { RC(dalloc(c, &c->pred_c[l], (size_t)Md * H)); HIPCHK(c, hipMemset(c->pred_c[l], 0, (size_t)Md * H * 4)); }
        RC(dalloc(c, (char**)&c->pred_y[l], (size_t)Md * H * c->esz)); HIPCHK(c, hipMemset(c->pred_y[l], 0, (size_t)Md * H * c->esz));
    }
    RC(dalloc(c, &c->pp, (size_t)Md * J)); HIPCHK(c, hipMemset(c->pp, 0, (size_t)Md * J * 4));
    if (c->W > 1) {       // second parity of every per-hypothesis buffer + the beam bookkeeping
        c->pred_c1.assign(d.pred_layers, nullptr); c->pred_y1.assign(d.pred_layers, nullptr);
        for (int l = 0; l < d.pred_layers; ++l) {
            if (d.pred_cell) { RC(dalloc(c, &c->pred_c1[l], (size_t)Md * H)); HIPCHK(c, hipMemset(c->pred_c1[l], 0, (size_t)Md * H * 4)); }
            RC(dalloc(c, (char**)&c->pred_y1[l], (size_t)Md * H * c->esz)); HIPCHK(c, hipMemset(c->pred_y1[l], 0, (size_t)Md * H * c->esz));
        }
        RC(dalloc(c, &c->pp1, (size_t)Md * J)); HIPCHK(c, hipMemset(c->pp1, 0, (size_t)Md * J * 4));
        RC(dalloc(c, &c->b_score, Md)); RC(dalloc(c, &c->b_alive, Md)); RC(dalloc(c, &c->b_inB, Md)); RC(dalloc(c, &c->b_parent, Md));
        HIPCHK(c, hipMemset(c->b_score, 0, sizeof(double) * Md)); HIPCHK(c, hipMemset(c->b_alive, 0, sizeof(int) * Md));
        HIPCHK(c, hipMemset(c->b_inB, 0, sizeof(int) * Md)); HIPCHK(c, hipMemset(c->b_parent, 0, sizeof(int) * Md));
        c->hyp.assign(M, std::vector<std::vector<int32_t>>(c->W));
        c->committed.assign(M, {}); c->committed_score.assign(M, 0.0); c->best_full.assign(M, {});
    }
    RC(dalloc(c, (char**)&c->ja, Mj * J * c->esz)); HIPCHK(c, hipMemset(c->ja, 0, Mj * J * c->esz));
    RC(dalloc(c, (char**)&c->cvt_a, (size_t)M * H * c->esz)); RC(dalloc(c, (char**)&c->cvt_b, (size_t)M * H * c->esz));
    RC(dalloc(c, &c->logits, Mj * V));
    RC(dalloc(c, &c->ds.t_idx, M)); RC(dalloc(c, &c->ds.iters, M)); RC(dalloc(c, &c->ds.token, Md));
    RC(dalloc(c, &c->ds.emit, Md)); RC(dalloc(c, &c->ds.logp_sum, M));
    RC(dalloc(c, &c->ds.sum_iters, M)); RC(dalloc(c, &c->ds.n_ones, M)); RC(dalloc(c, &c->zero_rows, M));
    HIPCHK(c, hipMemset(c->zero_rows, 0, sizeof(int) * M));
    c->T_row_dev = c->zero_rows;
    for (int q = 0; q < lasr_ctx::NFLY; ++q) {
        RC(dalloc(c, &c->T_row_ring[q], M));
        HIPCHK(c, hipMemset(c->T_row_ring[q], 0, sizeof(int) * M));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_enc[q], hipEventDisableTiming));
    }
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream_dec, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_misc, hipEventDisableTiming));
    RC(dalloc(c, &c->pe_ring, (size_t)lasr_ctx::RING * M * J));
    HIPCHK(c, hipMemset(c->pe_ring, 0, sizeof(float) * (size_t)lasr_ctx::RING * M * J));
    RC(dalloc(c, &c->c_cur, M)); RC(dalloc(c, &c->c_avail, M)); RC(dalloc(c, &c->c_iters, M)); RC(dalloc(c, &c->c_target, M));
    RC(dalloc(c, &c->c_ntotal, M)); RC(dalloc(c, &c->c_enc_frames, M)); RC(dalloc(c, &c->c_behind, 64));
    // (c_ntok_end / c_tok_ring live in pinned host memory, written by k_select directly: see below)
    for (int* p : {c->c_cur, c->c_avail, c->c_iters, c->c_target, c->c_ntotal, c->c_enc_frames})
        HIPCHK(c, hipMemset(p, 0, sizeof(int) * M));
    HIPCHK(c, hipMemset(c->c_behind, 0, sizeof(int) * 64));
    RC(dalloc(c, &c->c_done, 64)); HIPCHK(c, hipMemset(c->c_done, 0, sizeof(int) * 64));
    RC(dalloc(c, &c->c_iter, 4)); HIPCHK(c, hipMemset(c->c_iter, 0, sizeof(int) * 4));
    HIPCHK(c, hipHostMalloc((void**)&c->cont_host, sizeof(int) * (16 + (size_t)M * (lasr_ctx::NFLY + lasr_ctx::ENDSLOTS + lasr_ctx::TOKRING))));
    memset(c->cont_host, 0, sizeof(int) * (16 + (size_t)M * (lasr_ctx::NFLY + lasr_ctx::ENDSLOTS + lasr_ctx::TOKRING)));
    {   // continuous decode: the token ring and the per-step boundary marks are written by k_select straight
        // into this pinned block (zero-copy stores over PCIe, flushed at kernel end): a finished step needs
        // no result copy at all -- the host reads them as soon as the group's "rows behind" word says 0
        void* dp = nullptr;
        HIPCHK(c, hipHostGetDevicePointer(&dp, c->cont_host, 0));
        c->c_flag_dev = (int*)dp;
        c->c_ntok_end = (int*)dp + 16 + (size_t)lasr_ctx::NFLY * M;
        c->c_tok_ring = c->c_ntok_end + (size_t)M * lasr_ctx::ENDSLOTS;
    }
    c->h_frames_sub.assign(M, 0); c->h_fetched.assign(M, 0);
    c->dec_t_idx = c->ds.t_idx;
    c->T_row_dec = c->T_row_dev;
    for (int* p : {c->ds.t_idx, c->ds.iters, c->ds.sum_iters, c->ds.n_ones})
        HIPCHK(c, hipMemset(p, 0, sizeof(int) * M));
    HIPCHK(c, hipMemset(c->ds.token, 0, sizeof(int) * Md)); HIPCHK(c, hipMemset(c->ds.emit, 0, sizeof(int) * Md));
    HIPCHK(c, hipMemset(c->ds.logp_sum, 0, sizeof(double) * M));
    RC(dalloc(c, &c->win, (size_t)M * d.n_window * d.chunk)); HIPCHK(c, hipMemset(c->win, 0, (size_t)M * d.n_window * d.chunk * 4));
    RC(dalloc(c, &c->ring_pos, M)); HIPCHK(c, hipMemset(c->ring_pos, 0, sizeof(int) * M));
    RC(dalloc(c, &c->pend, (size_t)M * d.n_buffer * d.n_stack * d.n_mels));
    HIPCHK(c, hipMemset(c->pend, 0, (size_t)M * d.n_buffer * d.n_stack * d.n_mels * 4));

    lasr_ctx::Cmd tmp;
    c->cmd_bytes = cmd_layout(tmp, nullptr, M);
    HIPCHK(c, hipHostMalloc((void**)&c->cmd_host, c->cmd_bytes * NCMD));
    RC(dalloc(c, &c->cmd_dev, c->cmd_bytes * NCMD));
    RC(ensure_T(c, std::max(d.n_buffer, 4)));

    // ---- predictor input tables (one-time, on device, always exact f32: the table is f32 in both
    //      dtypes):  EF = ffn(embed);  tab = EF * Wx0^T + b
    {
        float* emb_dev = nullptr; float* EF = nullptr;
        RC(upload(c, &emb_dev, embed, (size_t)V * E));
        if (E != H) {
            Packed pk; void* wf = nullptr; float* bfn = nullptr;
            pack_tiles(pk, 0, H / 16, E, [&](int t, int ui, int k) { return ffn_w[(size_t)(16 * t + ui) * E + k]; });
            RC(upload_packed(c, &wf, pk)); RC(upload(c, &bfn, ffn_b, H));
            RC(dalloc(c, &EF, (size_t)V * H));
            GemmArgs g{}; g.A[0] = emb_dev; g.a_mt_total[0] = E; g.a_mt_off[0] = 0; g.KC[0] = E / 16; g.W[0] = wf; g.a_rows = V;
            EpiLinear::Args ea{}; ea.bias = bfn; ea.out = EF; ea.ldo = H; ea.n_rows = V; ea.t_idx = nullptr; ea.T_row = nullptr; ea.M = M;
            launch_gemm<OpsF32, EpiLinear, 1, true>(c, H / 16, V / 16, g, ea);
            HIPCHK(c, hipStreamSynchronize(c->stream));
            dfree(c, wf); dfree(c, bfn);
        } else {
            EF = emb_dev; emb_dev = nullptr;
        }
        const int G = c->G_pred;
        Packed pk; void* wt = nullptr; float* bt = nullptr;
        pack_tiles(pk, 0, G * H / 16, H, [&](int t, int ui, int k) { return in0_w[(size_t)(16 * t + ui) * H + k]; });
        RC(upload_packed(c, &wt, pk)); RC(upload(c, &bt, in0_b.data(), in0_b.size()));
        RC(dalloc(c, &c->pred[0].tab, (size_t)V * G * H));
        GemmArgs g{}; g.A[0] = EF; g.a_mt_total[0] = H; g.a_mt_off[0] = 0; g.KC[0] = H / 16; g.W[0] = wt; g.a_rows = V;
        EpiLinear::Args ea{}; ea.bias = bt; ea.out = c->pred[0].tab; ea.ldo = G * H; ea.n_rows = V; ea.t_idx = nullptr; ea.T_row = nullptr; ea.M = M;
        launch_gemm<OpsF32, EpiLinear, 1, true>(c, G * H / 16, V / 16, g, ea);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        dfree(c, wt); dfree(c, bt); dfree(c, EF); dfree(c, emb_dev);
    }

    c->open_.assign(M, 0); c->n_chunks.assign(M, 0); c->n_pend.assign(M, 0);
    c->queue.assign(M, {}); c->neg_logp.assign(M, 0.0); c->align.assign(M, 0.0);
    c->ev_ok = true;
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) c->ev_ok = false;
    return LASR_OK;
}

int lasr_create(int device, const lasr_model_desc* d, const float* weights, size_t n_weights, void* hip_stream,
                lasr_ctx** out) {
    if (!out) return LASR_EINVAL;
    *out = nullptr;
    lasr_ctx* c = new lasr_ctx();
    if (!valid_desc(d)) { int rc = fail(c, LASR_EINVAL, "invalid model description"); *out = c; return rc; }
    c->d = *d;
    c->device = device;
    *out = c;   // returned even on failure so that lasr_last_error() works; caller destroys it
    if (!weights || n_weights != lasr_weight_count(d))
        return fail(c, LASR_EINVAL, "weight blob has %zu floats, expected %zu", n_weights, lasr_weight_count(d));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(c, LASR_EHIP, "no HIP device available");
    HIPCHK(c, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(c, LASR_EHIP, "device %d is %s; liblasr_hip is built for gfx950 only", device, prop.gcnArchName);
    c->stream = (hipStream_t)hip_stream;
    return create_impl(c, weights, n_weights);
}

// ---------------------------------------------------------------------------- slots
int lasr_stream_open(lasr_ctx* c, int* slot) {
    if (!c || !slot) return LASR_EINVAL;
    for (int s = 0; s < c->d.max_streams; ++s)
        if (!c->open_[s]) {
            c->open_[s] = 1;
            *slot = s;
            return lasr_stream_reset(c, s, 1 | 2 | 4 | 8);
        }
    return fail(c, LASR_EFULL, "all %d stream slots are open", c->d.max_streams);
}

int lasr_stream_reset(lasr_ctx* c, int slot, int what) {
    if (!c) return LASR_EINVAL;
    if (slot < 0 || slot >= c->d.max_streams || !c->open_[slot]) return fail(c, LASR_ESTATE, "slot %d is not open", slot);
    for (const auto& p : c->pending)
        if (std::find(p.rows.begin(), p.rows.end(), slot) != p.rows.end())
            return fail(c, LASR_ESTATE, "slot %d has a submitted step in flight: call lasr_step_wait first", slot);
    HIPCHK(c, hipSetDevice(c->device));
    if (what & 8) { c->n_chunks[slot] = 0; c->n_pend[slot] = 0; c->queue[slot].clear(); c->neg_logp[slot] = 0.0; }
    if (what & 2) beam_host_reset(c, slot, (what & 8) != 0);
    if (what & 7) {
        RC(cmd_begin(c));
        c->hc.what[slot] = what & 7;
        RC(cmd_commit(c));
        if (c->pending.empty() && !c->group_inflight) {
            RC(apply_reset(c, (what & 2) != 0));
        } else {
            // other streams have steps in flight: the encoder side of the reset is ordered on the main
            // stream, the predictor / LM side (BOS pass) on the decode stream, between two iteration groups
            if (what & 1) RC(apply_reset(c, false, 1));
            if (what & 6) {
                HIPCHK(c, hipEventRecord(c->ev_misc, c->stream));
                hipStream_t keep = c->stream;
                HIPCHK(c, hipStreamWaitEvent(c->stream_dec, c->ev_misc, 0));
                c->stream = c->stream_dec;
                int rc = apply_reset(c, (what & 2) != 0, 2);
                c->stream = keep;
                if (rc) return rc;
            }
        }
    }
    return LASR_OK;
}

int lasr_stream_close(lasr_ctx* c, int slot) {
    if (!c) return LASR_EINVAL;
    if (slot < 0 || slot >= c->d.max_streams || !c->open_[slot]) return fail(c, LASR_ESTATE, "slot %d is not open", slot);
    for (const auto& p : c->pending)
        if (std::find(p.rows.begin(), p.rows.end(), slot) != p.rows.end())
            return fail(c, LASR_ESTATE, "slot %d has a submitted step in flight: call lasr_step_wait first", slot);
    c->open_[slot] = 0;
    c->queue[slot].clear();
    return LASR_OK;
}

// ---------------------------------------------------------------------------- streaming
int lasr_push_pcm(lasr_ctx* c, const int* slots, int n, const float* pcm) {
    if (!c) return LASR_EINVAL;
    RC(check_slots(c, slots, n, true));
    if (n == 0) return LASR_OK;
    if (!pcm) return fail(c, LASR_EINVAL, "pcm is null");
    HIPCHK(c, hipSetDevice(c->device));
    const int CH = c->d.chunk;
    const float* src = pcm;
    const bool from_host = !is_device_ptr(pcm);
    int stage_i = -1;
    if (from_host) {
        if (!c->push_stage) {
            RC(dalloc(c, &c->push_stage, (size_t)lasr_ctx::NSTAGE * c->M * CH));
            HIPCHK(c, hipHostMalloc((void**)&c->push_stage_host, sizeof(float) * (size_t)lasr_ctx::NSTAGE * c->M * CH));
            for (auto& e : c->push_ev) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto& e : c->push_copied) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HIPCHK(c, hipStreamCreateWithFlags(&c->stream_copy, hipStreamNonBlocking));
        }
        stage_i = c->push_next;
        c->push_next = (stage_i + 1) % lasr_ctx::NSTAGE;
        if (c->push_used[stage_i]) HIPCHK(c, hipEventSynchronize(c->push_ev[stage_i]));   // its last reader (16 pushes ago) is done
        float* dst = c->push_stage + (size_t)stage_i * c->M * CH;
        float* pin = c->push_stage_host + (size_t)stage_i * c->M * CH;
        memcpy(pin, pcm, sizeof(float) * (size_t)n * CH);      // the caller's buffer is free on return, whatever its kind
        HIPCHK(c, hipMemcpyAsync(dst, pin, sizeof(float) * (size_t)n * CH, hipMemcpyHostToDevice, c->stream_copy));   // truly asynchronous
        HIPCHK(c, hipEventRecord(c->push_copied[stage_i], c->stream_copy));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->push_copied[stage_i], 0));
        src = dst;
    }
    if (c->M <= 512) {      // slot -> staging-row map by value: no command-block copy for a push
        PushIdx pi;
        for (int r = 0; r < 512; ++r) pi.idx[r] = -1;
        for (int i = 0; i < n; ++i) pi.idx[slots[i]] = (short)i;
        hipLaunchKernelGGL(k_push_pcm, dim3(c->M), dim3(256), 0, c->stream, src, (const int*)nullptr, pi, c->win, c->ring_pos, CH, c->d.n_window);
    } else {
        RC(cmd_begin(c));
        for (int r = 0; r < c->M; ++r) c->hc.src_idx[r] = -1;
        for (int i = 0; i < n; ++i) c->hc.src_idx[slots[i]] = i;
        RC(cmd_commit(c));
        PushIdx pi;
        hipLaunchKernelGGL(k_push_pcm, dim3(c->M), dim3(256), 0, c->stream, src, (const int*)c->dc.src_idx, pi, c->win, c->ring_pos, CH, c->d.n_window);
    }
    for (int i = 0; i < n; ++i) c->n_chunks[slots[i]]++;
    if (from_host) {
        HIPCHK(c, hipEventRecord(c->push_ev[stage_i], c->stream));
        c->push_used[stage_i] = true;
    }
    return LASR_OK;
}

// front-end of one client chunk for the listed slots and, for the slots whose frame buffer filled up,
// LayerNorm + encoder + encoder half of the joint -- all enqueued on c->stream, nothing synchronises
static int enqueue_frontend_encoder(lasr_ctx* c, const int* slots, int n, std::vector<int>& model_rows, int& Tm) {
    const lasr_model_desc& d = c->d;
    // window geometry (api-server.py:95-102 + TransformTime + StreamPostprocess)
    const long long N = (long long)d.n_window * d.chunk;
    const int T = 1 + (int)(N / d.hop);
    const int a0 = T / 3 + 1;
    const int nf = std::min(d.n_stack, T - a0);
    if (nf < d.n_stack) return fail(c, LASR_EINVAL, "chunk of %d samples is too short: window yields %d < n_stack frames", d.chunk, nf);
    if (N <= d.n_fft / 2) return fail(c, LASR_EINVAL, "window shorter than the reflect padding");
    RC(cmd_begin(c));
    model_rows.clear();
    bool any_feat = false;
    for (int r = 0; r < c->M; ++r) c->hc.feat_sel[r] = -1;
    for (int i = 0; i < n; ++i) {
        const int s = slots[i];
        if (c->n_chunks[s] < d.n_window) continue;          // window not full: the servicer does not call the pipeline
        c->hc.feat_sel[s] = c->n_pend[s] * d.n_stack;
        any_feat = true;
        if (++c->n_pend[s] == d.n_buffer) {                  // Buffer.encodes: emit when n_buffer collected
            c->n_pend[s] = 0;
            c->hc.T_row[s] = d.n_buffer;
            model_rows.push_back(s);
        }
    }
    RC(cmd_commit(c));
    rec(c, 0);
    if (any_feat) {
        MelArgs m{};
        m.window = c->window; m.tw512 = c->tw512; m.tw1024 = c->tw1024; m.fb_start = c->fb_start; m.fb_off = c->fb_off;
        m.fb_w = c->fb_w; m.n_mels = d.n_mels; m.hop = d.hop; m.pcm = c->win; m.N = N; m.stream = 1;
        m.ring_head = c->ring_pos; m.chunk = d.chunk; m.n_window = d.n_window; m.row_sel = c->dc.feat_sel; m.frame0 = a0;
        m.frames_per_row = d.n_stack; m.out = c->pend; m.out_frames = d.n_buffer * d.n_stack;
        m.row_N = nullptr; m.row_src_off = nullptr; m.row_frames = nullptr;
        m.win_off = (d.n_fft - d.win) / 2; m.win_len = d.win; m.fb_nnz = c->fb_nnz;
        hipLaunchKernelGGL(k_logmel, dim3((d.n_stack + 3) / 4, c->M), dim3(256), 0, c->stream, m);
    }
    Tm = d.n_buffer;
    if (model_rows.empty()) return LASR_OK;
    RC(ensure_T(c, Tm));
    RC(commit_T_rows(c, Tm));
    {
        StackLnArgs a{};
        a.src = c->pend; a.mode = 0; a.src_frames = d.n_buffer * d.n_stack; a.frame_step = d.n_stack; a.row_off = nullptr;
        a.T_row = c->T_row_dev; a.ln_w = c->ln_w; a.ln_b = c->ln_b; a.x0 = c->x0; a.F = d.feat; a.n_mels = d.n_mels;
        a.n_stack = d.n_stack; a.M = c->M; a.MT = c->MT; a.mt_total = c->Tcap * c->MT; a.feats_out = nullptr; a.bf = c->bf; a.Tmax = Tm;
        LAUNCH_STACK_LN( dim3((Tm + 3) / 4, c->M), dim3(256), 0, c->stream, a);
    }
    rec(c, 1);
    run_encoder(c, Tm);
    rec(c, 2);
    return LASR_OK;
}

int lasr_step_stream(lasr_ctx* c, const int* slots, int n, int* n_ran) {
    if (!c) return LASR_EINVAL;
    if (n_ran) *n_ran = 0;
    RC(check_slots(c, slots, n, true));
    RC(require_idle(c));
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int> model_rows;
    int Tm = 0;
    RC(enqueue_frontend_encoder(c, slots, n, model_rows, Tm));
    if (model_rows.empty()) {
        HIPCHK(c, hipGetLastError());
        return LASR_OK;
    }
    RC(run_decode(c, Tm, c->d.max_iters_stream, false, model_rows));
    rec(c, 3);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    collect_stats(c, Tm);
    if (n_ran) *n_ran = (int)model_rows.size();
    return LASR_OK;
}

static int cont_launch_group(lasr_ctx* c, int G);
static void cont_poll(lasr_ctx* c);

// Pipelined + continuous form of lasr_step_stream.  submit: front-end + encoder of this chunk on the
// main stream (the encoder half of the joint goes to a per-row frame ring).  wait: keeps ONE greedy
// loop running on stream_dec until every row of the OLDEST submitted step has consumed that step's
// frames; rows that are done early continue with the frames of the later, already encoded steps, so
// the latency-bound tail of a bursty stream overlaps useful work instead of idling 63 rows.
// Tokens are attributed to the step whose frames produced them: per-step results are identical to
// lasr_step_stream.
int lasr_step_submit(lasr_ctx* c, const int* slots, int n) {
    if (!c) return LASR_EINVAL;
    RC(check_slots(c, slots, n, true));
    if (c->W > 1) return fail(c, LASR_ESTATE, "lasr_step_submit is greedy-only; use lasr_step_stream with beam > 1");
    if ((int)c->pending.size() >= lasr_ctx::NFLY - 1) return fail(c, LASR_ESTATE, "%d steps already in flight: call lasr_step_wait", (int)c->pending.size());
    HIPCHK(c, hipSetDevice(c->device));
    // keep the decode stream busy while the host enqueues (and the GPU runs) this chunk's encoder
    cont_poll(c);
    if (!c->pending.empty() && !c->group_inflight) {
        c->kick_iters = c->kick_n;
        RC(cont_launch_group(c, c->kick_iters));
    }
    const int idx = (int)(c->model_steps % lasr_ctx::NFLY);
    float* pe_keep = c->pe;
    c->pe = c->pe_ring;                     // run_encoder writes the joint's encoder half into the ring
    std::vector<int> model_rows;
    int Tm = 0;
    const bool prof = c->profiling;
    c->profiling = false;
    int rc = enqueue_frontend_encoder(c, slots, n, model_rows, Tm);
    c->profiling = prof;
    c->pe = pe_keep;
    if (rc) return rc;
    if (model_rows.empty()) {
        HIPCHK(c, hipGetLastError());
        return LASR_OK;
    }
    hipLaunchKernelGGL(k_advance, dim3(grid1(c->M)), dim3(256), 0, c->stream, c->c_enc_frames, (const int*)c->T_row_dev, c->M);
    HIPCHK(c, hipEventRecord(c->ev_enc[idx], c->stream));
    lasr_ctx::PendingStep p;
    p.rows = model_rows; p.Tm = Tm; p.idx = idx; p.admitted = false; p.target_set = false;
    p.T_row_ptr = c->T_row_dev;             // lives in the command ring until long after this step is collected
    p.serial = c->model_steps;
    p.target.assign(c->M, 0);
    for (int r : model_rows) { c->h_frames_sub[r] += Tm; p.target[r] = (int)c->h_frames_sub[r]; }
    c->pending.push_back(std::move(p));
    c->model_steps++;
    HIPCHK(c, hipGetLastError());
    return LASR_OK;
}

int lasr_step_pending(lasr_ctx* c) { return c ? (int)c->pending.size() : 0; }

static int spin_flag(lasr_ctx* c, volatile int* flag, hipStream_t st) {
    unsigned long long spins = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == -1) {
        __builtin_ia32_pause();
        if (++spins > (1ull << 27)) { HIPCHK(c, hipStreamSynchronize(st)); break; }
    }
    return LASR_OK;
}

// decode-side view of the ctx while the continuous loop is being fed (restored by the guard)
struct ContScope {
    lasr_ctx* c; hipStream_t st; float* pe; int ring; int* tidx; int* trow;
    explicit ContScope(lasr_ctx* c_) : c(c_), st(c_->stream), pe(c_->pe), ring(c_->pe_ring_R), tidx(c_->dec_t_idx), trow(c_->T_row_dec) {
        c->stream = c->stream_dec; c->pe = c->pe_ring; c->pe_ring_R = lasr_ctx::RING;
        c->dec_t_idx = c->c_cur; c->T_row_dec = c->c_avail;
    }
    ~ContScope() { c->stream = st; c->pe = pe; c->pe_ring_R = ring; c->dec_t_idx = tidx; c->T_row_dec = trow; }
};

// One group of G greedy iterations on stream_dec for whatever rows have frames to decode, followed by
// the copy of "rows of the oldest pending step still behind" into the pinned flag.  Does not wait.
static int cont_launch_group(lasr_ctx* c, int G) {
    const int M = c->M, V = c->d.vocab, J = c->d.joint;
    c->la = c->la_stream;
    ContScope scope(c);
    lasr_ctx::PendingStep& P = c->pending.front();
    DecState s = c->ds;
    s.t_idx = c->c_cur; s.iters = c->c_iters; s.step_ntok = c->c_ntotal; s.step_tok = c->c_tok_ring;
    s.tok_cap = lasr_ctx::TOKRING; s.unfinished = c->c_behind; s.cont = 1; s.target = c->c_target;
    s.ntok_end = c->c_ntok_end; s.step_T = P.Tm; s.end_slots = lasr_ctx::ENDSLOTS; s.done_blocks = c->c_done;
    s.iter_ctr = c->c_iter;
    int* flag = c->cont_host;
    int* tgt_stage = c->cont_host + 16;
    // admit encoded steps in order: the oldest unconditionally, later ones only if their encoder is done
    bool admitted_any = false;
    for (auto& q : c->pending) {
        if (q.admitted) continue;
        if (&q != &P && hipEventQuery(c->ev_enc[q.idx]) != hipSuccess) { (void)hipGetLastError(); break; }
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_enc[q.idx], 0));
        hipLaunchKernelGGL(k_advance, dim3(grid1(M)), dim3(256), 0, c->stream, c->c_avail, q.T_row_ptr, M);
        q.admitted = true;
        admitted_any = true;
    }
    if (admitted_any)   // rows that were idle need their joint activation for the new frames
        hipLaunchKernelGGL(k_ja, dim3(grid1((size_t)M * J)), dim3(256), 0, c->stream, c->pe, c->pp, c->c_cur,
                           c->c_avail, c->ja, J, M, c->MTj, c->pe_ring_R, c->bf, 1, M, c->la);
    if (!P.target_set) {
        int* st = tgt_stage + (size_t)P.idx * M;
        memcpy(st, P.target.data(), sizeof(int) * M);
        HIPCHK(c, hipMemcpyAsync(c->c_target, st, sizeof(int) * M, hipMemcpyHostToDevice, c->stream));
        P.target_set = true;
    }
    __atomic_store_n(flag, -1, __ATOMIC_RELEASE);      // before the launch that will overwrite it
    c->dbg_gate = false;
    // the G iterations are launch-invariant (the flag-ring slot comes from a device counter, the last k_select
    // publishes the "rows behind" word): replayed as one hipGraph per (G, ping-pong parities)
    auto enqueue = [&]() {
        for (int q = 0; q < G; ++q) {
            s.host_flag = (q == G - 1) ? c->c_flag_dev : nullptr;
            launch_logits(c, c->logits, c->la * M, true);
            hipLaunchKernelGGL((k_select<false>), dim3(M), dim3(256), 0, c->stream, c->logits, V, c->d.blank, c->d.max_iters_stream,
                               c->c_avail, s, 0, (float*)nullptr, (int*)nullptr, c->la, M);
            launch_predictor(c);
            launch_ppj(c);
            launch_lm(c);
        }
    };
    if (c->use_graphs && !c->dbg) {
        const auto key = std::make_tuple(G, c->pred_par, c->lm.par);
        auto it = c->cgraphs.find(key);
        if (it == c->cgraphs.end()) {
            const int pp0 = c->pred_par, lp0 = c->lm.par;
            hipGraph_t gr = nullptr;
            hipGraphExec_t ex = nullptr;
            HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            enqueue();
            hipError_t e = hipStreamEndCapture(c->stream, &gr);
            c->pred_par = pp0; c->lm.par = lp0;            // the capture only recorded; parities advance at launch
            if (e != hipSuccess || !gr) return fail(c, LASR_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
            e = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
            (void)hipGraphDestroy(gr);
            if (e != hipSuccess) return fail(c, LASR_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
            it = c->cgraphs.emplace(key, ex).first;
        }
        HIPCHK(c, hipGraphLaunch(it->second, c->stream));
        if (G & 1) { c->pred_par ^= 1; if (c->lm.on) c->lm.par ^= 1; }
    } else {
        enqueue();
    }
    c->cont_iters += G;
    c->group_inflight = true;
    c->inflight_for = P.serial;
    HIPCHK(c, hipGetLastError());
    return LASR_OK;
}

// non-blocking: if the in-flight group has finished, consume its flag
static void cont_poll(lasr_ctx* c) {
    if (!c->group_inflight) return;
    const int v = __atomic_load_n((volatile int*)c->cont_host, __ATOMIC_ACQUIRE);
    if (v == -1) return;
    c->group_inflight = false;
    if (v == 0) c->done_serial = c->inflight_for;
}

int lasr_step_wait(lasr_ctx* c, int* n_ran) {
    if (!c) return LASR_EINVAL;
    if (n_ran) *n_ran = 0;
    if (c->pending.empty()) return LASR_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const int M = c->M;
    int* flag = c->cont_host;
    int* h_end = c->cont_host + 16 + (size_t)lasr_ctx::NFLY * M;
    int* h_ring = h_end + (size_t)M * lasr_ctx::ENDSLOTS;
    const long long it0 = c->cont_iters - (c->group_inflight ? c->kick_iters : 0);
    const long long serial = c->pending.front().serial;
    for (int guard = 0; c->done_serial != serial; ++guard) {
        if (!c->group_inflight) RC(cont_launch_group(c, c->wait_n));
        RC(spin_flag(c, flag, c->stream_dec));
        c->group_inflight = false;
        // a group launched while an older step was the target says nothing about this one
        if (c->inflight_for == serial && *flag == 0) c->done_serial = serial;
        if (guard > 4096) return fail(c, LASR_EHIP, "decode loop did not converge");
    }
    // results of the oldest step: tokens between the previous and this step boundary of every row, already
    // in pinned memory (written by the kernels of the groups that completed before the flag said 0)
    lasr_ctx::PendingStep& P = c->pending.front();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int r : P.rows) {
        const int j = P.target[r] / P.Tm - 1;
        const long long end = h_end[(size_t)r * lasr_ctx::ENDSLOTS + (j % lasr_ctx::ENDSLOTS)];
        for (long long q = c->h_fetched[r]; q < end; ++q)
            c->queue[r].push_back(h_ring[(size_t)r * lasr_ctx::TOKRING + (q % lasr_ctx::TOKRING)]);
        c->h_fetched[r] = end;
    }
    c->stats.frames = P.Tm;
    c->stats.decode_iters = (int)(c->cont_iters - it0);
    if (n_ran) *n_ran = (int)P.rows.size();
    c->pending.erase(c->pending.begin());
    c->cmd_inflight = 0;
    return LASR_OK;
}

// ---------------------------------------------------------------------------- offline
static int transcribe_common(lasr_ctx* c, const int* slots, int n, int T_max) {
    // cmd block (T_row, what) already filled + committed by the caller; x0 holds the features
    const lasr_model_desc& d = c->d;
    std::vector<int> rows(slots, slots + n);
    RC(commit_T_rows(c, T_max));
    rec(c, 1);
    run_encoder(c, T_max);
    rec(c, 2);
    RC(run_decode(c, T_max, d.max_iters_offline, true, rows));
    rec(c, 3);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    collect_stats(c, T_max);
    return LASR_OK;
}

int lasr_transcribe_pcm(lasr_ctx* c, const int* slots, int n, const float* pcm, const int64_t* n_samples) {
    if (!c) return LASR_EINVAL;
    RC(require_idle(c));
    RC(check_slots(c, slots, n, true));
    if (n == 0) return LASR_OK;
    if (!pcm || !n_samples) return fail(c, LASR_EINVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    const lasr_model_desc& d = c->d;
    long long total = 0; int T_max = 0; int Tmel_max = 0;
    std::vector<int> Tp(n), Tm(n);
    for (int i = 0; i < n; ++i) {
        if (n_samples[i] <= d.n_fft / 2) return fail(c, LASR_EINVAL, "utterance %d too short (%lld samples)", i, (long long)n_samples[i]);
        Tm[i] = 1 + (int)(n_samples[i] / d.hop);
        if (Tm[i] < d.n_stack) return fail(c, LASR_EINVAL, "utterance %d yields no stacked frame", i);
        Tp[i] = (Tm[i] - d.n_stack) / d.stride + 1;
        T_max = std::max(T_max, Tp[i]); Tmel_max = std::max(Tmel_max, Tm[i]);
        total += n_samples[i];
    }
    RC(ensure_T(c, T_max));
    const float* src = pcm;
    if (!is_device_ptr(pcm)) {
        RC(ensure_buf(c, &c->stage_pcm, &c->stage_pcm_floats, (size_t)total));
        HIPCHK(c, hipMemcpyAsync(c->stage_pcm, pcm, sizeof(float) * (size_t)total, hipMemcpyHostToDevice, c->stream));
        src = c->stage_pcm;
    }
    RC(ensure_buf(c, &c->lm_buf, &c->lm_floats, (size_t)c->M * Tmel_max * d.n_mels));
    RC(cmd_begin(c));
    long long off = 0;
    for (int i = 0; i < n; ++i) {
        const int s = slots[i];
        c->hc.T_row[s] = Tp[i]; c->hc.what[s] = 7; c->hc.row_frames[s] = Tm[i];
        c->hc.row_N[s] = n_samples[i]; c->hc.row_src_off[s] = off;
        off += n_samples[i];
        c->queue[s].clear();
        c->neg_logp[s] = 0.0;
        beam_host_reset(c, s, true);
    }
    RC(cmd_commit(c));
    RC(apply_reset(c, true));
    rec(c, 0);
    MelArgs m{};
    m.window = c->window; m.tw512 = c->tw512; m.tw1024 = c->tw1024; m.fb_start = c->fb_start; m.fb_off = c->fb_off;
    m.fb_w = c->fb_w; m.n_mels = d.n_mels; m.hop = d.hop; m.pcm = src; m.N = 0; m.stream = 0;
    m.ring_head = nullptr; m.chunk = d.chunk; m.n_window = d.n_window; m.row_sel = nullptr; m.frame0 = 0;
    m.frames_per_row = Tmel_max; m.out = c->lm_buf; m.out_frames = Tmel_max;
    m.row_N = c->dc.row_N; m.row_src_off = c->dc.row_src_off; m.row_frames = c->dc.row_frames;
    m.win_off = (d.n_fft - d.win) / 2; m.win_len = d.win; m.fb_nnz = c->fb_nnz;
    hipLaunchKernelGGL(k_logmel, dim3((Tmel_max + 3) / 4, c->M), dim3(256), 0, c->stream, m);
    StackLnArgs a{};
    a.src = c->lm_buf; a.mode = 0; a.src_frames = Tmel_max; a.frame_step = d.stride; a.row_off = nullptr;
    a.T_row = c->dc.T_row; a.ln_w = c->ln_w; a.ln_b = c->ln_b; a.x0 = c->x0; a.F = d.feat; a.n_mels = d.n_mels;
    a.n_stack = d.n_stack; a.M = c->M; a.MT = c->MT; a.mt_total = c->Tcap * c->MT; a.feats_out = nullptr; a.bf = c->bf; a.Tmax = T_max;
    LAUNCH_STACK_LN( dim3((T_max + 3) / 4, c->M), dim3(256), 0, c->stream, a);
    return transcribe_common(c, slots, n, T_max);
}

int lasr_transcribe_feats(lasr_ctx* c, const int* slots, int n, const float* feats, const int32_t* n_frames) {
    if (!c) return LASR_EINVAL;
    RC(require_idle(c));
    RC(check_slots(c, slots, n, true));
    if (n == 0) return LASR_OK;
    if (!feats || !n_frames) return fail(c, LASR_EINVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    const lasr_model_desc& d = c->d;
    long long total = 0; int T_max = 0;
    for (int i = 0; i < n; ++i) {
        if (n_frames[i] < 1) return fail(c, LASR_EINVAL, "utterance %d has no frames", i);
        T_max = std::max(T_max, (int)n_frames[i]); total += n_frames[i];
    }
    RC(ensure_T(c, T_max));
    const float* src = feats;
    if (!is_device_ptr(feats)) {
        RC(ensure_buf(c, &c->feat_stage, &c->feat_stage_floats, (size_t)total * d.feat));
        HIPCHK(c, hipMemcpyAsync(c->feat_stage, feats, sizeof(float) * (size_t)total * d.feat, hipMemcpyHostToDevice, c->stream));
        src = c->feat_stage;
    }
    RC(cmd_begin(c));
    long long off = 0;
    for (int i = 0; i < n; ++i) {
        const int s = slots[i];
        c->hc.T_row[s] = n_frames[i]; c->hc.what[s] = 7; c->hc.row_feat_off[s] = off;
        off += n_frames[i];
        c->queue[s].clear();
        c->neg_logp[s] = 0.0;
        beam_host_reset(c, s, true);
    }
    RC(cmd_commit(c));
    RC(apply_reset(c, true));
    rec(c, 0);
    StackLnArgs a{};
    a.src = src; a.mode = 1; a.src_frames = 0; a.frame_step = 0; a.row_off = c->dc.row_feat_off;
    a.T_row = c->dc.T_row; a.ln_w = c->ln_w; a.ln_b = c->ln_b; a.x0 = c->x0; a.F = d.feat; a.n_mels = d.n_mels;
    a.n_stack = d.n_stack; a.M = c->M; a.MT = c->MT; a.mt_total = c->Tcap * c->MT; a.feats_out = nullptr; a.bf = c->bf; a.Tmax = T_max;
    LAUNCH_STACK_LN( dim3((T_max + 3) / 4, c->M), dim3(256), 0, c->stream, a);
    return transcribe_common(c, slots, n, T_max);
}

// Transducer.transcribe_stream on feature chunks (models.py:506-575): carried encoder / predictor
// state, max_iters_stream.  feats [n, T, feat] (host or device), the same T for every listed slot.
int lasr_step_feats(lasr_ctx* c, const int* slots, int n, const float* feats, int T) {
    if (!c) return LASR_EINVAL;
    RC(require_idle(c));
    RC(check_slots(c, slots, n, true));
    if (n == 0) return LASR_OK;
    if (!feats || T < 1) return fail(c, LASR_EINVAL, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const lasr_model_desc& d = c->d;
    RC(ensure_T(c, T));
    const float* src = feats;
    if (!is_device_ptr(feats)) {
        RC(ensure_buf(c, &c->feat_stage, &c->feat_stage_floats, (size_t)n * T * d.feat));
        HIPCHK(c, hipMemcpyAsync(c->feat_stage, feats, sizeof(float) * (size_t)n * T * d.feat, hipMemcpyHostToDevice, c->stream));
        src = c->feat_stage;
    }
    RC(cmd_begin(c));
    std::vector<int> rows(slots, slots + n);
    for (int i = 0; i < n; ++i) { c->hc.T_row[slots[i]] = T; c->hc.row_feat_off[slots[i]] = (long long)i * T; }
    RC(cmd_commit(c));
    RC(commit_T_rows(c, T));
    rec(c, 0);
    StackLnArgs a{};
    a.src = src; a.mode = 1; a.row_off = c->dc.row_feat_off; a.T_row = c->T_row_dev; a.ln_w = c->ln_w; a.ln_b = c->ln_b;
    a.x0 = c->x0; a.F = d.feat; a.n_mels = d.n_mels; a.n_stack = d.n_stack; a.M = c->M; a.MT = c->MT;
    a.mt_total = c->Tcap * c->MT; a.feats_out = nullptr; a.bf = c->bf; a.Tmax = T;
    LAUNCH_STACK_LN( dim3((T + 3) / 4, c->M), dim3(256), 0, c->stream, a);
    rec(c, 1);
    run_encoder(c, T);
    rec(c, 2);
    RC(run_decode(c, T, d.max_iters_stream, false, rows));
    rec(c, 3);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    collect_stats(c, T);
    return LASR_OK;
}

int lasr_fetch(lasr_ctx* c, int slot, int32_t* tokens, int cap, int* n_new, double* neg_logp, double* align) {
    if (!c || !n_new) return LASR_EINVAL;
    if (slot < 0 || slot >= c->d.max_streams || !c->open_[slot]) return fail(c, LASR_ESTATE, "slot %d is not open", slot);
    auto& q = c->queue[slot];
    if ((int)q.size() > cap || (!tokens && !q.empty())) {
        *n_new = (int)q.size();
        return fail(c, LASR_EFULL, "token buffer too small: need %d", (int)q.size());
    }
    if (!q.empty()) memcpy(tokens, q.data(), sizeof(int32_t) * q.size());
    *n_new = (int)q.size();
    q.clear();
    if (neg_logp) *neg_logp = c->neg_logp[slot];
    if (align) *align = c->align[slot];
    return LASR_OK;
}

int lasr_fetch_many(lasr_ctx* c, const int* slots, int n, int32_t* tokens, int cap, int* n_new) {
    if (!c || !n_new || (n > 0 && !slots)) return LASR_EINVAL;
    for (int i = 0; i < n; ++i) {
        const int slot = slots[i];
        if (slot < 0 || slot >= c->d.max_streams || !c->open_[slot]) return fail(c, LASR_ESTATE, "slot %d is not open", slot);
        if ((int)c->queue[slot].size() > cap) return fail(c, LASR_EFULL, "token buffer too small: slot %d needs %d", slot, (int)c->queue[slot].size());
    }
    for (int i = 0; i < n; ++i) {
        auto& q = c->queue[slots[i]];
        if (!q.empty()) memcpy(tokens + (size_t)i * cap, q.data(), sizeof(int32_t) * q.size());
        n_new[i] = (int)q.size();
        q.clear();
    }
    return LASR_OK;
}

// ---------------------------------------------------------------------------- op-level entry points
int lasr_logmel(lasr_ctx* c, const float* pcm, int B, int64_t N, float* logmel) {
    if (!c || !pcm || !logmel || B < 1 || B > c->M) return c ? fail(c, LASR_EINVAL, "bad argument") : LASR_EINVAL;
    if (N <= c->d.n_fft / 2) return fail(c, LASR_EINVAL, "signal shorter than the reflect padding");
    HIPCHK(c, hipSetDevice(c->device));
    const lasr_model_desc& d = c->d;
    const int T = 1 + (int)(N / d.hop);
    MelArgs m{};
    m.window = c->window; m.tw512 = c->tw512; m.tw1024 = c->tw1024; m.fb_start = c->fb_start; m.fb_off = c->fb_off;
    m.fb_w = c->fb_w; m.n_mels = d.n_mels; m.hop = d.hop; m.pcm = pcm; m.N = N; m.stream = 0;
    m.ring_head = nullptr; m.chunk = d.chunk; m.n_window = d.n_window; m.row_sel = nullptr; m.frame0 = 0;
    m.frames_per_row = T; m.out = logmel; m.out_frames = T;
    m.row_N = nullptr; m.row_src_off = nullptr; m.row_frames = nullptr;
    m.win_off = (d.n_fft - d.win) / 2; m.win_len = d.win; m.fb_nnz = c->fb_nnz;
    hipLaunchKernelGGL(k_logmel, dim3((T + 3) / 4, B), dim3(256), 0, c->stream, m);
    HIPCHK(c, hipGetLastError());
    return LASR_OK;
}

int lasr_stack(lasr_ctx* c, const float* logmel, int B, int T, float* feats, int* Tp) {
    if (!c) return LASR_EINVAL;
    const lasr_model_desc& d = c->d;
    const int tp = T < d.n_stack ? 0 : (T - d.n_stack) / d.stride + 1;
    if (Tp) *Tp = tp;
    if (tp == 0) return LASR_OK;                       // fewer than n_stack frames: no stacked frame
    if (!logmel || !feats || B < 1) return fail(c, LASR_EINVAL, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_stack, dim3(tp, B), dim3(256), 0, c->stream, logmel, T, d.n_mels, d.n_stack, d.stride, feats, tp, d.feat);
    HIPCHK(c, hipGetLastError());
    return LASR_OK;
}

int lasr_encoder(lasr_ctx* c, const float* feats, int B, int Tp, float* out, float* h_out, float* c_out) {
    if (!c || !feats || !out || B < 1 || B > c->d.max_streams || Tp < 1) return c ? fail(c, LASR_EINVAL, "bad argument") : LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    const lasr_model_desc& d = c->d;
    const int H = d.hidden;
    RC(ensure_T(c, Tp));
    RC(cmd_begin(c));
    for (int r = 0; r < B; ++r) { c->hc.T_row[r] = Tp; c->hc.what[r] = 1; c->hc.row_feat_off[r] = (long long)r * Tp; }
    RC(cmd_commit(c));
    RC(apply_reset(c, false));
    RC(commit_T_rows(c, Tp));
    StackLnArgs a{};
    a.src = feats; a.mode = 1; a.row_off = c->dc.row_feat_off; a.T_row = c->T_row_dev; a.ln_w = c->ln_w; a.ln_b = c->ln_b;
    a.x0 = c->x0; a.F = d.feat; a.n_mels = d.n_mels; a.n_stack = d.n_stack; a.M = c->M; a.MT = c->MT;
    a.mt_total = c->Tcap * c->MT; a.feats_out = nullptr; a.bf = c->bf; a.Tmax = Tp;
    LAUNCH_STACK_LN( dim3((Tp + 3) / 4, c->M), dim3(256), 0, c->stream, a);
    run_encoder(c, Tp);
    hipLaunchKernelGGL(k_enc_out, dim3(grid1((size_t)B * Tp * H)), dim3(256), 0, c->stream,
                       (const void*)c->ybuf[(d.enc_layers - 1) & 1], c->Tcap * c->MT, c->M, out, B, Tp, H, c->bf);
    for (int l = 0; l < d.enc_layers; ++l) {
        if (h_out)
            hipLaunchKernelGGL(k_from_frag, dim3(grid1((size_t)B * H)), dim3(256), 0, c->stream,
                               (const void*)c->enc_h[c->enc_par][l], c->MT, 0, h_out + (size_t)l * B * H, H, B, H, c->bf);
        if (c_out)
            hipLaunchKernelGGL(k_c_to_rows, dim3(grid1((size_t)B * H)), dim3(256), 0, c->stream,
                               (const float*)c->enc_c[l], c->M, c_out + (size_t)l * B * H, B, H);
    }
    HIPCHK(c, hipGetLastError());
    return LASR_OK;
}

int lasr_predictor(lasr_ctx* c, const int32_t* tok, int B, int U, float* out) {
    if (!c || !tok || !out || B < 1 || B > c->d.max_streams || U < 1) return c ? fail(c, LASR_EINVAL, "bad argument") : LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    const int H = c->d.hidden;
    for (int i = 0; i < B * U; ++i)
        if (tok[i] < 0 || tok[i] >= c->d.vocab) return fail(c, LASR_EINVAL, "token %d out of range", tok[i]);
    // learned initial state, no implicit BOS
    RC(cmd_begin(c));
    for (int r = 0; r < B; ++r) c->hc.what[r] = 2;
    RC(cmd_commit(c));
    RC(apply_reset(c, false, 3, true));
    for (int u = 0; u < U; ++u) {
        RC(cmd_begin(c));
        for (int r = 0; r < B; ++r) { c->hc.token[r] = tok[(size_t)r * U + u]; c->hc.emit[r] = 1; }
        RC(cmd_commit(c));
        if (c->Md > c->M) HIPCHK(c, hipMemsetAsync(c->ds.emit, 0, sizeof(int) * c->Md, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->ds.token, c->dc.token, sizeof(int) * c->M, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->ds.emit, c->dc.emit, sizeof(int) * c->M, hipMemcpyDeviceToDevice, c->stream));
        launch_predictor(c);
    }
    hipLaunchKernelGGL(k_from_elem, dim3(grid1((size_t)B * H)), dim3(256), 0, c->stream, (const void*)c->pred_y[c->d.pred_layers - 1],
                       out, (size_t)B * H, c->bf);
    HIPCHK(c, hipGetLastError());
    return LASR_OK;
}

int lasr_joint(lasr_ctx* c, const float* h_pred, const float* h_enc, int B, float* logits, float* logp_max, int32_t* argmax) {
    if (!c || !h_pred || !h_enc || !logits || B < 1 || B > c->d.max_streams) return c ? fail(c, LASR_EINVAL, "bad argument") : LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    const int H = c->d.hidden, J = c->d.joint, V = c->d.vocab;
    RC(ensure_T(c, 1));
    {   // pp = h_pred W1p^T + b1 ; pe[0] = h_enc W1e^T   (row-major A)
        const void* ap = h_pred; const void* ae = h_enc;
        if (c->bf) {   // bf16 operands: round the f32 inputs once
            hipLaunchKernelGGL(k_to_elem, dim3(grid1((size_t)B * H)), dim3(256), 0, c->stream, h_pred, c->cvt_a, (size_t)B * H, 1);
            hipLaunchKernelGGL(k_to_elem, dim3(grid1((size_t)B * H)), dim3(256), 0, c->stream, h_enc, c->cvt_b, (size_t)B * H, 1);
            ap = c->cvt_a; ae = c->cvt_b;
        }
        GemmArgs g{}; g.A[0] = ap; g.a_mt_total[0] = H; g.a_mt_off[0] = 0; g.W[0] = c->W1p; g.a_rows = B;
        EpiLinear::Args ea{}; ea.bias = c->b1; ea.out = c->pp; ea.ldo = J; ea.n_rows = B; ea.t_idx = nullptr; ea.T_row = nullptr; ea.M = c->M;
        launch_linear<true, 3>(c, J / 16, (B + 15) / 16, g, H, ea);
        g.A[0] = ae; g.W[0] = c->W1e; ea.bias = nullptr; ea.out = c->pe;
        launch_linear<true, 3>(c, J / 16, (B + 15) / 16, g, H, ea);
    }
    hipLaunchKernelGGL(k_ja, dim3(grid1((size_t)c->M * J)), dim3(256), 0, c->stream, (const float*)c->pe, (const float*)c->pp,
                       (const int*)nullptr, (const int*)nullptr, c->ja, J, c->M, c->MTj, 1 << 30, c->bf, 1, c->M, 1);
    launch_logits(c, logits, B, false);
    if (logp_max && argmax) {
        DecState s = c->ds;
        hipLaunchKernelGGL((k_select<true>), dim3(B), dim3(256), 0, c->stream, (const float*)logits, V, c->d.blank, 1,
                           (const int*)nullptr, s, 0, logp_max, argmax, 1, c->M);
    }
    HIPCHK(c, hipGetLastError());
    return LASR_OK;
}

// ---------------------------------------------------------------------------- resampling
static long long resample_num_out(long long n_in, int sr_in, int sr_out) {     // kaldi GetNumOutputSamples(flush = true)
    long long a = sr_in, b = sr_out;
    while (b) { const long long t = a % b; a = b; b = t; }
    const long long tick = (long long)sr_in / a * sr_out;
    const long long ticks_in = tick / sr_in, ticks_out = tick / sr_out;
    const long long length = n_in * ticks_in;
    if (length <= 0) return 0;
    long long last = length / ticks_out;
    if (last * ticks_out == length) last -= 1;
    return last + 1;
}

int lasr_resample(lasr_ctx* c, const float* pcm, int B, int64_t N_in, int sr_in, float* out, int64_t* N_out) {
    if (!c || !N_out) return LASR_EINVAL;
    const int sr_out = c->d.sample_rate;
    if (B < 1 || N_in < 1 || sr_in < 1000 || sr_in > 384000) return fail(c, LASR_EINVAL, "bad argument");
    *N_out = resample_num_out(N_in, sr_in, sr_out);
    if (!out) return LASR_OK;                          // size query
    if (!pcm) return fail(c, LASR_EINVAL, "pcm is null");
    HIPCHK(c, hipSetDevice(c->device));
    auto it = c->resamplers.find(sr_in);
    if (it == c->resamplers.end()) {
        // one windowed-sinc filter per output phase, float32 arithmetic like torchaudio 0.6.0's
        // compliance/kaldi.py::_get_LR_indices_and_weights (lowpass_filter_width = 6, cutoff = 0.99 * min(sr) / 2)
        long long a = sr_in, b = sr_out;
        while (b) { const long long t = a % b; a = b; b = t; }
        lasr_ctx::Resampler r;
        r.in_unit = (int)(sr_in / a); r.U = (int)(sr_out / a);
        const double cutoff = 0.99 * 0.5 * std::min(sr_in, sr_out);
        const float width = (float)(6.0 / (2.0 * cutoff));
        std::vector<float> lo(r.U), hi(r.U), ot(r.U);
        int taps = 0;
        for (int p = 0; p < r.U; ++p) {
            ot[p] = (float)p / (float)sr_out;
            lo[p] = std::ceil((ot[p] - width) * (float)sr_in);
            hi[p] = std::floor((ot[p] + width) * (float)sr_in);
            taps = std::max(taps, (int)(hi[p] - lo[p] + 1.f));
        }
        r.taps = taps;
        std::vector<int> first(r.U);
        std::vector<float> w((size_t)r.U * taps, 0.f);
        const float cw = (float)(2.0 * M_PI * cutoff / 6.0), cs = (float)(2.0 * M_PI * cutoff), pi = (float)M_PI;
        for (int p = 0; p < r.U; ++p) {
            first[p] = (int)lo[p];
            for (int j = 0; j < taps; ++j) {
                const float dt = (lo[p] + (float)j) / (float)sr_in - ot[p];
                float v = 0.f;
                if (std::fabs(dt) < width) v = 0.5f * (1.f + std::cos(cw * dt));
                if (dt != 0.f) v *= std::sin(cs * dt) / (pi * dt);
                else v *= (float)(2.0 * cutoff);
                w[(size_t)p * taps + j] = v / (float)sr_in;
            }
        }
        RC(upload(c, &r.first, first.data(), first.size()));
        RC(upload(c, &r.w, w.data(), w.size()));
        it = c->resamplers.emplace(sr_in, r).first;
    }
    const lasr_ctx::Resampler& r = it->second;
    hipLaunchKernelGGL(k_resample, dim3((unsigned)((*N_out + 255) / 256), B), dim3(256), 0, c->stream, pcm, (long long)N_in,
                       (const int*)r.first, (const float*)r.w, r.U, r.taps, r.in_unit, out, (long long)*N_out);
    HIPCHK(c, hipGetLastError());
    return LASR_OK;
}

// ---------------------------------------------------------------------------- LM shallow fusion
size_t lasr_lm_weight_count(const lasr_lm_desc* d) {
    if (!d || d->vocab <= 0 || d->embed <= 0 || d->hidden <= 0 || d->layers < 1 || d->layers > 8) return 0;
    const size_t V = d->vocab, E = d->embed, H = d->hidden;
    size_t n = V * E;
    for (int l = 0; l < d->layers; ++l) n += 4 * H * (l == 0 ? E : H) + 4 * H * H + 8 * H;
    return n + V * H + V;
}

int lasr_attach_lm(lasr_ctx* c, const lasr_lm_desc* d, const float* weights, size_t n_weights) {
    if (!c) return LASR_EINVAL;
    if (c->lm.on) return fail(c, LASR_ESTATE, "an LM is already attached");
    if (c->W > 1) return fail(c, LASR_ESTATE, "LM shallow fusion is implemented for greedy decoding (beam = 1)");
    if (!d || !weights || n_weights != lasr_lm_weight_count(d) || n_weights == 0)
        return fail(c, LASR_EINVAL, "LM weight blob has %zu floats, expected %zu", n_weights, d ? lasr_lm_weight_count(d) : (size_t)0);
    if (d->vocab != c->d.vocab) return fail(c, LASR_EINVAL, "LM vocabulary %d != model vocabulary %d", d->vocab, c->d.vocab);
    if (d->vocab > 4096) return fail(c, LASR_EINVAL, "LM fusion keeps a row's log-probs in registers: vocab <= 4096");
    if (d->embed % 16 || d->hidden % (c->bf ? 32 : 16)) return fail(c, LASR_EINVAL, "LM dims must be multiples of 16 (hidden: 32 for bf16)");
    RC(require_idle(c));
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    lasr_ctx::LM& m = c->lm;
    const int V = d->vocab, E = d->embed, H = d->hidden, L = d->layers, M = c->M;
    m.E = E; m.H = H; m.L = L; m.alpha = d->alpha; m.theta = d->theta; m.min_val = d->min_val;
    Reader rd{weights, n_weights};
    const float* embed = rd.take((size_t)V * E);
    m.cells.resize(L);
    std::vector<float> in0_w, in0_b;
    for (int l = 0; l < L; ++l) RC(load_lstm(c, rd, m.cells[l], l == 0 ? E : H, H, true, l == 0 ? &in0_w : nullptr, l == 0 ? &in0_b : nullptr));
    const float* wout = rd.take((size_t)V * H); const float* bout = rd.take(V);
    if (!bout || rd.left != 0) return fail(c, LASR_EINVAL, "LM weight blob layout mismatch");
    {
        Packed pk;
        pack_tiles(pk, c->bf, V / 16, H, [&](int t, int ui, int k) { return wout[(size_t)(16 * t + ui) * H + k]; });
        RC(upload_packed(c, &m.Wout, pk));
        RC(upload(c, &m.bout, bout, V));
        std::vector<float> one(H, 1.f), zero(H, 0.f);
        RC(upload(c, &m.ones, one.data(), H)); RC(upload(c, &m.zeros, zero.data(), H));
    }
    {   // layer-0 input table (exact f32, as for the predictor): tab[v] = embed[v] * W_ih0^T + (b_ih + b_hh)
        float* emb_dev = nullptr; void* wt = nullptr; float* bt = nullptr;
        RC(upload(c, &emb_dev, embed, (size_t)V * E));
        Packed pk;
        pack_tiles(pk, 0, 4 * H / 16, E, [&](int t, int ui, int k) { return in0_w[(size_t)(16 * t + ui) * E + k]; });
        RC(upload_packed(c, &wt, pk)); RC(upload(c, &bt, in0_b.data(), in0_b.size()));
        RC(dalloc(c, &m.cells[0].tab, (size_t)V * 4 * H));
        GemmArgs g{}; g.A[0] = emb_dev; g.a_mt_total[0] = E; g.a_mt_off[0] = 0; g.KC[0] = E / 16; g.W[0] = wt; g.a_rows = V;
        EpiLinear::Args ea{}; ea.bias = bt; ea.out = m.cells[0].tab; ea.ldo = 4 * H; ea.n_rows = V; ea.t_idx = nullptr; ea.T_row = nullptr; ea.M = M;
        launch_gemm<OpsF32, EpiLinear, 1, true>(c, 4 * H / 16, V / 16, g, ea);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        dfree(c, emb_dev); dfree(c, wt); dfree(c, bt);
        // the x-side weights of layer 0 are folded into the table; its packed copy is not needed
        dfree(c, m.cells[0].WxA); m.cells[0].WxA = nullptr;
    }
    for (int p = 0; p < 2; ++p) m.h[p].resize(L);
    m.y.resize(L); m.cst.resize(L);
    for (int l = 0; l < L; ++l) {
        for (int p = 0; p < 2; ++p) { RC(dalloc(c, (char**)&m.h[p][l], (size_t)M * H * c->esz)); HIPCHK(c, hipMemset(m.h[p][l], 0, (size_t)M * H * c->esz)); }
        RC(dalloc(c, (char**)&m.y[l], (size_t)M * H * c->esz)); HIPCHK(c, hipMemset(m.y[l], 0, (size_t)M * H * c->esz));
        RC(dalloc(c, &m.cst[l], (size_t)M * H)); HIPCHK(c, hipMemset(m.cst[l], 0, sizeof(float) * (size_t)M * H));
    }
    RC(dalloc(c, &m.raw, (size_t)M * V)); RC(dalloc(c, &m.lmz, (size_t)M * V)); RC(dalloc(c, &m.valid, M));
    HIPCHK(c, hipMemset(m.lmz, 0, sizeof(float) * (size_t)M * V)); HIPCHK(c, hipMemset(m.valid, 0, sizeof(int) * M));
    for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second);   // decode groups change shape
    c->graphs.clear();
    for (auto& kv : c->cgraphs) (void)hipGraphExecDestroy(kv.second);
    c->cgraphs.clear();
    c->la = c->la_stream = c->la_offline = 1;   // the fused re-pick needs the LM state of exactly this decision
    c->ds.lmz = m.lmz; c->ds.lm_valid = m.valid; c->ds.lm_alpha = m.alpha; c->ds.lm_theta = m.theta; c->ds.lm_min = m.min_val;
    m.on = true;
    return LASR_OK;
}

// ---------------------------------------------------------------------------- stats / bench
int lasr_get_stats(lasr_ctx* c, lasr_step_stats* s) {
    if (!c || !s) return LASR_EINVAL;
    *s = c->stats;
    return LASR_OK;
}
int lasr_set_profiling(lasr_ctx* c, int on) {
    if (!c) return LASR_EINVAL;
    c->profiling = on != 0;
    return LASR_OK;
}
// debug: copy the phase timestamps of the last launch of each GEMM kind (cell, pred0, pred1, ppj, logits)
int lasr_debug_timing(lasr_ctx* c, unsigned long long* out /*[5*4096*8]*/) {
    if (!c || !out) return LASR_EINVAL;
    if (!c->dbg) return fail(c, LASR_ESTATE, "set LASR_DBG_TIMING=1 before lasr_create");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->dbg, sizeof(unsigned long long) * 5 * 4096 * 16, hipMemcpyDeviceToHost));
    return LASR_OK;
}

int lasr_sync(lasr_ctx* c) {
    if (!c) return LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->cmd_inflight = 0;
    return LASR_OK;
}

int lasr_bench_cell(lasr_ctx* c, int layer, int iters, double* us) {
    if (!c || !us || layer < 0 || layer >= c->d.enc_layers || iters < 1) return c ? fail(c, LASR_EINVAL, "bad argument") : LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    RC(ensure_T(c, 1));
    const int H = c->d.hidden, I = c->enc[layer].I, M = c->M;
    // random (not zero) operands: zero-filled data inflates the clock (DVFS)
    hipLaunchKernelGGL(k_fill_rand, dim3(grid1((size_t)M * I)), dim3(256), 0, c->stream, layer == 0 ? c->x0 : c->ybuf[(layer - 1) & 1], (size_t)M * I, 17u, c->bf);
    for (int p = 0; p < 2; ++p)
        hipLaunchKernelGGL(k_fill_rand, dim3(grid1((size_t)M * H)), dim3(256), 0, c->stream, c->enc_h[p][layer], (size_t)M * H, 23u + p, c->bf);
    RC(cmd_begin(c));
    for (int r = 0; r < c->d.max_streams; ++r) c->hc.T_row[r] = 1;
    RC(cmd_commit(c));
    RC(commit_T_rows(c, 1));
    const void* xsrc = layer == 0 ? c->x0 : c->ybuf[(layer - 1) & 1];
    const int mt_total = c->Tcap * c->MT;
    for (int i = 0; i < 3; ++i) { launch_enc_cell(c, layer, 0, xsrc, mt_total, c->ybuf[layer & 1], mt_total); c->enc_par ^= 1; }
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    HIPCHK(c, hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; ++i) { launch_enc_cell(c, layer, 0, xsrc, mt_total, c->ybuf[layer & 1], mt_total); c->enc_par ^= 1; }
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *us = (double)ms * 1000.0 / iters;
    c->cmd_inflight = 0;
    HIPCHK(c, hipGetLastError());
    return LASR_OK;
}

}  // extern "C"
