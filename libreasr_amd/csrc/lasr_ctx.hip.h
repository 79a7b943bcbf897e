// lasr_ctx.hip.h -- the engine context (lasr_ctx), error / allocation helpers, operand packing, description checks
// Included through lasr_host.hip.h by every translation unit of the library.
#pragma once
#include <dlfcn.h>

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
    int la_sync = 1;           // synchronous per-chunk protocol: 1 (groups are sized for one frame per iteration; 2 measured -4 %)
    int la_stream = 2, la_offline = 3;   // measured on configs[1] (12 steps in flight, 32-row logits tiling): streaming 2 frames per
                               // iteration f32 +2 %, bf16 +9 % (2.5 instead of 3.1 iterations per model step; 3 frames: -8 %);
                               // offline (258 frames) 3 frames +10 %
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
    int cell_nw = 0;                // waves per encoder-cell workgroup (0: 4 for f32, 8 for bf16); LASR_CELL_NW
    bool enc_u12 = false;           // encoder cell tiling D (12 units x 64 rows per workgroup, EpiLSTMe): bf16, H % 12 == 0, M % 64 == 0,
                                    // H / 12 * M / 64 >= 256 workgroups; LASR_ENC_U12 overrides
    int dec_prio = 1, cell_prio = 0;   // s_setprio of the decode-stream GEMMs (+4 % at 6 steps in flight, round 2) / of everything else
    int logits_mt = 2;              // m-tiles per workgroup of the logits GEMM (la x 64 rows must not re-read W2 per 16-row tile)
    // beam search: c, BN(h) and pp ping-pong like h (every slot may be re-parented each round):
    // parity 0 = pred_c / pred_y / pp, parity 1 = the *1 buffers; all follow pred_par
    std::vector<float*> pred_c1;
    std::vector<void*> pred_y1;
    float* pp1 = nullptr;
    double* b_score = nullptr; int *b_alive = nullptr, *b_inB = nullptr, *b_parent = nullptr, *b_trellis = nullptr;
    // host: token history of every hypothesis slot as a shared-prefix tree per stream (a round re-parents W slots: copying W
    // token vectors per round grows with the length of the stream; a node per emitted token does not)
    struct BeamHost { std::vector<int> par, tok; std::vector<int> cur; };
    std::vector<BeamHost> bh;
    // continuous beam loop (lasr_step_submit / lasr_step_wait with beam > 1): pinned rings written by k_beam_select
    static constexpr int TRING = 128;
    int *b_tre_host = nullptr, *b_tre_dev = nullptr;         // [TRING][Md] records of a round
    int *b_fdone_host = nullptr, *b_fdone_dev = nullptr;     // [TRING][M]  frame finished in that round
    double *b_endsc_host = nullptr, *b_endsc_dev = nullptr;  // [M][ENDSLOTS][W] slot scores at the end of a model step
    int *b_endal_host = nullptr, *b_endal_dev = nullptr;     // [M][ENDSLOTS]    alive mask
    long long b_rounds_replayed = 0;
    std::vector<long long> b_frames_done;                    // per stream: frames the host has seen finished
    struct BeamResult { std::vector<int32_t> tokens; double score; };
    std::vector<std::deque<BeamResult>> b_results;           // per stream: finished model steps not yet collected
    std::vector<std::vector<int32_t>> committed;          // host: best hypothesis at the last predictor reset(s)
    std::vector<double> committed_score;
    std::vector<std::vector<int32_t>> best_full;          // host: committed + current best hypothesis
    int* trellis_host = nullptr; size_t trellis_host_ints = 0;
    int enc_par = 0;
    int pred_par = 0;               // predictor h ping-pong parity (row-major [M][H] buffers)
    void *cvt_a = nullptr, *cvt_b = nullptr;   // [M][H] element-typed staging of f32 op-level inputs
    bool dbg_gate = true;           // decode kernels record timestamps only in the first iteration of a step
    unsigned long long* dbg = nullptr;   // LASR_DBG_TIMING: [5 kinds][4096 blocks][16] phase timestamps
    // Unused dynamic LDS given to the streaming log-mel launch (k_fe_mel: 46 592 B of its own) so that its workgroups never share a
    // CU with a workgroup of the wide decode tilings (EpiLSTMw / EpiNBRCw / EpiLinearT<4>: 66-72 KB of LDS each, launched for
    // >= 256 hypothesis rows or >= 512 logits rows): 98 304 B in total, and 98 304 + 65 536 > 160 KB.  Next to those kernels (bf16
    // operands, beam search over >= 512 rows) a few waves of the log-mel kernel per thousand returned wrong spectra -- one 16-lane
    // LDS read pass each, no shared data, cause not understood -- and configs[4] was not reproducible run to run; with the CU to
    // themselves: never (profiles/r06/r06_experiments.txt R; detectors: lasr_debug_enclog, lasr_debug_fe_race, tests/test_gpu_race.py).
    // 0 where no such kernel can run (configs[1]); LASR_FE_LDS_PAD overrides (bytes).  Since the same round the wide tilings run on
    // the older matrix instruction (OpsBF16k16: the neighbour that disturbed was v_mfma_f32_16x16x32_bf16 at their density, and only
    // it), after which the probe and the race log are clean WITHOUT the pad as well: the pad is the second line of defence.
    int fe_lds_pad = 0, logmel_lds_pad = 0;       // (k_fe_mel / the per-chunk k_logmel: 98 304 B minus the kernel's own LDS)
    unsigned* enclog = nullptr;     // LASR_DBG_ENCLOG=N: per model step and row, exact checksums (sum of the element bit patterns) of the
    float* pendlog = nullptr;       // LASR_DBG_PENDLOG=1 (with LASR_DBG_ENCLOG): a copy of the pending log-mel frames per logged step
    int enclog_cap = 0, enclog_n = 0;   // encoder's inputs and state behind that step: [N][2 T + 2 L][M], see lasr_debug_enclog
    std::vector<unsigned long long> tile_masks;   // per step t: m-tiles with an active row (from the host's T_row)
    float *pp = nullptr, *logits = nullptr;
    void* ja = nullptr;             // joint activation, fragment-major, element-typed
    DecState ds{};
    int n_iter_slots = 0;
    int* T_row_dev = nullptr;       // [M] current step's frames per row: points INTO the step's device command block
    int* zero_rows = nullptr;       // [M] zeros (reset passes: "no row is decoding")
    // greedy decode: predictor state after the BOS step + its joint half, captured once (see ResetArgs); LASR_BOS_CACHE=0: BOS pass per reset
    std::vector<float*> bos_h, bos_c; float* bos_pp = nullptr; bool bos_ready = false;
    int* T_row_dec = nullptr;       // what the decode kernels read (T_row_fix; frames-available counters when continuous)
    int* T_row_fix = nullptr;       // [M] fixed-address copy of the current synchronous step's T_row: the decode kernels are
                                    // replayed from cached hipGraphs, which bake their pointer arguments in, while the
                                    // command block (T_row_dev) moves with every step
    int* dec_t_idx = nullptr;       // frame cursor array the decode kernels use (ds.t_idx, or c_cur when continuous)
    int pe_ring_R = 1 << 30;        // pe frame t lives at slot t % pe_ring_R
    // continuous decode (lasr_step_submit / lasr_step_wait): front-end + encoder of later chunks run on
    // the main stream while ONE greedy loop keeps running on stream_dec across chunk boundaries: a row
    // that finished chunk k moves on to chunk k+1's frames while a bursty row is still on chunk k.
    // a GEMM launch recorded instead of issued (launch_gemm with `cap` set): what pair launches are assembled from (k_gemm2)
    struct Captured {
        const void* fn = nullptr;             // the k_gemm instantiation that would have run
        unsigned gx = 0, gy = 0, threads = 0;
        alignas(16) unsigned char g[256];     // GemmArgs
        alignas(16) unsigned char ea[768];    // the epilogue's Args
        size_t g_size = 0, ea_size = 0;
    };
    Captured* cap = nullptr;
    // lasr_bench_neighbour (experiment): a third stream on a hardware queue of its own, the neighbour's buffers
    hipStream_t stream_nb = nullptr; unsigned long long* nb_done = nullptr; float* nb_buf = nullptr; size_t nb_floats = 0;
    int nb_kind = 0, nb_wgs = 0; double nb_ms = 0.0;
    hipStream_t stream_dec = nullptr;
    // LM branch of a decode iteration (pipelined protocol): the LM step and the predictor / joint chain both start from the
    // token a selection kernel has just written and both end at the next selection -- two branches of the group's hipGraph
    hipStream_t stream_lm = nullptr;
    hipEvent_t ev_lm_fork = nullptr, ev_lm_join = nullptr;
    double lm_stream_ratio[2] = {0.0, 0.0};   // overlap probe of stream_lm against the main / the decode stream
    int dec_stream_attempts = 0;    // streams tried at creation until one ran concurrently with the ctx stream (see create_impl)
    double dec_stream_ratio = 0.0;  // the chosen stream's probe (wall / delay: ~1 concurrent, ~2 one hardware queue)
    static constexpr int NFLY = 32; // ring of encoder-done events: up to NFLY - 1 steps in flight (round 5: 16 -> 32; the token ring binds first: 25)
    static constexpr int RING = 64; // pe ring, frames per row
    static constexpr int TOKRING = 512, ENDSLOTS = 32;
    hipEvent_t ev_enc[NFLY] = {};
    hipEvent_t ev_misc = nullptr;
    int* T_row_ring[NFLY] = {};
    float* pe_ring = nullptr;
    int *c_cur = nullptr, *c_avail = nullptr, *c_iters = nullptr, *c_target = nullptr, *c_ntotal = nullptr;
    int* c_hcur_dev = nullptr;      // device view of the pinned per-row frame cursors (cont_host + 16)
    int *c_ntok_end = nullptr, *c_tok_ring = nullptr, *c_behind = nullptr, *c_enc_frames = nullptr, *c_enc_base = nullptr;
    int *c_done = nullptr, *c_flag_dev = nullptr;   // workgroups finished per iteration; device view of cont_host[0]
    int* c_done2 = nullptr;                         // second arrival ring (k_beam_fuse publishes the round when an LM is attached)
    int* c_iter = nullptr;                          // device-side iteration counter of the continuous loop
    std::map<std::tuple<int, int, int>, hipGraphExec_t> cgraphs;   // (iterations, predictor parity, LM parity) -> group
    int* cont_host = nullptr;       // pinned: [0] flag, [16..16+M) per-row frame cursors, then (after NFLY*M ints) ntok_end + token ring
    struct PendingStep { std::vector<int> rows; int Tm; int idx; bool admitted; std::vector<int> target; const int* T_row_ptr; long long serial; };
    std::vector<PendingStep> pending;
    std::vector<long long> h_frames_sub, h_fetched;
    std::vector<int> h_avail;       // per-row frames admitted to the decode loop (device copy: c_avail)
    std::vector<int> h_cur_seen;    // per-row frame cursors as of the last consumed group (step j of row r is decoded iff >= its target)
    int work_left = 0;              // rows that still had encoded frames to decode when the last consumed group ended
    long long model_steps = 0, cont_iters = 0, iters_reported = 0;
    bool group_inflight = false;    // a decode group has been launched and its flag not yet consumed
    // The flag of a group is published by its LAST SELECTION kernel; the predictor cells and the joint half of that iteration run
    // behind it on the decode stream and write the rows' predictor state (emitting rows: the cell; the others: the carry into the
    // other parity).  Work that touches the decode state from the ctx stream once the pipeline has drained (a reset of an idle
    // context, the synchronous protocol) first makes the ctx stream wait for the decode stream (order_after_decode_tail).  Round 5:
    // a reset right behind lasr_step_wait at ONE step in flight raced with that tail -- 1 reset of 303 lost in the first run of a
    // fresh context, found by the native front's depth-12 == depth-1 check against the numpy oracle.
    bool dec_tail_open = false;
    // native pump thread of the pipelined protocol (lasr_engine.hip, pump_main): owns the group launches while steps are in flight.
    // mu guards the decode-side host state: pending, h_avail, h_cur_seen, work_left, group_inflight, cont_iters, pred_par / lm.par,
    // the beam's host trees and results
    std::mutex mu;
    std::condition_variable cv_pump;
    std::condition_variable cv_prog;          // lasr_step_wait sleeping on the pump's progress counter (under mu)
    std::thread pump_th;
    bool pump_started = false, pump_on = false;
    std::atomic<bool> pump_stop{false};
    int pump_G = 3;                 // iterations per group launched by the pump; LASR_PUMP_G
    std::atomic<long long> kick{0}, progress{0};   // steps handed over by the API thread / groups consumed by the pump
    int pump_rc = 0; std::string pump_err;
    int kick_n = 3, wait_n = 1;     // iterations per group without the pump: kicked from submit / launched while waiting (swept on configs[1])
    // hipGraph cache of streaming decode groups: key = (first iteration, iterations, pe/T_row buffer,
    // predictor parity at group start, frames)
    std::map<std::tuple<int, int, int, int, int>, hipGraphExec_t> graphs;
    bool use_graphs = true;
    // main stream: the cell sequence of a pipelined model step as one graph per (active m-tiles, parity); default on for bf16
    // operands only (LASR_MAIN_GRAPH overrides)
    bool main_graph = false;
    hipStream_t stream_cap = nullptr;
    std::map<std::vector<unsigned long long>, hipGraphExec_t> mgraphs;
    int* T_row_main = nullptr;      // [M] fixed home of the step's per-row frame counts on the main stream (pipelined protocol)

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
        // beam search (rows = hypothesis slots, Md of them): c, y, lmz and valid ping-pong like h (a slot may continue any parent)
        std::vector<void*> y1; std::vector<float*> cst1; float* lmz1 = nullptr; int* valid1 = nullptr;
        // int8-served form (lasr_attach_lm_int8): integer-valued bf16 weights (row-major tiles of 16 outputs, K padded to 32),
        // per-tensor weight scales, separate biases; fp32 state (h[0][l] == h[1][l] row-major [M][H], cst[l] as [H][M])
        bool q8 = false;
        std::vector<void*> qWih, qWhh; void* qWout = nullptr;
        std::vector<float> s_ih, s_hh; float s_out = 0.f;
        std::vector<float*> b_ih, b_hh;
        std::vector<int> Kp_ih; int Kp_h = 0;          // padded K of the x side per layer / of every H-wide operand
        float *gx = nullptr, *gh = nullptr;            // [M][4H]
        unsigned short* qa = nullptr; float* sx = nullptr;   // quantised activations [M][Kmax], per-row scales [M]
        std::vector<unsigned short*> qh; std::vector<float*> sxh;   // per layer: the quantised image of h [M][Kp_h] and its scales [M]
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
    // fused streaming front-end (k_fe_mel + k_ln_tile): nothing is computed on a client chunk that does not complete a model
    // step; the step's launch works through the row's last n_buffer windows, which the PCM ring (ring_chunks =
    // n_window + n_buffer - 1 chunks) still holds.  pend_serial[s * n_buffer + j] = chunk count of slot s when its
    // pending frame j was taken; pend_mat = that frame had to be computed early into `pend` (the client pushed more
    // chunks without stepping, the ring was about to lose its window)
    bool fe_fused = false;
    int ring_chunks = 0;
    std::vector<int> pend_serial;
    std::vector<char> pend_mat;
    float* win = nullptr; int* ring_pos = nullptr;
    float* pend = nullptr;          // [M][n_buffer*n_stack][n_mels]
    float* stage_pcm = nullptr; size_t stage_pcm_floats = 0;
    float* win_rs = nullptr; size_t win_rs_floats = 0;       // resampled client windows (lasr_step_window)
    // streaming pushes from host memory: ring of pinned staging entries (read by the ring-append / front-end kernel over PCIe)
    // + one event per host push ("its source has been read": staging entry free, ticket consumed)
    static constexpr int NSTAGE = 64;     // > 2 pushes x 15 model steps in flight: a push never waits for its ring entry
    hipEvent_t push_ev[NSTAGE] = {}; bool push_used[NSTAGE] = {};
    long long push_serial = 0;            // tickets handed out so far
    float* push_stage_host = nullptr;     // pinned [NSTAGE][M][chunk]
    float* push_stage_host_dev = nullptr; // its device view
    float* push_stage_dev = nullptr;      // device [NSTAGE][M][chunk]: target of the per-push DMA
    hipStream_t stream_copy = nullptr; hipEvent_t push_copied[NSTAGE] = {}; bool push_dma[NSTAGE] = {};
    struct CopyPool {                     // helper threads of the staging copy (lazy; LASR_PUSH_THREADS, default 2)
        bool init = false;
        std::vector<std::thread> th;
        std::mutex m; std::condition_variable cv;                 // the job below is read and claimed under m
        const char* src = nullptr; char* dst = nullptr; size_t bytes = 0, part_bytes = 0;
        const float* const* rows = nullptr; size_t row_bytes = 0;   // gather job: destination row r from rows[r]
        int parts = 0, next = 0, done = 0; long long gen = 0;
        std::atomic<long long> gen_hint{0}, done_hint{0};         // spin targets: a new job exists / job g is complete
        std::atomic<bool> stop{false};
    } pool;
    std::vector<int> h_ring_pos;          // host mirror of ring_pos (every append goes through the host: +1 per pushed chunk)
    // Deferred append (round 5): the chunk of a lasr_push_submit that completes no model step is not appended by a launch of its
    // own when its source stays readable (the engine's device staging entry of a host push; a device buffer pushed with
    // LASR_PUSH_DEVICE_STABLE): the NEXT lasr_push_submit of the same slots appends both chunks in its front-end launch
    // (k_fe_mel src2) -- one launch and one kernel boundary less per model step on the stream that binds the job.  Every other
    // entry point that touches the PCM ring flushes it first (flush_lazy: the plain append launch).
    struct LazyPush { bool on = false; const float* src = nullptr; std::vector<int> slots; int ev_i = -1; bool dma = false; } lazy;
    int pump_nap_pct = 0;                 // LASR_PUMP_NAP_PCT (see pump_main); default 75 when several ranks share the host, else 0
    int lazy_taken = 0, lazy_flushed = 0; // deferred chunks appended by a front-end launch / by the plain launch after all (lasr_debug_config)
    bool lazy_on = true;                  // LASR_PUSH_LAZY=0 turns the deferred append off (A/B switch)
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
    int* res_dev = nullptr;         // its device view (k_publish stores the step's results there)
    int* pub_arrivals = nullptr;    // device: workgroups of the running k_publish that have stored their part
    size_t res_bytes = 0;

    // host mirrors
    std::vector<char> open_;
    std::vector<int> n_chunks, n_pend;
    std::vector<std::vector<int32_t>> queue;
    std::vector<double> neg_logp, align;

    // in-job timing of the dominant kernel (lasr_cell_prof): one HIP-event pair around the encoder-cell sequence of
    // every model step, on the stream the cells are launched on; harvested lazily (ring of pairs)
    static constexpr int NCELLEV = 64;
    bool cell_prof = false;
    bool cell_prof_events = true;   // lasr_cell_prof(c, 2): in-kernel clocks only, no HIP events on the stream
    hipEvent_t cp_ev[NCELLEV][2] = {};
    bool cp_ok = false;
    int cp_head = 0, cp_n = 0;      // ring: cp_n pairs outstanding, oldest at (cp_head - cp_n) mod NCELLEV
    int cp_cells[NCELLEV] = {};
    double cp_us = 0.0;
    long long cp_launches = 0;
    // ... and the kernels' own durations: every cell launch gets a slot {min entry, max exit} of the device's constant
    // wall clock over its workgroups (what a kernel trace calls the kernel's duration: no launch gaps, no event overhead)
    static constexpr int NCELLSLOT = 1 << 12;
    unsigned long long* cp_slots = nullptr;   // device [2][NCELLSLOT][PROF_W]: per launch and workgroup, entry clocks then exit clocks
    long long cp_slot_next = 0;
    std::vector<unsigned char> cp_slot_cells;  // cells computed by the launch of slot i (layer-wavefront launches: up to 8)
    int enc_wave = 0;               // encoder pass as a layer wavefront (cells of an anti-diagonal share a launch): default on for bf16; LASR_ENC_WAVE
    double cp_clock_mhz = 100.0;

    // stream timeline (lasr_trace): timestamped marks on the main and the decode stream of the pipelined protocol
    static constexpr int NTRACE = 8192;
    bool tr_on = false;
    std::vector<hipEvent_t> tr_ev;
    std::vector<int> tr_tag;
    std::vector<double> tr_val;
    std::vector<int> tr_prev_cur, tr_prev_ntot;    // per-row cursors / token counts at the previous consumed group
    int tr_last_G = 0;
    hipEvent_t tr_base = nullptr;
    std::atomic<int> tr_n{0};

    // stats
    bool profiling = false;
    hipEvent_t ev[8];
    bool ev_ok = false;
    lasr_step_stats stats{};
};

namespace {

// c->err belongs to the API thread (lasr_last_error hands out its c_str()); the pump thread points this at its own buffer
// (c->pump_err, read by the API thread under c->mu), so two threads never write one std::string
thread_local std::string* tl_err_sink = nullptr;
int fail(lasr_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (tl_err_sink) *tl_err_sink = buf;
    else if (c) c->err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(c, LASR_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// roctx ranges (tracing): LASR_ROCTX=1 resolves roctxRangePushA / roctxRangePop from the profiler's marker library
// at the first lasr_create (no link-time dependency; silently off when the library is absent) and the protocol's host phases push
// named ranges -- "lasr_push_submit", "lasr frontend+encoder", "lasr encoder cells", "lasr decode group", "lasr_step_wait",
// "lasr_step_stream", "lasr_transcribe" -- which `rocprofv3 --marker-trace --kernel-trace` lays over the kernels of both streams.
struct lasr_roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false;
};
inline lasr_roctx& roctx_state() { static lasr_roctx r; return r; }      // (per translation unit: every range is pushed from lasr_engine.hip)
inline void roctx_init() {
    lasr_roctx& r = roctx_state();
    if (r.tried) return;
    r.tried = true;
    if (!getenv("LASR_ROCTX") || atoi(getenv("LASR_ROCTX")) == 0) return;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
        void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        auto push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        auto pop = (int (*)())dlsym(h, "roctxRangePop");
        if (push && pop) { r.push = push; r.pop = pop; return; }
    }
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(roctx_state().push != nullptr) { if (on) roctx_state().push(name); }
    ~RoctxRange() { if (on) roctx_state().pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

template <class T>
int dalloc(lasr_ctx* c, T** p, size_t n) {
    void* q = nullptr;
    if (n == 0) n = 1;
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess) return fail(c, LASR_ENOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
    c->dev_allocs.push_back(q);
    *p = (T*)q;
    // LASR_POISON=1 (tests / soak): every device allocation starts as 0xff bytes (f32 / bf16 NaN, int -1) instead of whatever the
    // previous owner of the memory left: a kernel that reads a buffer it should first have written shows up at once
    static const int poison = getenv("LASR_POISON") ? atoi(getenv("LASR_POISON")) : 0;
    if (poison && hipMemset(q, 0xff, n * sizeof(T)) != hipSuccess) return fail(c, LASR_EHIP, "poison fill failed");
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
    if (d->beam > 1 && d->vocab > 2048) return false;                         // k_beam_select_rw keeps a hypothesis row's logits in one wave's registers
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


}  // namespace
