// lasr_engine.hip -- host side of liblasr_hip.so: weight packing, per-stream device state, the
// per-chunk step (front-end -> encoder -> greedy decode loop) and the C ABI of include/lasr.h.
// gfx950 only.  No CPU fallback: every numeric result comes from the kernels in lasr_kernels.hip.h.
#include "lasr_host.hip.h"
#include <sys/prctl.h>
#include <time.h>

static void cont_poll(lasr_ctx* c);       // (pipelined protocol, below; require_idle consumes a group that was still running)
static int flush_lazy(lasr_ctx* c);       // (deferred ring append of lasr_push_submit, below)
#include "lasr_cmd.hip.h"
#include "lasr_decode.hip.h"
#include "lasr_weights.hip.h"


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
    if (c->pump_started) {
        { std::lock_guard<std::mutex> lk(c->mu); c->pump_stop.store(true); }
        c->cv_pump.notify_all();
        c->pump_th.join();
    }
    (void)hipStreamSynchronize(c->stream);
    if (c->stream_dec) (void)hipStreamSynchronize(c->stream_dec);
    for (void* p : c->dev_allocs) (void)hipFree(p);
    if (c->cmd_host) (void)hipHostFree(c->cmd_host);
    if (c->res_host) (void)hipHostFree(c->res_host);
    if (c->cont_host) (void)hipHostFree(c->cont_host);
    for (auto& e : c->tr_ev) (void)hipEventDestroy(e);
    if (c->tr_base) (void)hipEventDestroy(c->tr_base);
    if (c->trellis_host) (void)hipHostFree(c->trellis_host);
    if (c->push_stage_host) (void)hipHostFree(c->push_stage_host);
    if (c->stream_copy) { (void)hipStreamSynchronize(c->stream_copy); (void)hipStreamDestroy(c->stream_copy); }
    for (auto& e : c->push_copied)
        if (e) (void)hipEventDestroy(e);
    if (!c->pool.th.empty()) {
        { std::lock_guard<std::mutex> lk(c->pool.m); c->pool.stop.store(true); }
        c->pool.cv.notify_all();
        for (auto& t : c->pool.th) t.join();
    }
    for (auto& e : c->push_ev)
        if (e) (void)hipEventDestroy(e);
    for (void* p : c->host_allocs) (void)hipHostFree(p);
    if (c->ev_ok)
        for (auto& e : c->ev) (void)hipEventDestroy(e);
    for (auto& e : c->ev_enc)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_misc) (void)hipEventDestroy(c->ev_misc);
    if (c->cp_ok)
        for (auto& p : c->cp_ev)
            for (auto& e : p) (void)hipEventDestroy(e);
    for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second);
    for (auto& kv : c->cgraphs) (void)hipGraphExecDestroy(kv.second);
    for (auto& kv : c->mgraphs) (void)hipGraphExecDestroy(kv.second);
    if (c->stream_cap) (void)hipStreamDestroy(c->stream_cap);
    if (c->stream_dec) { (void)hipStreamSynchronize(c->stream_dec); (void)hipStreamDestroy(c->stream_dec); }
    if (c->stream_lm) { (void)hipStreamSynchronize(c->stream_lm); (void)hipStreamDestroy(c->stream_lm); }
    if (c->stream_nb) { (void)hipStreamSynchronize(c->stream_nb); (void)hipStreamDestroy(c->stream_nb); }
    if (c->ev_lm_fork) (void)hipEventDestroy(c->ev_lm_fork);
    if (c->ev_lm_join) (void)hipEventDestroy(c->ev_lm_join);
    delete c;
}

const char* lasr_last_error(const lasr_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

static int overlap_probe_impl(lasr_ctx* c, int delay_us, double* ratio, hipStream_t sa = nullptr, hipStream_t sb = nullptr);

static int create_impl(lasr_ctx* c, const float* weights, size_t n_weights) {
    roctx_init();
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
        if (e) c->la_stream = c->la_offline = c->la_sync = std::min(lasr_ctx::LA_MAX, std::max(1, atoi(e)));
        if (c->W > 1) c->la_stream = c->la_offline = c->la_sync = 1;
        c->la = c->la_stream;
        // beam: a model step takes ~5 selection rounds per frame; with short rounds (configs[2]: 77 us) the host round trip per
        // group is worth amortising (4 rounds per group: 13.4-13.8 -> 14.3-14.5 k audio-s/s), with long ones (configs[4]: 1024
        // hypothesis rows x 1536) the rounds launched past the need cost more (10.3 -> 10.0 k): profiles/r03/r03_experiments.txt O
        if (c->W > 1 && (size_t)Md * H <= (size_t)256 * 1024) c->wait_n = 4;
    }
    // groups launched by the pump thread (see pump_main): greedy 2 iterations (f32 52.1-52.6 k at 2, 51.7-52.1 at 3, 51.8 at 4 against
    // 51.8-51.9 without the pump; bf16 96.0 / 93.4 / 93.1 against 92.6: profiles/r04/r04_pump_ab.txt); beam: the group size of
    // round 3's wait path for short rounds (configs[2]: 4; re-checked in round 6: 2 / 4 / 6 / 8 within 1 %), 2 for long ones under
    // the pump (configs[4], 5 / 6 steps in flight: G = 1 25.4 / 27.2 k, 2 26.0 / 27.6 k, 3 25.8 / 27.7 k, 4 25.2 / 26.9 k:
    // profiles/r06/r06_experiments.txt G)
    c->pump_G = c->W > 1 ? std::max(2, c->wait_n) : 2;
    if (getenv("LASR_PUMP_G")) c->pump_G = std::max(1, std::min(8, atoi(getenv("LASR_PUMP_G"))));
    {   // The pump thread spins on its group's flag by design (one host core per context).  When several engine processes share the
        // host (one rank per GPU: the launcher's LOCAL_WORLD_SIZE / WORLD_SIZE say so) it instead sleeps through 75 % of a group's
        // expected duration (moving average) before it polls: 2.04 -> 1.32 host cores per rank in the 8-rank dry run at -1.7 %
        // aggregate, -0.4 % at one rank (profiles/r05/pump_nap).  A single process keeps the spin: the headline takes the rate.
        const char* lw = getenv("LOCAL_WORLD_SIZE") ? getenv("LOCAL_WORLD_SIZE") : getenv("WORLD_SIZE");
        c->pump_nap_pct = (lw && atoi(lw) > 1) ? 75 : 0;
        if (getenv("LASR_PUMP_NAP_PCT")) c->pump_nap_pct = std::max(0, std::min(90, atoi(getenv("LASR_PUMP_NAP_PCT"))));
    }
    if (getenv("LASR_PUSH_LAZY")) c->lazy_on = atoi(getenv("LASR_PUSH_LAZY")) != 0;
    const size_t Mj = (size_t)c->MTj * 16;
    c->G_pred = d.pred_cell ? 4 : 3;
    c->bf = d.dtype == 1; c->kch = c->bf ? 32 : 16; c->esz = c->bf ? 2 : 4;
    Reader rd{weights, n_weights};
    {   // encoder cell tiling D (see EpiLSTMe): where tiling C needs more than one 32 x 32 workgroup per CU and the hidden size divides
        c->enc_u12 = c->bf && H % 12 == 0 && M % 64 == 0 && (H / 12) * (M / 64) >= 256 && (H / 8) * (M / 32) > 256
                     && F % 32 == 0 && (F / 32) % 8 == 0 && (H / 32) % 8 == 0;
        if (getenv("LASR_ENC_U12")) c->enc_u12 = atoi(getenv("LASR_ENC_U12")) != 0 && H % 12 == 0 && M % 64 == 0;
    }
    {
        if (getenv("LASR_NO_GRAPH")) c->use_graphs = false;
        if (getenv("LASR_CELL_NW")) c->cell_nw = atoi(getenv("LASR_CELL_NW")) == 4 ? 4 : 8;
        // encoder pass as a layer wavefront: bf16 cells are load-paced with idle MFMA time, two of them per CU overlap (streaming
        // +4 %, offline +14 %); f32 cells are MFMA-paced, two per CU only contend (-4 %)
        c->enc_wave = c->bf ? 1 : 0;
        // (tiling D -- one 100 KB-LDS workgroup per CU and cell -- has nothing to overlap inside a launch: plain launches.  cfg5,
        //  128 streams, greedy / beam 8: 46.8 / 14.6 k against 45.9 / 14.4 k as a wavefront, profiles/r04/r04_cell_tiling_d.txt)
        if (c->enc_u12) c->enc_wave = 0;
        if (getenv("LASR_ENC_WAVE")) c->enc_wave = atoi(getenv("LASR_ENC_WAVE"));
        // decode-stream GEMMs (predictor cells, PPJ, logits): 8 waves per workgroup with either operand type.  (Rounds 2-5 ran them on
        // 4 waves with f32 operands: +5 % whole job when the decode loop had slack; with the loop the binding stream -- 18 steps in
        // flight, 92 % busy -- 8 waves win: 53.5-53.6 -> 54.2-54.6 k timed, 57.3-57.5 -> 57.6-58.3 k sustained, same box, 4 runs each:
        // profiles/r06/r06_experiments.txt T)
        if (getenv("LASR_DBG_TIMING")) {
            RC(dalloc(c, &c->dbg, (size_t)5 * 4096 * 16));
            HIPCHK(c, hipMemset(c->dbg, 0, sizeof(unsigned long long) * 5 * 4096 * 16));
        }
        if (getenv("LASR_DBG_ENCLOG") && atoi(getenv("LASR_DBG_ENCLOG")) > 0) {
            c->enclog_cap = atoi(getenv("LASR_DBG_ENCLOG"));
            RC(dalloc(c, &c->enclog, (size_t)c->enclog_cap * 32 * c->M));
            HIPCHK(c, hipMemset(c->enclog, 0, sizeof(unsigned) * (size_t)c->enclog_cap * 32 * c->M));
            if (getenv("LASR_DBG_PENDLOG") && atoi(getenv("LASR_DBG_PENDLOG")) > 0)
                RC(dalloc(c, &c->pendlog, (size_t)c->enclog_cap * c->M * d.n_buffer * d.n_stack * d.n_mels));
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
        c->bh.assign(M, lasr_ctx::BeamHost{});
        for (auto& b : c->bh) bh_reset(b, c->W);
        c->committed.assign(M, {}); c->committed_score.assign(M, 0.0); c->best_full.assign(M, {});
        c->b_frames_done.assign(M, 0); c->b_results.assign(M, {});
        {   // continuous beam loop: the rounds' records, frame marks and step-end scores in pinned host memory (zero-copy stores)
            const size_t n_tre = (size_t)lasr_ctx::TRING * Md, n_fd = (size_t)lasr_ctx::TRING * M;
            const size_t n_es = (size_t)M * lasr_ctx::ENDSLOTS * c->W, n_ea = (size_t)M * lasr_ctx::ENDSLOTS;
            char* blk = nullptr;
            const size_t bytes = sizeof(double) * n_es + sizeof(int) * (n_tre + n_fd + n_ea) + 64;
            HIPCHK(c, hipHostMalloc((void**)&blk, bytes));
            c->host_allocs.push_back(blk);
            memset(blk, 0, bytes);
            void* dp = nullptr;
            HIPCHK(c, hipHostGetDevicePointer(&dp, blk, 0));
            char* dblk = (char*)dp;
            c->b_endsc_host = (double*)blk; c->b_endsc_dev = (double*)dblk;
            size_t off = sizeof(double) * n_es;
            c->b_tre_host = (int*)(blk + off); c->b_tre_dev = (int*)(dblk + off); off += sizeof(int) * n_tre;
            c->b_fdone_host = (int*)(blk + off); c->b_fdone_dev = (int*)(dblk + off); off += sizeof(int) * n_fd;
            c->b_endal_host = (int*)(blk + off); c->b_endal_dev = (int*)(dblk + off);
        }
    }
    RC(dalloc(c, (char**)&c->ja, Mj * J * c->esz)); HIPCHK(c, hipMemset(c->ja, 0, Mj * J * c->esz));
    RC(dalloc(c, (char**)&c->cvt_a, (size_t)M * H * c->esz)); RC(dalloc(c, (char**)&c->cvt_b, (size_t)M * H * c->esz));
    HIPCHK(c, hipMemset(c->cvt_a, 0, (size_t)M * H * c->esz)); HIPCHK(c, hipMemset(c->cvt_b, 0, (size_t)M * H * c->esz));
    RC(dalloc(c, &c->logits, Mj * V)); HIPCHK(c, hipMemset(c->logits, 0, sizeof(float) * Mj * V));
    RC(dalloc(c, &c->ds.t_idx, M)); RC(dalloc(c, &c->ds.iters, M)); RC(dalloc(c, &c->ds.token, Md));
    RC(dalloc(c, &c->ds.emit, Md)); RC(dalloc(c, &c->ds.logp_sum, M));
    RC(dalloc(c, &c->ds.sum_iters, M)); RC(dalloc(c, &c->ds.n_ones, M)); RC(dalloc(c, &c->zero_rows, M));
    HIPCHK(c, hipMemset(c->zero_rows, 0, sizeof(int) * M));
    RC(dalloc(c, &c->pub_arrivals, 4)); HIPCHK(c, hipMemset(c->pub_arrivals, 0, sizeof(int) * 4));
    RC(dalloc(c, &c->T_row_fix, M));
    HIPCHK(c, hipMemset(c->T_row_fix, 0, sizeof(int) * M));
    RC(dalloc(c, &c->T_row_main, M));
    HIPCHK(c, hipMemset(c->T_row_main, 0, sizeof(int) * M));
    // measured (profiles/r03/r03_experiments.txt I): host time per model step 66 -> 45 us, but the replay starts its first cell ~6 us
    // later than a plain launch does: f32 -2 % (52.8 against 54.0 k audio-s/s), bf16 +0.5 %  =>  on for bf16, off for f32
    {   // see lasr_ctx::fe_lds_pad.  The kernels' own (static) LDS is asked of the runtime, not assumed
        hipFuncAttributes fa{}, la_{};
        HIPCHK(c, hipFuncGetAttributes(&fa, (const void*)k_fe_mel<10>));
        HIPCHK(c, hipFuncGetAttributes(&la_, (const void*)k_logmel));
        const int lds_cu = 160 * 1024, excl = 98304;             // 98 304 + 65 536 > 160 KB
        const bool wide_decode = c->Md >= 256 || lasr_ctx::LA_MAX * M >= 512;
        c->fe_lds_pad = wide_decode ? std::max(0, excl - (int)fa.sharedSizeBytes) : 0;
        if (getenv("LASR_FE_LDS_PAD")) c->fe_lds_pad = std::max(0, std::min(lds_cu - (int)fa.sharedSizeBytes, atoi(getenv("LASR_FE_LDS_PAD"))));
        c->logmel_lds_pad = c->fe_lds_pad ? std::max(0, excl - (int)la_.sharedSizeBytes) : 0;
        HIPCHK(c, hipFuncSetAttribute((const void*)k_fe_mel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_cu - (int)fa.sharedSizeBytes));
        HIPCHK(c, hipFuncSetAttribute((const void*)k_logmel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_cu - (int)la_.sharedSizeBytes));
    }
    c->main_graph = c->bf != 0;
    if (getenv("LASR_MAIN_GRAPH")) c->main_graph = atoi(getenv("LASR_MAIN_GRAPH")) != 0;
    c->T_row_dev = c->zero_rows;
    for (int q = 0; q < lasr_ctx::NFLY; ++q) {
        RC(dalloc(c, &c->T_row_ring[q], M));
        HIPCHK(c, hipMemset(c->T_row_ring[q], 0, sizeof(int) * M));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_enc[q], hipEventDisableTiming));
    }
    // (round 3 measured CU masks for either stream, a high-priority decode stream and per-stream delay probes: every one of them
    //  lost or was a probe -- profiles/r03/r03_experiments.txt B, C; the switches are gone, the patch of the masks is in the history)
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream_dec, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_misc, hipEventDisableTiming));
    {   // The pipelined protocol needs the decode stream and the ctx stream on DIFFERENT hardware queues.  The runtime multiplexes
        // the process's streams onto a few queues (GPU_MAX_HW_QUEUES, 4 by default); in a process that already owns several
        // streams (other contexts, torch, an RCCL communicator) a new stream can land on the ctx stream's queue, and the two
        // streams then run strictly one after the other (the job at 0.73 of its rate: profiles/r04/r04_dist_ab.txt).  So the
        // stream is probed (two 0.3 ms delay kernels, one per stream) and replaced until it overlaps; the rejected ones are kept
        // alive until then, so that the next one is placed elsewhere.  LASR_DEC_STREAM_PICK=0 switches this off.
        static const int pick = getenv("LASR_DEC_STREAM_PICK") ? atoi(getenv("LASR_DEC_STREAM_PICK")) : 1;
        std::vector<hipStream_t> rejected;
        c->dec_stream_attempts = 1;
        for (int attempt = 0; pick && attempt < 8; ++attempt) {
            double ratio = 0.0;
            if (overlap_probe_impl(c, 300, &ratio) != LASR_OK) { (void)hipGetLastError(); break; }
            c->dec_stream_ratio = ratio;
            if (ratio < 1.5) break;
            rejected.push_back(c->stream_dec);
            c->stream_dec = nullptr;
            HIPCHK(c, hipStreamCreateWithFlags(&c->stream_dec, hipStreamNonBlocking));
            c->dec_stream_attempts++;
        }
        for (hipStream_t st : rejected) (void)hipStreamDestroy(st);
        if (getenv("LASR_VERBOSE"))
            fprintf(stderr, "[lasr] decode stream: %d stream(s) tried, overlap probe %.2f\n", c->dec_stream_attempts, c->dec_stream_ratio);
    }
    RC(dalloc(c, &c->pe_ring, (size_t)lasr_ctx::RING * M * J));
    HIPCHK(c, hipMemset(c->pe_ring, 0, sizeof(float) * (size_t)lasr_ctx::RING * M * J));
    RC(dalloc(c, &c->c_cur, M)); RC(dalloc(c, &c->c_avail, M)); RC(dalloc(c, &c->c_iters, M)); RC(dalloc(c, &c->c_target, M));
    RC(dalloc(c, &c->c_ntotal, M)); RC(dalloc(c, &c->c_enc_frames, M)); RC(dalloc(c, &c->c_enc_base, M));
    HIPCHK(c, hipMemset(c->c_enc_base, 0, sizeof(int) * M)); RC(dalloc(c, &c->c_behind, 64));
    // (c_ntok_end / c_tok_ring live in pinned host memory, written by k_select directly: see below)
    for (int* p : {c->c_cur, c->c_avail, c->c_iters, c->c_target, c->c_ntotal, c->c_enc_frames})
        HIPCHK(c, hipMemset(p, 0, sizeof(int) * M));
    HIPCHK(c, hipMemset(c->c_behind, 0, sizeof(int) * 64));
    RC(dalloc(c, &c->c_done, 64)); HIPCHK(c, hipMemset(c->c_done, 0, sizeof(int) * 64));
    RC(dalloc(c, &c->c_done2, 64)); HIPCHK(c, hipMemset(c->c_done2, 0, sizeof(int) * 64));
    RC(dalloc(c, &c->c_iter, 4)); HIPCHK(c, hipMemset(c->c_iter, 0, sizeof(int) * 4));
    HIPCHK(c, hipHostMalloc((void**)&c->cont_host, sizeof(int) * (16 + (size_t)M * (lasr_ctx::NFLY + lasr_ctx::ENDSLOTS + lasr_ctx::TOKRING))));
    memset(c->cont_host, 0, sizeof(int) * (16 + (size_t)M * (lasr_ctx::NFLY + lasr_ctx::ENDSLOTS + lasr_ctx::TOKRING)));
    {   // continuous decode: the token ring and the per-step boundary marks are written by k_select straight
        // into this pinned block (zero-copy stores over PCIe, flushed at kernel end): a finished step needs
        // no result copy at all -- the host reads them as soon as the group's "rows behind" word says 0
        void* dp = nullptr;
        HIPCHK(c, hipHostGetDevicePointer(&dp, c->cont_host, 0));
        c->c_flag_dev = (int*)dp;
        c->c_hcur_dev = (int*)dp + 16;
        c->c_ntok_end = (int*)dp + 16 + (size_t)lasr_ctx::NFLY * M;
        c->c_tok_ring = c->c_ntok_end + (size_t)M * lasr_ctx::ENDSLOTS;
    }
    c->h_frames_sub.assign(M, 0); c->h_fetched.assign(M, 0); c->h_cur_seen.assign(M, 0); c->h_avail.assign(M, 0);
    c->dec_t_idx = c->ds.t_idx;
    c->T_row_dec = c->T_row_dev;
    for (int* p : {c->ds.t_idx, c->ds.iters, c->ds.sum_iters, c->ds.n_ones})
        HIPCHK(c, hipMemset(p, 0, sizeof(int) * M));
    HIPCHK(c, hipMemset(c->ds.token, 0, sizeof(int) * Md)); HIPCHK(c, hipMemset(c->ds.emit, 0, sizeof(int) * Md));
    HIPCHK(c, hipMemset(c->ds.logp_sum, 0, sizeof(double) * M));
    // (the reference front-end: 10 frames of 128 mels per stacked frame; other shapes take the per-chunk kernels)
    c->fe_fused = M <= 512 && d.n_buffer <= 4 && d.n_stack == 10 && d.n_mels <= 128 && d.feat == 1280 && !getenv("LASR_FE_LEGACY");
    c->h_ring_pos.assign(M, 0);
    c->ring_chunks = c->fe_fused ? d.n_window + d.n_buffer - 1 : d.n_window;
    c->pend_serial.assign((size_t)M * d.n_buffer, 0); c->pend_mat.assign((size_t)M * d.n_buffer, 0);
    RC(dalloc(c, &c->win, (size_t)M * c->ring_chunks * d.chunk)); HIPCHK(c, hipMemset(c->win, 0, (size_t)M * c->ring_chunks * d.chunk * 4));
    RC(dalloc(c, &c->ring_pos, M)); HIPCHK(c, hipMemset(c->ring_pos, 0, sizeof(int) * M));
    RC(dalloc(c, &c->pend, (size_t)M * d.n_buffer * d.n_stack * d.n_mels));
    HIPCHK(c, hipMemset(c->pend, 0, (size_t)M * d.n_buffer * d.n_stack * d.n_mels * 4));

    lasr_ctx::Cmd tmp;
    c->cmd_bytes = cmd_layout(tmp, nullptr, M);
    HIPCHK(c, hipHostMalloc((void**)&c->cmd_host, c->cmd_bytes * NCMD));
    RC(dalloc(c, &c->cmd_dev, c->cmd_bytes * NCMD)); HIPCHK(c, hipMemset(c->cmd_dev, 0, c->cmd_bytes * NCMD));
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
            launch_table_gemm_f32(c, H / 16, V / 16, g, ea);
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
        launch_table_gemm_f32(c, G * H / 16, V / 16, g, ea);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        dfree(c, wt); dfree(c, bt); dfree(c, EF); dfree(c, emb_dev);
    }

    c->open_.assign(M, 0); c->n_chunks.assign(M, 0); c->n_pend.assign(M, 0);
    c->queue.assign(M, {}); c->neg_logp.assign(M, 0.0); c->align.assign(M, 0.0);
    c->ev_ok = true;
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) c->ev_ok = false;
    // Greedy decode: what a reset leaves in a row's predictor state -- the learned initial state advanced by one step on BOS, and the
    // joint's predictor half of it (models.py:489) -- does not depend on the row: computed ONCE here, through the same kernels a
    // reset used to launch (row 0), and stored by k_reset_rows from then on.  A reset then costs the decode stream one small launch
    // instead of one + three weight-streaming GEMMs (the servicer's reset rule resets ~2 streams per model step in a replay).
    static const int bos_cache = getenv("LASR_BOS_CACHE") ? atoi(getenv("LASR_BOS_CACHE")) : 1;
    if (bos_cache && c->W == 1) {
        RC(cmd_begin(c));
        c->hc.what[0] = 2;
        RC(cmd_commit(c));
        RC(apply_reset(c, true));
        BosArgs b{};
        b.H = H; b.J = J; b.Lp = d.pred_layers; b.M = M; b.lstm = d.pred_cell; b.bf = c->bf;
        c->bos_h.assign(d.pred_layers, nullptr); c->bos_c.assign(d.pred_layers, nullptr);
        for (int l = 0; l < d.pred_layers; ++l) {
            RC(dalloc(c, &c->bos_h[l], H));
            if (d.pred_cell) RC(dalloc(c, &c->bos_c[l], H));
            b.pred_h[l] = c->pred_h[c->pred_par][l]; b.pred_c[l] = d.pred_cell ? c->pred_c[l] : nullptr;
            b.bos_h[l] = c->bos_h[l]; b.bos_c[l] = c->bos_c[l];
        }
        RC(dalloc(c, &c->bos_pp, J));
        b.pp = c->pp; b.bos_pp = c->bos_pp;
        hipLaunchKernelGGL(k_bos_capture, dim3(grid1(std::max(H, J))), dim3(256), 0, c->stream, b);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        c->bos_ready = true;
    }
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
static void cont_poll(lasr_ctx* c);

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

// lasr_stream_reset for n slots: all of them are checked before anything is changed (an error leaves every slot as it was), then
// ONE command block and one set of launches serve them all
static int reset_impl(lasr_ctx* c, const int* slots, int n, int what) {
    if (n <= 0) return LASR_OK;
    for (int i = 0; i < n; ++i)
        if (slots[i] < 0 || slots[i] >= c->d.max_streams || !c->open_[slots[i]]) return fail(c, LASR_ESTATE, "slot %d is not open", slots[i]);
    {
        // With LASR_RESET_IF_DECODED a slot with submitted steps can be reset once the decode loop has FINISHED them for this slot
        // (lasr_peek_slot says so), collected or not: its encoder steps are behind on the ctx stream, its frames are decoded, the tokens sit in the pinned
        // ring until lasr_step_wait hands them out.  Model state only (bits 1 | 2 | 4); greedy decode.
        const bool if_decoded = (what & LASR_RESET_IF_DECODED) != 0;
        what &= ~LASR_RESET_IF_DECODED;
        std::lock_guard<std::mutex> lk(c->mu);
        bool polled = false;
        for (int i = 0; i < n; ++i) {
            const int slot = slots[i];
            bool inflight = false, undecoded = !if_decoded;      // (without the flag: refused whenever a step of the slot is uncollected)
            for (const auto& p : c->pending)
                if (std::find(p.rows.begin(), p.rows.end(), slot) != p.rows.end()) {
                    inflight = true;
                    if (!c->pump_on && !polled) { cont_poll(c); polled = true; }
                    undecoded |= c->h_cur_seen[slot] < p.target[slot];
                }
            if (inflight && (undecoded || (what & 8) || c->W > 1))
                return fail(c, LASR_ESTATE, "slot %d has a submitted step in flight: call lasr_step_wait first", slot);
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    if (what & 8) RC(flush_lazy(c));         // (the PCM ring of these slots starts over)
    for (int i = 0; i < n; ++i) {
        const int slot = slots[i];
        if (what & 8) { c->n_chunks[slot] = 0; c->n_pend[slot] = 0; c->queue[slot].clear(); c->neg_logp[slot] = 0.0; }
        if (what & 2) beam_host_reset(c, slot, (what & 8) != 0);
    }
    if (what & 7) {
        RC(cmd_begin(c));
        for (int i = 0; i < n; ++i) c->hc.what[slots[i]] = what & 7;
        RC(cmd_commit(c));
        std::lock_guard<std::mutex> lk(c->mu);            // (decode-side launches: the pump thread stays out)
        if (c->pending.empty() && !c->group_inflight) {
            RC(order_after_decode_tail(c));          // (the last group's predictor cells may still be running on the decode stream)
            RC(apply_reset(c, (what & 2) != 0));
        } else {
            // other streams have steps in flight: the encoder side of the reset is ordered on the main
            // stream, the predictor / LM side (BOS pass) on the decode stream, between two iteration groups
            if (what & 1) RC(apply_reset(c, false, 1));
            if (what & 6) {
                // the command block reaches the decode stream by a copy of its own: an event edge from the main stream would make
                // the decode loop wait for everything queued there (up to `steps in flight` encoder passes, ~2 ms at depth 12)
                hipStream_t keep = c->stream;
                HIPCHK(c, hipMemcpyAsync((char*)c->dc.T_row, (char*)c->hc.T_row, c->cmd_bytes, hipMemcpyHostToDevice, c->stream_dec));
                c->stream = c->stream_dec;
                int rc = apply_reset(c, (what & 2) != 0, 2);
                c->stream = keep;
                if (rc) return rc;
            }
        }
    }
    return LASR_OK;
}
int lasr_stream_reset(lasr_ctx* c, int slot, int what) {
    if (!c) return LASR_EINVAL;
    return reset_impl(c, &slot, 1, what);
}
int lasr_stream_reset_many(lasr_ctx* c, const int* slots, int n, int what) {
    if (!c) return LASR_EINVAL;
    RC(check_slots(c, slots, n, true));
    return reset_impl(c, slots, n, what);
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
static int cont_pump(lasr_ctx* c, int G);
static void cont_poll(lasr_ctx* c);

static void fill_mel_args(lasr_ctx* c, MelArgs& m) {
    const lasr_model_desc& d = c->d;
    m.window = c->window; m.tw512 = c->tw512; m.tw1024 = c->tw1024; m.fb_start = c->fb_start; m.fb_off = c->fb_off;
    m.fb_w = c->fb_w; m.n_mels = d.n_mels; m.hop = d.hop;
    m.win_off = (d.n_fft - d.win) / 2; m.win_len = d.win; m.fb_nnz = c->fb_nnz;
}
// window geometry of the streaming front-end (api-server.py:95-102 + TransformTime + StreamPostprocess): first frame picked
static int stream_frame0(lasr_ctx* c, int* nf_out) {
    const lasr_model_desc& d = c->d;
    const long long N = (long long)d.n_window * d.chunk;
    const int T = 1 + (int)(N / d.hop);
    const int a0 = T / 3 + 1;
    if (nf_out) *nf_out = std::min(d.n_stack, T - a0);
    return a0;
}

// Fused front-end, irregular clients: a slot that is about to be pushed again although it still has a pending frame whose
// window the ring would lose (more than one chunk pushed per lasr_step_* call) gets that frame computed NOW into `pend`
// (the per-chunk log-mel kernel, window selected by its age); the step's k_fe_mel launch then skips it (age 15).
// the per-chunk log-mel kernel of the streaming protocols (more than 512 slots, non-standard front-end shapes, irregular clients) runs
// beside the decode stream like k_fe_mel: the same CU exclusion against the wide decode tilings (lasr_ctx::fe_lds_pad)
static int logmel_lds_pad(lasr_ctx* c) { return c->logmel_lds_pad; }
static int materialize_pending(lasr_ctx* c, const int* slots, int n) {
    const lasr_model_desc& d = c->d;
    const int slack = c->ring_chunks - d.n_window;
    for (int j = 0; j < d.n_buffer; ++j) {
        MelArgs m{};
        bool any = false;
        for (int i = 0; i < n; ++i) {
            const int s = slots[i];
            if (j >= c->n_pend[s] || c->pend_mat[(size_t)s * d.n_buffer + j]) continue;
            const int age = c->n_chunks[s] - c->pend_serial[(size_t)s * d.n_buffer + j];
            if (age + 1 <= slack) continue;                      // still in the ring after this push
            if (!any) { for (int r = 0; r < 512; ++r) m.sel_v[r] = -1; any = true; }
            m.sel_v[s] = (short)(j * d.n_stack);
            m.age_v[s] = (unsigned char)age;
            c->pend_mat[(size_t)s * d.n_buffer + j] = 1;
        }
        if (!any) continue;
        RC(flush_lazy(c));                                       // (the kernel below reads the ring)
        fill_mel_args(c, m);
        m.pcm = c->win; m.N = (long long)d.n_window * d.chunk; m.stream = 1; m.ring_head = c->ring_pos; m.chunk = d.chunk;
        m.n_window = d.n_window; m.ring_chunks = c->ring_chunks; m.frame0 = stream_frame0(c, nullptr);
        m.frames_per_row = d.n_stack; m.out = c->pend; m.out_frames = d.n_buffer * d.n_stack;
        m.by_value = 1; m.trow_out = nullptr;
        hipLaunchKernelGGL(k_logmel, dim3((d.n_stack + 3) / 4, c->M), dim3(256), logmel_lds_pad(c), c->stream, m);
    }
    return LASR_OK;
}

// ---- client chunks -> PCM ring.  Three kinds of caller memory:
//   device           read by the ring-append (or front-end) kernel in stream order, like any stream-ordered copy;
//   host (default)   pageable OR pinned: copied into the engine's own pinned staging ring before the call returns (the caller's
//                    buffer is free on return), DMA'd from there into a device staging entry on a copy-only stream; the kernel
//                    reads HBM;
//   pinned, no copy  LASR_PUSH_PINNED_NOCOPY: the DMA (or, zero-copy, the kernel) reads the CALLER's pinned buffer after the call
//                    has returned; the buffer must stay untouched until lasr_push_consumed(ticket) says so.
struct PushSrc { const float* src = nullptr; int ev_i = -1; long long ticket = -1; bool dma = false; };

// threaded copy into the staging ring: 328 KB per push at 64 streams is 40 us of one core when the source is cold (pageable
// client buffers), which made the single host thread the limit of the PCIe-inclusive rate.  The calling thread takes parts as
// well, so a helper that wakes up late costs nothing.
static void pool_worker(lasr_ctx::CopyPool* P) {
    long long seen = 0;
    for (;;) {
        // wait for a new job: spin briefly on the hint (pushes come every ~100 us in steady state), then sleep
        for (int spins = 0; P->gen_hint.load(std::memory_order_acquire) == seen && !P->stop.load(std::memory_order_acquire); ++spins) {
            if (spins < 20000) { __builtin_ia32_pause(); continue; }
            std::unique_lock<std::mutex> lk(P->m);
            P->cv.wait_for(lk, std::chrono::milliseconds(50), [&] { return P->gen != seen || P->stop.load(); });
            break;
        }
        if (P->stop.load(std::memory_order_acquire)) return;
        std::unique_lock<std::mutex> lk(P->m);               // parts are claimed under the lock: the job's fields cannot change under a helper
        seen = P->gen;
        while (P->next < P->parts) {
            const int i = P->next++;
            const size_t lo = (size_t)i * P->part_bytes, hi = std::min(P->bytes, lo + P->part_bytes);
            const char* s = P->src; char* d = P->dst;
            const float* const* rows = P->rows; const size_t rb = P->row_bytes;
            lk.unlock();
            if (rows) for (size_t r = lo / rb; r < hi / rb; ++r) memcpy(d + r * rb, rows[r], rb);
            else memcpy(d + lo, s + lo, hi - lo);
            lk.lock();
            if (++P->done == P->parts) P->done_hint.store(P->gen, std::memory_order_release);
        }
    }
}
// rows != nullptr: gather -- row r of the destination (row_bytes each) comes from rows[r]
static void staged_copy(lasr_ctx* c, void* dst, const void* src, size_t bytes, const float* const* rows = nullptr, size_t row_bytes = 0) {
    lasr_ctx::CopyPool& P = c->pool;
    if (!P.init) {
        P.init = true;
        int nth = 2;
        if (getenv("LASR_PUSH_THREADS")) nth = std::max(0, std::min(8, atoi(getenv("LASR_PUSH_THREADS"))));
        for (int i = 0; i < nth; ++i) P.th.emplace_back(pool_worker, &P);
    }
    if (P.th.empty() || bytes < (size_t)(96 << 10)) {
        if (rows) for (size_t r = 0; r < bytes / row_bytes; ++r) memcpy((char*)dst + r * row_bytes, rows[r], row_bytes);
        else memcpy(dst, src, bytes);
        return;
    }
    std::unique_lock<std::mutex> lk(P.m);                    // (the previous job is complete: done == parts, nobody is copying)
    P.src = (const char*)src; P.dst = (char*)dst; P.bytes = bytes; P.rows = rows; P.row_bytes = row_bytes;
    P.part_bytes = ((bytes / (4 * (P.th.size() + 1))) + 4095) & ~(size_t)4095;
    if (rows) P.part_bytes = std::max<size_t>(1, (bytes / row_bytes + 4 * (P.th.size() + 1) - 1) / (4 * (P.th.size() + 1))) * row_bytes;   // whole rows
    P.parts = (int)((bytes + P.part_bytes - 1) / P.part_bytes);
    P.next = 0; P.done = 0;
    const long long g = ++P.gen;
    P.gen_hint.store(g, std::memory_order_release);
    P.cv.notify_all();
    while (P.next < P.parts) {                               // the calling thread takes parts as well
        const int i = P.next++;
        const size_t lo = (size_t)i * P.part_bytes, hi = std::min(bytes, lo + P.part_bytes);
        lk.unlock();
        if (rows) for (size_t r = lo / row_bytes; r < hi / row_bytes; ++r) memcpy((char*)dst + r * row_bytes, rows[r], row_bytes);
        else memcpy((char*)dst + lo, (const char*)src + lo, hi - lo);
        lk.lock();
        if (++P.done == P.parts) P.done_hint.store(g, std::memory_order_release);
    }
    lk.unlock();
    while (P.done_hint.load(std::memory_order_acquire) != g) __builtin_ia32_pause();      // parts still being copied by helpers
}

static int push_prepare(lasr_ctx* c, const int* slots, int n, const float* pcm, int flags, PushSrc& ps, const float* const* rows = nullptr) {
    const int CH = c->d.chunk;
    ps.src = pcm;
    if (!rows && is_device_ptr(pcm)) {
        if (flags & LASR_PUSH_PINNED_NOCOPY) return fail(c, LASR_EINVAL, "LASR_PUSH_PINNED_NOCOPY needs pinned HOST memory");
        return LASR_OK;
    }
    if (!c->push_ev[0])
        for (auto& e : c->push_ev) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // one event per host push: "the kernel that reads this push's source has finished" (staging entry free / ticket consumed)
    ps.ticket = c->push_serial++;
    ps.ev_i = (int)(ps.ticket % lasr_ctx::NSTAGE);
    if (c->push_used[ps.ev_i]) HIPCHK(c, hipEventSynchronize(c->push_ev[ps.ev_i]));       // 64 pushes ago: long done
    // How the bytes cross PCIe: a DMA (hipMemcpyAsync on a copy-only stream) into a device staging entry, issued NOW -- the ctx
    // stream is usually a model step behind the host, so the copy is long done when the front-end / ring-append kernel gets
    // there, and that kernel reads HBM.  (Round 2 let the kernel read the pinned memory itself, 16 bytes per lane over PCIe:
    // ~20 us of the critical stream per 328 KB chunk batch, 43 k against 51 k audio-s/s resident.)
    const float* host_src = nullptr;
    if (flags & LASR_PUSH_PINNED_NOCOPY) {
        const void* pinned = pinned_host_dev_ptr(pcm);
        if (!pinned) return fail(c, LASR_EINVAL, "LASR_PUSH_PINNED_NOCOPY: the buffer is not pinned (device-mapped) host memory");
        ps.src = (const float*)pinned;
        host_src = pcm;
    } else {
        if (!c->push_stage_host) {
            HIPCHK(c, hipHostMalloc((void**)&c->push_stage_host, sizeof(float) * (size_t)lasr_ctx::NSTAGE * c->M * CH));
            void* dp = nullptr;
            HIPCHK(c, hipHostGetDevicePointer(&dp, c->push_stage_host, 0));
            c->push_stage_host_dev = (float*)dp;
        }
        staged_copy(c, c->push_stage_host + (size_t)ps.ev_i * c->M * CH, pcm, sizeof(float) * (size_t)n * CH, rows, sizeof(float) * (size_t)CH);
        ps.src = c->push_stage_host_dev + (size_t)ps.ev_i * c->M * CH;
        host_src = c->push_stage_host + (size_t)ps.ev_i * c->M * CH;
    }
    {
        if (!c->push_stage_dev) {
            RC(dalloc(c, &c->push_stage_dev, (size_t)lasr_ctx::NSTAGE * c->M * CH));
            HIPCHK(c, hipStreamCreateWithFlags(&c->stream_copy, hipStreamNonBlocking));
            for (auto& e : c->push_copied) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        // (the device entry was last read by the kernel of push serial - NSTAGE: waited for above through push_ev)
        float* dst = c->push_stage_dev + (size_t)ps.ev_i * c->M * CH;
        HIPCHK(c, hipMemcpyAsync(dst, host_src, sizeof(float) * (size_t)n * CH, hipMemcpyHostToDevice, c->stream_copy));
        HIPCHK(c, hipEventRecord(c->push_copied[ps.ev_i], c->stream_copy));
        // (the ctx stream waits for the copy where the first kernel that reads `dst` is launched: wait_copied.  A chunk whose append is
        //  deferred is read by the NEXT call's front-end launch, behind that call's own copy on the same in-order copy stream: one
        //  cross-stream wait per model step instead of one per chunk)
        ps.src = dst;
        ps.dma = true;
    }
    return LASR_OK;
}
// the ctx stream waits until the DMA of host push `ev_i` has landed in its device staging entry
static int wait_copied(lasr_ctx* c, const PushSrc& ps) {
    if (ps.dma && ps.ev_i >= 0) HIPCHK(c, hipStreamWaitEvent(c->stream, c->push_copied[ps.ev_i], 0));
    return LASR_OK;
}
// the plain ring append (one launch)
static int push_append_launch(lasr_ctx* c, const int* slots, int n, const PushSrc& ps) {
    const int CH = c->d.chunk;
    RC(wait_copied(c, ps));
    if (c->M <= 512) {      // slot -> staging-row map by value: no command-block copy for a push
        PushIdx pi;
        for (int r = 0; r < 512; ++r) pi.idx[r] = -1;
        for (int i = 0; i < n; ++i) pi.idx[slots[i]] = (short)i;
        hipLaunchKernelGGL(k_push_pcm, dim3(c->M), dim3(256), 0, c->stream, ps.src, (const int*)nullptr, pi, c->win, c->ring_pos, CH, c->ring_chunks);
    } else {
        RC(cmd_begin(c));
        for (int r = 0; r < c->M; ++r) c->hc.src_idx[r] = -1;
        for (int i = 0; i < n; ++i) c->hc.src_idx[slots[i]] = i;
        RC(cmd_commit(c));
        PushIdx pi;
        hipLaunchKernelGGL(k_push_pcm, dim3(c->M), dim3(256), 0, c->stream, ps.src, (const int*)c->dc.src_idx, pi, c->win, c->ring_pos, CH, c->ring_chunks);
    }
    return LASR_OK;
}
// the "source consumed" event of a host push whose append was deferred: recorded behind the launch that finally read it
static int lazy_consumed(lasr_ctx* c) {
    if (c->lazy.ev_i >= 0) {
        HIPCHK(c, hipEventRecord(c->push_ev[c->lazy.ev_i], c->stream));
        c->push_used[c->lazy.ev_i] = true;
        c->push_dma[c->lazy.ev_i] = c->lazy.dma;
    }
    c->lazy = lasr_ctx::LazyPush{};
    return LASR_OK;
}
// a deferred append that the next call cannot take along: the plain append launch, now (see lasr_ctx::LazyPush)
static int flush_lazy(lasr_ctx* c) {
    if (!c->lazy.on) return LASR_OK;
    HIPCHK(c, hipSetDevice(c->device));
    PushSrc ps;
    ps.src = c->lazy.src; ps.ev_i = c->lazy.ev_i; ps.dma = c->lazy.dma;
    const std::vector<int> slots = c->lazy.slots;
    RC(push_append_launch(c, slots.data(), (int)slots.size(), ps));
    c->lazy_flushed++;
    return lazy_consumed(c);
}
// after the launch that reads ps.src: the host mirrors and the "source consumed" event (defer_event: the append was deferred)
static int push_finish(lasr_ctx* c, const int* slots, int n, const PushSrc& ps, long long* ticket, bool counted = false, bool defer_event = false) {
    for (int i = 0; i < n; ++i) {
        if (!counted) c->n_chunks[slots[i]]++;
        c->h_ring_pos[slots[i]] = (c->h_ring_pos[slots[i]] + 1) % c->ring_chunks;
    }
    if (ps.ev_i >= 0) {
        c->push_dma[ps.ev_i] = ps.dma;
        if (!defer_event) {
            HIPCHK(c, hipEventRecord(c->push_ev[ps.ev_i], c->stream));
            c->push_used[ps.ev_i] = true;
        }
    }
    if (ticket) *ticket = ps.ticket;
    return LASR_OK;
}

int lasr_push_pcm_ex(lasr_ctx* c, const int* slots, int n, const float* pcm, int flags, long long* ticket) {
    if (!c) return LASR_EINVAL;
    if (ticket) *ticket = -1;
    RC(check_slots(c, slots, n, true));
    if (n == 0) return LASR_OK;
    if (!pcm) return fail(c, LASR_EINVAL, "pcm is null");
    HIPCHK(c, hipSetDevice(c->device));
    RC(flush_lazy(c));
    if (!c->pending.empty()) RC(cont_pump(c, c->kick_n));      // steps in flight: keep the decode loop fed
    PushSrc ps;
    RC(push_prepare(c, slots, n, pcm, flags, ps));
    tr_mark(c, 1, c->stream);
    if (c->fe_fused) RC(materialize_pending(c, slots, n));      // irregular clients only: see there
    RC(push_append_launch(c, slots, n, ps));
    return push_finish(c, slots, n, ps, ticket);
}
int lasr_push_pcm(lasr_ctx* c, const int* slots, int n, const float* pcm) { return lasr_push_pcm_ex(c, slots, n, pcm, 0, nullptr); }

// 1: the source of the push that returned `ticket` has been read (a LASR_PUSH_PINNED_NOCOPY buffer may be reused); 0: not yet
int lasr_push_consumed(lasr_ctx* c, long long ticket) {
    if (!c) return LASR_EINVAL;
    if (ticket < 0 || ticket >= c->push_serial) return fail(c, LASR_EINVAL, "unknown push ticket %lld", ticket);
    if (c->push_serial - ticket > lasr_ctx::NSTAGE) return 1;        // its event slot has been waited for and reused since
    HIPCHK(c, hipSetDevice(c->device));
    // (DMA path: the source has been read once the copy is done; zero-copy path: once the reading kernel is)
    hipEvent_t ev = c->push_dma[ticket % lasr_ctx::NSTAGE] ? c->push_copied[ticket % lasr_ctx::NSTAGE] : c->push_ev[ticket % lasr_ctx::NSTAGE];
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return 1;
    (void)hipGetLastError();
    if (e == hipErrorNotReady) return 0;
    return fail(c, LASR_EHIP, "hipEventQuery failed: %s", hipGetErrorString(e));
}

// front-end of one client chunk for the listed slots and, for the slots whose frame buffer filled up,
// LayerNorm + encoder + encoder half of the joint -- all enqueued on c->stream, nothing synchronises
// fused (lasr_push_submit): the listed slots' newest chunk is still in the caller's buffer `fused->src` (row i of it belongs to
// slots[i]); the front-end launch appends it to the PCM ring itself.  *fused_done tells the caller whether that happened.
static int enqueue_frontend_encoder(lasr_ctx* c, const int* slots, int n, std::vector<int>& model_rows, int& Tm,
                                    const PushSrc* fused = nullptr, bool* fused_done = nullptr) {
    RoctxRange roctx_range_("lasr frontend+encoder");
    const lasr_model_desc& d = c->d;
    // window geometry (api-server.py:95-102 + TransformTime + StreamPostprocess)
    const long long N = (long long)d.n_window * d.chunk;
    int nf = 0;
    const int a0 = stream_frame0(c, &nf);
    if (nf < d.n_stack) return fail(c, LASR_EINVAL, "chunk of %d samples is too short: window yields %d < n_stack frames", d.chunk, nf);
    if (N <= d.n_fft / 2) return fail(c, LASR_EINVAL, "window shorter than the reflect padding");
    RC(cmd_begin(c));
    model_rows.clear();
    Tm = d.n_buffer;
    if (c->fe_fused) {
        // nothing runs on a chunk that does not complete a model step: the frame is only noted (chunk serial); the step's
        // single launch works through the last n_buffer windows of every model row
        for (int i = 0; i < n; ++i) {
            const int s = slots[i];
            if (c->n_chunks[s] < d.n_window) continue;          // window not full: the servicer does not call the pipeline
            const size_t q = (size_t)s * d.n_buffer + c->n_pend[s];
            c->pend_serial[q] = c->n_chunks[s]; c->pend_mat[q] = 0;
            if (++c->n_pend[s] == d.n_buffer) {                  // Buffer.encodes: emit when n_buffer collected
                c->n_pend[s] = 0;
                c->hc.T_row[s] = d.n_buffer;
                model_rows.push_back(s);
            }
        }
        rec(c, 0);
        if (model_rows.empty()) return LASR_OK;
        RC(ensure_T(c, Tm));
        int* enc_frames = nullptr; int* enc_base = nullptr;
        if (c->pe == c->pe_ring) { enc_frames = c->c_enc_frames; enc_base = c->c_enc_base; }
        // per (t', row): chunks pushed since the window of stacked frame t' was current; 255: the frame is already in pend
        std::vector<unsigned char> age_v((size_t)d.n_buffer * c->M, 0);
        for (int s : model_rows) {
            for (int j = 0; j < d.n_buffer; ++j) {
                const size_t q = (size_t)s * d.n_buffer + j;
                const int age = c->n_chunks[s] - c->pend_serial[q];
                if (!c->pend_mat[q] && age > c->ring_chunks - d.n_window)
                    return fail(c, LASR_ESTATE, "slot %d: the PCM ring no longer holds the window of pending frame %d", s, j);
                age_v[(size_t)j * c->M + s] = c->pend_mat[q] ? 255 : (unsigned char)age;
            }
        }
        {
            // log-mel halves (+ the ring append of the newest chunk when fused) on 2 x n_buffer x rows workgroups, then stack + LayerNorm
            FeMelArgs m{};
            m.window = c->window; m.tw512 = c->tw512; m.tw1024 = c->tw1024; m.fb_start = c->fb_start; m.fb_off = c->fb_off; m.fb_w = c->fb_w;
            m.n_mels = d.n_mels; m.hop = d.hop; m.fb_nnz = c->fb_nnz; m.win_off = (d.n_fft - d.win) / 2; m.win_len = d.win;
            m.pcm = c->win; m.ring_pos = c->ring_pos; m.chunk = d.chunk; m.n_window = d.n_window; m.ring_chunks = c->ring_chunks; m.frame0 = a0;
            m.pend = c->pend; m.pend_frames = d.n_buffer * d.n_stack;
            int* trow_home = (c->pe == c->pe_ring) ? c->T_row_main : nullptr;      // pipelined: one fixed buffer (see commit_T_rows)
            m.trow_out = trow_home ? trow_home : c->dc.T_row; m.enc_frames = enc_frames; m.enc_base = enc_base;
            m.src = fused ? fused->src : nullptr;
            const bool with_lazy = fused && c->lazy.on;          // (push_submit_impl: the deferred chunk belongs to exactly `slots`)
            m.src2 = with_lazy ? c->lazy.src : nullptr;
            for (int r = 0; r < 512; ++r) { m.idx[r] = -1; m.tp_pk[r] = 0; m.age_pk[r] = 0; }
            for (int r = 0; r < c->M; ++r) m.tp_pk[r] = (unsigned char)(c->h_ring_pos[r] << 4);
            if (fused)
                for (int i = 0; i < n; ++i) {
                    m.idx[slots[i]] = (short)i;
                    // (the host mirror already counts the deferred chunk; the ring on the device does not hold it yet)
                    if (with_lazy) m.tp_pk[slots[i]] = (unsigned char)(((c->h_ring_pos[slots[i]] - 1 + c->ring_chunks) % c->ring_chunks) << 4);
                }
            for (int s : model_rows) {
                m.tp_pk[s] |= (unsigned char)d.n_buffer;
                unsigned pk = 0;
                for (int j = 0; j < d.n_buffer; ++j) { const unsigned a8 = age_v[(size_t)j * c->M + s]; pk |= (a8 == 255 ? 15u : a8) << (4 * j); }
                m.age_pk[s] = (unsigned short)pk;
            }
            hipStream_t fe_st = c->stream;
            if (fused) {
                RC(wait_copied(c, *fused));
                if (with_lazy && c->lazy.dma && !fused->dma) { PushSrc lz; lz.ev_i = c->lazy.ev_i; lz.dma = true; RC(wait_copied(c, lz)); }
            }
            // c->fe_lds_pad bytes of unused dynamic LDS: the workgroup then shares its CU with no workgroup of the wide decode tilings
            // (see lasr_ctx::fe_lds_pad)
            hipLaunchKernelGGL((k_fe_mel<10>), dim3(2 * d.n_buffer, c->M), dim3(320), c->fe_lds_pad, fe_st, m);
            if (fused_done) *fused_done = fused != nullptr;
            if (with_lazy) { c->lazy_taken++; RC(lazy_consumed(c)); }
            RC(commit_T_rows(c, Tm, /*fixed_copy=*/c->pe != c->pe_ring, trow_home));    // the continuous loop reads its own frame counters
            StackLnArgs a{};
            a.src = c->pend; a.mode = 0; a.src_frames = d.n_buffer * d.n_stack; a.frame_step = d.n_stack; a.row_off = nullptr;
            a.T_row = c->T_row_dev; a.ln_w = c->ln_w; a.ln_b = c->ln_b; a.x0 = c->x0; a.F = d.feat; a.n_mels = d.n_mels;
            a.n_stack = d.n_stack; a.M = c->M; a.MT = c->MT; a.mt_total = c->Tcap * c->MT; a.feats_out = nullptr; a.bf = c->bf; a.Tmax = Tm;
            if (d.feat == 1280 && d.n_stack == 10 && d.n_mels == 128) {
                LnTileArgs t{};
                t.pend = c->pend; t.pend_frames = d.n_buffer * d.n_stack; t.T_row = c->T_row_dev; t.ln_w = c->ln_w; t.ln_b = c->ln_b;
                t.x0 = c->x0; t.MT = c->MT; t.mt_total = c->Tcap * c->MT; t.bf = c->bf;
                static bool attr = false;
                if (!attr) { (void)hipFuncSetAttribute((const void*)k_ln_tile, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); attr = true; }
                // store phase of the tile kernel on 4 z-slices (8 -> 32 workgroups; bit-identical): f32 52.5-52.7 -> 52.6-53.3 k, bf16
                // 92.8 -> 95.4 k (profiles/r04/r04_lnz_ab.txt)
                constexpr int ln_z = 4;
                // (with wide decode tilings around: 98 304 B instead of the 82 176 the tile needs -- the same CU exclusion as the log-mel
                //  launch's; this kernel reads its tile back with wide LDS reads as well and has never been seen wrong)
                hipLaunchKernelGGL(k_ln_tile, dim3(c->MT, Tm, ln_z), dim3(1024), c->fe_lds_pad ? 98304 : 16 * 1284 * 4, fe_st, t);
            } else {
                LAUNCH_STACK_LN( dim3((Tm + 3) / 4, c->M), dim3(256), 0, fe_st, a);
            }
        }
        rec(c, 1);
        run_encoder(c, Tm);
        rec(c, 2);
        return LASR_OK;
    }
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
    // <= 512 rows: the command (frame slot + frames of this model step per row) rides in the log-mel launch's
    // arguments; otherwise it goes through the command ring (one host->device copy)
    const bool by_value = any_feat && c->M <= 512 && d.n_buffer * d.n_stack < 32768 && d.n_buffer < 256;
    if (!by_value) RC(cmd_commit(c));
    rec(c, 0);
    if (any_feat) {
        MelArgs m{};
        fill_mel_args(c, m);
        m.pcm = c->win; m.N = N; m.stream = 1;
        m.ring_head = c->ring_pos; m.chunk = d.chunk; m.n_window = d.n_window; m.ring_chunks = c->ring_chunks; m.row_sel = c->dc.feat_sel; m.frame0 = a0;
        m.frames_per_row = d.n_stack; m.out = c->pend; m.out_frames = d.n_buffer * d.n_stack;
        m.row_N = nullptr; m.row_src_off = nullptr; m.row_frames = nullptr;
        if (by_value) {
            m.by_value = 1;
            m.trow_out = model_rows.empty() ? nullptr : c->dc.T_row;
            for (int r = 0; r < c->M; ++r) { m.sel_v[r] = (short)c->hc.feat_sel[r]; m.trow_v[r] = (unsigned char)c->hc.T_row[r]; }
        }
        hipLaunchKernelGGL(k_logmel, dim3((d.n_stack + 3) / 4, c->M), dim3(256), logmel_lds_pad(c), c->stream, m);
    }
    if (model_rows.empty()) return LASR_OK;
    RC(ensure_T(c, Tm));
    RC(commit_T_rows(c, Tm, /*fixed_copy=*/c->pe != c->pe_ring));    // the continuous loop reads its own frame counters
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
    RoctxRange roctx_range_("lasr_step_stream");
    if (n_ran) *n_ran = 0;
    RC(check_slots(c, slots, n, true));
    RC(flush_lazy(c));
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

static int cont_pump(lasr_ctx* c, int G);
static int pump_start(lasr_ctx* c);
static void pump_kick(lasr_ctx* c);

// Pipelined + continuous form of lasr_step_stream.  submit: front-end + encoder of this chunk on the
// main stream (the encoder half of the joint goes to a per-row frame ring).  ONE greedy loop runs on
// stream_dec across chunk boundaries, in groups of iterations (one hipGraph launch each); rows that are
// done early continue with the frames of the later, already encoded steps, so the latency-bound tail of
// a bursty stream overlaps useful work instead of idling 63 rows.  The last k_select of a group stores
// every row's frame cursor and the number of rows that still have encoded frames into pinned memory:
// the host derives "step j is decoded" for EVERY step in flight from the cursors (no per-step target on
// the device), so lasr_step_wait returns at once when the loop is ahead and the host can keep the main
// stream fed `depth` steps deep; groups are launched from push / submit / wait whenever none is in
// flight and frames are waiting (or about to be: a group may start with a stream-side wait for the next
// encoder, so decoding resumes without the host).
// Tokens are attributed to the step whose frames produced them: per-step results are identical to
// lasr_step_stream.
static int submit_impl(lasr_ctx* c, const int* slots, int n, const PushSrc* fused, bool* fused_done);
int lasr_step_submit(lasr_ctx* c, const int* slots, int n) {
    if (!c) return LASR_EINVAL;
    RC(check_slots(c, slots, n, true));
    RC(flush_lazy(c));
    return submit_impl(c, slots, n, nullptr, nullptr);
}

// lasr_push_pcm_ex + lasr_step_submit in one call: when the chunk completes a model step, the front-end launch reads the newest
// chunk straight from the source buffer and appends it to the PCM ring itself (one launch less per model step).
static int push_submit_impl(lasr_ctx* c, const int* slots, int n, const float* pcm, int flags, long long* ticket, const float* const* rows);
int lasr_push_submit(lasr_ctx* c, const int* slots, int n, const float* pcm, int flags, long long* ticket) {
    if (!c) return LASR_EINVAL;
    return push_submit_impl(c, slots, n, pcm, flags, ticket, nullptr);
}
// lasr_push_submit with the chunk of slots[i] at rows[i] (host memory, pageable or pinned; chunk floats each): what a server has
// when every connection's frame sits in its own receive buffer (api-server.py:88-91 tensorizes one message per stream) -- the
// gather into one batch is the copy into the staging ring that a host push makes anyway, not an extra pass on the caller's side.
int lasr_push_submit_rows(lasr_ctx* c, const int* slots, int n, const float* const* rows, long long* ticket) {
    if (!c) return LASR_EINVAL;
    if (n > 0 && !rows) return fail(c, LASR_EINVAL, "rows is null");
    for (int i = 0; i < n; ++i)
        if (!rows[i]) return fail(c, LASR_EINVAL, "rows[%d] is null", i);
    // (the runtime is asked about the first row only: a pointer query costs ~1 us, a server's rows all come from one allocator)
    if (n > 0 && is_device_ptr(rows[0])) return fail(c, LASR_EINVAL, "lasr_push_submit_rows takes host memory (rows[0] is a device pointer)");
    return push_submit_impl(c, slots, n, n > 0 ? rows[0] : nullptr, 0, ticket, rows);
}
static int push_submit_impl(lasr_ctx* c, const int* slots, int n, const float* pcm, int flags, long long* ticket, const float* const* rows) {
    RoctxRange roctx_range_("lasr_push_submit");
    if (ticket) *ticket = -1;
    RC(check_slots(c, slots, n, true));
    if (n == 0) return LASR_OK;
    if (!pcm) return fail(c, LASR_EINVAL, "pcm is null");
    // every check without side effects comes BEFORE the push: LASR_ESTATE / LASR_EINVAL from this call mean nothing was pushed
    // (what can still fail after the push is the HIP runtime: LASR_EHIP, with the chunk pushed)
    if ((int)c->pending.size() + 1 > lasr_max_inflight(c))
        return fail(c, LASR_ESTATE, "%d steps already in flight (limit %d): call lasr_step_wait (nothing was pushed)", (int)c->pending.size(), lasr_max_inflight(c));
    if (c->W > 1 && c->M > 512) return fail(c, LASR_ESTATE, "the pipelined protocol with beam > 1 takes up to 512 stream slots (nothing was pushed)");
    {
        int nf = 0;
        (void)stream_frame0(c, &nf);
        if (nf < c->d.n_stack || (long long)c->d.n_window * c->d.chunk <= c->d.n_fft / 2)
            return fail(c, LASR_EINVAL, "chunk of %d samples is too short for the streaming window (nothing was pushed)", c->d.chunk);
    }
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->pending.empty()) RC(cont_pump(c, c->kick_n));
    PushSrc ps;
    RC(push_prepare(c, slots, n, pcm, flags, ps, rows));
    tr_mark(c, 1, c->stream);
    if (c->fe_fused) RC(materialize_pending(c, slots, n));
    const bool can_fuse = c->fe_fused;
    // a deferred chunk rides in this call's front-end launch only when it belongs to exactly these slots, in this order
    if (c->lazy.on && !(can_fuse && (int)c->lazy.slots.size() == n && std::equal(slots, slots + n, c->lazy.slots.begin()))) RC(flush_lazy(c));
    if (!can_fuse) RC(push_append_launch(c, slots, n, ps));
    for (int i = 0; i < n; ++i) c->n_chunks[slots[i]]++;
    bool fused_done = false;
    int rc = submit_impl(c, slots, n, can_fuse ? &ps : nullptr, &fused_done);
    if (rc) {       // (argument errors were caught above: what can fail here is the runtime)
        (void)flush_lazy(c);
        if (can_fuse && !fused_done) (void)push_append_launch(c, slots, n, ps);
        (void)push_finish(c, slots, n, ps, ticket, true);
        return rc;
    }
    if (can_fuse && !fused_done) {
        // no model step from this chunk.  Its append waits for the next call's front-end launch when the source stays readable
        // until then (the engine's own device staging entry of a host push; a device buffer the caller declared stable);
        // otherwise the plain append launch, now.
        RC(flush_lazy(c));                       // (an older deferred chunk goes first: ring order)
        const bool stable = c->lazy_on && (ps.dma || (!rows && (flags & LASR_PUSH_DEVICE_STABLE) && ps.ev_i < 0));
        if (stable) {
            c->lazy.on = true; c->lazy.src = ps.src; c->lazy.slots.assign(slots, slots + n); c->lazy.ev_i = ps.ev_i; c->lazy.dma = ps.dma;
            return push_finish(c, slots, n, ps, ticket, true, /*defer_event=*/true);
        }
        RC(push_append_launch(c, slots, n, ps));
    }
    return push_finish(c, slots, n, ps, ticket, true);
}

static int submit_impl(lasr_ctx* c, const int* slots, int n, const PushSrc* fused, bool* fused_done) {
    if (c->W > 1 && c->M > 512) return fail(c, LASR_ESTATE, "the pipelined protocol with beam > 1 takes up to 512 stream slots");
    {   // in-flight limit: the event / T_row rings (NFLY) and the per-row rings the decode loop runs through --
        // encoder frames not yet decoded (pe ring), tokens not yet collected (token ring), step boundary marks
        const int inflight = (int)c->pending.size() + 1, Tm = c->d.n_buffer;
        if (inflight > lasr_ctx::NFLY - 1 || inflight > lasr_ctx::ENDSLOTS || inflight * Tm > lasr_ctx::RING ||
            inflight * Tm * c->d.max_iters_stream > lasr_ctx::TOKRING)
            return fail(c, LASR_ESTATE, "%d steps already in flight (limit %d for n_buffer %d, max_iters_stream %d): call lasr_step_wait",
                        (int)c->pending.size(), lasr_max_inflight(c), Tm, c->d.max_iters_stream);
    }
    HIPCHK(c, hipSetDevice(c->device));
    // keep the decode stream busy while the host enqueues (and the GPU runs) this chunk's encoder (a no-op with the pump thread)
    RC(cont_pump(c, c->kick_n));
    {   // the pump and every graph it can need BEFORE anything of this step is enqueued: a capture / instantiation failure then
        // leaves the host frame targets and the device frame counters in agreement
        std::lock_guard<std::mutex> lk(c->mu);
        if (!c->pump_on || c->cgraphs.empty()) RC(pump_start(c));
    }
    const int idx = (int)(c->model_steps % lasr_ctx::NFLY);
    float* pe_keep = c->pe;
    c->pe = c->pe_ring;                     // run_encoder writes the joint's encoder half into the ring
    std::vector<int> model_rows;
    int Tm = 0;
    const bool prof = c->profiling;
    c->profiling = false;
    int rc = enqueue_frontend_encoder(c, slots, n, model_rows, Tm, fused, fused_done);
    c->profiling = prof;
    c->pe = pe_keep;
    if (rc) return rc;
    if (model_rows.empty()) {
        HIPCHK(c, hipGetLastError());
        return LASR_OK;
    }
    if (!c->fe_fused)       // (the fused front-end launch has already advanced the frame counters)
        hipLaunchKernelGGL(k_advance, dim3(grid1(c->M)), dim3(256), 0, c->stream, c->c_enc_frames, (const int*)c->T_row_dev, c->M);
    HIPCHK(c, hipEventRecord(c->ev_enc[idx], c->stream));
    tr_mark(c, 5, c->stream);
    lasr_ctx::PendingStep p;
    p.rows = model_rows; p.Tm = Tm; p.idx = idx; p.admitted = false;
    p.T_row_ptr = c->T_row_dev;             // lives in the command ring until long after this step is collected
    p.serial = c->model_steps;
    p.target.assign(c->M, 0);
    for (int r : model_rows) { c->h_frames_sub[r] += Tm; p.target[r] = (int)c->h_frames_sub[r]; }
    bool kicked = false;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->pending.push_back(std::move(p));
        if (c->pump_on) { c->kick.fetch_add(1, std::memory_order_release); kicked = true; }     // (under c->mu: see pump_kick)
    }
    c->model_steps++;
    HIPCHK(c, hipGetLastError());
    if (kicked) { c->cv_pump.notify_one(); return LASR_OK; }
    // the enqueue above took tens of microseconds of host time: a group may have finished meanwhile.  If nothing is
    // left to decode the next group is queued behind this step's encoder event (stream-side wait)
    RC(cont_pump(c, c->kick_n));
    return LASR_OK;
}

// Non-consuming look at the submitted, uncollected model steps of one slot (greedy decode): how many there are, how many of
// them (oldest first) the decode loop has finished for this slot, and the tokens of those -- counts[k] tokens for step k,
// concatenated in `tokens`.  lasr_step_wait / lasr_fetch hand the same tokens out later, in step order, as if nobody had looked.
// What it is for: a scheduler that must judge a stream's step before it may submit the stream's next one (the servicer's
// reset rule, api-server.py:131-134) learns the verdict when the row is decoded, not `steps in flight` collections later.
int lasr_peek_slot(lasr_ctx* c, int slot, int32_t* tokens, int cap, int32_t* counts, int cap_steps, int* n_decoded, int* n_inflight) {
    if (!c || !n_decoded || !n_inflight) return LASR_EINVAL;
    *n_decoded = 0; *n_inflight = 0;
    if (slot < 0 || slot >= c->d.max_streams || !c->open_[slot]) return fail(c, LASR_ESTATE, "slot %d is not open", slot);
    if (c->W > 1) return fail(c, LASR_ESTATE, "lasr_peek_slot serves greedy decode (beam = 1)");
    const int M = c->M;
    const int* h_end = c->cont_host + 16 + (size_t)lasr_ctx::NFLY * M;
    const int* h_ring = h_end + (size_t)M * lasr_ctx::ENDSLOTS;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->pump_on) cont_poll(c);
    long long from = c->h_fetched[slot];
    int used = 0;
    bool open_run = true;
    for (const auto& P : c->pending) {
        if (std::find(P.rows.begin(), P.rows.end(), slot) == P.rows.end()) continue;
        const int k = (*n_inflight)++;
        if (!open_run || c->h_cur_seen[slot] < P.target[slot]) { open_run = false; continue; }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const int j = P.target[slot] / P.Tm - 1;
        const long long end = h_end[(size_t)slot * lasr_ctx::ENDSLOTS + (j % lasr_ctx::ENDSLOTS)];
        const int n = (int)(end - from);
        if (k >= cap_steps || used + n > cap) return fail(c, LASR_EFULL, "lasr_peek_slot: buffers too small");
        if (counts) counts[k] = n;
        for (long long q = from; q < end; ++q)
            if (tokens) tokens[used++] = h_ring[(size_t)slot * lasr_ctx::TOKRING + (q % lasr_ctx::TOKRING)];
        from = end;
        (*n_decoded)++;
    }
    return LASR_OK;
}

// lasr_peek_slot for n slots in one call: skip[i] oldest steps of slots[i] are of no interest (the caller has seen them);
// row i of tokens [n][cap] / counts [n][cap_steps] holds the decoded steps behind them; n_decoded[i] counts ALL decoded steps
// of the slot (the skipped ones included), n_inflight[i] its submitted steps.
int lasr_peek_many(lasr_ctx* c, const int* slots, int n, const int* skip, int32_t* tokens, int cap, int32_t* counts, int cap_steps,
                   int* n_decoded, int* n_inflight) {
    if (!c || !slots || !n_decoded || !n_inflight || n < 0) return LASR_EINVAL;
    if (c->W > 1) return fail(c, LASR_ESTATE, "lasr_peek_many serves greedy decode (beam = 1)");
    const int M = c->M;
    const int* h_end = c->cont_host + 16 + (size_t)lasr_ctx::NFLY * M;
    const int* h_ring = h_end + (size_t)M * lasr_ctx::ENDSLOTS;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->pump_on) cont_poll(c);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int i = 0; i < n; ++i) {
        const int slot = slots[i];
        n_decoded[i] = 0; n_inflight[i] = 0;
        if (slot < 0 || slot >= c->d.max_streams || !c->open_[slot]) return fail(c, LASR_ESTATE, "slot %d is not open", slot);
        long long from = c->h_fetched[slot];
        int used = 0, kept = 0;
        bool open_run = true;
        const int sk = skip ? skip[i] : 0;
        for (const auto& P : c->pending) {
            if (P.target[slot] == 0) continue;                       // the slot is not part of this step
            const int k = n_inflight[i]++;
            if (!open_run || c->h_cur_seen[slot] < P.target[slot]) { open_run = false; continue; }
            const int j = P.target[slot] / P.Tm - 1;
            const long long end = h_end[(size_t)slot * lasr_ctx::ENDSLOTS + (j % lasr_ctx::ENDSLOTS)];
            if (k >= sk) {
                const int cnt = (int)(end - from);
                if (kept >= cap_steps || used + cnt > cap) return fail(c, LASR_EFULL, "lasr_peek_many: buffers too small");
                if (counts) counts[(size_t)i * cap_steps + kept] = cnt;
                if (tokens)
                    for (long long q = from; q < end; ++q) tokens[(size_t)i * cap + used++] = h_ring[(size_t)slot * lasr_ctx::TOKRING + (q % lasr_ctx::TOKRING)];
                kept++;
            }
            from = end;
            n_decoded[i]++;
        }
    }
    return LASR_OK;
}

int lasr_step_pending(lasr_ctx* c) { return c ? (int)c->pending.size() : 0; }     // (only the API thread changes the size)

int lasr_max_inflight(const lasr_ctx* c) {
    if (!c) return 0;
    const int Tm = c->d.n_buffer;
    int n = std::min(lasr_ctx::NFLY - 1, lasr_ctx::ENDSLOTS);
    n = std::min(n, lasr_ctx::RING / Tm);
    n = std::min(n, lasr_ctx::TOKRING / (Tm * c->d.max_iters_stream));
    return std::max(n, 0);
}

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

// ---- the decode loop of the pipelined protocol: groups of G iterations on stream_dec --------------------------------
// Decode-side state (c->pending, h_avail, h_cur_seen, work_left, group_inflight, cont_iters, the predictor / LM parities, the
// beam's host trees) is guarded by c->mu.  Groups are launched by the NATIVE PUMP THREAD (pump_main) while steps are in flight:
// it spins on the pinned flag word and replays the next group's hipGraph the moment the previous one has published its cursors,
// whatever the API thread is doing (its push_submit spends ~60 us per chunk enqueueing the front-end and the encoder cells on
// the main stream; round 3 launched groups only from inside API calls, so a finished group waited for the host: 5 % of the
// decode stream).  The pump never uses the launch helpers (they take the stream and the pe buffer from the ctx, which the API
// thread is using for the main stream at that moment): its path is event waits, one admission kernel with explicit arguments
// and a graph replay; the graphs are captured by the API thread (ensure_group_graphs) before it hands the first step over.
// Without graphs (LASR_NO_GRAPH, LASR_DBG_TIMING), with more than 512 slots, or with LASR_PUMP=0 there is no pump thread and
// the API calls launch the groups themselves, as in round 3.

// decode-side kernel state of the continuous loop
static void cont_states(lasr_ctx* c, DecState& s, BeamState& bs) {
    const int M = c->M, V = c->d.vocab;
    s = c->ds;
    s.t_idx = c->c_cur; s.iters = c->c_iters; s.step_ntok = c->c_ntotal; s.step_tok = c->c_tok_ring;
    s.tok_cap = lasr_ctx::TOKRING; s.unfinished = c->c_behind; s.cont = 1; s.host_cur = c->c_hcur_dev;
    s.host_ntot = c->tr_on ? c->c_hcur_dev + M : nullptr;
    s.ntok_end = c->c_ntok_end; s.step_T = c->d.n_buffer; s.end_slots = lasr_ctx::ENDSLOTS; s.done_blocks = c->c_done;
    s.iter_ctr = c->c_iter;
    s.dbg = c->dbg ? c->dbg + ((size_t)4 * 4096 + 4095) * 16 : nullptr;
    bs = BeamState{};
    if (c->W > 1) {
        bs.W = c->W; bs.V = V; bs.blank = c->d.blank; bs.max_iters = c->d.max_iters_stream; bs.Md = c->Md;
        bs.t_idx = c->c_cur; bs.iters = c->c_iters; bs.T_row = c->c_avail;
        bs.score = c->b_score; bs.alive = c->b_alive; bs.inB = c->b_inB; bs.token = c->ds.token; bs.emit = c->ds.emit;
        bs.parent = c->b_parent; bs.trellis = c->b_tre_dev; bs.unfinished = c->c_behind; bs.dbg = nullptr;
        bs.cont = 1; bs.tring = lasr_ctx::TRING; bs.frame_done = c->b_fdone_dev; bs.iter_ctr = c->c_iter; bs.done_blocks = c->c_done;
        bs.host_cur = c->c_hcur_dev; bs.step_T = c->d.n_buffer; bs.end_slots = lasr_ctx::ENDSLOTS;
        bs.end_score = c->b_endsc_dev; bs.end_alive = c->b_endal_dev;
    }
}

// the G iterations of a group through the launch helpers (API thread only, c->mu held, inside a ContScope): launch-invariant
// (the flag-ring slot comes from a device counter, the last selection kernel publishes the cursors and the "rows with frames
// left" word), so a group is one hipGraph per (G, ping-pong parities)
static void cont_enqueue(lasr_ctx* c, int G) {
    const int M = c->M, V = c->d.vocab;
    DecState s; BeamState bs;
    cont_states(c, s, bs);
    c->dbg_gate = false;
    // LM branch (stream_lm): forked behind the selection kernel whose tokens it consumes, joined in front of the next one (which
    // reads its scores and rewrites token / emit) or at the end of the group -- the predictor, the joint half and the next
    // logits GEMM run beside it (captured: two branches of the group graph)
    const bool side = c->lm.on && c->stream_lm != nullptr;
    // pair launches (LASR_LM_PAIR, greedy, fp32 / bf16 LM of >= 3 layers beside a 2 x NBRC predictor -- configs[1] with the
    // reference's LM; other shapes keep the LM step in line)
    static const int pair_env = getenv("LASR_LM_PAIR") ? atoi(getenv("LASR_LM_PAIR")) : 1;
    const bool pair = pair_env && !side && c->W == 1 && c->lm.on && !c->lm.q8 && c->d.pred_cell == 0 && c->d.pred_layers == 2 && c->lm.L >= 3;
    bool lm_tail = false;
    bool lm_open = false;
    auto lm_join = [&]() {
        if (lm_open) (void)hipStreamWaitEvent(c->stream, c->ev_lm_join, 0);
        lm_open = false;
    };
    auto lm_step = [&](bool beam) {
        if (!side) { launch_lm(c, beam); return; }
        hipStream_t keep = c->stream;
        (void)hipEventRecord(c->ev_lm_fork, keep);
        (void)hipStreamWaitEvent(c->stream_lm, c->ev_lm_fork, 0);
        c->stream = c->stream_lm;
        launch_lm(c, beam);
        c->stream = keep;
        (void)hipEventRecord(c->ev_lm_join, c->stream_lm);
        lm_open = true;
    };
    for (int q = 0; q < G; ++q) {
        if (c->W > 1) {                 // one selection round: logits of every hypothesis slot -> ordered top-W -> predictor / joint
            bs.host_flag = (q == G - 1) ? c->c_flag_dev : nullptr;
            launch_logits(c, c->logits, c->Md, true);
            lm_join();
            launch_beam_select(c, bs, 0);
            if (side) lm_step(true);
            launch_predictor(c, true);
            launch_ppj(c, true);
            if (!side) lm_step(true);
            continue;
        }
        s.host_flag = (q == G - 1) ? c->c_flag_dev : nullptr;
        if (pair) {
            // LM step and predictor / joint chain of an iteration as PAIR launches (k_gemm2): LM layer l beside stage l of the chain;
            // the step's last layers run beside the NEXT iteration's logits GEMM (or alone, at the end of the group)
            lasr_ctx::Captured A, B;
            auto rec = [&](lasr_ctx::Captured& k, auto&& fn) { c->cap = &k; fn(); c->cap = nullptr; };
            if (lm_tail) {
                rec(A, [&] { launch_logits(c, c->logits, c->la * M, true); });
                rec(B, [&] { launch_lm(c, false, 3, 4, false); });
                launch_pair(c, 3, false, A, B);
                launch_lm(c, false, 4, -1);                       // deeper layers (if any), output layer, k_lm_post, parity
                lm_tail = false;
            } else {
                launch_logits(c, c->logits, c->la * M, true);
            }
            launch_select<false>(c->stream, M, c->logits, V, c->d.blank, c->d.max_iters_stream, c->c_avail, s, 0, nullptr, nullptr, c->la, M);
            rec(A, [&] { launch_predictor(c, false, 0, 1); });
            rec(B, [&] { launch_lm(c, false, 0, 1, false); });
            launch_pair(c, 0, true, A, B);
            rec(A, [&] { launch_predictor(c, false, 1, 2); });
            rec(B, [&] { launch_lm(c, false, 1, 2, false); });
            launch_pair(c, 1, false, A, B);
            rec(A, [&] { launch_ppj(c); });
            rec(B, [&] { launch_lm(c, false, 2, 3, false); });
            launch_pair(c, 2, false, A, B);
            if (c->lm.L > 3) lm_tail = true;
            else launch_lm(c, false, 3, -1);                      // (3 layers: only the output layer is left)
            if (q == G - 1 && lm_tail) { launch_lm(c, false, 3, -1); lm_tail = false; }
            continue;
        }
        launch_logits(c, c->logits, c->la * M, true);
        lm_join();
        launch_select<false>(c->stream, M, c->logits, V, c->d.blank, c->d.max_iters_stream, c->c_avail, s, 0, nullptr, nullptr, c->la, M);
        if (side) lm_step(false);
        launch_predictor(c);
        launch_ppj(c);
        if (!side) lm_step(false);
    }
    lm_join();
}

// the group graph for (G, current parities): captured on first use (API thread, c->mu held)
static int cont_group_graph(lasr_ctx* c, int G, int pred_par, int lm_par, hipGraphExec_t* out) {
    const auto key = std::make_tuple(G, pred_par, lm_par);
    auto it = c->cgraphs.find(key);
    if (it == c->cgraphs.end()) {
        const int pp0 = c->pred_par, lp0 = c->lm.par;
        c->la = c->la_stream;
        ContScope scope(c);
        c->pred_par = pred_par; c->lm.par = lm_par;
        hipGraph_t gr = nullptr;
        hipGraphExec_t ex = nullptr;
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        cont_enqueue(c, G);
        hipError_t e = hipStreamEndCapture(c->stream, &gr);
        c->pred_par = pp0; c->lm.par = lp0;            // the capture only recorded; parities advance at launch
        if (e != hipSuccess || !gr) return fail(c, LASR_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
        (void)hipGraphDestroy(gr);
        if (e != hipSuccess) return fail(c, LASR_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
        it = c->cgraphs.emplace(key, ex).first;
    }
    *out = it->second;
    return LASR_OK;
}
// every graph the pump thread can need (it never captures): G = pump_G at each parity the loop can be in
static int ensure_group_graphs(lasr_ctx* c) {
    hipGraphExec_t ex;
    for (int pp = 0; pp < 2; ++pp)
        for (int lp = 0; lp < (c->lm.on ? 2 : 1); ++lp)
            RC(cont_group_graph(c, c->pump_G, pp, c->lm.on ? lp : c->lm.par, &ex));
    return LASR_OK;
}

// One group of G iterations on stream_dec for whatever rows have frames to decode; its last selection kernel publishes the
// rows' frame cursors and the "rows with frames left" word to pinned memory.  Does not wait.  c->mu held.  from_pump: called
// by the pump thread -- explicit stream / buffers only, never a capture (rc 1 = the graph is missing, nothing was launched)
static int cont_launch_group(lasr_ctx* c, int G, bool from_pump = false) {
    const int M = c->M, J = c->d.joint;
    hipStream_t sd = c->stream_dec;
    float* pe = c->pe_ring;
    const int R = lasr_ctx::RING, la = c->W > 1 ? 1 : c->la_stream;
    int* flag = c->cont_host;
    hipGraphExec_t ex = nullptr;
    const bool graphs = c->use_graphs && !c->dbg;
    if (graphs) {
        if (from_pump) {
            auto it = c->cgraphs.find(std::make_tuple(G, c->pred_par, c->lm.par));
            if (it == c->cgraphs.end()) return 1;
            ex = it->second;
        } else {
            RC(cont_group_graph(c, G, c->pred_par, c->lm.par, &ex));
        }
    } else if (from_pump) {
        return 1;
    }
    // admit encoded steps in order.  While rows still have frames to decode only steps whose encoder has finished are
    // admitted (the loop must not stall behind an encoder); when nothing is left the first one is admitted unconditionally:
    // the decode stream then waits for that encoder on the GPU and resumes by itself
    bool admitted_any = false;
    tr_mark(c, 10, sd);
    const bool by_value = M <= 512;
    for (auto& q : c->pending) {
        if (q.admitted) continue;
        const bool must = c->work_left == 0 && !admitted_any;
        if (!must && hipEventQuery(c->ev_enc[q.idx]) != hipSuccess) { (void)hipGetLastError(); break; }
        HIPCHK(c, hipStreamWaitEvent(sd, c->ev_enc[q.idx], 0));
        if (by_value) { for (int r : q.rows) c->h_avail[r] = q.target[r]; }
        else hipLaunchKernelGGL(k_advance, dim3(grid1(M)), dim3(256), 0, sd, c->c_avail, q.T_row_ptr, M);
        q.admitted = true;
        admitted_any = true;
    }
    if (admitted_any && c->W > 1) {     // beam: every hypothesis slot of the streams that have a frame to decode
        AvailV av;
        for (int r = 0; r < 512; ++r) av.v[r] = r < M ? c->h_avail[r] : 0;
        hipLaunchKernelGGL(k_ja_admit_beam, dim3(grid1((size_t)c->Md * J)), dim3(256), 0, sd, (const float*)pe, (const float*)cur_pp(c),
                           (const int*)c->c_cur, av, c->c_avail, c->ja, J, c->Md, c->W, M, c->MTj, R, c->bf);
    } else
    if (admitted_any) {  // rows that were idle need their joint activation for the new frames
        if (by_value) {
            AvailV av;
            for (int r = 0; r < 512; ++r) av.v[r] = r < M ? c->h_avail[r] : 0;
            hipLaunchKernelGGL(k_ja_admit, dim3(grid1((size_t)M * J)), dim3(256), 0, sd, (const float*)pe, (const float*)c->pp,
                               (const int*)c->c_cur, av, c->c_avail, c->ja, J, M, c->MTj, R, c->bf, la);
        } else {
            hipLaunchKernelGGL(k_ja, dim3(grid1((size_t)M * J)), dim3(256), 0, sd, pe, c->pp, c->c_cur,
                               c->c_avail, c->ja, J, M, c->MTj, R, c->bf, 1, M, la);
        }
    }
    tr_mark(c, 11 + 100 * G + (admitted_any ? 1000 : 0), sd);
    __atomic_store_n(flag, -1, __ATOMIC_RELEASE);      // before the launch that will overwrite it
    if (graphs) {
        { RoctxRange roctx_range_("lasr decode group"); HIPCHK(c, hipGraphLaunch(ex, sd)); }
        if (G & 1) { c->pred_par ^= 1; if (c->lm.on) c->lm.par ^= 1; }
    } else {
        c->la = c->la_stream;
        ContScope scope(c);
        cont_enqueue(c, G);
    }
    tr_mark(c, 12, sd);
    c->tr_last_G = G;
    c->cont_iters += G;
    c->group_inflight = true;
    c->dec_tail_open = true;
    HIPCHK(c, hipGetLastError());
    return LASR_OK;
}

// continuous beam loop: replay the rounds of the groups that have completed into the streams' hypothesis trees; a stream that
// finished a model step in a round gets that step's result (best alive slot by the scores the kernel stored for the step)
static void beam_replay(lasr_ctx* c) {
    const int W = c->W, Md = c->Md, M = c->M, Tm = c->d.n_buffer;
    for (long long it = c->b_rounds_replayed; it < c->cont_iters; ++it) {
        const int slot = (int)(it % lasr_ctx::TRING);
        const int* tre = c->b_tre_host + (size_t)slot * Md;
        const int* fd = c->b_fdone_host + (size_t)slot * M;
        for (int q = 0; q < c->d.max_streams; ++q) {
            const int* e = tre + (size_t)q * W;
            if (e[0] == -1) continue;                          // stream idle in this round
            auto& H = c->bh[q];
            bh_apply(H, e, W);
            if (!fd[q]) continue;
            const long long frames = ++c->b_frames_done[q];
            if (frames % Tm) continue;
            const int es = (int)((frames / Tm - 1) % lasr_ctx::ENDSLOTS);
            const double* sc = c->b_endsc_host + ((size_t)q * lasr_ctx::ENDSLOTS + es) * W;
            const int am = c->b_endal_host[(size_t)q * lasr_ctx::ENDSLOTS + es];
            int best = -1;
            for (int j = 0; j < W; ++j)
                if (((am >> j) & 1) && (best < 0 || sc[j] > sc[best])) best = j;
            lasr_ctx::BeamResult r;
            r.tokens = c->committed[q];
            r.score = c->committed_score[q];
            if (best >= 0) { bh_tokens(H, H.cur[best], r.tokens); r.score += sc[best]; }
            c->b_results[q].push_back(std::move(r));
        }
    }
    c->b_rounds_replayed = c->cont_iters;
}

// non-blocking: if the in-flight group has finished, consume its flag and snapshot the rows' frame cursors
// (nothing writes them again until the next group is launched).  c->mu held.
static void cont_poll(lasr_ctx* c) {
    if (!c->group_inflight) return;
    const int v = __atomic_load_n((volatile int*)c->cont_host, __ATOMIC_ACQUIRE);
    if (v == -1) return;
    c->group_inflight = false;
    c->work_left = v;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    memcpy(c->h_cur_seen.data(), c->cont_host + 16, sizeof(int) * c->M);
    if (c->W > 1) beam_replay(c);
    if (c->tr_on) {     // iterations this group needed = most decisions (frames + tokens) any row made in it; rows that moved
        const int* nt = c->cont_host + 16 + c->M;
        int need = 0, rows = 0;
        for (int r = 0; r < c->M; ++r) {
            const int dlt = (c->h_cur_seen[r] - c->tr_prev_cur[r]) + (nt[r] - c->tr_prev_ntot[r]);
            need = std::max(need, dlt);
            rows += dlt > 0;
            c->tr_prev_cur[r] = c->h_cur_seen[r]; c->tr_prev_ntot[r] = nt[r];
        }
        tr_note(c, 20, need * 1000.0 + c->tr_last_G * 100.0 + v + rows / 1000.0);
    }
    c->progress.fetch_add(1, std::memory_order_release);
    c->cv_prog.notify_all();           // (c->mu held: a lasr_step_wait that sleeps on the counter checks it under the same mutex)
}

// non-blocking: launch the next group of G iterations if none is in flight and frames are waiting (or the encoder
// of a submitted step is still to be admitted).  c->mu held.
static int cont_pump_locked(lasr_ctx* c, int G, bool from_pump = false) {
    cont_poll(c);
    if (c->group_inflight || c->pending.empty()) return LASR_OK;
    bool unadmitted = false;
    for (const auto& q : c->pending) unadmitted |= !q.admitted;
    if (c->work_left == 0 && !unadmitted) return LASR_OK;        // everything encoded so far is decoded
    return cont_launch_group(c, G, from_pump);
}
// API-thread form: a no-op while the pump thread owns the launches
static int cont_pump(lasr_ctx* c, int G) {
    if (c->pump_on) return LASR_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    return cont_pump_locked(c, G);
}

// ---- the pump thread
static void pump_main(lasr_ctx* c) {
    (void)hipSetDevice(c->device);
    tl_err_sink = &c->pump_err;             // fail() on this thread writes the pump's own buffer (under c->mu), never c->err
    long long seen_kick = -1;
    // LASR_PUMP_NAP_PCT > 0: the pump sleeps through that share of a group's expected duration (moving average of the last groups)
    // before it starts to spin on the group's flag -- a group of 2 iterations runs ~170 us, and polling it from the first
    // microsecond keeps one host core per context at 100 %.  The timer slack of this thread goes down to 1 us for that.
    double ema_us = 0.0;
    if (c->pump_nap_pct > 0) (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
    for (;;) {
        bool inflight = false, idle = false;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            if (c->pump_stop.load(std::memory_order_acquire)) return;
            if (c->pump_on && c->pump_rc == 0) {
                const int rc = cont_pump_locked(c, c->pump_G, true);
                if (rc < 0) { c->pump_rc = rc; c->progress.fetch_add(1, std::memory_order_release); c->cv_prog.notify_all(); }
                // (rc 1: the graph for this parity is missing -- the API thread captures it with its next call)
            }
            inflight = c->group_inflight;
            idle = !inflight;
        }
        if (inflight) {
            // the group's flag: a plain spin (the pump is the only thing this thread does); a submit in the meantime changes nothing
            // before the group has finished
            const auto t_launch = std::chrono::steady_clock::now();
            if (c->pump_nap_pct > 0 && ema_us > 20.0) {
                const long ns = (long)(ema_us * 10.0 * c->pump_nap_pct);        // us * 1000 * pct / 100
                struct timespec ts{0, std::min(ns, 2000000L)};
                (void)nanosleep(&ts, nullptr);
            }
            unsigned long long spins = 0;
            while (__atomic_load_n((volatile int*)c->cont_host, __ATOMIC_ACQUIRE) == -1 && !c->pump_stop.load(std::memory_order_relaxed)) {
                __builtin_ia32_pause();
                if (++spins > (1ull << 31)) break;
            }
            if (c->pump_nap_pct > 0) {
                // (a group found finished right after the nap may have overslept: its duration is an upper bound, and the
                //  average then shrinks the next nap only through the shorter groups -- so count such a group at 80 %)
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_launch).count();
                if (spins < 4) us *= 0.8;
                ema_us = ema_us == 0.0 ? us : 0.875 * ema_us + 0.125 * us;
            }
            continue;
        }
        if (idle) {
            // nothing to launch: wait for the next submitted step.  Spin briefly (steps come every ~200 us under load), then PARK on
            // the condition variable: the kick counter moves under c->mu, so a kick cannot fall between the check and the sleep, and
            // an idle engine costs no CPU (the 2 ms timed wait of round 4 re-ran the 1-2 ms spin phase after every
            // time-out -- 30-50 % of a core per idle context).  The 250 ms time-out is a backstop only (lasr_step_wait kicks too).
            bool parked = false;
            for (int spins = 0; c->kick.load(std::memory_order_acquire) == seen_kick && !c->pump_stop.load(std::memory_order_relaxed); ++spins) {
                if (!parked && spins < 40000) { __builtin_ia32_pause(); continue; }
                std::unique_lock<std::mutex> lk(c->mu);
                c->cv_pump.wait_for(lk, std::chrono::milliseconds(250), [&] { return c->kick.load() != seen_kick || c->pump_stop.load(); });
                parked = true;               // (a time-out goes straight back to sleep: no second spin phase)
            }
            seen_kick = c->kick.load(std::memory_order_acquire);
        }
    }
}
// called by the API thread (c->mu held) when the first step is handed to the decode loop
static int pump_start(lasr_ctx* c) {
    static const int pump_env = getenv("LASR_PUMP") ? atoi(getenv("LASR_PUMP")) : 1;
    const bool can = pump_env != 0 && c->use_graphs && !c->dbg && c->M <= 512;
    if (!can) { c->pump_on = false; return LASR_OK; }
    RC(ensure_group_graphs(c));
    if (!c->pump_started) {
        c->pump_started = true;
        c->pump_th = std::thread(pump_main, c);
    }
    c->pump_on = true;
    return LASR_OK;
}
// The counter moves UNDER c->mu: the sleeping pump checks it and blocks under the same mutex, so a kick cannot fall between its
// check and its sleep.  (Round 4 first bumped it without the mutex: a kick in that window was lost and the pump slept out its
// timeout -- 20 ms then -- while a step waited; seen once, as a 25 ms stall in an 81 ms timed region.  The timeout is 2 ms now.)
static void pump_kick(lasr_ctx* c) {
    { std::lock_guard<std::mutex> lk(c->mu); c->kick.fetch_add(1, std::memory_order_release); }
    c->cv_pump.notify_one();
}

static bool cont_step_done(const lasr_ctx* c, const lasr_ctx::PendingStep& P) {
    for (int r : P.rows)
        if (c->h_cur_seen[r] < P.target[r]) return false;
    return true;
}

int lasr_step_wait(lasr_ctx* c, int* n_ran) {
    if (!c) return LASR_EINVAL;
    RoctxRange roctx_range_("lasr_step_wait");
    if (n_ran) *n_ran = 0;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->pending.empty()) return LASR_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    const int M = c->M;
    int* flag = c->cont_host;
    int* h_end = c->cont_host + 16 + (size_t)lasr_ctx::NFLY * M;
    int* h_ring = h_end + (size_t)M * lasr_ctx::ENDSLOTS;
    if (c->pump_on) {
        // the pump thread launches the groups and consumes their flags: wait for its progress counter
        for (unsigned long long guard = 0;; ++guard) {
            const long long seen = c->progress.load(std::memory_order_acquire);
            {
                std::lock_guard<std::mutex> lk(c->mu);
                if (c->pump_rc) return fail(c, c->pump_rc, "decode pump: %s", c->pump_err.c_str());
                if (cont_step_done(c, c->pending.front())) break;
            }
            // a short spin (a group that is about to publish), then the thread sleeps until the pump has consumed the next group's
            // flag (cont_poll notifies): with 20 steps in flight a collect has slack, and a rank no longer burns a core waiting
            // (round 5: 2.65 -> see profiles/r05 host cores per rank).  The time-out kicks a sleeping pump (backstop).
            unsigned long long spins = 0;
            while (c->progress.load(std::memory_order_acquire) == seen) {
                __builtin_ia32_pause();
                if (++spins > 3000) {
                    std::unique_lock<std::mutex> lk(c->mu);
                    const bool moved = c->cv_prog.wait_for(lk, std::chrono::milliseconds(10), [&] { return c->progress.load(std::memory_order_acquire) != seen || c->pump_rc != 0; });
                    lk.unlock();
                    if (!moved) pump_kick(c);
                    break;
                }
            }
            if (guard > (1u << 16)) return fail(c, LASR_EHIP, "decode loop did not converge");
        }
    } else {
        for (int guard = 0;; ++guard) {
            {
                std::lock_guard<std::mutex> lk(c->mu);
                cont_poll(c);
                if (cont_step_done(c, c->pending.front())) break;
                if (!c->group_inflight) RC(cont_launch_group(c, c->wait_n));
            }
            RC(spin_flag(c, flag, c->stream_dec));
            if (guard > 4096) return fail(c, LASR_EHIP, "decode loop did not converge");
        }
    }
    // results of the oldest step: tokens between the previous and this step boundary of every row, already
    // in pinned memory (written by the kernels of the groups that completed before the cursors were published)
    std::lock_guard<std::mutex> lk(c->mu);
    lasr_ctx::PendingStep& P = c->pending.front();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (c->W > 1) {      // beam: the whole best hypothesis as of this model step (lasr_fetch semantics of beam > 1)
        for (int r : P.rows) {
            if (c->b_results[r].empty()) return fail(c, LASR_EHIP, "beam: no result for slot %d although its step is decoded", r);
            lasr_ctx::BeamResult& br = c->b_results[r].front();
            c->queue[r] = br.tokens; c->best_full[r] = br.tokens;
            c->neg_logp[r] = -br.score; c->align[r] = 0.0;
            c->b_results[r].pop_front();
        }
    } else
    for (int r : P.rows) {
        const int j = P.target[r] / P.Tm - 1;
        const long long end = h_end[(size_t)r * lasr_ctx::ENDSLOTS + (j % lasr_ctx::ENDSLOTS)];
        for (long long q = c->h_fetched[r]; q < end; ++q)
            c->queue[r].push_back(h_ring[(size_t)r * lasr_ctx::TOKRING + (q % lasr_ctx::TOKRING)]);
        c->h_fetched[r] = end;
    }
    c->stats.frames = P.Tm;
    c->stats.decode_iters = (int)(c->cont_iters - c->iters_reported);
    c->iters_reported = c->cont_iters;
    if (n_ran) *n_ran = (int)P.rows.size();
    c->pending.erase(c->pending.begin());
    c->cmd_inflight = 0;
    if (!c->pump_on) RC(cont_pump_locked(c, c->wait_n));
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
    RoctxRange roctx_range_("lasr_transcribe");
    if (!c) return LASR_EINVAL;
    RC(flush_lazy(c));
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
    RC(flush_lazy(c));
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
    RC(flush_lazy(c));
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
        launch_select<true>(c->stream, B, logits, V, c->d.blank, 1, nullptr, s, 0, logp_max, argmax, 1, c->M);
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

int lasr_resample(lasr_ctx* c, const float* pcm, int B, int64_t N_in, int sr_in, float* out, int64_t* N_out);

// Generic client windows (any chunk length, any sample rate): what ASRServicer.TranscribeStream + x_tfm_stream do per call
// (api-server.py:83-115; transforms.py:141-144 Resample of the WHOLE window, :306-323 log-mel of the window with reflect
// padding, :335-342 frames T//3 + 1 .. + n_stack, :436-441 stack, :463-471 Buffer) for windows the caller has concatenated:
// pcm = [n][N] float32 (host or device) at `sr` Hz, one window per listed slot.  Synchronous protocol; the fast path
// (lasr_push_pcm + lasr_step_*) is the same computation specialised to 16 kHz / fixed chunks with the window kept on the GPU.
int lasr_step_window(lasr_ctx* c, const int* slots, int n, const float* pcm, int64_t N, int sr, int* n_ran) {
    if (!c) return LASR_EINVAL;
    if (n_ran) *n_ran = 0;
    RC(check_slots(c, slots, n, true));
    RC(flush_lazy(c));
    RC(require_idle(c));
    if (n == 0) return LASR_OK;
    if (!pcm || N < 1) return fail(c, LASR_EINVAL, "bad window");
    if (c->M > 512) return fail(c, LASR_EINVAL, "lasr_step_window supports up to 512 stream slots");
    HIPCHK(c, hipSetDevice(c->device));
    const lasr_model_desc& d = c->d;
    const float* src = pcm;
    if (!is_device_ptr(pcm)) {
        RC(ensure_buf(c, &c->stage_pcm, &c->stage_pcm_floats, (size_t)n * N));
        HIPCHK(c, hipMemcpyAsync(c->stage_pcm, pcm, sizeof(float) * (size_t)n * N, hipMemcpyHostToDevice, c->stream));
        src = c->stage_pcm;
    }
    long long Nw = N;
    if (sr != d.sample_rate) {                              // Resample.encodes on the whole window (transforms.py:141-144)
        int64_t No = 0;
        RC(lasr_resample(c, src, n, N, sr, nullptr, &No));
        RC(ensure_buf(c, &c->win_rs, &c->win_rs_floats, (size_t)n * No));
        RC(lasr_resample(c, src, n, N, sr, c->win_rs, &No));
        src = c->win_rs; Nw = No;
    }
    const int T = 1 + (int)(Nw / d.hop);
    const int a0 = T / 3 + 1;                               // StreamPostprocess (transforms.py:335-342)
    if (T - a0 < d.n_stack) return fail(c, LASR_EINVAL, "window of %lld samples at 16 kHz yields %d < n_stack frames after the cut", Nw, T - a0);
    if (Nw <= d.n_fft / 2) return fail(c, LASR_EINVAL, "window shorter than the reflect padding");
    for (int i = 0; i < n; ++i) {           // validate every slot BEFORE any state changes: an error leaves nothing half-done
        const int s = slots[i];
        for (int j = 0; j < c->n_pend[s]; ++j)
            if (c->fe_fused && !c->pend_mat[(size_t)s * d.n_buffer + j])
                return fail(c, LASR_ESTATE, "slot %d has frames pending from lasr_push_pcm steps: do not mix the two streaming forms", s);
    }
    RC(cmd_begin(c));
    std::vector<int> model_rows;
    MelArgs m{};
    fill_mel_args(c, m);
    for (int i = 0; i < n; ++i) {
        const int s = slots[i];
        m.dst_row_v[i] = (short)s;
        m.sel_v[i] = (short)(c->n_pend[s] * d.n_stack);
        c->pend_mat[(size_t)s * d.n_buffer + c->n_pend[s]] = 1;
        c->pend_serial[(size_t)s * d.n_buffer + c->n_pend[s]] = c->n_chunks[s];
        if (++c->n_pend[s] == d.n_buffer) {
            c->n_pend[s] = 0;
            c->hc.T_row[s] = d.n_buffer;
            model_rows.push_back(s);
        }
    }
    RC(cmd_commit(c));
    rec(c, 0);
    m.pcm = src; m.N = Nw; m.stream = 0; m.by_value = 1; m.frame0 = a0; m.frames_per_row = d.n_stack;
    m.out = c->pend; m.out_frames = d.n_buffer * d.n_stack; m.chunk = d.chunk; m.n_window = d.n_window;
    hipLaunchKernelGGL(k_logmel, dim3((d.n_stack + 3) / 4, n), dim3(256), 0, c->stream, m);
    if (model_rows.empty()) {
        HIPCHK(c, hipGetLastError());
        return LASR_OK;
    }
    const int Tm = d.n_buffer;
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
    RC(run_decode(c, Tm, d.max_iters_stream, false, model_rows));
    rec(c, 3);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    collect_stats(c, Tm);
    if (n_ran) *n_ran = (int)model_rows.size();
    return LASR_OK;
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

// The LM's own stream (see cont_enqueue), LASR_LM_SIDE=1 (default: the LM step in line on the decode stream): created when an
// LM is attached, on a hardware queue shared with neither the main nor the decode stream (probed like the decode stream at
// lasr_create; a stream that cannot be placed is given up).  Opt-in because what it buys depends on where the RUNTIME places the
// branches of the replayed group graph, which the probe does not control: the same build measured 32.7 against 30.7 k in line on
// one lease / process shape and 27.0 against 30.5 k on another (profiles/r04/r04_experiments.txt N).
static int lm_side_setup(lasr_ctx* c) {
    static const int on = getenv("LASR_LM_SIDE") ? atoi(getenv("LASR_LM_SIDE")) : 0;
    if (!on || !c->stream_dec || c->stream_lm) return LASR_OK;
    static const int pick = getenv("LASR_DEC_STREAM_PICK") ? atoi(getenv("LASR_DEC_STREAM_PICK")) : 1;
    std::vector<hipStream_t> rejected;
    bool ok = false;
    for (int attempt = 0; attempt < 8 && !ok; ++attempt) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream_lm, hipStreamNonBlocking));
        if (!pick) { ok = true; break; }
        double r0 = 0.0, r1 = 0.0;
        if (overlap_probe_impl(c, 300, &r0, c->stream, c->stream_lm) != LASR_OK ||
            overlap_probe_impl(c, 300, &r1, c->stream_dec, c->stream_lm) != LASR_OK) { (void)hipGetLastError(); break; }
        c->lm_stream_ratio[0] = r0; c->lm_stream_ratio[1] = r1;
        ok = r0 < 1.5 && r1 < 1.5;
        if (!ok) { rejected.push_back(c->stream_lm); c->stream_lm = nullptr; }
    }
    for (hipStream_t st : rejected) (void)hipStreamDestroy(st);
    if (!ok && c->stream_lm) { (void)hipStreamDestroy(c->stream_lm); c->stream_lm = nullptr; }
    if (c->stream_lm) {
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_lm_fork, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_lm_join, hipEventDisableTiming));
    }
    if (getenv("LASR_VERBOSE"))
        fprintf(stderr, "[lasr] LM stream: %s, overlap probe %.2f (main) %.2f (decode)\n", c->stream_lm ? "own" : "in line",
                c->lm_stream_ratio[0], c->lm_stream_ratio[1]);
    return LASR_OK;
}

int lasr_attach_lm(lasr_ctx* c, const lasr_lm_desc* d, const float* weights, size_t n_weights) {
    if (!c) return LASR_EINVAL;
    if (c->lm.on) return fail(c, LASR_ESTATE, "an LM is already attached");
    if (!d || !weights || n_weights != lasr_lm_weight_count(d) || n_weights == 0)
        return fail(c, LASR_EINVAL, "LM weight blob has %zu floats, expected %zu", n_weights, d ? lasr_lm_weight_count(d) : (size_t)0);
    if (d->vocab != c->d.vocab) return fail(c, LASR_EINVAL, "LM vocabulary %d != model vocabulary %d", d->vocab, c->d.vocab);
    if (d->vocab > 4096) return fail(c, LASR_EINVAL, "LM fusion keeps a row's log-probs in registers: vocab <= 4096");
    if (d->embed % 16 || d->hidden % (c->bf ? 32 : 16)) return fail(c, LASR_EINVAL, "LM dims must be multiples of 16 (hidden: 32 for bf16)");
    RC(flush_lazy(c));
    RC(require_idle(c));
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    lasr_ctx::LM& m = c->lm;
    const int V = d->vocab, E = d->embed, H = d->hidden, L = d->layers;
    const int M = c->Md;                               // LM rows: streams, or hypothesis slots with beam > 1 (Md = M x W)
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
        launch_table_gemm_f32(c, 4 * H / 16, V / 16, g, ea);
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
    if (c->W > 1) {                                    // second parity of everything a re-parented slot inherits
        m.y1.assign(L, nullptr); m.cst1.assign(L, nullptr);
        for (int l = 0; l < L; ++l) {
            RC(dalloc(c, (char**)&m.y1[l], (size_t)M * H * c->esz)); HIPCHK(c, hipMemset(m.y1[l], 0, (size_t)M * H * c->esz));
            RC(dalloc(c, &m.cst1[l], (size_t)M * H)); HIPCHK(c, hipMemset(m.cst1[l], 0, sizeof(float) * (size_t)M * H));
        }
        RC(dalloc(c, &m.lmz1, (size_t)M * V)); RC(dalloc(c, &m.valid1, M));
        HIPCHK(c, hipMemset(m.lmz1, 0, sizeof(float) * (size_t)M * V)); HIPCHK(c, hipMemset(m.valid1, 0, sizeof(int) * M));
    }
    for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second);   // decode groups change shape
    c->graphs.clear();
    for (auto& kv : c->cgraphs) (void)hipGraphExecDestroy(kv.second);
    c->cgraphs.clear();
    // (round 4: lookahead stays on with an LM -- blank frames change neither the predictor nor the LM state, k_select re-picks the
    //  token of the first non-blank frame of its window; LASR_LM_LOOKAHEAD=0 restores one frame per iteration)
    if (getenv("LASR_LM_LOOKAHEAD") && atoi(getenv("LASR_LM_LOOKAHEAD")) == 0) c->la = c->la_stream = c->la_offline = c->la_sync = 1;
    c->ds.lmz = m.lmz; c->ds.lm_valid = m.valid; c->ds.lm_alpha = m.alpha; c->ds.lm_theta = m.theta; c->ds.lm_min = m.min_val;
    m.on = true;
    RC(lm_side_setup(c));
    return LASR_OK;
}

// The LM as the reference serves it: load_lm (lm.py:86-100) runs maybe_quantize (utils.py:197-210) =
// torch.quantization.quantize_dynamic({nn.LSTM, nn.Linear}, qint8) on it.  Same blob as lasr_attach_lm; the weights are
// quantised here (per tensor, symmetric: scale = max|w| / 127.5, q = clamp(rint(w * (1 / scale)), -128, 127)), activations per
// row and per matmul (at run time the quantised image of every h is made once, by the cell kernel that produces it: k_lm_cell_q).  Numerics of the installed torch's x86 / fbgemm engine (restated in
// oracle/rnnt_oracle.py:dq_linear and pinned to the reference's quantised LM there).
int lasr_attach_lm_int8(lasr_ctx* c, const lasr_lm_desc* d, const float* weights, size_t n_weights) {
    if (!c) return LASR_EINVAL;
    if (c->lm.on) return fail(c, LASR_ESTATE, "an LM is already attached");
    if (c->W > 1) return fail(c, LASR_ESTATE, "LM shallow fusion is implemented for greedy decoding (beam = 1)");
    if (!d || !weights || n_weights != lasr_lm_weight_count(d) || n_weights == 0)
        return fail(c, LASR_EINVAL, "LM weight blob has %zu floats, expected %zu", n_weights, d ? lasr_lm_weight_count(d) : (size_t)0);
    if (d->vocab != c->d.vocab) return fail(c, LASR_EINVAL, "LM vocabulary %d != model vocabulary %d", d->vocab, c->d.vocab);
    if (d->vocab > 4096 || d->vocab % 16) return fail(c, LASR_EINVAL, "LM fusion keeps a row's log-probs in registers: vocab <= 4096, multiple of 16");
    if (d->hidden % 4 || d->hidden > 1024 || d->embed > 1024 || d->embed < 1 || d->layers < 1 || d->layers > 8)
        return fail(c, LASR_EINVAL, "int8 LM: hidden a multiple of 4, embed / hidden <= 1024 (exact integer accumulation), 1..8 layers");
    RC(flush_lazy(c));
    RC(require_idle(c));
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    lasr_ctx::LM& m = c->lm;
    const int V = d->vocab, E = d->embed, H = d->hidden, L = d->layers, M = c->M;
    m.E = E; m.H = H; m.L = L; m.alpha = d->alpha; m.theta = d->theta; m.min_val = d->min_val;
    auto pad32 = [](int k) { return (k + 31) / 32 * 32; };
    m.Kp_h = pad32(H);
    Reader rd{weights, n_weights};
    const float* embed = rd.take((size_t)V * E);
    // one quantised matrix [N][K] -> integer-valued bf16 tiles (16 outputs x padded K), its scale
    auto quantise = [&](const float* w, int N, int K, void** dev, float* scale) -> int {
        float amax = 0.f;
        for (size_t i = 0; i < (size_t)N * K; ++i) amax = std::max(amax, std::fabs(w[i]));
        float sw = (float)((double)amax / 127.5);
        if (sw < 1.1920928955078125e-07f) sw = 1.1920928955078125e-07f;           // MinMaxObserver: scale >= eps
        const float inv = 1.0f / sw;
        const int Kp = pad32(K);
        Packed pk;
        pack_tiles(pk, 1, N / 16, Kp, [&](int t, int ui, int k) {
            if (k >= K) return 0.f;
            const float q = std::nearbyint(w[(size_t)(16 * t + ui) * K + k] * inv);
            return std::min(127.f, std::max(-128.f, q));
        });
        *scale = sw;
        return upload_packed(c, dev, pk);
    };
    m.cells.assign(L, Cell{});
    m.qWih.assign(L, nullptr); m.qWhh.assign(L, nullptr); m.s_ih.assign(L, 0.f); m.s_hh.assign(L, 0.f);
    m.b_ih.assign(L, nullptr); m.b_hh.assign(L, nullptr); m.Kp_ih.assign(L, 0);
    for (int l = 0; l < L; ++l) {
        const int I = l == 0 ? E : H;
        const float* wih = rd.take((size_t)4 * H * I); const float* whh = rd.take((size_t)4 * H * H);
        const float* bih = rd.take(4 * H); const float* bhh = rd.take(4 * H);
        if (!bhh) return fail(c, LASR_EINVAL, "LM weight blob too short");
        m.Kp_ih[l] = pad32(I);
        RC(quantise(wih, 4 * H, I, &m.qWih[l], &m.s_ih[l]));
        RC(quantise(whh, 4 * H, H, &m.qWhh[l], &m.s_hh[l]));
        RC(upload(c, &m.b_ih[l], bih, 4 * H)); RC(upload(c, &m.b_hh[l], bhh, 4 * H));
    }
    const float* wout = rd.take((size_t)V * H); const float* bout = rd.take(V);
    if (!bout || rd.left != 0) return fail(c, LASR_EINVAL, "LM weight blob layout mismatch");
    RC(quantise(wout, V, H, &m.qWout, &m.s_out));
    RC(upload(c, &m.bout, bout, V));
    const int Kmax = std::max(m.Kp_h, m.Kp_ih[0]);
    {   // layer-0 input table: tab[v] = linear_dynamic(embed[v]; W_ih0) + b_ih0 -- a pure function of the token, quantisation included
        float* emb_dev = nullptr; unsigned short* qa = nullptr; float* sx = nullptr;
        RC(upload(c, &emb_dev, embed, (size_t)V * E));
        RC(dalloc(c, &qa, (size_t)V * m.Kp_ih[0])); RC(dalloc(c, &sx, V));
        RC(dalloc(c, &m.cells[0].tab, (size_t)V * 4 * H));
        lm_q_gemv(c, emb_dev, E, E, m.Kp_ih[0], m.qWih[0], m.s_ih[0], m.b_ih[0], m.cells[0].tab, 4 * H, V, qa, sx);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipGetLastError());
        dfree(c, emb_dev); dfree(c, qa); dfree(c, sx);
    }
    for (int p = 0; p < 2; ++p) m.h[p].resize(L);
    m.y.assign(L, nullptr); m.cst.resize(L);
    for (int l = 0; l < L; ++l) {
        float* h = nullptr;
        RC(dalloc(c, &h, (size_t)M * H)); HIPCHK(c, hipMemset(h, 0, sizeof(float) * (size_t)M * H));
        m.h[0][l] = m.h[1][l] = h;                              // row-local in-place update: no ping-pong
        RC(dalloc(c, &m.cst[l], (size_t)M * H)); HIPCHK(c, hipMemset(m.cst[l], 0, sizeof(float) * (size_t)M * H));
    }
    m.qh.assign(L, nullptr); m.sxh.assign(L, nullptr);
    {   // quantised image of h = 0: zeros with scale 0.1 (what ChooseQuantizationParams makes of an all-zero row)
        std::vector<float> tenth(M, 0.1f);
        for (int l = 0; l < L; ++l) {
            RC(dalloc(c, &m.qh[l], (size_t)M * m.Kp_h)); HIPCHK(c, hipMemset(m.qh[l], 0, sizeof(unsigned short) * (size_t)M * m.Kp_h));
            RC(upload(c, &m.sxh[l], tenth.data(), (size_t)M));
        }
    }
    RC(dalloc(c, &m.gx, (size_t)M * 4 * H)); RC(dalloc(c, &m.gh, (size_t)M * 4 * H));
    RC(dalloc(c, &m.qa, (size_t)M * Kmax)); RC(dalloc(c, &m.sx, M));
    RC(dalloc(c, &m.raw, (size_t)M * V)); RC(dalloc(c, &m.lmz, (size_t)M * V)); RC(dalloc(c, &m.valid, M));
    HIPCHK(c, hipMemset(m.lmz, 0, sizeof(float) * (size_t)M * V)); HIPCHK(c, hipMemset(m.valid, 0, sizeof(int) * M));
    for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second);   // decode groups change shape
    c->graphs.clear();
    for (auto& kv : c->cgraphs) (void)hipGraphExecDestroy(kv.second);
    c->cgraphs.clear();
    // (round 4: lookahead stays on with an LM -- blank frames change neither the predictor nor the LM state, k_select re-picks the
    //  token of the first non-blank frame of its window; LASR_LM_LOOKAHEAD=0 restores one frame per iteration)
    if (getenv("LASR_LM_LOOKAHEAD") && atoi(getenv("LASR_LM_LOOKAHEAD")) == 0) c->la = c->la_stream = c->la_offline = c->la_sync = 1;
    c->ds.lmz = m.lmz; c->ds.lm_valid = m.valid; c->ds.lm_alpha = m.alpha; c->ds.lm_theta = m.theta; c->ds.lm_min = m.min_val;
    m.q8 = true;
    m.on = true;
    RC(lm_side_setup(c));
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

// debug (LASR_DBG_ENCLOG=N at create): the checksum log of the first N model steps since create or the last read; see lasr_debug.h
int lasr_debug_enclog(lasr_ctx* c, unsigned* out, size_t cap, int* steps) {
    if (!c || !steps) return LASR_EINVAL;
    if (!c->enclog) return fail(c, LASR_ESTATE, "set LASR_DBG_ENCLOG=<steps> before lasr_create");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)c->enclog_n * 32 * c->M;
    *steps = c->enclog_n;
    if (n > cap) return fail(c, LASR_EFULL, "checksum log needs %zu words", n);
    if (n && out) HIPCHK(c, hipMemcpy(out, c->enclog, sizeof(unsigned) * n, hipMemcpyDeviceToHost));
    c->enclog_n = 0;
    return LASR_OK;
}

// debug: see lasr_debug.h.  The streaming front-end's log-mel launch (k_fe_mel) back to back on the main stream over the resident PCM
// ring, each launch followed by per-row checksums of its output, while the decode stream runs `aggressor` in a loop.
int lasr_debug_fe_race(lasr_ctx* c, int iters, int aggressor, int per_iter, int lds_pad, int* bad_launches, int* bad_rows) {
    if (!c || !bad_launches || !bad_rows || iters < 1) return LASR_EINVAL;
    *bad_launches = *bad_rows = 0;
    if (lds_pad < 0) lds_pad = c->fe_lds_pad;
    if (lds_pad > 160 * 1024 - 48 * 1024) return fail(c, LASR_EINVAL, "lds_pad too large");
    RC(flush_lazy(c));
    RC(require_idle(c));
    if (!c->fe_fused || !c->stream_dec || c->M > 512) return fail(c, LASR_ESTATE, "the race probe needs the fused front-end and the decode stream");
    HIPCHK(c, hipSetDevice(c->device));
    const lasr_model_desc& d = c->d;
    int nf = 0;
    const int a0 = stream_frame0(c, &nf);
    FeMelArgs m{};
    m.window = c->window; m.tw512 = c->tw512; m.tw1024 = c->tw1024; m.fb_start = c->fb_start; m.fb_off = c->fb_off; m.fb_w = c->fb_w;
    m.n_mels = d.n_mels; m.hop = d.hop; m.fb_nnz = c->fb_nnz; m.win_off = (d.n_fft - d.win) / 2; m.win_len = d.win;
    m.pcm = c->win; m.ring_pos = c->ring_pos; m.chunk = d.chunk; m.n_window = d.n_window; m.ring_chunks = c->ring_chunks; m.frame0 = a0;
    // (outputs go to scratch buffers: the slots' pending frames and the step's frame counts stay as they are)
    const size_t n_pend = (size_t)c->M * d.n_buffer * d.n_stack * d.n_mels;
    float* pend_x = nullptr; int* trow_x = nullptr;
    RC(dalloc(c, &pend_x, n_pend));
    RC(dalloc(c, &trow_x, (size_t)c->M));
    m.pend = pend_x; m.pend_frames = d.n_buffer * d.n_stack; m.trow_out = trow_x;
    for (int r = 0; r < 512; ++r) { m.idx[r] = -1; m.tp_pk[r] = 0; m.age_pk[r] = 0; }
    for (int r = 0; r < c->M; ++r) {
        m.tp_pk[r] = (unsigned char)((c->h_ring_pos[r] << 4) | d.n_buffer);
        unsigned pk = 0;
        for (int j = 0; j < d.n_buffer; ++j) pk |= (unsigned)(d.n_buffer - 1 - j) << (4 * j);
        m.age_pk[r] = (unsigned short)pk;
    }
    unsigned* log = nullptr;
    RC(dalloc(c, &log, (size_t)(iters + 1) * c->M));
    RowSumArgs ra{};
    ra.s[0] = RowSumSrc{pend_x, 2, 0, 0, d.n_buffer * d.n_stack * d.n_mels};
    hipStream_t keep = c->stream;
    auto aggress = [&]() {
        c->stream = c->stream_dec;
        if (aggressor == 1) launch_logits(c, c->logits, c->Md, false);
        else if (aggressor == 2) launch_predictor(c, c->W > 1);
        else if (aggressor == 3) launch_ppj(c, c->W > 1);
        c->stream = keep;
    };
    const int pp0 = c->pred_par;
    for (int i = 0; i <= iters; ++i) {
        if (i > 0) for (int q = 0; q < per_iter; ++q) aggress();
        hipLaunchKernelGGL((k_fe_mel<10>), dim3(2 * d.n_buffer, c->M), dim3(320), lds_pad, c->stream, m);
        hipLaunchKernelGGL(k_dbg_rowsum, dim3(c->M, 1), dim3(256), 0, c->stream, ra, c->M, c->bf, log + (size_t)i * c->M);
        if (i == 0) HIPCHK(c, hipStreamSynchronize(c->stream));       // the reference launch runs alone
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream_dec));
    c->pred_par = pp0;
    std::vector<unsigned> h((size_t)(iters + 1) * c->M);
    hipError_t e = hipMemcpy(h.data(), log, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
    dfree(c, log); dfree(c, pend_x); dfree(c, trow_x);
    if (e != hipSuccess) return fail(c, LASR_EHIP, "race probe copy failed: %s", hipGetErrorString(e));
    for (int i = 1; i <= iters; ++i) {
        int nb = 0;
        for (int r = 0; r < c->M; ++r) nb += h[(size_t)i * c->M + r] != h[r];
        *bad_rows += nb;
        *bad_launches += nb != 0;
    }
    return LASR_OK;
}

// In-job timing of the dominant kernel: while on, every model step's encoder-cell sequence (enc_layers x frames
// back-to-back launches of k_gemm<EpiLSTM>) is bracketed by a HIP-event pair on the ctx stream.  lasr_cell_prof_read
// drains the outstanding pairs and returns the accumulated microseconds / cell launches since the last on-switch.
int lasr_cell_prof(lasr_ctx* c, int on) {
    if (!c) return LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    // on = 1: HIP-event pairs + in-kernel clocks; on = 2: in-kernel clocks only (an event record between two kernels
    // costs the stream a bubble of several microseconds: twice per model step in mode 1)
    c->cell_prof_events = on == 1 || on == 3;
    if ((on == 1 || on == 3) && !c->cp_ok) {
        for (auto& p : c->cp_ev)
            for (auto& e : p) HIPCHK(c, hipEventCreate(&e));
        c->cp_ok = true;
    }
    if (on) { cell_prof_harvest(c, true); c->cp_us = 0.0; c->cp_launches = 0; }
    if (on == 3) { dfree(c, c->cp_slots); c->cp_slots = nullptr; c->cp_slot_next = 0; }      // events only: the cells stay graph-replayable
    if (on && on != 3) {
        const size_t half = (size_t)PROF_W * lasr_ctx::NCELLSLOT;
        if (!c->cp_slots) {
            RC(dalloc(c, &c->cp_slots, 2 * half));
            int khz = 0;
            if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) c->cp_clock_mhz = khz / 1e3;
            else (void)hipGetLastError();
        }
        HIPCHK(c, hipMemset(c->cp_slots, 0xff, half * sizeof(unsigned long long)));            // entry clocks: ~0 = not written
        HIPCHK(c, hipMemset(c->cp_slots + half, 0, half * sizeof(unsigned long long)));         // exit clocks
        c->cp_slot_next = 0;
        c->cp_slot_cells.assign(lasr_ctx::NCELLSLOT, 0);
    }
    c->cell_prof = on != 0;
    return LASR_OK;
}
// the cell kernels' own durations since lasr_cell_prof(c, 1): per launch, max exit - min entry of the device's constant
// wall clock over the kernel's workgroups (comparable with a kernel trace's duration column)
int lasr_cell_prof_kernel(lasr_ctx* c, double* us_total, long long* launches, long long* cells) {
    if (!c || !us_total || !launches) return LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    *us_total = 0.0; *launches = 0;
    if (cells) *cells = 0;
    if (!c->cp_slots || c->cp_slot_next == 0) return LASR_OK;
    HIPCHK(c, hipDeviceSynchronize());
    const size_t half = (size_t)PROF_W * lasr_ctx::NCELLSLOT, used = (size_t)PROF_W * c->cp_slot_next;
    std::vector<unsigned long long> he(used), hx(used);
    HIPCHK(c, hipMemcpy(he.data(), c->cp_slots, used * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(hx.data(), c->cp_slots + half, used * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (long long i = 0; i < c->cp_slot_next; ++i) {
        unsigned long long lo = ~0ull, hi = 0ull;
        for (int w = 0; w < PROF_W; ++w) {
            lo = std::min(lo, he[(size_t)i * PROF_W + w]);
            hi = std::max(hi, hx[(size_t)i * PROF_W + w]);
        }
        if (lo == ~0ull || hi < lo) continue;
        *us_total += (double)(hi - lo) / c->cp_clock_mhz;
        *launches += 1;
        if (cells) *cells += c->cp_slot_cells[i];
    }
    return LASR_OK;
}
int lasr_cell_prof_read(lasr_ctx* c, double* us_total, long long* launches) {
    if (!c || !us_total || !launches) return LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    cell_prof_harvest(c, true);
    *us_total = c->cp_us; *launches = c->cp_launches;
    return LASR_OK;
}

// Stream timeline of the pipelined protocol: while on, timestamped marks (HIP events) are recorded on the main stream
// (1 push, 3 first cell, 4 last cell done, 5 model step enqueued) and on the decode stream (10 group reached,
// 11 + 100 G [+ 1000 if steps were admitted] admission done, 12 group done).  lasr_trace_read synchronises and
// returns the marks in record order with their time in microseconds since lasr_trace(on).
int lasr_trace(lasr_ctx* c, int on) {
    if (!c) return LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (on && c->tr_ev.empty()) {
        c->tr_ev.resize(lasr_ctx::NTRACE); c->tr_tag.assign(lasr_ctx::NTRACE, 0); c->tr_val.assign(lasr_ctx::NTRACE, 0.0);
        for (auto& e : c->tr_ev) HIPCHK(c, hipEventCreate(&e));
        HIPCHK(c, hipEventCreate(&c->tr_base));
    }
    if (on) {
        c->tr_n = 0;
        HIPCHK(c, hipEventRecord(c->tr_base, c->stream));
        c->tr_prev_cur = c->h_cur_seen;
        c->tr_prev_ntot.assign(c->M, 0);        // the first group after the switch reports a bogus token delta
    }
    c->tr_on = on != 0;
    return LASR_OK;
}
int lasr_trace_read(lasr_ctx* c, double* us, int* tags, int cap, int* n) {
    if (!c || !us || !tags || !n) return LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    *n = 0;
    const int n_marks = std::min(c->tr_n.load(), (int)lasr_ctx::NTRACE);
    for (int i = 0; i < n_marks && i < cap; ++i) {
        float ms = 0.f;
        tags[i] = c->tr_tag[i];
        if (tags[i] == 20) { us[i] = c->tr_val[i]; *n = i + 1; continue; }      // value record
        if (hipEventElapsedTime(&ms, c->tr_base, c->tr_ev[i]) != hipSuccess) { (void)hipGetLastError(); ms = -1.f; }
        us[i] = 1e3 * (double)ms;
        *n = i + 1;
    }
    return LASR_OK;
}

// Debug / test read-out of the resident state (synchronises the ctx stream and the decode stream): [rows][K] row-major f32
// into `out` (host).  what: 0 x0 (LayerNorm'ed features) of frame `index`; 1 enc_h / 2 enc_c of layer `index` (current
// parity); 3 pe (joint encoder half, synchronous buffer) of frame `index`; 4 pp; 5 pred_h of layer `index` (current parity,
// decoder rows); 6 PCM ring [M][ring_chunks * chunk]; 7 pending log-mel frames [M][n_buffer * n_stack * n_mels];
// 8 last encoder layer's output of frame `index`; 9 integers as floats [8][M]: ring_pos, host ring mirror, n_chunks,
// n_pend, T_row_fix, t_idx, token, emit.  *rows / *cols (optional) describe the matrix; cap = floats `out` holds.
int lasr_debug_read(lasr_ctx* c, int what, int index, float* out, size_t cap, int* rows, int* cols) {
    if (!c || !out) return LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    RC(flush_lazy(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->stream_dec) HIPCHK(c, hipStreamSynchronize(c->stream_dec));
    const lasr_model_desc& d = c->d;
    const int M = c->M, H = d.hidden, F = d.feat, J = d.joint;
    int R = M, K = 0;
    const int mt_total = c->Tcap * c->MT;
    enum { FRAG, ROWMAJ_ELEM, ROWMAJ_F32, UNITMAJ, INTS } kind = ROWMAJ_F32;
    const void* src = nullptr; int frag_mt_total = 0, frag_mt_off = 0;
    switch (what) {
        case 0: if (index < 0 || index >= c->Tcap) return fail(c, LASR_EINVAL, "frame out of range");
                kind = FRAG; src = c->x0; K = F; frag_mt_total = mt_total; frag_mt_off = index * c->MT; break;
        case 1: if (index < 0 || index >= d.enc_layers) return fail(c, LASR_EINVAL, "layer out of range");
                kind = FRAG; src = c->enc_h[c->enc_par][index]; K = H; frag_mt_total = c->MT; frag_mt_off = 0; break;
        case 2: if (index < 0 || index >= d.enc_layers) return fail(c, LASR_EINVAL, "layer out of range");
                kind = UNITMAJ; src = c->enc_c[index]; K = H; break;
        case 3: if (index < 0 || index >= c->Tcap) return fail(c, LASR_EINVAL, "frame out of range");
                kind = ROWMAJ_F32; src = c->pe_sync + (size_t)index * M * J; K = J; break;
        case 4: kind = ROWMAJ_F32; src = cur_pp(c); K = J; R = c->Md; break;
        case 5: if (index < 0 || index >= d.pred_layers) return fail(c, LASR_EINVAL, "layer out of range");
                kind = ROWMAJ_ELEM; src = c->pred_h[c->pred_par][index]; K = H; R = c->Md; break;
        case 6: kind = ROWMAJ_F32; src = c->win; K = c->ring_chunks * d.chunk; break;
        case 7: kind = ROWMAJ_F32; src = c->pend; K = d.n_buffer * d.n_stack * d.n_mels; break;
        case 8: if (index < 0 || index >= c->Tcap) return fail(c, LASR_EINVAL, "frame out of range");
                kind = FRAG; src = c->ybuf[(d.enc_layers - 1) & 1]; K = H; frag_mt_total = mt_total; frag_mt_off = index * c->MT; break;
        case 9: kind = INTS; R = 8; K = M; break;
        case 10: if (!c->pendlog || index < 0 || index >= c->enclog_cap) return fail(c, LASR_EINVAL, "no pending-frame log (LASR_DBG_ENCLOG + LASR_DBG_PENDLOG) or step out of range");
                kind = ROWMAJ_F32; K = d.n_buffer * d.n_stack * d.n_mels; src = c->pendlog + (size_t)index * M * K; break;
        default: return fail(c, LASR_EINVAL, "unknown debug read %d", what);
    }
    if (rows) *rows = R;
    if (cols) *cols = K;
    if ((size_t)R * K > cap) return fail(c, LASR_EFULL, "debug read needs %zu floats", (size_t)R * K);
    if (kind == INTS) {
        std::vector<int> tmp(M);
        auto put = [&](int row, const int* v) { for (int r = 0; r < M; ++r) out[(size_t)row * M + r] = (float)v[r]; };
        HIPCHK(c, hipMemcpy(tmp.data(), c->ring_pos, sizeof(int) * M, hipMemcpyDeviceToHost)); put(0, tmp.data());
        put(1, c->h_ring_pos.data()); put(2, c->n_chunks.data()); put(3, c->n_pend.data());
        HIPCHK(c, hipMemcpy(tmp.data(), c->T_row_fix, sizeof(int) * M, hipMemcpyDeviceToHost)); put(4, tmp.data());
        HIPCHK(c, hipMemcpy(tmp.data(), c->ds.t_idx, sizeof(int) * M, hipMemcpyDeviceToHost)); put(5, tmp.data());
        HIPCHK(c, hipMemcpy(tmp.data(), c->ds.token, sizeof(int) * M, hipMemcpyDeviceToHost)); put(6, tmp.data());
        HIPCHK(c, hipMemcpy(tmp.data(), c->ds.emit, sizeof(int) * M, hipMemcpyDeviceToHost)); put(7, tmp.data());
        return LASR_OK;
    }
    if (kind == ROWMAJ_F32) {
        HIPCHK(c, hipMemcpy(out, src, sizeof(float) * (size_t)R * K, hipMemcpyDeviceToHost));
        return LASR_OK;
    }
    float* tmp = nullptr;
    RC(dalloc(c, &tmp, (size_t)R * K));
    if (kind == FRAG)
        hipLaunchKernelGGL(k_from_frag, dim3(grid1((size_t)R * K)), dim3(256), 0, c->stream, src, frag_mt_total, frag_mt_off, tmp, K, R, K, c->bf);
    else if (kind == ROWMAJ_ELEM)
        hipLaunchKernelGGL(k_from_elem, dim3(grid1((size_t)R * K)), dim3(256), 0, c->stream, src, tmp, (size_t)R * K, c->bf);
    else
        hipLaunchKernelGGL(k_c_to_rows, dim3(grid1((size_t)R * K)), dim3(256), 0, c->stream, (const float*)src, M, tmp, R, K);
    hipError_t e = hipMemcpy(out, tmp, sizeof(float) * (size_t)R * K, hipMemcpyDeviceToHost);
    dfree(c, tmp);
    if (e != hipSuccess) return fail(c, LASR_EHIP, "debug read copy failed: %s", hipGetErrorString(e));
    return LASR_OK;
}

int lasr_sync(lasr_ctx* c) {
    if (!c) return LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    RC(flush_lazy(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->cmd_inflight = 0;
    return LASR_OK;
}

// Do the two streams of the pipelined protocol (the ctx stream and the decode stream) run CONCURRENTLY on this process's
// hardware queues?  One wave per stream holds its stream for delay_us; *ratio = wall time / delay_us: ~1 when the streams
// overlap, ~2 when the runtime has mapped both onto one hardware queue (seen with an eagerly initialised RCCL communicator:
// the job then runs at 0.74 of its rate, profiles/r03/r03_experiments.txt M).  bench.py reports it per rank.
int lasr_overlap_probe(lasr_ctx* c, int delay_us, double* ratio) {
    if (!c || !ratio || delay_us < 1 || delay_us > 1000000) return c ? fail(c, LASR_EINVAL, "bad argument") : LASR_EINVAL;
    RC(flush_lazy(c));
    RC(require_idle(c));
    HIPCHK(c, hipSetDevice(c->device));
    return overlap_probe_impl(c, delay_us, ratio);
}
static int overlap_probe_impl(lasr_ctx* c, int delay_us, double* ratio, hipStream_t sa, hipStream_t sb) {
    if (!sb) { sa = c->stream; sb = c->stream_dec; }
    hipEvent_t e0, e1, e2;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1)); HIPCHK(c, hipEventCreate(&e2));
    const unsigned long long ticks = (unsigned long long)delay_us * 100ull;      // 100 MHz wall clock
    for (int rep = 0; rep < 2; ++rep) {                                            // (first pass: code object load, queue creation)
        HIPCHK(c, hipEventRecord(e0, sa));
        HIPCHK(c, hipStreamWaitEvent(sb, e0, 0));
        hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, sa, rep ? ticks : 100ull);
        hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, sb, rep ? ticks : 100ull);
        HIPCHK(c, hipEventRecord(e2, sb));
        HIPCHK(c, hipStreamWaitEvent(sa, e2, 0));
        HIPCHK(c, hipEventRecord(e1, sa));
        HIPCHK(c, hipEventSynchronize(e1));
    }
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    *ratio = 1e3 * (double)ms / (double)delay_us;
    return LASR_OK;
}

// Experiment: a synthetic neighbour beside the job (see k_nb_mfma / k_nb_load).  kind 1 = MFMA only, 2 = loads from HBM (non-temporal,
// 512 MB), 3 = loads that hit in L2 (96 KB per wave, again and again), 4 = loads from the Infinity Cache (128 MB): n_wg one-wave
// workgroups run for `ms` on a third stream whose hardware queue is shared with neither engine stream (probed; a long kernel on a
// shared queue would simply block the stream behind it) -- the call returns at once.  kind 0: wait for the neighbour and report what
// it got done in *rate (kind 1: TFLOP/s, kind 2: GB/s).  Which resource the two-stream job is short of shows in what a neighbour
// that takes ONLY that resource costs it (bench.py --neighbour).
int lasr_bench_neighbour(lasr_ctx* c, int kind, int n_wg, int ms, double* rate) {
    if (!c || kind < 0 || kind > 4) return c ? fail(c, LASR_EINVAL, "bad argument") : LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (kind == 0) {
        if (rate) *rate = 0.0;
        if (!c->stream_nb || !c->nb_kind) return LASR_OK;
        HIPCHK(c, hipStreamSynchronize(c->stream_nb));
        std::vector<unsigned long long> done(c->nb_wgs);
        HIPCHK(c, hipMemcpy(done.data(), c->nb_done, sizeof(unsigned long long) * c->nb_wgs, hipMemcpyDeviceToHost));
        double total = 0.0;
        for (auto v : done) total += (double)v;
        // k_nb_mfma: iterations are MFMAs of 2 * 16 * 16 * 4 flop; k_nb_load: 1 KB loads
        if (rate) *rate = c->nb_kind == 1 ? total * 2048.0 / (c->nb_ms * 1e-3) * 1e-12 : total * 1024.0 / (c->nb_ms * 1e-3) * 1e-9;
        c->nb_kind = 0;
        return LASR_OK;
    }
    if (n_wg < 1 || n_wg > 4096 || ms < 1 || ms > 5000) return fail(c, LASR_EINVAL, "neighbour: 1..4096 workgroups, 1..5000 ms");
    if (c->nb_kind) return fail(c, LASR_ESTATE, "a neighbour is already running: collect it with kind 0");
    if (!c->stream_nb) {
        std::vector<hipStream_t> rejected;
        for (int attempt = 0; attempt < 8 && !c->stream_nb; ++attempt) {
            hipStream_t st = nullptr;
            HIPCHK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            double r0 = 0.0, r1 = 0.0;
            if (overlap_probe_impl(c, 300, &r0, c->stream, st) != LASR_OK || overlap_probe_impl(c, 300, &r1, c->stream_dec, st) != LASR_OK) {
                (void)hipGetLastError(); rejected.push_back(st); break;
            }
            if (r0 < 1.5 && r1 < 1.5) c->stream_nb = st; else rejected.push_back(st);
        }
        for (hipStream_t st : rejected) (void)hipStreamDestroy(st);
        if (!c->stream_nb) return fail(c, LASR_EHIP, "no hardware queue left for the neighbour stream");
        RC(dalloc(c, &c->nb_done, 4096));
        c->nb_floats = (size_t)128 << 20;                 // 512 MB: twice the Infinity Cache
        RC(dalloc(c, &c->nb_buf, c->nb_floats));
        HIPCHK(c, hipMemset(c->nb_buf, 0, sizeof(float) * c->nb_floats));
        HIPCHK(c, hipDeviceSynchronize());
    }
    HIPCHK(c, hipMemsetAsync(c->nb_done, 0, sizeof(unsigned long long) * 4096, c->stream_nb));
    const unsigned long long ticks = (unsigned long long)ms * 100000ull;
    if (kind == 1) hipLaunchKernelGGL(k_nb_mfma, dim3(n_wg), dim3(64), 0, c->stream_nb, ticks, c->nb_done, c->nb_buf);
    else if (kind == 2) hipLaunchKernelGGL(k_nb_load, dim3(n_wg), dim3(64), 0, c->stream_nb, ticks, (const f32x4*)c->nb_buf, c->nb_floats / 4, (size_t)0, 0, c->nb_done, c->nb_buf);
    else if (kind == 3) hipLaunchKernelGGL(k_nb_load, dim3(n_wg), dim3(64), 0, c->stream_nb, ticks, (const f32x4*)c->nb_buf, c->nb_floats / 4, (size_t)(96 << 10) / 16, 1, c->nb_done, c->nb_buf);
    else hipLaunchKernelGGL(k_nb_load, dim3(n_wg), dim3(64), 0, c->stream_nb, ticks, (const f32x4*)c->nb_buf, (size_t)(128 << 20) / 16, (size_t)0, 1, c->nb_done, c->nb_buf);
    HIPCHK(c, hipGetLastError());
    c->nb_kind = kind; c->nb_wgs = n_wg; c->nb_ms = (double)ms;
    return LASR_OK;
}

// engine configuration as resolved at lasr_create (include/lasr_debug.h)
int lasr_debug_config(lasr_ctx* c, const char* key, int* value) {
    if (!c || !key || !value) return LASR_EINVAL;
    const struct { const char* k; int v; } tab[] = {
        {"enc_wave", c->enc_wave}, {"enc_u12", (int)c->enc_u12}, {"main_graph", (int)c->main_graph},
        {"pump_G", c->pump_G}, {"la_stream", c->la_stream}, {"la_offline", c->la_offline},
        {"cell_nw", c->cell_nw ? c->cell_nw : (c->bf ? 8 : 4)}, {"use_graphs", (int)c->use_graphs}, {"M", c->M},
        {"push_lazy", (int)c->lazy_on}, {"pump_nap_pct", c->pump_nap_pct}, {"lazy_taken", c->lazy_taken}, {"lazy_flushed", c->lazy_flushed},
        {"fe_lds_pad", c->fe_lds_pad}, {"roctx", roctx_state().push != nullptr ? 1 : 0},
    };
    for (const auto& e : tab)
        if (!strcmp(e.k, key)) { *value = e.v; return LASR_OK; }
    return fail(c, LASR_EINVAL, "lasr_debug_config: unknown key '%s'", key);
}

int lasr_bench_cell(lasr_ctx* c, int layer, int iters, double* us) {
    if (!c || !us || layer < 0 || layer >= c->d.enc_layers || iters < 1) return c ? fail(c, LASR_EINVAL, "bad argument") : LASR_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    const int Tn = 1;
    RC(ensure_T(c, Tn));
    const int H = c->d.hidden, I = c->enc[layer].I, M = c->M;
    // random (not zero) operands: zero-filled data inflates the clock (DVFS)
    hipLaunchKernelGGL(k_fill_rand, dim3(grid1((size_t)c->Tcap * M * I)), dim3(256), 0, c->stream, layer == 0 ? c->x0 : c->ybuf[(layer - 1) & 1],
                       (size_t)c->Tcap * M * I, 17u, c->bf);
    for (int p = 0; p < 2; ++p)
        hipLaunchKernelGGL(k_fill_rand, dim3(grid1((size_t)M * H)), dim3(256), 0, c->stream, c->enc_h[p][layer], (size_t)M * H, 23u + p, c->bf);
    RC(cmd_begin(c));
    for (int r = 0; r < c->d.max_streams; ++r) c->hc.T_row[r] = Tn;
    RC(cmd_commit(c));
    RC(commit_T_rows(c, Tn));
    const void* xsrc = layer == 0 ? c->x0 : c->ybuf[(layer - 1) & 1];
    const int mt_total = c->Tcap * c->MT;
    auto one = [&]() { launch_enc_cell(c, layer, 0, xsrc, mt_total, c->ybuf[layer & 1], mt_total); c->enc_par ^= 1; };
    for (int i = 0; i < 3; ++i) one();
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    HIPCHK(c, hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; ++i) one();
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *us = (double)ms * 1000.0 / ((double)iters * Tn);
    c->cmd_inflight = 0;
    HIPCHK(c, hipGetLastError());
    return LASR_OK;
}

}  // extern "C"

#include "lasr_front.hip.h"
