// lasr_decode.hip.h -- state reset, encoder pass, greedy and beam decode loops of the synchronous protocol
// Engine unit only (lasr_engine.hip), included after lasr_host.hip.h and lasr_cmd.hip.h.
#pragma once

namespace {

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
    const bool use_bos = c->bos_ready && !beam && !plain_rows && c->W == 1 && (mask & 2);
    if (use_bos) {
        for (int l = 0; l < a.Lp; ++l) { a.bos_h[l] = c->bos_h[l]; a.bos_c[l] = c->d.pred_cell ? c->bos_c[l] : nullptr; }
        a.bos_pp = c->bos_pp; a.pp = c->pp; a.J = c->d.joint;
        any_pred = false;                    // the state after the BOS step is stored, not computed
    }
    hipLaunchKernelGGL(k_reset_rows, dim3(grid1((size_t)c->M * c->d.hidden)), dim3(256), 0, c->stream, a);
    if (c->lm.on && (mask & 2)) {      // LM state lives on the decode side, like the predictor's
        LmResetArgs la{};
        const bool lb = c->W > 1 && !plain_rows;         // beam: W slots per stream, current parity of every ping-pong buffer
        la.what = c->dc.what; la.M = c->M; la.H = c->lm.H; la.L = c->lm.L; la.bf = c->lm.q8 ? 0 : c->bf;
        la.W = lb ? c->W : 1; la.Md = lb ? c->Md : c->M;
        la.lm_valid = (lb && c->lm.par) ? c->lm.valid1 : c->lm.valid;
        for (int l = 0; l < c->lm.L; ++l) { la.h[l] = c->lm.h[c->lm.par][l]; la.c[l] = (lb && c->lm.par) ? c->lm.cst1[l] : c->lm.cst[l]; }
        if (c->lm.q8) { la.Kp = c->lm.Kp_h; for (int l = 0; l < c->lm.L; ++l) { la.qh[l] = c->lm.qh[l]; la.sxh[l] = c->lm.sxh[l]; } }
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

// ---------------------------------------------------------------------------- in-job cell timing
void cell_prof_harvest(lasr_ctx* c, bool all) {
    while (c->cp_n > 0) {
        const int i = ((c->cp_head - c->cp_n) % lasr_ctx::NCELLEV + lasr_ctx::NCELLEV) % lasr_ctx::NCELLEV;
        if (!all && c->cp_n < lasr_ctx::NCELLEV && hipEventQuery(c->cp_ev[i][1]) != hipSuccess) { (void)hipGetLastError(); break; }
        (void)hipEventSynchronize(c->cp_ev[i][1]);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->cp_ev[i][0], c->cp_ev[i][1]) == hipSuccess) {
            c->cp_us += 1e3 * (double)ms;
            c->cp_launches += c->cp_cells[i];
        } else {
            (void)hipGetLastError();
        }
        c->cp_n--;
    }
}

// ---------------------------------------------------------------------------- encoder + decode
// Encoder over T_max frames for rows with T_row > 0 (x0 already holds LayerNorm'ed features).
void run_encoder(lasr_ctx* c, int T_max) {
    RoctxRange roctx_range_("lasr encoder cells");
    const int L = c->d.enc_layers;
    const int mt_total = c->Tcap * c->MT;
    const int par0 = c->enc_par;
    int cp_slot = -1;
    if (c->cell_prof && c->cp_ok && c->cell_prof_events) {
        if (c->cp_n >= lasr_ctx::NCELLEV) cell_prof_harvest(c, false);
        cp_slot = c->cp_head;
        c->cp_head = (c->cp_head + 1) % lasr_ctx::NCELLEV;
        c->cp_cells[cp_slot] = L * T_max;
        (void)hipEventRecord(c->cp_ev[cp_slot][0], c->stream);
    }
    tr_mark(c, 3, c->stream);
    // layer wavefront: the cells (l, t) with l + t = d depend only on diagonal d - 1, so a diagonal is ONE launch
    // (k_gemm_multi, up to NPMAX cells): L + T - 1 launches instead of L * T, and the per-launch fixed costs of a cell
    // overlap its neighbours' K loops.  Cell (l, t) reads h parity par0 ^ (t & 1); all layers end on par0 ^ (T & 1).
    // Otherwise layer-major order: every layer starts from parity par0 and toggles T_max times (enc_h[par][l] is
    // indexed by the parity at launch time, so all layers end on par0 ^ (T_max & 1)).
    auto enqueue_cells = [&]() {
        if (c->enc_wave && L > 1 && T_max > 1) {
            EncCellRef cells[NPMAX];
            for (int d = 0; d < L + T_max - 1; ++d) {
                int n = 0;
                for (int l = std::min(d, L - 1); l >= 0 && d - l < T_max; --l) {
                    cells[n++] = EncCellRef{l, d - l};
                    if (n == NPMAX) { launch_enc_wave(c, cells, n, par0, mt_total); n = 0; }
                }
                if (n) launch_enc_wave(c, cells, n, par0, mt_total);
            }
        } else {
            for (int l = 0; l < L; ++l) {
                c->enc_par = par0;
                const void* xsrc = (l == 0) ? c->x0 : c->ybuf[(l - 1) & 1];
                void* ydst = c->ybuf[l & 1];
                for (int t = 0; t < T_max; ++t) {
                    launch_enc_cell(c, l, t, xsrc, mt_total, ydst, mt_total);
                    c->enc_par ^= 1;
                }
            }
        }
        c->enc_par = par0 ^ (T_max & 1);
    };
    // Pipelined protocol: the cell sequence of a model step (launches whose arguments only depend on which m-tiles are active
    // and on the ping-pong parity) is replayed as ONE hipGraph on the caller's stream: 8 (bf16 wavefront: 5) launches -> 1 per
    // model step on the host.  The graph is recorded by capturing on an internal stream (the caller's may be the legacy NULL
    // stream, which cannot capture) and launched on the caller's.  Not with the in-kernel timers (a per-launch slot pointer) or
    // the debug stamps.
    bool replayed = false;
    if (c->main_graph && c->use_graphs && c->pe == c->pe_ring && !(c->cell_prof && c->cp_slots) && !c->dbg && T_max <= 8) {
        std::vector<unsigned long long> key{(unsigned long long)T_max, (unsigned long long)par0, (unsigned long long)(uintptr_t)c->T_row_dev,
                                            (unsigned long long)(uintptr_t)c->x0, (unsigned long long)mt_total, (unsigned long long)c->enc_wave};
        for (int t = 0; t < T_max; ++t) key.push_back(c->tile_masks.empty() ? ~0ull : c->tile_masks[t]);
        auto it = c->mgraphs.find(key);
        bool ok = true;
        if (it == c->mgraphs.end() && c->mgraphs.size() >= 64) {
            // the key holds the per-frame masks of the active m-tiles: with many slots and churning streams the combinations do
            // not repeat, and every new one costs a capture + instantiation on the submit path and memory for good.  Past 64
            // cached graphs the cache is dropped (the steady state of a full server -- all tiles active -- is one entry)
            for (auto& kv : c->mgraphs) (void)hipGraphExecDestroy(kv.second);
            c->mgraphs.clear();
        }
        if (it == c->mgraphs.end()) {
            if (!c->stream_cap && hipStreamCreateWithFlags(&c->stream_cap, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ok = false; }
            hipGraph_t gr = nullptr;
            hipGraphExec_t ex = nullptr;
            hipStream_t keep = c->stream;
            if (ok && hipStreamBeginCapture(c->stream_cap, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                c->stream = c->stream_cap;
                enqueue_cells();
                c->stream = keep;
                c->enc_par = par0;
                if (hipStreamEndCapture(c->stream_cap, &gr) != hipSuccess || !gr) ok = false;
                if (ok && hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0) != hipSuccess) ok = false;
                if (gr) (void)hipGraphDestroy(gr);
            } else ok = false;
            if (!ok) { (void)hipGetLastError(); c->main_graph = false; }      // (fall back to plain launches for good)
            else it = c->mgraphs.emplace(key, ex).first;
        }
        if (ok && hipGraphLaunch(it->second, c->stream) == hipSuccess) {
            c->enc_par = par0 ^ (T_max & 1);
            replayed = true;
        }
    }
    if (!replayed) enqueue_cells();
    if (cp_slot >= 0) { (void)hipEventRecord(c->cp_ev[cp_slot][1], c->stream); c->cp_n++; }
    tr_mark(c, 4, c->stream);
    if (c->enclog && c->enclog_n < c->enclog_cap && 2 * T_max + 2 * L + 2 <= 32) {       // LASR_DBG_ENCLOG: checksums behind this step's cells
        RowSumArgs ra{};
        int e = 0;
        for (int t = 0; t < T_max; ++t) ra.s[e++] = RowSumSrc{c->x0, 0, mt_total, t * c->MT, c->d.feat};
        for (int l = 0; l < L; ++l) ra.s[e++] = RowSumSrc{c->enc_c[l], 1, 0, 0, c->d.hidden};
        for (int l = 0; l < L; ++l) ra.s[e++] = RowSumSrc{c->enc_h[c->enc_par][l], 0, c->MT, 0, c->d.hidden};
        for (int t = 0; t < T_max; ++t) ra.s[e++] = RowSumSrc{c->ybuf[(L - 1) & 1], 0, mt_total, t * c->MT, c->d.hidden};
        ra.s[e++] = RowSumSrc{c->pend, 2, 0, 0, c->d.n_buffer * c->d.n_stack * c->d.n_mels};
        ra.s[e++] = RowSumSrc{c->win, 2, 0, 0, c->ring_chunks * c->d.chunk};
        hipLaunchKernelGGL(k_dbg_rowsum, dim3(c->M, e), dim3(256), 0, c->stream, ra, c->M, c->bf,
                           c->enclog + (size_t)c->enclog_n * 32 * c->M);
        if (c->pendlog) {
            const size_t n = (size_t)c->M * c->d.n_buffer * c->d.n_stack * c->d.n_mels;
            (void)hipMemcpyAsync(c->pendlog + (size_t)c->enclog_n * n, c->pend, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream);
        }
        c->enclog_n++;
    }
    // encoder half of the joint for all frames: pe[t][r] = W1e * enc[t][r]
    const int H = c->d.hidden, J = c->d.joint;
    GemmArgs g{};
    g.A[0] = c->ybuf[(L - 1) & 1]; g.a_mt_total[0] = mt_total; g.a_mt_off[0] = 0; g.W[0] = c->W1e;
    EpiLinear::Args ea{};
    ea.bias = nullptr; ea.out = c->pe; ea.ldo = J; ea.n_rows = T_max * c->M; ea.t_idx = nullptr; ea.T_row = nullptr; ea.M = c->M;
    if (c->pe == c->pe_ring) { ea.ring_base = c->fe_fused ? c->c_enc_base : c->c_enc_frames; ea.ring = lasr_ctx::RING; }   // continuous mode: per-row frame ring
    launch_linear<false, 3>(c, J / 16, T_max * c->MT, g, H, ea);
}

// ---- host side of the beam: hypotheses as a shared-prefix tree per stream
void bh_reset(lasr_ctx::BeamHost& B, int W) { B.par.clear(); B.tok.clear(); B.cur.assign(W, -1); }
void bh_tokens(const lasr_ctx::BeamHost& B, int node, std::vector<int32_t>& out) {      // appends root -> leaf
    const size_t at = out.size();
    for (int n = node; n >= 0; n = B.par[n]) out.push_back(B.tok[n]);
    std::reverse(out.begin() + at, out.end());
}
// one selection round of a stream: e[j] = (parent slot << 16) | (token + 1 if extended else 0); -2 dead slot
void bh_apply(lasr_ctx::BeamHost& B, const int* e, int W) {
    int nh[8];
    for (int j = 0; j < W; ++j) {
        if (e[j] < 0) { nh[j] = -1; continue; }
        const int p = B.cur[e[j] >> 16], tok = e[j] & 0xffff;
        if (tok) { B.par.push_back(p); B.tok.push_back(tok - 1); nh[j] = (int)B.par.size() - 1; }
        else nh[j] = p;
    }
    for (int j = 0; j < W; ++j) B.cur[j] = nh[j];
    if (B.par.size() > (size_t)1 << 18) {           // compaction: keep the live hypotheses only
        std::vector<std::vector<int32_t>> live(W);
        for (int j = 0; j < W; ++j) bh_tokens(B, B.cur[j], live[j]);
        B.par.clear(); B.tok.clear();
        for (int j = 0; j < W; ++j) {
            int n = -1;
            for (int32_t t : live[j]) { B.par.push_back(n); B.tok.push_back(t); n = (int)B.par.size() - 1; }
            B.cur[j] = n;
        }
    }
}

// Greedy decode of the current step (T_row_dev, pe ready).  Blocks until done; fills host queues.
int run_decode_beam(lasr_ctx* c, int T_max, int max_iters, bool offline, const std::vector<int>& rows);

int run_decode(lasr_ctx* c, int T_max, int max_iters, bool offline, const std::vector<int>& rows) {
    if (c->W > 1) return run_decode_beam(c, T_max, max_iters, offline, rows);
    const int M = c->M, J = c->d.joint, V = c->d.vocab;
    c->la = offline ? c->la_offline : c->la_sync;
    DecState s = c->ds;
    s.tok_cap = T_max * max_iters;
    const int total_cap = T_max * max_iters;
    int iter = 0;
    // iterations are launched in even-sized groups (the predictor ping-pong parity then returns to
    // its start); after each group the "rows still decoding" counter and the step's tokens so far
    // come back in the same round trip.  In streaming mode every group is a cached hipGraph: one
    // launch instead of 4 kernels per iteration, so the GPU is not fed at host launch speed.
    // (with lookahead a row consumes up to `la` blank frames per iteration: fewer iterations up front)
    constexpr int sync_first = 4;   // extra iterations of the first group
    constexpr int sync_next = 2;    // (swept in round 2: 4 + 2 best, +2 %)
    int group = offline ? std::min(total_cap, ((T_max + c->la - 1) / c->la + 16) & ~1) : std::min(total_cap, (T_max + sync_first) & ~1);
    const int next_group = offline ? 32 : sync_next;
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
            launch_select<false>(c->stream, M, c->logits, V, c->d.blank, max_iters, c->T_row_dec, s, it, nullptr, nullptr, c->la, M);
            launch_predictor(c);
            launch_ppj(c);
            launch_lm(c);
        }
        // The host spins on the "rows still decoding" word in pinned memory.  Streaming: the step's tokens so far travel with it --
        // k_publish stores them into the pinned block and releases the word last (system scope).  Offline (whole utterances,
        // large token blocks): only the word comes back per group; the results are copied once at the end, behind a real
        // synchronisation.  A word that arrives by hipMemcpyAsync carries nothing but itself: no ordering against any other
        // copy is assumed anywhere (round 3's "payload copy, then flag copy" let the flag overtake the payload: tests/soak.py).
        const int n_pay = M + M * s.tok_cap;
        if (!offline) {
            const int nb = std::max(1, std::min(64, (n_pay + 1023) / 1024));
            hipLaunchKernelGGL(k_publish, dim3(nb), dim3(256), 0, c->stream, (const int*)c->ds.step_ntok, c->res_dev + 4, n_pay,
                               (const int*)(c->ds.unfinished + (first + n - 1)), c->res_dev, c->pub_arrivals);
            return LASR_OK;
        }
        HIPCHK(c, hipMemcpyAsync(res, c->ds.unfinished + (first + n - 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
        return LASR_OK;
    };
    while (iter < total_cap) {
        const int n = std::min(group, total_cap - iter);
        bool launched = false;
        __atomic_store_n(&res[0], -1, __ATOMIC_RELEASE);      // sentinel, overwritten by the last copy of the group (stored BEFORE anything
                                                              // of the group is enqueued: a fast group must not be overwritten by it)
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
        if (!launched) RC(enqueue_group(iter, n));
        iter += n;
        // spin on the pinned word instead of hipStreamSynchronize (interrupt wake-up costs ~10-20 us per
        // round trip, and there are 2-4 per step); fall back to a real sync if nothing arrives.  LASR_SPIN=0: always a real sync
        static const bool spin = !(getenv("LASR_SPIN") && atoi(getenv("LASR_SPIN")) == 0);
        if (!spin) HIPCHK(c, hipStreamSynchronize(c->stream));
        else {
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
    if (offline) {
        // offline results: one copy each, read only after the stream has been synchronised
        HIPCHK(c, hipMemcpyAsync(ntok, c->ds.step_ntok, sizeof(int) * ((size_t)M + (size_t)M * s.tok_cap), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(sum_iters, c->ds.sum_iters, sizeof(int) * M, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(n_ones, c->ds.n_ones, sizeof(int) * M, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(logp, c->ds.logp_sum, sizeof(double) * M, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
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
    c->dbg_gate = c->dbg && getenv("LASR_DBG_BEAM");        // (LASR_DBG_TIMING + LASR_DBG_BEAM: phase stamps of the beam round's GEMMs)
    while (iter < total_cap) {
        const int n = std::min(group, total_cap - iter);
        for (int q = 0; q < n; ++q) {
            launch_logits(c, c->logits, Md, true);
            launch_beam_select(c, b, iter + q);
            launch_predictor(c, true);
            launch_ppj(c, true);
            launch_lm(c, true);
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
    for (int r : rows) {
        auto& H = c->bh[r];
        for (int it = 0; it < iter; ++it) {
            const int* e = tre + (size_t)it * Md + (size_t)r * W;
            if (e[0] == -1) continue;                          // stream idle in this round
            bh_apply(H, e, W);
        }
        int best = -1;
        for (int j = 0; j < W; ++j)
            if (alive[(size_t)r * W + j] && (best < 0 || sc[(size_t)r * W + j] > sc[(size_t)r * W + best])) best = j;
        auto& q = c->best_full[r];
        q = c->committed[r];                                   // what earlier predictor resets froze
        double score = c->committed_score[r];
        if (best >= 0) { bh_tokens(H, H.cur[best], q); score += sc[(size_t)r * W + best]; }
        c->queue[r] = q;                                       // beam mode: lasr_fetch hands out the whole best hypothesis
        c->neg_logp[r] = -score;
        c->align[r] = 0.0;                                     // alignment_score is a greedy-loop metric
    }
    return LASR_OK;
}

// host side of a predictor reset in beam mode: the best hypothesis so far is frozen, the beam restarts
void beam_host_reset(lasr_ctx* c, int slot, bool forget) {
    if (c->W <= 1) return;
    if (forget) { c->committed[slot].clear(); c->committed_score[slot] = 0.0; c->best_full[slot].clear(); }
    else { c->committed[slot] = c->best_full[slot]; c->committed_score[slot] = -c->neg_logp[slot]; }
    bh_reset(c->bh[slot], c->W);
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

// c->mu held.  The ctx stream waits (on the GPU, the host does not block) for whatever the last decode group still has queued
// behind its published flag: see lasr_ctx::dec_tail_open
int order_after_decode_tail(lasr_ctx* c) {
    if (!c->dec_tail_open || !c->stream_dec || !c->ev_misc) return LASR_OK;
    HIPCHK(c, hipEventRecord(c->ev_misc, c->stream_dec));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_misc, 0));
    c->dec_tail_open = false;
    return LASR_OK;
}
int require_idle(lasr_ctx* c) {
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->pending.empty()) return fail(c, LASR_ESTATE, "%d submitted step(s) not collected: call lasr_step_wait first", (int)c->pending.size());
    if (c->group_inflight) {        // the last group of the pipelined protocol may still be running: the calls that need an idle engine
        HIPCHK(c, hipStreamSynchronize(c->stream_dec));      // use the same decode state on the ctx stream
        cont_poll(c);                                        // (its flag and cursors are there now)
        c->group_inflight = false;
        c->dec_tail_open = false;
    }
    return order_after_decode_tail(c);
}

bool is_device_ptr(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeDevice;
}

// pinned (page-locked, device-mapped) host memory: returns the address the device can read it at, else nullptr
const void* pinned_host_dev_ptr(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
    return at.devicePointer;
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


}  // namespace
