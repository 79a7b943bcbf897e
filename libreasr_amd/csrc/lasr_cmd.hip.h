// lasr_cmd.hip.h -- command blocks (per-step host -> device parameters), per-step tile masks, buffers that grow.
// Engine unit only (lasr_engine.hip), included after lasr_host.hip.h.
#pragma once

namespace {

// the selection kernel of one beam round (+ the LM re-pick of the extended slots' tokens)
void launch_beam_select(lasr_ctx* c, BeamState& b, int iter_slot) {
    const int M = c->M;
    b.lm_on = c->lm.on ? 1 : 0; b.done2 = c->c_done2;
    const float* lg = (const float*)c->logits;
    // one wave per hypothesis row (k_beam_select_rw; V <= 2048 is checked at lasr_create for beam > 1)
    if (c->W <= 2) hipLaunchKernelGGL((k_beam_select_rw<2>), dim3(M), dim3(128), 0, c->stream, lg, b, iter_slot);
    else if (c->W <= 4) hipLaunchKernelGGL((k_beam_select_rw<4>), dim3(M), dim3(256), 0, c->stream, lg, b, iter_slot);
    else hipLaunchKernelGGL((k_beam_select_rw<8>), dim3(M), dim3(512), 0, c->stream, lg, b, iter_slot);
    if (c->lm.on)
        LAUNCH_BEAM_FUSE(c->d.vocab, dim3(c->Md), dim3(256), 0, c->stream, (const float*)c->logits, b, iter_slot, cur_lmz(c), cur_lm_valid(c),
                           c->lm.alpha, c->lm.theta, c->lm.min_val);
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
int commit_T_rows(lasr_ctx* c, int T_max, bool fixed_copy = true, int* fixed_home = nullptr) {
    // (fixed_home: the front-end launch wrote the counts there itself -- the pipelined protocol keeps ONE buffer on the main
    //  stream, so the cell launches of every step have the same arguments and can be replayed as a graph)
    c->T_row_dev = fixed_home ? fixed_home : c->dc.T_row;             // the command ring (NCMD blocks) outlives every step in flight
    // decode kernels of the synchronous protocols read a FIXED buffer (cached graphs replay baked-in pointers)
    if (fixed_copy) {
        HIPCHK(c, hipMemcpyAsync(c->T_row_fix, c->T_row_dev, sizeof(int) * c->M, hipMemcpyDeviceToDevice, c->stream));
        c->T_row_dec = c->T_row_fix;
    }
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
    for (auto& kv : c->mgraphs) (void)hipGraphExecDestroy(kv.second);
    c->mgraphs.clear();
    const int M = c->M, H = c->d.hidden, F = c->d.feat, J = c->d.joint;
    int cap = std::max(T, std::max(2 * c->Tcap, c->d.n_buffer));
    dfree(c, c->x0); dfree(c, c->ybuf[0]); dfree(c, c->ybuf[1]); dfree(c, c->pe_sync);
    c->pe_sync = nullptr;
    dfree(c, c->ds.step_ntok); dfree(c, c->ds.unfinished);
    c->x0 = c->ybuf[0] = c->ybuf[1] = c->pe = nullptr; c->ds.step_ntok = nullptr; c->ds.step_tok = nullptr; c->ds.unfinished = nullptr;
    RC(dalloc(c, (char**)&c->x0, (size_t)cap * M * F * c->esz));
    RC(dalloc(c, (char**)&c->ybuf[0], (size_t)cap * M * H * c->esz));
    RC(dalloc(c, (char**)&c->ybuf[1], (size_t)cap * M * H * c->esz));
    RC(dalloc(c, &c->pe_sync, (size_t)cap * M * J)); HIPCHK(c, hipMemset(c->pe_sync, 0, sizeof(float) * (size_t)cap * M * J));
    c->pe = c->pe_sync;
    const int mi = std::max(c->d.max_iters_offline, c->d.max_iters_stream);
    c->tok_cap_alloc = cap * mi;
    // [ntok M][tokens M x tok_cap]: one contiguous block so a group's results reach the host in one copy
    RC(dalloc(c, &c->ds.step_ntok, (size_t)M + (size_t)M * c->tok_cap_alloc));
    HIPCHK(c, hipMemset(c->ds.step_ntok, 0, sizeof(int) * ((size_t)M + (size_t)M * c->tok_cap_alloc)));
    c->ds.step_tok = c->ds.step_ntok + M;
    c->n_iter_slots = cap * mi + 8;
    RC(dalloc(c, &c->ds.unfinished, (size_t)c->n_iter_slots)); HIPCHK(c, hipMemset(c->ds.unfinished, 0, sizeof(int) * (size_t)c->n_iter_slots));
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
    memset(c->res_host, 0, c->res_bytes);
    {
        void* dp = nullptr;
        HIPCHK(c, hipHostGetDevicePointer(&dp, c->res_host, 0));
        c->res_dev = (int*)dp;
    }
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


}  // namespace
