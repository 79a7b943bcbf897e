// lasr_launch.hip.h -- kernel launch helpers (GEMM variants, predictor / joint / LM passes), command blocks, growing buffers
// Part of the single translation unit lasr_engine.hip (textual include, in this order:
// lasr_ctx, lasr_launch, lasr_decode, lasr_weights); not a stand-alone header.
#pragma once

namespace {

// ---------------------------------------------------------------------------- launch helpers
template <class Ops, class Epi, int MT, bool AROW, int D = 3, int NWV = NW>
void launch_gemm(lasr_ctx* c, int n_groups, int m_groups, const GemmArgs& g0, const typename Epi::Args& ea) {
    GemmArgs g = g0;
    g.prio = (c->stream && (c->stream == c->stream_dec || c->stream == c->stream_lm)) ? c->dec_prio : c->cell_prio;
    if (c->cap) {           // recorded for a pair launch (see launch_pair)
        static_assert(sizeof(GemmArgs) <= sizeof(c->cap->g) && sizeof(typename Epi::Args) <= sizeof(c->cap->ea), "Captured too small");
        lasr_ctx::Captured& k = *c->cap;
        k.fn = (const void*)&k_gemm<Ops, Epi, MT, NWV, AROW, D>;
        k.gx = (unsigned)n_groups; k.gy = (unsigned)m_groups; k.threads = NWV * 64;
        memcpy(k.g, &g, sizeof(g)); k.g_size = sizeof(g);
        memcpy(k.ea, (const void*)&ea, sizeof(ea)); k.ea_size = sizeof(ea);
        return;
    }
    hipLaunchKernelGGL((k_gemm<Ops, Epi, MT, NWV, AROW, D>), dim3(n_groups, m_groups), dim3(NWV * 64), 0, c->stream, g, ea);
}
// a recorded launch issued on its own
void replay_captured(lasr_ctx* c, lasr_ctx::Captured& k) {
    if (!k.fn) return;
    void* args[2] = {(void*)k.g, (void*)k.ea};
    (void)hipLaunchKernel(k.fn, dim3(k.gx, k.gy), dim3(k.threads), args, 0, c->stream);
    k.fn = nullptr;
}
// two recorded launches as ONE (k_gemm2) when they are the kinds the template names; otherwise one after the other
template <class Ops, class EpiA, int MTa, int NWa, bool AROWa, int Da, class EpiB, int MTb, int NWb, bool AROWb, int Db>
bool launch_pair_t(lasr_ctx* c, lasr_ctx::Captured& A, lasr_ctx::Captured& B) {
    if (A.fn != (const void*)&k_gemm<Ops, EpiA, MTa, NWa, AROWa, Da> || B.fn != (const void*)&k_gemm<Ops, EpiB, MTb, NWb, AROWb, Db>) return false;
    GemmArgs ga, gb;
    typename EpiA::Args ea; typename EpiB::Args eb;
    memcpy(&ga, A.g, sizeof(ga)); memcpy(&gb, B.g, sizeof(gb));
    memcpy((void*)&ea, A.ea, sizeof(ea)); memcpy((void*)&eb, B.ea, sizeof(eb));
    const int na = (int)(A.gx * A.gy), nb = (int)(B.gx * B.gy);
    hipLaunchKernelGGL((k_gemm2<Ops, EpiA, MTa, NWa, AROWa, Da, EpiB, MTb, NWb, AROWb, Db>), dim3(na + nb), dim3((NWa > NWb ? NWa : NWb) * 64), 0,
                       c->stream, ga, ea, (int)A.gx, na, gb, eb, (int)B.gx);
    A.fn = B.fn = nullptr;
    return true;
}

int grid1(size_t n, int b = 256) { return (int)((n + b - 1) / b); }

// lasr_trace: a value record (no event): lasr_trace_read returns `val` in the time field
// (marks come from the API thread -- main stream -- and from the pump thread -- decode stream: slots are drawn atomically)
void tr_note(lasr_ctx* c, int tag, double val) {
    if (!c->tr_on) return;
    const int i = c->tr_n.fetch_add(1, std::memory_order_relaxed);
    if (i >= lasr_ctx::NTRACE) return;
    c->tr_val[i] = val;
    c->tr_tag[i] = tag;
}
// lasr_trace: one timestamped mark on stream `st` (no-op unless tracing)
void tr_mark(lasr_ctx* c, int tag, hipStream_t st) {
    if (!c->tr_on) return;
    const int i = c->tr_n.fetch_add(1, std::memory_order_relaxed);
    if (i >= lasr_ctx::NTRACE) return;
    (void)hipEventRecord(c->tr_ev[i], st);
    c->tr_tag[i] = tag;
}

// encoder LSTM cell (layer l, step t): x from `xsrc` (fragment-major, K = I); tiling "C"
// gx_row0 >= 0 (c->enc_xg): the x side of this frame was computed by launch_enc_xg into c->gx (row gx_row0 + stream);
// the cell's K loop is the recurrent half only
template <class Ops, bool GX>
void launch_enc_cell_t(lasr_ctx* c, int l, int t, const void* xsrc, int x_mt_total, void* ydst, int y_mt_total, int gx_row0) {
    const Cell& L = c->enc[l];
    const int H = c->d.hidden;
    GemmArgs g{};
    g.A[0] = xsrc; g.a_mt_total[0] = x_mt_total; g.a_mt_off[0] = t * c->MT; g.KC[0] = GX ? 0 : L.I / Ops::KCH; g.W[0] = L.WxC;
    g.A[1] = c->enc_h[c->enc_par][l]; g.a_mt_total[1] = c->MT; g.a_mt_off[1] = 0; g.KC[1] = H / Ops::KCH; g.W[1] = L.WhC;
    g.M = c->M; g.dbg = c->dbg;
    if (c->cell_prof && c->cp_slots && c->cp_slot_next < lasr_ctx::NCELLSLOT) {
        c->cp_slot_cells[c->cp_slot_next] = 1;
        g.prof = c->cp_slots + (size_t)PROF_W * c->cp_slot_next;
        g.prof_x = g.prof + (size_t)PROF_W * lasr_ctx::NCELLSLOT;
        c->cp_slot_next++;
    }
    using E = EpiLSTM<Ops, false, false, 8, GX>;
    typename E::Args ea{};
    ea.bias = L.bias; ea.flag = c->T_row_dev; ea.t = t; ea.tile_mask = c->tile_masks.empty() ? ~0ull : c->tile_masks[t];
    ea.c = c->enc_c[l]; ea.h_in = c->enc_h[c->enc_par][l]; ea.h_out = c->enc_h[c->enc_par ^ 1][l];
    ea.y = ydst; ea.y_mt_total = y_mt_total; ea.y_mt_off = t * c->MT;
    ea.bn_s = L.bn_s; ea.bn_t = L.bn_t; ea.H = H; ea.M = c->M; ea.MT = c->MT;
    ea.gx = GX ? c->gx : nullptr; ea.gx_ld = c->gx_rows; ea.gx_row0 = gx_row0;
    // K split over 4 waves for f32 operands (12.8 us against 17.2 us with 8: fewer requests in flight, half the
    // LDS reduction), 8 waves for bf16 (5.4 us against 6.6 us); LASR_CELL_NW overrides
    const int nw = c->cell_nw ? c->cell_nw : (Ops::BF ? 8 : 4);
    if (c->enc_u12) {
        using E12 = EpiLSTMe<Ops, 12, GX>;
        typename E12::Args e12;
        static_assert(sizeof(e12) == sizeof(ea), "same Args layout");
        memcpy((void*)&e12, (const void*)&ea, sizeof(e12));
        if (c->cell_nw == 4) launch_gemm<Ops, E12, 4, false, 3, 4>(c, H / 12, c->M / 64, g, e12);
        else launch_gemm<Ops, E12, 4, false, 3, NW>(c, H / 12, c->M / 64, g, e12);
        return;
    }
    if (nw == 4) launch_gemm<Ops, E, 2, false, 3, 4>(c, H / 8, c->M / 32, g, ea);
    else launch_gemm<Ops, E, 2, false>(c, H / 8, c->M / 32, g, ea);
}
// x side of layer l for frames t0 .. t0 + Tn - 1 (rows frame-major in c->gx): ONE GEMM with the cell tiling's packed W_ih
template <class Ops>
void launch_enc_xg_t(lasr_ctx* c, int l, int t0, int Tn, const void* xsrc, int x_mt_total) {
    const Cell& L = c->enc[l];
    const int H = c->d.hidden, R = Tn * c->M;
    GemmArgs g{};
    g.A[0] = xsrc; g.a_mt_total[0] = x_mt_total; g.a_mt_off[0] = t0 * c->MT; g.KC[0] = L.I / Ops::KCH; g.W[0] = L.WxC;
    g.A[1] = nullptr; g.KC[1] = 0; g.W[1] = nullptr;
    g.M = c->M; g.dbg = nullptr;
    if (c->cell_prof && c->cp_slots && c->cp_slot_next < lasr_ctx::NCELLSLOT) {
        c->cp_slot_cells[c->cp_slot_next] = 0;            // (no cell finished by this launch: its time is spread over the frames' cells)
        g.prof = c->cp_slots + (size_t)PROF_W * c->cp_slot_next;
        g.prof_x = g.prof + (size_t)PROF_W * lasr_ctx::NCELLSLOT;
        c->cp_slot_next++;
    }
    const int nw = c->cell_nw ? c->cell_nw : (Ops::BF ? 8 : 4);
    auto fill = [&](auto& ea) {
        ea.gx = c->gx; ea.gx_ld = c->gx_rows; ea.R = R; ea.MTm = c->MT;
        for (int i = 0; i < XG_TMAX; ++i)
            ea.mask[i] = i < Tn ? (c->tile_masks.empty() ? ~0ull : c->tile_masks[t0 + i]) : 0ull;
    };
    const int mg = (R + 63) / 64;
    if (c->enc_u12) {
        typename EpiXG<12>::Args ea{};
        fill(ea);
        launch_gemm<Ops, EpiXG<12>, 4, false, 3, NW>(c, H / 12, mg, g, ea);
        return;
    }
    typename EpiXG<8>::Args ea{};
    fill(ea);
    if (nw == 4) launch_gemm<Ops, EpiXG<8>, 4, false, 3, 4>(c, H / 8, mg, g, ea);
    else launch_gemm<Ops, EpiXG<8>, 4, false, 3, NW>(c, H / 8, mg, g, ea);
}
void launch_enc_xg(lasr_ctx* c, int l, int t0, int Tn, const void* xsrc, int x_mt_total) {
    if (c->bf) launch_enc_xg_t<OpsBF16>(c, l, t0, Tn, xsrc, x_mt_total);
    else launch_enc_xg_t<OpsF32>(c, l, t0, Tn, xsrc, x_mt_total);
}
void launch_enc_cell(lasr_ctx* c, int l, int t, const void* xsrc, int x_mt_total, void* ydst, int y_mt_total, int gx_row0 = -1) {
    if (gx_row0 >= 0) {
        if (c->bf) launch_enc_cell_t<OpsBF16, true>(c, l, t, xsrc, x_mt_total, ydst, y_mt_total, gx_row0);
        else launch_enc_cell_t<OpsF32, true>(c, l, t, xsrc, x_mt_total, ydst, y_mt_total, gx_row0);
        return;
    }
    if (c->bf) launch_enc_cell_t<OpsBF16, false>(c, l, t, xsrc, x_mt_total, ydst, y_mt_total, 0);
    else launch_enc_cell_t<OpsF32, false>(c, l, t, xsrc, x_mt_total, ydst, y_mt_total, 0);
}

// One anti-diagonal of the encoder's (layer, time) grid in ONE launch: cells (l, d - l), independent of each other.
// par0 = h ping-pong parity before the pass (cell (l, t) reads parity par0 ^ (t & 1)).
struct EncCellRef { int l, t; };
template <class Ops>
void launch_enc_wave_t(lasr_ctx* c, const EncCellRef* cells, int n, int par0, int mt_total) {
    using E = EpiLSTM<Ops, false, false, 8>;
    const int H = c->d.hidden;
    MultiArgs<E> m{};
    unsigned long long* prof = nullptr;
    if (c->cell_prof && c->cp_slots && c->cp_slot_next < lasr_ctx::NCELLSLOT) {
        c->cp_slot_cells[c->cp_slot_next] = (unsigned char)n;
        prof = c->cp_slots + (size_t)PROF_W * (c->cp_slot_next++);
    }
    for (int i = 0; i < n; ++i) {
        const int l = cells[i].l, t = cells[i].t, par = par0 ^ (t & 1);
        const Cell& L = c->enc[l];
        const void* xsrc = (l == 0) ? c->x0 : c->ybuf[(l - 1) & 1];
        GemmArgs& g = m.g[i];
        g.A[0] = xsrc; g.a_mt_total[0] = mt_total; g.a_mt_off[0] = t * c->MT; g.KC[0] = L.I / Ops::KCH; g.W[0] = L.WxC;
        g.A[1] = c->enc_h[par][l]; g.a_mt_total[1] = c->MT; g.a_mt_off[1] = 0; g.KC[1] = H / Ops::KCH; g.W[1] = L.WhC;
        g.M = c->M; g.prof = prof; g.prof_x = prof ? prof + (size_t)PROF_W * lasr_ctx::NCELLSLOT : nullptr; g.prio = c->cell_prio;
        typename E::Args& ea = m.ea[i];
        ea.bias = L.bias; ea.flag = c->T_row_dev; ea.t = t; ea.tile_mask = c->tile_masks.empty() ? ~0ull : c->tile_masks[t];
        ea.c = c->enc_c[l]; ea.h_in = c->enc_h[par][l]; ea.h_out = c->enc_h[par ^ 1][l];
        ea.y = c->ybuf[l & 1]; ea.y_mt_total = mt_total; ea.y_mt_off = t * c->MT;
        ea.bn_s = L.bn_s; ea.bn_t = L.bn_t; ea.H = H; ea.M = c->M; ea.MT = c->MT;
    }
    const int nw = c->cell_nw ? c->cell_nw : (Ops::BF ? 8 : 4);
    if (c->enc_u12) {
        using E12 = EpiLSTMe<Ops, 12>;
        MultiArgs<E12> m12;
        static_assert(sizeof(m12) == sizeof(m), "same Args layout");
        memcpy((void*)&m12, (const void*)&m, sizeof(m12));
        if (c->cell_nw == 4) hipLaunchKernelGGL((k_gemm_multi<Ops, E12, 4, 4, false, 3>), dim3(H / 12, c->M / 64, n), dim3(256), 0, c->stream, m12);
        else hipLaunchKernelGGL((k_gemm_multi<Ops, E12, 4, NW, false, 3>), dim3(H / 12, c->M / 64, n), dim3(NW * 64), 0, c->stream, m12);
        return;
    }
    const dim3 grid(H / 8, c->M / 32, n);
    if (nw == 4) hipLaunchKernelGGL((k_gemm_multi<Ops, E, 2, 4, false, 3>), grid, dim3(256), 0, c->stream, m);
    else hipLaunchKernelGGL((k_gemm_multi<Ops, E, 2, NW, false, 3>), grid, dim3(NW * 64), 0, c->stream, m);
}
void launch_enc_wave(lasr_ctx* c, const EncCellRef* cells, int n, int par0, int mt_total) {
    if (c->bf) launch_enc_wave_t<OpsBF16>(c, cells, n, par0, mt_total);
    else launch_enc_wave_t<OpsF32>(c, cells, n, par0, mt_total);
}

// beam search: the slots that are not extended are carried by k_beam_carry instead of the GEMM epilogues (LASR_BEAM_CARRY=0: as in round 3)
// LASR_BEAM_CARRY: 0 = non-extended hypothesis slots carried inside the cell / joint kernels' epilogues (round 3), 1 = by a launch of
// their own (k_beam_carry), 2 = as extra workgroups of the joint-half GEMM's launch (k_gemm_carry)
int beam_carry_mode() {
    static const int v = getenv("LASR_BEAM_CARRY") ? atoi(getenv("LASR_BEAM_CARRY")) : 2;
    return v;
}
bool beam_carry_on() { return beam_carry_mode() != 0; }
void fill_beam_carry(lasr_ctx* c, BeamCarryArgs& a) {
    const int H = c->d.hidden, p = c->pred_par;
    a.emit = c->ds.emit; a.parent = c->b_parent; a.W = c->W; a.Md = c->Md; a.H = H; a.J = c->d.joint; a.Lp = c->d.pred_layers;
    a.bf = c->bf; a.lstm = c->d.pred_cell;
    for (int l = 0; l < a.Lp; ++l) {
        a.h_in[l] = c->pred_h[p][l]; a.h_out[l] = c->pred_h[p ^ 1][l];
        a.y_in[l] = p ? c->pred_y1[l] : c->pred_y[l]; a.y_out[l] = p ? c->pred_y[l] : c->pred_y1[l];
        if (a.lstm) { a.c_in[l] = p ? c->pred_c1[l] : c->pred_c[l]; a.c_out[l] = p ? c->pred_c[l] : c->pred_c1[l]; }
    }
    a.pp_in = p ? c->pp1 : c->pp; a.pp_out = p ? c->pp : c->pp1;
    a.pe = c->pe; a.t_idx = c->dec_t_idx; a.T_row = c->T_row_dec; a.ja = c->ja; a.MTj = c->MTj; a.ring = c->pe_ring_R; a.M_enc = c->M;
}
// carry blocks of a launch: Md slot blocks + (LSTM predictor) the cell-state blocks
int beam_carry_blocks(lasr_ctx* c) {
    return c->Md + (c->d.pred_cell ? ((c->d.hidden + 15) / 16) * ((c->Md + 255) / 256) : 0);
}
// a GEMM launch whose grid carries the round's carry blocks behind its own m-groups (see k_gemm_carry)
template <class Ops, class Epi, int MT, bool AROW, int D = 3, int NWV = NW>
void launch_gemm_carry(lasr_ctx* c, int n_groups, int m_groups, const GemmArgs& g0, const typename Epi::Args& ea) {
    GemmArgs g = g0;
    g.prio = (c->stream && (c->stream == c->stream_dec || c->stream == c->stream_lm)) ? c->dec_prio : c->cell_prio;
    BeamCarryArgs ca{};
    fill_beam_carry(c, ca);
    const int extra = (beam_carry_blocks(c) + n_groups - 1) / n_groups;
    hipLaunchKernelGGL((k_gemm_carry<Ops, Epi, MT, NWV, AROW, D>), dim3(n_groups, m_groups + extra), dim3(NWV * 64), 0, c->stream, g, ea, ca, m_groups);
}

// one predictor pass (all layers) for rows with emit != 0 (compacted inside the kernels); predictor
// state is row-major [M][H]; toggles pred_par
// (l0, l1: layers [l0, l1) of the pass -- the pair launches of cont_enqueue issue a pass layer by layer; the parity flips with the last one)
template <class Ops>
void launch_predictor_t(lasr_ctx* c, bool beam, int l0 = 0, int l1 = -1) {
    const int H = c->d.hidden;
    if (l1 < 0) l1 = c->d.pred_layers;
    const int mgroups = c->Md / (16 * MTA);
    const int p = c->pred_par;
    // many decoder rows (beam 8 x 64+ streams, >= 512 streams): 16-unit workgroups, a quarter of the activation traffic
    // (configs[4], 1024 rows: predictor cells 135 -> ~50 us, whole job +60 %; at 256 rows: bf16 equal, f32 -22 %; at 64: -20 %)
    const bool wide = c->Md >= 512;
    const bool wide8 = c->bf && c->Md >= 256 && c->Md < 512;   // 8 units per workgroup, 8 waves (configs[2]: 6.4 -> 7.2 k in round 2)
    const bool split_carry = beam && beam_carry_on();
    if (split_carry && beam_carry_mode() == 1 && l0 == 0) {      // the slots that are not extended: whole-row copies by their own launch (see k_beam_carry)
        BeamCarryArgs a{};
        fill_beam_carry(c, a);
        hipLaunchKernelGGL(k_beam_carry, dim3(std::max(c->Md, ((H + 15) / 16) * ((c->Md + 255) / 256)), 2), dim3(256), 0, c->stream, a);
    }                                                   // (mode 2: the carry rides in launch_ppj's launch of the same pass)
    for (int l = l0; l < l1; ++l) {
        const Cell& L = c->pred[l];
        GemmArgs g{};
        g.skip_idle = split_carry ? 1 : 0;
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
            ea.no_carry = split_carry ? 1 : 0;
            if (l == 0) {
                if (wide8) launch_gemm<Ops, EpiLSTMw<Ops, true, 2>, MTA, true, -1>(c, H / 8, mgroups, g, ea);
                else if (wide) launch_gemm<Ops, EpiLSTMw<Ops, true>, MTA, true, -1, 4>(c, H / 16, mgroups, g, ea);
                else if (c->dec_nw_mask & 1) launch_gemm<Ops, EpiLSTM<Ops, true, true, 4>, MTA, true, -1, 4>(c, H / 4, mgroups, g, ea); else launch_gemm<Ops, EpiLSTM<Ops, true, true, 4>, MTA, true, -1>(c, H / 4, mgroups, g, ea);
            } else {
                typename EpiLSTM<Ops, true, false, 4>::Args eb{};
                static_assert(sizeof(eb) == sizeof(ea), "same Args layout");
                memcpy(&eb, &ea, sizeof(eb));
                if (wide8) launch_gemm<Ops, EpiLSTMw<Ops, false, 2>, MTA, true, -1>(c, H / 8, mgroups, g, eb);
                else if (wide) launch_gemm<Ops, EpiLSTMw<Ops, false>, MTA, true, -1, 4>(c, H / 16, mgroups, g, eb);
                else if (c->dec_nw_mask & 1) launch_gemm<Ops, EpiLSTM<Ops, true, false, 4>, MTA, true, -1, 4>(c, H / 4, mgroups, g, eb); else launch_gemm<Ops, EpiLSTM<Ops, true, false, 4>, MTA, true, -1>(c, H / 4, mgroups, g, eb);
            }
        } else {
            typename EpiNBRC<Ops, true>::Args ea{};
            ea.bias = L.bias; ea.rbias = L.rbias; ea.tab = L.tab; ea.token = c->ds.token; ea.emit = c->ds.emit;
            ea.h_in = c->pred_h[p][l]; ea.h_out = c->pred_h[p ^ 1][l]; ea.y = y_out;
            ea.bn_s = L.bn_s; ea.bn_t = L.bn_t; ea.H = H; ea.M = c->Md;
            if (beam) { ea.parent = c->b_parent; ea.W = c->W; ea.y_in = y_in; }
            ea.no_carry = split_carry ? 1 : 0;
            if (l == 0) {
                if (wide8) launch_gemm<Ops, EpiNBRCw<Ops, true, 2>, MTA, true, -1>(c, H / 8, mgroups, g, ea);
                else if (wide) launch_gemm<Ops, EpiNBRCw<Ops, true>, MTA, true, -1, 4>(c, H / 16, mgroups, g, ea);
                else if (c->dec_nw_mask & 1) launch_gemm<Ops, EpiNBRC<Ops, true>, MTA, true, -1, 4>(c, H / 4, mgroups, g, ea); else launch_gemm<Ops, EpiNBRC<Ops, true>, MTA, true, -1>(c, H / 4, mgroups, g, ea);
            } else {
                typename EpiNBRC<Ops, false>::Args eb{};
                static_assert(sizeof(eb) == sizeof(ea), "same Args layout");
                memcpy(&eb, &ea, sizeof(eb));
                if (wide8) launch_gemm<Ops, EpiNBRCw<Ops, false, 2>, MTA, true, -1>(c, H / 8, mgroups, g, eb);
                else if (wide) launch_gemm<Ops, EpiNBRCw<Ops, false>, MTA, true, -1, 4>(c, H / 16, mgroups, g, eb);
                else if (c->dec_nw_mask & 1) launch_gemm<Ops, EpiNBRC<Ops, false>, MTA, true, -1, 4>(c, H / 4, mgroups, g, eb); else launch_gemm<Ops, EpiNBRC<Ops, false>, MTA, true, -1>(c, H / 4, mgroups, g, eb);
            }
        }
    }
    if (!beam && l1 == c->d.pred_layers) c->pred_par ^= 1;      // beam: launch_ppj (same pass, same parities) toggles
}
void launch_predictor(lasr_ctx* c, bool beam = false, int l0 = 0, int l1 = -1) {
    if (c->bf) launch_predictor_t<OpsBF16>(c, beam, l0, l1);
    else launch_predictor_t<OpsF32>(c, beam, l0, l1);
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
    if (beam && beam_carry_on()) { ea.no_carry = 1; g.skip_idle = 1; }      // (k_beam_carry, launched with the predictor pass)
    const bool ppj_wide = c->Md >= 512 && c->MTd % 4 == 0;   // 64-row workgroups for many decoder rows (64-column ones measured slower:
                                                             // 19.9 against 14.3 us at 1024 rows, round 4)
    if (beam && beam_carry_mode() == 2) {      // the round's carry as extra workgroups of this launch
        if (ppj_wide) launch_gemm_carry<Ops, EpiPPJ<Ops>, 4, true, -1, 4>(c, J / 16, c->MTd / 4, g, ea);
        else if (c->dec_nw_mask & 2) launch_gemm_carry<Ops, EpiPPJ<Ops>, 1, true, -1, 4>(c, J / 16, c->MTd, g, ea);
        else launch_gemm_carry<Ops, EpiPPJ<Ops>, 1, true, -1>(c, J / 16, c->MTd, g, ea);
    } else
    if (ppj_wide) launch_gemm<Ops, EpiPPJ<Ops>, 4, true, -1, 4>(c, J / 16, c->MTd / 4, g, ea);
    else if (c->dec_nw_mask & 2) launch_gemm<Ops, EpiPPJ<Ops>, 1, true, -1, 4>(c, J / 16, c->MTd, g, ea); else launch_gemm<Ops, EpiPPJ<Ops>, 1, true, -1>(c, J / 16, c->MTd, g, ea);
    if (beam) c->pred_par ^= 1;
}
void launch_ppj(lasr_ctx* c, bool beam = false) {
    if (c->bf) launch_ppj_t<OpsBF16>(c, beam);
    else launch_ppj_t<OpsF32>(c, beam);
}
float* cur_pp(lasr_ctx* c) { return (c->W > 1 && c->pred_par) ? c->pp1 : c->pp; }

template <bool AROW, int D>
void launch_linear(lasr_ctx* c, int n_groups, int m_groups, GemmArgs g, int K, const EpiLinear::Args& ea);

// k_lm_post / k_beam_fuse with the register slots their vocabulary needs (bit-identical either way, see k_lm_post)
inline bool keep16(int V) {
    static const int force = getenv("LASR_KEEP16") ? atoi(getenv("LASR_KEEP16")) : 0;
    return force || V > 2048;
}
#define LAUNCH_LM_POST(V_, ...) do { if (keep16(V_)) hipLaunchKernelGGL(k_lm_post<16>, __VA_ARGS__); else hipLaunchKernelGGL(k_lm_post<8>, __VA_ARGS__); } while (0)
#define LAUNCH_BEAM_FUSE(V_, ...) do { if (keep16(V_)) hipLaunchKernelGGL(k_beam_fuse<16>, __VA_ARGS__); else hipLaunchKernelGGL(k_beam_fuse<8>, __VA_ARGS__); } while (0)

// LMFuser.advance (lm.py:49-53) for the rows with emit != 0: LM step on the token just emitted, then
// log_softmax + standardise + [0] = MIN_VAL into lmz (read by the next k_select of that row)
// (l0, l1: LSTM layers [l0, l1) of the step; the output layer, k_lm_post and the parity flip come with the last one unless tail = false)
template <class Ops>
void launch_lm_t(lasr_ctx* c, bool beam, int l0 = 0, int l1 = -1, bool tail = true) {
    lasr_ctx::LM& m = c->lm;
    const int H = m.H, V = c->d.vocab, p = m.par;
    const int R = beam ? c->Md : c->M;                   // LM rows: streams, or hypothesis slots (beam: parity p -> p ^ 1, parent-indirected)
    if (l1 < 0) l1 = m.L;
    for (int l = l0; l < l1; ++l) {
        const Cell& L = m.cells[l];
        GemmArgs g{};
        void* y_out = (beam && !p) ? m.y1[l] : m.y[l];
        const void* y_in = (beam && p) ? m.y1[l] : m.y[l];
        if (l > 0) { g.A[0] = (beam && !p) ? m.y1[l - 1] : m.y[l - 1]; g.a_mt_total[0] = H; g.a_mt_off[0] = 0; g.KC[0] = H / Ops::KCH; g.W[0] = L.WxA; }
        g.A[1] = m.h[p][l]; g.a_mt_total[1] = H; g.a_mt_off[1] = 0; g.KC[1] = H / Ops::KCH; g.W[1] = L.WhA;
        if (beam) { g.parent = c->b_parent; g.beam_w = c->W; }
        g.compact = c->ds.emit; g.M = R;
        typename EpiLSTM<Ops, true, true, 4>::Args ea{};
        ea.bias = L.bias; ea.tab = L.tab; ea.token = c->ds.token; ea.flag = c->ds.emit; ea.t = 0;
        ea.c = (beam && !p) ? m.cst1[l] : m.cst[l]; ea.h_in = m.h[p][l]; ea.h_out = m.h[p ^ 1][l]; ea.y = y_out;
        ea.bn_s = m.ones; ea.bn_t = m.zeros; ea.H = H; ea.M = R; ea.MT = R / 16;
        if (beam) { ea.parent = c->b_parent; ea.W = c->W; ea.c_in = p ? m.cst1[l] : m.cst[l]; ea.y_in = y_in; }
        if (l == 0) {
            launch_gemm<Ops, EpiLSTM<Ops, true, true, 4>, MTA, true, -1>(c, H / 4, R / (16 * MTA), g, ea);
        } else {
            typename EpiLSTM<Ops, true, false, 4>::Args eb{};
            memcpy(&eb, &ea, sizeof(eb));
            launch_gemm<Ops, EpiLSTM<Ops, true, false, 4>, MTA, true, -1>(c, H / 4, R / (16 * MTA), g, eb);
        }
    }
    if (l1 < m.L || !tail) return;
    GemmArgs g{};
    g.A[0] = (beam && !p) ? m.y1[m.L - 1] : m.y[m.L - 1]; g.a_mt_total[0] = H; g.a_mt_off[0] = 0; g.W[0] = m.Wout; g.a_rows = R;
    EpiLinear::Args ea{};
    ea.bias = m.bout; ea.out = m.raw; ea.ldo = V; ea.n_rows = R; ea.t_idx = nullptr; ea.T_row = nullptr; ea.M = R;
    launch_linear<true, -1>(c, V / 16, R / 16, g, H, ea);
    if (beam)
        LAUNCH_LM_POST(V, dim3(R), dim3(256), 0, c->stream, (const float*)m.raw, (const int*)c->ds.emit, p ? m.lmz : m.lmz1,
                           p ? m.valid : m.valid1, V, m.min_val, (const int*)c->b_parent, c->W, (const float*)(p ? m.lmz1 : m.lmz),
                           (const int*)(p ? m.valid1 : m.valid));
    else
        LAUNCH_LM_POST(V, dim3(R), dim3(256), 0, c->stream, (const float*)m.raw, (const int*)c->ds.emit, m.lmz, m.valid, V, m.min_val,
                           (const int*)nullptr, 1, (const float*)m.lmz, (const int*)m.valid);
    m.par ^= 1;
}
// int8-served LM step (see k_lm_quant): per layer  quantise -> GEMV -> dequantise  for the x side (layers > 0) and the h side,
// element-wise cell for the rows that emitted; then the output layer the same way and k_lm_post.  The GEMVs run for all M
// rows (the non-emitting rows' results are dropped by the cell kernel): this path is about arithmetic parity, not speed.
void lm_q_gemv(lasr_ctx* c, const float* src, int lds, int K, int Kp, const void* Wq, float w_scale, const float* bias, float* out,
               int N, int rows, unsigned short* qa, float* sx) {
    hipLaunchKernelGGL(k_lm_quant, dim3(rows), dim3(256), 0, c->stream, src, lds, K, qa, Kp, sx);
    GemmArgs g{};
    g.A[0] = qa; g.a_mt_total[0] = Kp; g.a_mt_off[0] = 0; g.KC[0] = Kp / 32; g.W[0] = Wq; g.a_rows = rows;
    EpiLinear::Args ea{};
    ea.bias = bias; ea.out = out; ea.ldo = N; ea.n_rows = rows; ea.t_idx = nullptr; ea.T_row = nullptr; ea.M = c->M;
    ea.row_scale = sx; ea.w_scale = w_scale;
    launch_gemm<OpsBF16, EpiLinear, 1, true, -1>(c, N / 16, (rows + 15) / 16, g, ea);
}
// the integer GEMV + dequantisation on an input that is already quantised (qh / sxh of a layer's h, kept by k_lm_cell_q)
void lm_q_gemv_pre(lasr_ctx* c, const unsigned short* qa, const float* sx, int Kp, const void* Wq, float w_scale, const float* bias,
                   float* out, int N, int rows) {
    GemmArgs g{};
    g.A[0] = qa; g.a_mt_total[0] = Kp; g.a_mt_off[0] = 0; g.KC[0] = Kp / 32; g.W[0] = Wq; g.a_rows = rows;
    EpiLinear::Args ea{};
    ea.bias = bias; ea.out = out; ea.ldo = N; ea.n_rows = rows; ea.t_idx = nullptr; ea.T_row = nullptr; ea.M = c->M;
    ea.row_scale = sx; ea.w_scale = w_scale;
    launch_gemm<OpsBF16, EpiLinear, 1, true, -1>(c, N / 16, (rows + 15) / 16, g, ea);
}
void launch_lm_q8(lasr_ctx* c) {
    lasr_ctx::LM& m = c->lm;
    const int H = m.H, M = c->M, V = c->d.vocab;
    for (int l = 0; l < m.L; ++l) {
        // x side: the quantised image of the layer below's NEW h (its cell kernel has just written it); h side: this layer's own
        if (l > 0) lm_q_gemv_pre(c, m.qh[l - 1], m.sxh[l - 1], m.Kp_h, m.qWih[l], m.s_ih[l], m.b_ih[l], m.gx, 4 * H, M);
        lm_q_gemv_pre(c, m.qh[l], m.sxh[l], m.Kp_h, m.qWhh[l], m.s_hh[l], m.b_hh[l], m.gh, 4 * H, M);
        hipLaunchKernelGGL(k_lm_cell_q, dim3(M), dim3(256), 0, c->stream, (const float*)m.gx, (const float*)(l == 0 ? m.cells[0].tab : nullptr),
                           (const int*)c->ds.token, (const float*)m.gh, (const int*)c->ds.emit, (float*)m.h[0][l], m.cst[l], H, M,
                           m.qh[l], m.sxh[l], m.Kp_h);
    }
    m.par ^= 1;
    lm_q_gemv_pre(c, m.qh[m.L - 1], m.sxh[m.L - 1], m.Kp_h, m.qWout, m.s_out, m.bout, m.raw, V, M);
    LAUNCH_LM_POST(V, dim3(M), dim3(256), 0, c->stream, (const float*)m.raw, (const int*)c->ds.emit, m.lmz, m.valid, V, m.min_val,
                       (const int*)nullptr, 1, (const float*)m.lmz, (const int*)m.valid);
}
void launch_lm(lasr_ctx* c, bool beam = false, int l0 = 0, int l1 = -1, bool tail = true) {
    if (!c->lm.on) return;
    if (c->lm.q8) { launch_lm_q8(c); return; }
    if (c->bf) launch_lm_t<OpsBF16>(c, beam, l0, l1, tail);
    else launch_lm_t<OpsF32>(c, beam, l0, l1, tail);
}
// Pair launches of the greedy loop with an fp32 / bf16 LM (cont_enqueue): stage `kind` of the predictor / joint chain (0, 1: NBRC
// layers 0, 1; 2: joint half; 3: the next iteration's logits GEMM), recorded in A, with LM layer l (0: through the token table),
// recorded in B.  The kinds the templates name are the ones configs[1] runs (2 x NBRC predictor, decode GEMMs on 4 waves with f32
// operands and 8 with bf16); anything else is issued one after the other.
void launch_pair(lasr_ctx* c, int kind, bool lm_first, lasr_ctx::Captured& A, lasr_ctx::Captured& B) {
    bool ok = false;
    if (!c->bf) {
        using LT = EpiLSTM<OpsF32, true, true, 4>; using LF = EpiLSTM<OpsF32, true, false, 4>;
        if (kind == 0 && lm_first) ok = launch_pair_t<OpsF32, EpiNBRC<OpsF32, true>, MTA, 4, true, -1, LT, MTA, NW, true, -1>(c, A, B);
        else if (kind == 1 && !lm_first) ok = launch_pair_t<OpsF32, EpiNBRC<OpsF32, false>, MTA, 4, true, -1, LF, MTA, NW, true, -1>(c, A, B);
        else if (kind == 2 && !lm_first) ok = launch_pair_t<OpsF32, EpiPPJ<OpsF32>, 1, 4, true, -1, LF, MTA, NW, true, -1>(c, A, B);
        else if (kind == 3 && !lm_first) ok = launch_pair_t<OpsF32, EpiLinear, 2, 4, false, -1, LF, MTA, NW, true, -1>(c, A, B);
    } else {
        using LT = EpiLSTM<OpsBF16, true, true, 4>; using LF = EpiLSTM<OpsBF16, true, false, 4>;
        if (kind == 0 && lm_first) ok = launch_pair_t<OpsBF16, EpiNBRC<OpsBF16, true>, MTA, NW, true, -1, LT, MTA, NW, true, -1>(c, A, B);
        else if (kind == 1 && !lm_first) ok = launch_pair_t<OpsBF16, EpiNBRC<OpsBF16, false>, MTA, NW, true, -1, LF, MTA, NW, true, -1>(c, A, B);
        else if (kind == 2 && !lm_first) ok = launch_pair_t<OpsBF16, EpiPPJ<OpsBF16>, 1, NW, true, -1, LF, MTA, NW, true, -1>(c, A, B);
        else if (kind == 3 && !lm_first) ok = launch_pair_t<OpsBF16, EpiLinear, 2, NW, false, -1, LF, MTA, NW, true, -1>(c, A, B);
    }
    if (!ok) { replay_captured(c, A); replay_captured(c, B); }
}
// current-parity LM output of the hypothesis slots (beam): parity 0 = lmz / valid, parity 1 = lmz1 / valid1
const float* cur_lmz(lasr_ctx* c) { return (c->W > 1 && c->lm.par) ? c->lm.lmz1 : c->lm.lmz; }
const int* cur_lm_valid(lasr_ctx* c) { return (c->W > 1 && c->lm.par) ? c->lm.valid1 : c->lm.valid; }
// the selection kernel of one beam round (+ the LM re-pick of the extended slots' tokens)
void launch_beam_select(lasr_ctx* c, BeamState& b, int iter_slot) {
    const int M = c->M;
    b.lm_on = c->lm.on ? 1 : 0; b.done2 = c->c_done2;
    const bool small = c->d.vocab <= 2048;      // 512 threads fill their register slots with real logits (1024: half padding; round 3, P)
    const float* lg = (const float*)c->logits;
    // one wave per hypothesis row (k_beam_select_rw; LASR_BEAM_SELECT_RW=0: every row spread over all waves, round 3's kernel)
    static const int rw_env = getenv("LASR_BEAM_SELECT_RW") ? atoi(getenv("LASR_BEAM_SELECT_RW")) : 1;
    if (small && rw_env) {
        if (c->W <= 2) hipLaunchKernelGGL((k_beam_select_rw<2>), dim3(M), dim3(128), 0, c->stream, lg, b, iter_slot);
        else if (c->W <= 4) hipLaunchKernelGGL((k_beam_select_rw<4>), dim3(M), dim3(256), 0, c->stream, lg, b, iter_slot);
        else hipLaunchKernelGGL((k_beam_select_rw<8>), dim3(M), dim3(512), 0, c->stream, lg, b, iter_slot);
    } else
    if (small) {
        if (c->W <= 2) hipLaunchKernelGGL((k_beam_select<2, 512>), dim3(M), dim3(512), 0, c->stream, lg, b, iter_slot);
        else if (c->W <= 4) hipLaunchKernelGGL((k_beam_select<4, 512>), dim3(M), dim3(512), 0, c->stream, lg, b, iter_slot);
        else hipLaunchKernelGGL((k_beam_select<8, 512>), dim3(M), dim3(512), 0, c->stream, lg, b, iter_slot);
    } else {
        if (c->W <= 2) hipLaunchKernelGGL((k_beam_select<2, 1024>), dim3(M), dim3(1024), 0, c->stream, lg, b, iter_slot);
        else if (c->W <= 4) hipLaunchKernelGGL((k_beam_select<4, 1024>), dim3(M), dim3(1024), 0, c->stream, lg, b, iter_slot);
        else hipLaunchKernelGGL((k_beam_select<8, 1024>), dim3(M), dim3(1024), 0, c->stream, lg, b, iter_slot);
    }
    if (c->lm.on)
        LAUNCH_BEAM_FUSE(c->d.vocab, dim3(c->Md), dim3(256), 0, c->stream, (const float*)c->logits, b, iter_slot, cur_lmz(c), cur_lm_valid(c),
                           c->lm.alpha, c->lm.theta, c->lm.min_val);
}

// plain linear over element-typed A (fragment-major, or row-major when AROW); f32 row-major output
template <bool AROW, int D>
void launch_linear(lasr_ctx* c, int n_groups, int m_groups, GemmArgs g, int K, const EpiLinear::Args& ea) {
    g.KC[0] = K / c->kch;
    if (c->dec_nw_mask & 4) {
        if (c->bf) launch_gemm<OpsBF16, EpiLinear, 1, AROW, D, 4>(c, n_groups, m_groups, g, ea);
        else launch_gemm<OpsF32, EpiLinear, 1, AROW, D, 4>(c, n_groups, m_groups, g, ea);
        return;
    }
    if (c->bf) launch_gemm<OpsBF16, EpiLinear, 1, AROW, D>(c, n_groups, m_groups, g, ea);
    else launch_gemm<OpsF32, EpiLinear, 1, AROW, D>(c, n_groups, m_groups, g, ea);
}

// vocabulary projection of the joint for n_rows rows of ja.  m-tiles per workgroup (c->logits_mt): 1 = a 16-row x
// 16-column tile per workgroup (every m-tile re-reads the workgroup's 64 KB of W2 from L2); 2 / 4 = 32 / 64 rows per
// workgroup, W2 fragments fetched once per 2 / 4 m-tiles -- what a lookahead pass (la x M rows) wants
template <int MTL>
void launch_logits_t(lasr_ctx* c, const GemmArgs& g0, int n_rows, int K, const EpiLinear::Args& ea) {
    GemmArgs g = g0;
    g.KC[0] = K / c->kch;
    const int ng = c->d.vocab / 16, mg = (n_rows + 16 * MTL - 1) / (16 * MTL);
    if (c->dec_nw_mask & 4) {
        if (c->bf) launch_gemm<OpsBF16, EpiLinear, MTL, false, -1, 4>(c, ng, mg, g, ea);
        else launch_gemm<OpsF32, EpiLinear, MTL, false, -1, 4>(c, ng, mg, g, ea);
        return;
    }
    if (c->bf) launch_gemm<OpsBF16, EpiLinear, MTL, false, -1>(c, ng, mg, g, ea);
    else launch_gemm<OpsF32, EpiLinear, MTL, false, -1>(c, ng, mg, g, ea);
}
void launch_logits(lasr_ctx* c, float* out, int n_rows, bool gated) {
    const int J = c->d.joint, V = c->d.vocab;
    GemmArgs g{};
    g.A[0] = c->ja; g.a_mt_total[0] = c->MTj; g.a_mt_off[0] = 0; g.W[0] = c->W2; g.M = c->Md;
    g.dbg = (c->dbg && c->dbg_gate) ? c->dbg + (size_t)4 * 4096 * 16 : nullptr;
    EpiLinear::Args ea{};
    ea.bias = c->b2; ea.out = out; ea.ldo = V; ea.n_rows = n_rows;
    ea.t_idx = gated ? c->dec_t_idx : nullptr; ea.T_row = c->T_row_dec; ea.M = c->M; ea.W = c->W;
    if (n_rows >= 512 && V % 64 == 0) {      // 64 x 64 workgroups for the beam's hundreds of hypothesis rows (round 4: logits 28 -> 20 us)
        GemmArgs g4 = g;
        g4.KC[0] = J / c->kch;
        EpiLinearT<4>::Args e4{};
        static_assert(sizeof(e4) == sizeof(ea), "same Args layout");
        memcpy((void*)&e4, (const void*)&ea, sizeof(e4));
        if (c->bf) launch_gemm<OpsBF16, EpiLinearT<4>, 4, false, -1, 4>(c, V / 64, (n_rows + 63) / 64, g4, e4);
        else launch_gemm<OpsF32, EpiLinearT<4>, 4, false, -1, 4>(c, V / 64, (n_rows + 63) / 64, g4, e4);
        return;
    }
    if (c->logits_mt == 4 || (c->logits_mt == 2 && n_rows >= 512)) { launch_logits_t<4>(c, g, n_rows, J, ea); return; }
    if (c->logits_mt == 2) { launch_logits_t<2>(c, g, n_rows, J, ea); return; }
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
