// lasr_launch_dec.hip.h -- decode-side GEMM launches (predictor cells, joint half, logits, LM step, pair launches), templates over
// the operand type; included by lasr_launch_dec_f32.hip and lasr_launch_dec_bf16.hip, which instantiate them (LASR_DECL_OPS).
#pragma once

// two recorded launches as ONE (k_gemm2) when they are the kinds the template names; otherwise one after the other
template <class Ops, class EpiA, int MTa, int NWa, bool AROWa, int Da, class EpiB, int MTb, int NWb, bool AROWb, int Db>
static bool launch_pair_t(lasr_ctx* c, lasr_ctx::Captured& A, lasr_ctx::Captured& B) {
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


static void fill_beam_carry(lasr_ctx* c, BeamCarryArgs& a) {
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
static int beam_carry_blocks(lasr_ctx* c) {
    return c->Md + (c->d.pred_cell ? ((c->d.hidden + 15) / 16) * ((c->Md + 255) / 256) : 0);
}
// a GEMM launch whose grid carries the round's carry blocks behind its own m-groups (see k_gemm_carry)
template <class Ops, class Epi, int MT, bool AROW, int D = 3, int NWV = NW>
static void launch_gemm_carry(lasr_ctx* c, int n_groups, int m_groups, const GemmArgs& g0, const typename Epi::Args& ea) {
    GemmArgs g = g0;
    g.prio = (c->stream && (c->stream == c->stream_dec || c->stream == c->stream_lm)) ? c->dec_prio : c->cell_prio;
    BeamCarryArgs ca{};
    fill_beam_carry(c, ca);
    const int extra = (beam_carry_blocks(c) + n_groups - 1) / n_groups;
    hipLaunchKernelGGL((k_gemm_carry<Ops, Epi, MT, NWV, AROW, D>), dim3(n_groups, m_groups + extra), dim3(NWV * 64), 0, c->stream, g, ea, ca, m_groups);
}


// plain linear over element-typed A (fragment-major, or row-major when AROW); f32 row-major output
template <class Ops, bool AROW, int D>
void launch_linear_ops(lasr_ctx* c, int n_groups, int m_groups, GemmArgs g, int K, const EpiLinear::Args& ea) {
    g.KC[0] = K / Ops::KCH;
    launch_gemm<Ops, EpiLinear, 1, AROW, D>(c, n_groups, m_groups, g, ea);
}

// vocabulary projection of the joint for n_rows rows of ja.  m-tiles per workgroup (c->logits_mt): 1 = a 16-row x
// 16-column tile per workgroup (every m-tile re-reads the workgroup's 64 KB of W2 from L2); 2 / 4 = 32 / 64 rows per
// workgroup, W2 fragments fetched once per 2 / 4 m-tiles -- what a lookahead pass (la x M rows) wants
template <class Ops, int MTL>
static void launch_logits_t(lasr_ctx* c, const GemmArgs& g0, int n_rows, int K, const EpiLinear::Args& ea) {
    GemmArgs g = g0;
    g.KC[0] = K / Ops::KCH;
    const int ng = c->d.vocab / 16, mg = (n_rows + 16 * MTL - 1) / (16 * MTL);
    launch_gemm<Ops, EpiLinear, MTL, false, -1>(c, ng, mg, g, ea);
}
template <class Ops>
void launch_logits_ops(lasr_ctx* c, float* out, int n_rows, bool gated) {
    const int J = c->d.joint, V = c->d.vocab;
    GemmArgs g{};
    g.A[0] = c->ja; g.a_mt_total[0] = c->MTj; g.a_mt_off[0] = 0; g.W[0] = c->W2; g.M = c->Md;
    g.dbg = (c->dbg && c->dbg_gate) ? c->dbg + (size_t)4 * 4096 * 16 : nullptr;
    EpiLinear::Args ea{};
    ea.bias = c->b2; ea.out = out; ea.ldo = V; ea.n_rows = n_rows;
    ea.t_idx = gated ? c->dec_t_idx : nullptr; ea.T_row = c->T_row_dec; ea.M = c->M; ea.W = c->W;
    if (n_rows >= 512 && V % 64 == 0) {      // 64 x 64 workgroups for the beam's hundreds of hypothesis rows (round 4: logits 28 -> 20 us)
        GemmArgs g4 = g;
        g4.KC[0] = J / Ops::KCH;
        EpiLinearT<4>::Args e4{};
        static_assert(sizeof(e4) == sizeof(ea), "same Args layout");
        memcpy((void*)&e4, (const void*)&ea, sizeof(e4));
        launch_gemm<typename WideOps<Ops>::type, EpiLinearT<4>, 4, false, -1, 4>(c, V / 64, (n_rows + 63) / 64, g4, e4);
        return;
    }
    if (c->logits_mt == 4 || (c->logits_mt == 2 && n_rows >= 512)) { launch_logits_t<Ops, 4>(c, g, n_rows, J, ea); return; }
    if (c->logits_mt == 2) { launch_logits_t<Ops, 2>(c, g, n_rows, J, ea); return; }
    launch_linear_ops<Ops, false, -1>(c, V / 16, (n_rows + 15) / 16, g, J, ea);
}

// launch of a wide tiling through its own operand type (WideOps): the epilogue's Args are the same struct under another template
// argument -- copied bit for bit
template <class OW, class Epi, class ArgsIn>
static void launch_wide(lasr_ctx* c, int n_groups, int m_groups, const GemmArgs& g, const ArgsIn& ea_in) {
    typename Epi::Args ea;
    static_assert(sizeof(ea) == sizeof(ea_in), "same Args layout");
    memcpy((void*)&ea, (const void*)&ea_in, sizeof(ea));
    launch_gemm<OW, Epi, MTA, true, -1, 4>(c, n_groups, m_groups, g, ea);
}

// one predictor pass (all layers) for rows with emit != 0 (compacted inside the kernels); predictor
// state is row-major [M][H]; toggles pred_par
// (l0, l1: layers [l0, l1) of the pass -- the pair launches of cont_enqueue issue a pass layer by layer; the parity flips with the last one)
template <class Ops>
void launch_predictor_t(lasr_ctx* c, bool beam, int l0, int l1) {
    const int H = c->d.hidden;
    if (l1 < 0) l1 = c->d.pred_layers;
    const int mgroups = c->Md / (16 * MTA);
    const int p = c->pred_par;
    // many decoder rows (beam 8 x 64+ streams, >= 512 streams): 16-unit workgroups, a quarter of the activation traffic
    // (configs[4], 1024 rows: predictor cells 135 -> ~50 us, whole job +60 %; at 256 rows: bf16 equal, f32 -22 %; at 64: -20 %)
    const bool wide = c->Md >= 512;
    using OW = typename WideOps<Ops>::type;            // the wide tilings' matrix instruction (see OpsBF16k16)
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
                else if (wide) launch_wide<OW, EpiLSTMw<OW, true>>(c, H / 16, mgroups, g, ea);
                else launch_gemm<Ops, EpiLSTM<Ops, true, true, 4>, MTA, true, -1>(c, H / 4, mgroups, g, ea);
            } else {
                typename EpiLSTM<Ops, true, false, 4>::Args eb{};
                static_assert(sizeof(eb) == sizeof(ea), "same Args layout");
                memcpy(&eb, &ea, sizeof(eb));
                if (wide8) launch_gemm<Ops, EpiLSTMw<Ops, false, 2>, MTA, true, -1>(c, H / 8, mgroups, g, eb);
                else if (wide) launch_wide<OW, EpiLSTMw<OW, false>>(c, H / 16, mgroups, g, eb);
                else launch_gemm<Ops, EpiLSTM<Ops, true, false, 4>, MTA, true, -1>(c, H / 4, mgroups, g, eb);
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
                else if (wide) launch_wide<OW, EpiNBRCw<OW, true>>(c, H / 16, mgroups, g, ea);
                else launch_gemm<Ops, EpiNBRC<Ops, true>, MTA, true, -1>(c, H / 4, mgroups, g, ea);
            } else {
                typename EpiNBRC<Ops, false>::Args eb{};
                static_assert(sizeof(eb) == sizeof(ea), "same Args layout");
                memcpy(&eb, &ea, sizeof(eb));
                if (wide8) launch_gemm<Ops, EpiNBRCw<Ops, false, 2>, MTA, true, -1>(c, H / 8, mgroups, g, eb);
                else if (wide) launch_wide<OW, EpiNBRCw<OW, false>>(c, H / 16, mgroups, g, eb);
                else launch_gemm<Ops, EpiNBRC<Ops, false>, MTA, true, -1>(c, H / 4, mgroups, g, eb);
            }
        }
    }
    if (!beam && l1 == c->d.pred_layers) c->pred_par ^= 1;      // beam: launch_ppj (same pass, same parities) toggles
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
        else launch_gemm_carry<Ops, EpiPPJ<Ops>, 1, true, -1>(c, J / 16, c->MTd, g, ea);
    } else
    if (ppj_wide) launch_gemm<Ops, EpiPPJ<Ops>, 4, true, -1, 4>(c, J / 16, c->MTd / 4, g, ea);
    else launch_gemm<Ops, EpiPPJ<Ops>, 1, true, -1>(c, J / 16, c->MTd, g, ea);
    if (beam) c->pred_par ^= 1;
}
// LMFuser.advance (lm.py:49-53) for the rows with emit != 0: LM step on the token just emitted, then
// log_softmax + standardise + [0] = MIN_VAL into lmz (read by the next k_select of that row)
// (l0, l1: LSTM layers [l0, l1) of the step; the output layer, k_lm_post and the parity flip come with the last one unless tail = false)
template <class Ops>
void launch_lm_t(lasr_ctx* c, bool beam, int l0, int l1, bool tail) {
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
    launch_linear_ops<Ops, true, -1>(c, V / 16, R / 16, g, H, ea);
    if (beam)
        LAUNCH_LM_POST(V, dim3(R), dim3(256), 0, c->stream, (const float*)m.raw, (const int*)c->ds.emit, p ? m.lmz : m.lmz1,
                           p ? m.valid : m.valid1, V, m.min_val, (const int*)c->b_parent, c->W, (const float*)(p ? m.lmz1 : m.lmz),
                           (const int*)(p ? m.valid1 : m.valid));
    else
        LAUNCH_LM_POST(V, dim3(R), dim3(256), 0, c->stream, (const float*)m.raw, (const int*)c->ds.emit, m.lmz, m.valid, V, m.min_val,
                           (const int*)nullptr, 1, (const float*)m.lmz, (const int*)m.valid);
    m.par ^= 1;
}
// the pair kinds of one operand type (see launch_pair): false = not a kind the templates name
template <class Ops>
bool launch_pair_ops(lasr_ctx* c, int kind, bool lm_first, lasr_ctx::Captured& A, lasr_ctx::Captured& B) {
    constexpr int NWD = NW;                         // decode GEMMs: 8 waves with either operand type (round 6: T)
    using LT = EpiLSTM<Ops, true, true, 4>; using LF = EpiLSTM<Ops, true, false, 4>;
    if (kind == 0 && lm_first) return launch_pair_t<Ops, EpiNBRC<Ops, true>, MTA, NWD, true, -1, LT, MTA, NW, true, -1>(c, A, B);
    if (kind == 1 && !lm_first) return launch_pair_t<Ops, EpiNBRC<Ops, false>, MTA, NWD, true, -1, LF, MTA, NW, true, -1>(c, A, B);
    if (kind == 2 && !lm_first) return launch_pair_t<Ops, EpiPPJ<Ops>, 1, NWD, true, -1, LF, MTA, NW, true, -1>(c, A, B);
    if (kind == 3 && !lm_first) return launch_pair_t<Ops, EpiLinear, 2, NWD, false, -1, LF, MTA, NW, true, -1>(c, A, B);
    return false;
}
