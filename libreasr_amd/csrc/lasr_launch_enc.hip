// lasr_launch_enc.hip -- the encoder LSTM-cell GEMMs (tilings C and D, plain and layer-wavefront launches, both operand types):
// one of the translation units of liblasr_hip.so (see lasr_launch.hip.h).
#include "lasr_host.hip.h"

// encoder LSTM cell (layer l, step t): x from `xsrc` (fragment-major, K = I); tiling "C"
template <class Ops>
static void launch_enc_cell_t(lasr_ctx* c, int l, int t, const void* xsrc, int x_mt_total, void* ydst, int y_mt_total) {
    const Cell& L = c->enc[l];
    const int H = c->d.hidden;
    GemmArgs g{};
    g.A[0] = xsrc; g.a_mt_total[0] = x_mt_total; g.a_mt_off[0] = t * c->MT; g.KC[0] = L.I / Ops::KCH; g.W[0] = L.WxC;
    g.A[1] = c->enc_h[c->enc_par][l]; g.a_mt_total[1] = c->MT; g.a_mt_off[1] = 0; g.KC[1] = H / Ops::KCH; g.W[1] = L.WhC;
    g.M = c->M; g.dbg = c->dbg;
    if (c->cell_prof && c->cp_slots && c->cp_slot_next < lasr_ctx::NCELLSLOT) {
        c->cp_slot_cells[c->cp_slot_next] = 1;
        g.prof = c->cp_slots + (size_t)PROF_W * c->cp_slot_next;
        g.prof_x = g.prof + (size_t)PROF_W * lasr_ctx::NCELLSLOT;
        c->cp_slot_next++;
    }
    using E = EpiLSTM<Ops, false, false, 8>;
    typename E::Args ea{};
    ea.bias = L.bias; ea.flag = c->T_row_dev; ea.t = t; ea.tile_mask = c->tile_masks.empty() ? ~0ull : c->tile_masks[t];
    ea.c = c->enc_c[l]; ea.h_in = c->enc_h[c->enc_par][l]; ea.h_out = c->enc_h[c->enc_par ^ 1][l];
    ea.y = ydst; ea.y_mt_total = y_mt_total; ea.y_mt_off = t * c->MT;
    ea.bn_s = L.bn_s; ea.bn_t = L.bn_t; ea.H = H; ea.M = c->M; ea.MT = c->MT;
    // K split over 4 waves for f32 operands (12.8 us against 17.2 us with 8: fewer requests in flight, half the
    // LDS reduction), 8 waves for bf16 (5.4 us against 6.6 us); LASR_CELL_NW overrides
    const int nw = c->cell_nw ? c->cell_nw : (Ops::BF ? 8 : 4);
    if (c->enc_u12) {
        using E12 = EpiLSTMe<Ops, 12>;
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
void launch_enc_cell(lasr_ctx* c, int l, int t, const void* xsrc, int x_mt_total, void* ydst, int y_mt_total) {
    if (c->bf) launch_enc_cell_t<OpsBF16>(c, l, t, xsrc, x_mt_total, ydst, y_mt_total);
    else launch_enc_cell_t<OpsF32>(c, l, t, xsrc, x_mt_total, ydst, y_mt_total);
}

// One anti-diagonal of the encoder's (layer, time) grid in ONE launch: cells (l, d - l), independent of each other.
// par0 = h ping-pong parity before the pass (cell (l, t) reads parity par0 ^ (t & 1)).
template <class Ops>
static void launch_enc_wave_t(lasr_ctx* c, const EncCellRef* cells, int n, int par0, int mt_total) {
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

