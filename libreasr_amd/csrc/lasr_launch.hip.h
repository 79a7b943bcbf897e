// lasr_launch.hip.h -- launch interface of the GEMM kernels, shared by the translation units of liblasr_hip.so.
// The library is FOUR translation units (round 6; one 3.5-minute unit before): lasr_engine.hip (C ABI, protocols, small kernels),
// lasr_launch_enc.hip (encoder cells), lasr_launch_dec_f32.hip / lasr_launch_dec_bf16.hip (predictor, joint, LM and their pair
// launches per operand type).  Every k_gemm instantiation lives in exactly one of the last three; this header declares the
// host functions that launch them (explicit instantiations per operand type) and the small inline dispatchers on c->bf.
#pragma once

// ---------------------------------------------------------------------------- launch helpers
template <class Ops, class Epi, int MT, bool AROW, int D = 3, int NWV = NW>
inline void launch_gemm(lasr_ctx* c, int n_groups, int m_groups, const GemmArgs& g0, const typename Epi::Args& ea) {
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
inline void replay_captured(lasr_ctx* c, lasr_ctx::Captured& k) {
    if (!k.fn) return;
    void* args[2] = {(void*)k.g, (void*)k.ea};
    (void)hipLaunchKernel(k.fn, dim3(k.gx, k.gy), dim3(k.threads), args, 0, c->stream);
    k.fn = nullptr;
}
inline int grid1(size_t n, int b = 256) { return (int)((n + b - 1) / b); }

// lasr_trace: a value record (no event): lasr_trace_read returns `val` in the time field
// (marks come from the API thread -- main stream -- and from the pump thread -- decode stream: slots are drawn atomically)
inline void tr_note(lasr_ctx* c, int tag, double val) {
    if (!c->tr_on) return;
    const int i = c->tr_n.fetch_add(1, std::memory_order_relaxed);
    if (i >= lasr_ctx::NTRACE) return;
    c->tr_val[i] = val;
    c->tr_tag[i] = tag;
}
// lasr_trace: one timestamped mark on stream `st` (no-op unless tracing)
inline void tr_mark(lasr_ctx* c, int tag, hipStream_t st) {
    if (!c->tr_on) return;
    const int i = c->tr_n.fetch_add(1, std::memory_order_relaxed);
    if (i >= lasr_ctx::NTRACE) return;
    (void)hipEventRecord(c->tr_ev[i], st);
    c->tr_tag[i] = tag;
}


// ---- encoder cells (lasr_launch_enc.hip)
void launch_enc_cell(lasr_ctx* c, int l, int t, const void* xsrc, int x_mt_total, void* ydst, int y_mt_total);
struct EncCellRef { int l, t; };
void launch_enc_wave(lasr_ctx* c, const EncCellRef* cells, int n, int par0, int mt_total);

// ---- decode-side GEMMs, one set per operand type (lasr_launch_dec.hip.h, instantiated in lasr_launch_dec_{f32,bf16}.hip)
// LASR_BEAM_CARRY: 0 = non-extended hypothesis slots carried inside the cell / joint kernels' epilogues (round 3), 1 = by a launch of
// their own (k_beam_carry), 2 = as extra workgroups of the joint-half GEMM's launch (k_gemm_carry)
inline int beam_carry_mode() {
    static const int v = getenv("LASR_BEAM_CARRY") ? atoi(getenv("LASR_BEAM_CARRY")) : 2;
    return v;
}
inline bool beam_carry_on() { return beam_carry_mode() != 0; }
template <class Ops> void launch_predictor_t(lasr_ctx* c, bool beam, int l0, int l1);
template <class Ops> void launch_ppj_t(lasr_ctx* c, bool beam);
template <class Ops> void launch_lm_t(lasr_ctx* c, bool beam, int l0, int l1, bool tail);
template <class Ops> bool launch_pair_ops(lasr_ctx* c, int kind, bool lm_first, lasr_ctx::Captured& A, lasr_ctx::Captured& B);
template <class Ops> void launch_logits_ops(lasr_ctx* c, float* out, int n_rows, bool gated);
template <class Ops, bool AROW, int D> void launch_linear_ops(lasr_ctx* c, int n_groups, int m_groups, GemmArgs g, int K, const EpiLinear::Args& ea);
void launch_lm_q8(lasr_ctx* c);                                     // (integer-valued bf16 operands whatever the model's type: bf16 unit)
// quantise `rows` rows of src -> integer GEMV -> dequantise (+ bias) into out; also builds the int8-served LM's token table at attach
void lm_q_gemv(lasr_ctx* c, const float* src, int lds, int K, int Kp, const void* Wq, float w_scale, const float* bias, float* out,
               int N, int rows, unsigned short* qa, float* sx);
void launch_table_gemm_f32(lasr_ctx* c, int n_groups, int m_groups, const GemmArgs& g, const EpiLinear::Args& ea);   // f32 unit
#define LASR_DECL_OPS(X, Ops)                                                                                              \
    X template void launch_predictor_t<Ops>(lasr_ctx*, bool, int, int);                                                     \
    X template void launch_ppj_t<Ops>(lasr_ctx*, bool);                                                                     \
    X template void launch_lm_t<Ops>(lasr_ctx*, bool, int, int, bool);                                                      \
    X template bool launch_pair_ops<Ops>(lasr_ctx*, int, bool, lasr_ctx::Captured&, lasr_ctx::Captured&);                   \
    X template void launch_logits_ops<Ops>(lasr_ctx*, float*, int, bool);                                                   \
    X template void launch_linear_ops<Ops, true, -1>(lasr_ctx*, int, int, GemmArgs, int, const EpiLinear::Args&);           \
    X template void launch_linear_ops<Ops, false, -1>(lasr_ctx*, int, int, GemmArgs, int, const EpiLinear::Args&);          \
    X template void launch_linear_ops<Ops, true, 3>(lasr_ctx*, int, int, GemmArgs, int, const EpiLinear::Args&);            \
    X template void launch_linear_ops<Ops, false, 3>(lasr_ctx*, int, int, GemmArgs, int, const EpiLinear::Args&);
LASR_DECL_OPS(extern, OpsF32)
LASR_DECL_OPS(extern, OpsBF16)

// dispatchers on the context's operand type
inline void launch_predictor(lasr_ctx* c, bool beam = false, int l0 = 0, int l1 = -1) {
    if (c->bf) launch_predictor_t<OpsBF16>(c, beam, l0, l1);
    else launch_predictor_t<OpsF32>(c, beam, l0, l1);
}
inline void launch_ppj(lasr_ctx* c, bool beam = false) {
    if (c->bf) launch_ppj_t<OpsBF16>(c, beam);
    else launch_ppj_t<OpsF32>(c, beam);
}
inline float* cur_pp(lasr_ctx* c) { return (c->W > 1 && c->pred_par) ? c->pp1 : c->pp; }
inline void launch_lm(lasr_ctx* c, bool beam = false, int l0 = 0, int l1 = -1, bool tail = true) {
    if (!c->lm.on) return;
    if (c->lm.q8) { launch_lm_q8(c); return; }
    if (c->bf) launch_lm_t<OpsBF16>(c, beam, l0, l1, tail);
    else launch_lm_t<OpsF32>(c, beam, l0, l1, tail);
}
// Pair launches of the greedy loop with an fp32 / bf16 LM (cont_enqueue): stage `kind` of the predictor / joint chain (0, 1: NBRC
// layers 0, 1; 2: joint half; 3: the next iteration's logits GEMM), recorded in A, with LM layer l (0: through the token table),
// recorded in B.  The kinds the templates name are the ones configs[1] runs (2 x NBRC predictor, decode GEMMs on 8 waves);
// anything else is issued one after the other.
inline void launch_pair(lasr_ctx* c, int kind, bool lm_first, lasr_ctx::Captured& A, lasr_ctx::Captured& B) {
    const bool ok = c->bf ? launch_pair_ops<OpsBF16>(c, kind, lm_first, A, B) : launch_pair_ops<OpsF32>(c, kind, lm_first, A, B);
    if (!ok) { replay_captured(c, A); replay_captured(c, B); }
}
// current-parity LM output of the hypothesis slots (beam): parity 0 = lmz / valid, parity 1 = lmz1 / valid1
inline const float* cur_lmz(lasr_ctx* c) { return (c->W > 1 && c->lm.par) ? c->lm.lmz1 : c->lm.lmz; }
inline const int* cur_lm_valid(lasr_ctx* c) { return (c->W > 1 && c->lm.par) ? c->lm.valid1 : c->lm.valid; }
// plain linear over element-typed A (fragment-major, or row-major when AROW); f32 row-major output
template <bool AROW, int D>
inline void launch_linear(lasr_ctx* c, int n_groups, int m_groups, GemmArgs g, int K, const EpiLinear::Args& ea) {
    if (c->bf) launch_linear_ops<OpsBF16, AROW, D>(c, n_groups, m_groups, g, K, ea);
    else launch_linear_ops<OpsF32, AROW, D>(c, n_groups, m_groups, g, K, ea);
}
inline void launch_logits(lasr_ctx* c, float* out, int n_rows, bool gated) {
    if (c->bf) launch_logits_ops<OpsBF16>(c, out, n_rows, gated);
    else launch_logits_ops<OpsF32>(c, out, n_rows, gated);
}

// k_lm_post / k_beam_fuse with the register slots their vocabulary needs (bit-identical either way, see k_lm_post)
inline bool keep16(int V) {
    static const int force = getenv("LASR_KEEP16") ? atoi(getenv("LASR_KEEP16")) : 0;
    return force || V > 2048;
}
#define LAUNCH_LM_POST(V_, ...) do { if (keep16(V_)) hipLaunchKernelGGL(k_lm_post<16>, __VA_ARGS__); else hipLaunchKernelGGL(k_lm_post<8>, __VA_ARGS__); } while (0)
#define LAUNCH_BEAM_FUSE(V_, ...) do { if (keep16(V_)) hipLaunchKernelGGL(k_beam_fuse<16>, __VA_ARGS__); else hipLaunchKernelGGL(k_beam_fuse<8>, __VA_ARGS__); } while (0)

