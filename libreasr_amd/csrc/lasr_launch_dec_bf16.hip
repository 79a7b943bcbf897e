// lasr_launch_dec_bf16.hip -- decode-side GEMMs with bf16 operands (v_mfma_f32_16x16x32_bf16) + the int8-served LM step (integer-
// valued bf16 operands): a translation unit of liblasr_hip.so
#include "lasr_host.hip.h"
#include "lasr_launch_dec.hip.h"

LASR_DECL_OPS(, OpsBF16)

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
static void lm_q_gemv_pre(lasr_ctx* c, const unsigned short* qa, const float* sx, int Kp, const void* Wq, float w_scale, const float* bias,
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
