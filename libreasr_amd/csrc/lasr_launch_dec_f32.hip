// lasr_launch_dec_f32.hip -- decode-side GEMMs with f32 operands (v_mfma_f32_16x16x4_f32): a translation unit of liblasr_hip.so
#include "lasr_host.hip.h"
#include "lasr_launch_dec.hip.h"

LASR_DECL_OPS(, OpsF32)

// row-major f32 GEMM with the default 8-wave K split (token tables at lasr_create / lasr_attach_lm: embed -> layer-0 input projection)
void launch_table_gemm_f32(lasr_ctx* c, int n_groups, int m_groups, const GemmArgs& g, const EpiLinear::Args& ea) {
    launch_gemm<OpsF32, EpiLinear, 1, true>(c, n_groups, m_groups, g, ea);
}
