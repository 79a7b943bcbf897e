/* A plain-C consumer of include/lasr.h: the drop-in boundary does not need Python, torch or C++.
 * Built and run by tests/test_abi.py (no GPU needed: lasr_create must fail loudly with LASR_EHIP). */
#include <stdio.h>
#include <stdlib.h>
#include "lasr.h"

int main(void) {
    lasr_model_desc d;
    lasr_default_desc(&d);
    size_t n = lasr_weight_count(&d);
    printf("weights %zu\n", n);
    if (n != 53039616u) return 2;                    /* the reference 4+2-layer shape: 53.03 M parameters + BatchNorm running statistics */
    d.hidden = 1000;                                 /* not a multiple of 16: rejected */
    if (lasr_weight_count(&d) != 0) return 3;
    lasr_default_desc(&d);
    float* w = (float*)calloc(n, sizeof(float));
    lasr_ctx* ctx = NULL;
    int rc = lasr_create(0, &d, w, n, NULL, &ctx);
    printf("create rc %d: %s\n", rc, ctx ? lasr_last_error(ctx) : "(no ctx)");
    if (ctx) lasr_destroy(ctx);
    free(w);
    lasr_lm_desc lm = {2048, 768, 768, 4, 0.1f, 1.0f, -10.0f};
    printf("lm weights %zu\n", lasr_lm_weight_count(&lm));
    return rc == 0 ? 0 : (rc == LASR_EHIP ? 10 : 4); /* 10 = "no usable GPU", the expected outcome on a CPU box */
}
