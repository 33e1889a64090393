/* Plain-C consumer of the drop-in boundary: includes the public header, links libgrafx_amd.so, and calls the
 * entry points that need no GPU (version, size queries, argument validation).  Built and run by
 * tests/test_abi.py with gcc -- the header must be valid C, not just C++. */
#include <stdio.h>
#include <stdlib.h>

#include "grafx_amd.h"

#define CHECK(cond)                                                  \
    do {                                                             \
        if (!(cond)) {                                               \
            fprintf(stderr, "abi_smoke: failed: %s\n", #cond);       \
            return 1;                                                \
        }                                                            \
    } while (0)

int main(void) {
    gfx_rowmap_t m = {1, 0, 0, 0};
    CHECK(gfx_abi_version() > 0);
    CHECK(gfx_fftconv_nparts(4001) == 1);
    CHECK(gfx_fftconv_nparts(60001) == 8);
    CHECK(gfx_fir_spectrum_bytes(3, 4001) == (size_t)3 * 17 * 256 * 16);
    CHECK(gfx_fftconv_workspace_bytes(4, 2, 131072, 131072, 0, 4001) == 0);
    CHECK(gfx_fftconv_workspace_bytes(4, 2, 131072, 131072, 0, 60001) > 0);
    CHECK(gfx_iir_fsm_plan_bytes(4001) > 0 && gfx_iir_fsm_plan_bytes(4097) == 0);
    CHECK(gfx_istft_basis_bytes(384) > 0 && gfx_istft_basis_bytes(383) == 0);
    /* argument validation happens before any device call */
    CHECK(gfx_fftconv_f32(NULL, m, NULL, NULL, m, 1, 1, 1, 16, 16, 0, 8, NULL, 0, NULL) == GFX_EINVAL);
    CHECK(gfx_biquad_cascade_f32(NULL, m, NULL, m, NULL, NULL, 1, 1, 1, 1, 16, 0, NULL) == GFX_EINVAL);
    CHECK(gfx_waveshaper_f32(NULL, m, NULL, m, 1, 1, 16, GFX_WS_TANH, 0, 0, NULL, NULL, NULL, NULL, 0, NULL, NULL) == GFX_EINVAL);
    printf("abi_smoke ok (abi version %d)\n", gfx_abi_version());
    return 0;
}
