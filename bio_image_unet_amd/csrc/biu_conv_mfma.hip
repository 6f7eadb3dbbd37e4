// MFMA implicit-GEMM convolution kernels (placeholder translation unit: everything reports "not covered"
// until the kernels land, so every layer is served by the direct kernels).
#include "biu_internal.h"

size_t biu_mfma_packed_bytes(int, int, int, int, int, int, int, int) { return 0; }
int biu_mfma_pack(int, const float*, int, int, int, int, int, int, void*, hipStream_t) {
    return biu_fail(BIU_ERR_UNSUPPORTED, "mfma pack: not built");
}
bool biu_mfma_conv_ok(const biu_act*, const biu_act*, int, int, int, int, int) { return false; }
int biu_mfma_conv(const biu_act*, const biu_xform*, const void*, const float*, int, int, int, const biu_act*, int, int,
                  hipStream_t) {
    return biu_fail(BIU_ERR_UNSUPPORTED, "mfma conv: not built");
}
size_t biu_mfma_wgrad_workspace(int, int, int, int, int, int) { return 0; }
bool biu_mfma_wgrad_ok(const biu_act*, const biu_act*, int, int, int, int, int) { return false; }
int biu_mfma_wgrad(const biu_act*, const biu_xform*, const biu_act*, int, int, int, float*, float*, void*, size_t, int,
                   hipStream_t) {
    return biu_fail(BIU_ERR_UNSUPPORTED, "mfma wgrad: not built");
}
