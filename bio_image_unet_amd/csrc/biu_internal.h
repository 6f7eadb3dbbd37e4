// Internal (non-exported) interfaces between the translation units of libbiu_hip.so.
#pragma once
#include "biu_common.h"

// biu_direct.hip (C linkage, hidden: these cross translation units of the library, they are not part of include/biu.h)
#define BIU_HIDDEN __attribute__((visibility("hidden")))
extern "C" BIU_HIDDEN int biu_conv_fwd_direct(const biu_act* x, const biu_xform* xf, const float* w, const float* bias, int kd,
                                   int kh, int kw, int dil, const biu_act* y, int dtype, hipStream_t st);
extern "C" BIU_HIDDEN int biu_conv_bwd_data_direct(const biu_act* dy, const float* w, int kd, int kh, int kw, int dil,
                                        const biu_act* dx, int accumulate, int dtype, hipStream_t st);
extern "C" BIU_HIDDEN int biu_conv_bwd_weight_direct(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, int kh,
                                          int kw, int dil, float* dw, float* dbias, int dtype, hipStream_t st);
extern "C" BIU_HIDDEN int biu_convt_fwd_direct(const biu_act* x, const biu_xform* xf, const float* w, const float* bias, int kd,
                                    const biu_act* y, int dtype, hipStream_t st);
extern "C" BIU_HIDDEN int biu_convt_bwd_data_direct(const biu_act* dy, const float* w, int kd, const biu_act* dx, int accumulate,
                                         int dtype, hipStream_t st);
extern "C" BIU_HIDDEN int biu_convt_bwd_weight_direct(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, float* dw,
                                           float* dbias, int dtype, hipStream_t st);
int biu_chan_sum(const biu_act* a, float* out, int dtype, hipStream_t st);
bool biu_convt_shapes_ok(const biu_act* lo, const biu_act* hi, int kd);

// biu_conv_mfma.hip
size_t biu_mfma_packed_bytes(int kind, int cin, int cout, int kd, int kh, int kw, int dilation, int dtype);
int biu_mfma_pack(int kind, const float* w, int cin, int cout, int kd, int kh, int kw, int dtype, void* packed,
                  hipStream_t st);
bool biu_mfma_conv_ok(const biu_act* x, const biu_act* y, int kd, int kh, int kw, int dilation, int dtype);
struct BnRedFuse {            // BatchNorm-backward sums of the layer that PRODUCED the tensor whose gradient is being written
    const biu_act* y;
    const float *scale, *shift, *slope, *mean, *invstd;
};
bool biu_mfma_conv_cat_ok(const biu_act* x0, const biu_act* x1, const biu_act* y, int kd, int kh, int kw, int dilation, int dtype);
struct ConvCat {              // channel concatenation without a concat buffer: second input source and / or second output
    const biu_act* x1; const biu_xform* xf1;
    const biu_act* y1; int accumulate1;
};
int biu_mfma_conv(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, int kd, int kh, int kw,
                  const biu_act* y, int accumulate, float* bn_partial, int dtype, hipStream_t st, const BnRedFuse* red = nullptr,
                  const ConvCat* cat = nullptr, void* split_ws = nullptr, size_t split_ws_bytes = 0);
// biu_conv_roll.hip: rolling-window 3x3x3 bf16 convolution with register-resident weights (narrow full-resolution layers)
bool biu_conv_roll_ok(const biu_act* x, const biu_act* y, int dtype, bool has_cat, int accumulate, bool red);
int biu_conv_roll_mshape(const biu_act* x, const biu_act* y, int dtype);       // 32 | 16: which packed fragment image the launch reads
int biu_conv_roll_rows(const biu_act* x, const biu_act* y, int dtype);         // partial rows of its epilogue sums (= blocks per column)
int biu_conv_roll(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, const biu_act* y, float* bn_partial, const BnRedFuse* red,
                  hipStream_t st, int accumulate = 0);
// the up half of a folded decoder level (64 -> 32 channels) as the first writer of y: composed 8-class weights in registers, border-state bias table
bool biu_fold_roll_ok(const biu_act* x_low, const biu_act* y, int dtype);
int biu_fold_roll(const biu_act* x_low, const biu_xform* xf, const void* packed_fwd, const float* bias_sum, const float* fix, const biu_act* y, hipStream_t st);
// biu_fold_gemm.hip: the weight-space products of a folded level (composed weights; chain rule back to W_conv / W_T) on the fp32 matrix pipe
size_t biu_fold_gemm_layout_floats(int cin_low, int cup, int cout);
bool biu_fold_gemm_ok(int cin_low, int cup, int cout);
int biu_fold_gemm_layouts(const float* w_conv, int ccat, int cup, int cout, const float* w_t, int cin_low, float* layouts, hipStream_t st);
int biu_fold_gemm_compose(const float* layouts, int cin_low, int cup, int cout, const float* b_t, float* wfold, float* wb, hipStream_t st);
int biu_fold_gemm_chain(const float* G, size_t slice_f, const float* layouts, int cin_low, int cup, int cout, int ccat, float* dw_conv, float* dw_t,
                        const float* b_t, const float* Sk, hipStream_t st);
size_t biu_mfma_conv_split_bytes(int cin, const biu_act* y, const biu_act* y1, int kd, int dtype);   // 0: the launch is not split
int biu_mfma_convt_dgrad_bricks(const biu_act* dx, int kd);
int biu_mfma_convt_dgrad_rows(const biu_act* dx, int kd);
int biu_mfma_pack_batch(const biu_pack_job* jobs_device, int n, int dtype, hipStream_t st);
int biu_mfma_set_fp32_products(int mode);
int biu_mfma_conv_ksplit(int cin, const biu_act* y, int kd, int dtype);
int biu_mfma_conv_stat_rows(const biu_act* y, int kd, const biu_act* x = nullptr, int dtype = -1, bool red = false);
int biu_mfma_conv_bricks(const biu_act* y, int kd, const biu_act* x = nullptr, int dtype = -1);
size_t biu_mfma_wgrad_workspace(int cin, int cout, int kd, int kh, int kw, int dtype);
bool biu_mfma_wgrad_ok(const biu_act* x, const biu_act* dy, int kd, int kh, int kw, int dilation, int dtype);
struct BnBwdFuse {            // BatchNorm(+LeakyReLU) backward fused into the weight-gradient loader
    const biu_act* y;
    const float *scale, *shift, *slope, *cA, *cB, *cC;
};
int biu_mfma_wgrad(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, int kh, int kw, float* dw,
                   float* dbias, void* ws, size_t ws_bytes, int dtype, hipStream_t st, const BnBwdFuse* bn = nullptr,
                   const biu_act* x1 = nullptr, const biu_xform* xf1 = nullptr, int dw_ld_cols = 0, int dw_c_off = 0);

size_t biu_mfma_convt_packed_bytes(int kind, int cin, int cout, int kd, int dtype);
int biu_mfma_convt_pack(int kind, const float* w, int cin, int cout, int kd, int dtype, void* packed, hipStream_t st);
// nearest-neighbour up-sampling folded into the 3x3x3 convolution behind it (forward)
bool biu_mfma_upconv_ok(const biu_act* x, const biu_act* y, int dtype);
size_t biu_mfma_upconv_packed_bytes(int kind, int cin, int cout, int dtype);
int biu_mfma_upconv_pack(int kind, const float* w, int cin, int cout, int dtype, void* packed, hipStream_t st, const float* wf = nullptr);
int biu_mfma_upconv_dgrad_rows(const biu_act* dx);
int biu_mfma_upconv_dgrad(const biu_act* dy, const void* packed, const biu_act* dx, int accumulate, int dtype, hipStream_t st, float* bn_partial = nullptr,
                          const BnRedFuse* red = nullptr);
struct BnBwdFuse;
size_t biu_mfma_upconv_wgrad_workspace(int cin, int cout, int dtype);
int biu_mfma_upconv_wgrad(const biu_act* x, const biu_xform* xf, const biu_act* dy, float* dw, void* ws, size_t ws_bytes, int dtype,
                          hipStream_t st, const BnBwdFuse* bn);
int biu_mfma_upconv_stat_rows(const biu_act* x, const biu_act* y);
int biu_mfma_upconv_fwd(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, const biu_act* y, float* bn_partial,
                        int dtype, hipStream_t st, int accumulate = 0);
// ConvTranspose + concat + 3x3x3 conv of a decoder level with the up half folded onto the coarse tensor
bool biu_mfma_foldt_ok(const biu_act* x_low, const biu_act* skip, const biu_act* y, int dtype);
bool biu_mfma_foldt_worth(const biu_act* x_low, const biu_act* y);
size_t biu_mfma_foldt_packed_bytes(int cin_low, int cskip, int cout, int dtype);
int biu_mfma_foldt_pack(const float* w_conv, const float* b_conv, const float* w_t, const float* b_t, int cin_low, int cup, int cskip, int cout, int dtype,
                        void* packed, hipStream_t st);
int biu_mfma_foldt_stat_rows(const biu_act* x_low, const biu_act* y, const biu_act* skip = nullptr, int dtype = -1);
int biu_mfma_foldt_form(const biu_act* x_low, const biu_act* skip, const biu_act* y, int dtype);
int biu_mfma_foldt_fwd(const biu_act* x_low, const biu_xform* xf_low, const biu_act* skip, const biu_xform* xf_skip, const void* packed, const biu_act* y,
                       float* bn_partial, int dtype, hipStream_t st);
int biu_mfma_foldt_dgrad(const biu_act* dy, const void* packed, const biu_act* dx_low, int acc_low, const biu_act* dskip, int acc_skip, int dtype,
                         hipStream_t st, float* bn_partial_low, const BnRedFuse* red_low, void* ws, size_t ws_bytes);
size_t biu_mfma_foldt_wgrad_workspace(int cin_low, int cskip, int cout, int dtype);
int biu_mfma_foldt_wgrad(const biu_act* x_low, const biu_xform* xf_low, const biu_act* skip, const biu_xform* xf_skip, const biu_act* da, const BnBwdFuse* bn,
                         const float* dy_sum, const float* w_conv, const float* w_t, const float* b_t, int cup, float* dw_conv, float* dw_t, float* db_t, void* ws, size_t ws_bytes,
                         int dtype, hipStream_t st, int phases = 7);
bool biu_mfma_convt_ok(int kind, const biu_act* lo, const biu_act* hi, int kd, int dtype);
int biu_mfma_convt_fwd(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, int kd, const biu_act* y,
                       int dtype, hipStream_t st);
int biu_mfma_convt_dgrad(const biu_act* dy, const void* packed, int kd, const biu_act* dx, int accumulate, int dtype, hipStream_t st,
                         float* bn_partial = nullptr, const BnRedFuse* red = nullptr);
bool biu_mfma_convt_wgrad_ok(const biu_act* x, const biu_act* dy, int kd, int dtype);
int biu_mfma_convt_wgrad(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, float* dw, float* dbias, void* ws,
                         size_t ws_bytes, int dtype, hipStream_t st);

// biu_special.hip
bool biu_c1_conv_ok(const biu_act* x, const biu_act* y, int kd, int kh, int kw, int dil, int dtype);
int biu_c1_conv_fwd(const biu_act* x, const biu_xform* xf, const float* w, const float* bias, int kd, const biu_act* y, int dtype, hipStream_t st);
size_t biu_c1_wgrad_workspace(int cout, int kd);
int biu_c1_conv_wgrad(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, float* dw, void* ws, size_t ws_bytes, int dtype, hipStream_t st);
bool biu_vec_reduce_ok(const biu_act* a, int dtype);
int biu_bn_stats_vec(const biu_act* y, float* partial, int* nblk_out, int dtype, hipStream_t st);
int biu_bn_bwd_reduce_vec(const biu_act* da, const biu_act* y, const float* scale, const float* shift, const float* slope,
                          const float* mean, const float* invstd, float* partial, int* nblk_out, int dtype, hipStream_t st);
size_t biu_chan_sum_workspace(int c);
int biu_chan_sum_vec(const biu_act* a, float* out, void* ws, int dtype, hipStream_t st);
bool biu_head_bwd_fused_ok(const biu_act* x, const biu_act* dx, int cout, int dtype);
size_t biu_head_bwd_fused_workspace(int cin);
bool biu_head_bwd_bnred_ok(const biu_act* x, const biu_act* dx, int cout, int dtype);
int biu_head_bwd_bnred_fused(const biu_act* x, const biu_xform* xf, const float* w, int cout, const float* dl, const biu_act* dx, float* dw,
                             float* db, void* ws, const float* mean, const float* invstd, float* bn_partial, int* bn_nblk, int dtype,
                             hipStream_t st);
int biu_head_bwd_fused(const biu_act* x, const biu_xform* xf, const float* w, int cout, const float* dl, const biu_act* dx, float* dw,
                       float* db, void* ws, size_t ws_bytes, int dtype, hipStream_t st);
bool biu_rowvec_ok(const biu_act* a, int dtype);
int biu_bn_bwd_apply_rv(const biu_act* da, const biu_act* y, const float* scale, const float* shift, const float* slope,
                        const float* A, const float* B, const float* Cc, const biu_act* dy, int dtype, hipStream_t st);
int biu_xform_apply_rv(const biu_act* x, const biu_xform* xf, const biu_act* out, int dtype, hipStream_t st);
int biu_maxpool_fwd_rv(const biu_act* x, const biu_xform* xf, const biu_act* out, int pd, int dtype, hipStream_t st);
int biu_maxpool_bwd_rv(const biu_act* x, const biu_xform* xf, const biu_act* dout, const biu_act* dx, int pd, int accumulate, int dtype, hipStream_t st);
int biu_maxpool_bwd_bnred_rv(const biu_act* x, const biu_xform* xf, const biu_act* dout, const biu_act* dx, int pd, int accumulate,
                             const float* mean, const float* invstd, float* partial, size_t partial_floats, int* nblk, int dtype,
                             hipStream_t st);
int biu_nearest_rv(int mode, const biu_act* src, const biu_xform* xf, const biu_act* dst, int pd, int accumulate, int dtype, hipStream_t st);

// biu_convt.hip: bf16 ConvTranspose k2 s2 forward / data gradient with all parities from one resident tile
bool biu_convt_all_ok(const biu_act* lo, const biu_act* hi, int kd, int dtype);
int biu_convt_all_fwd(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, int kd, const biu_act* y, hipStream_t st);
int biu_convt_all_dgrad_rows(const biu_act* dx, const biu_act* dy, int kd);
int biu_convt_all_dgrad(const biu_act* dy, const void* packed, int kd, const biu_act* dx, int accumulate, hipStream_t st,
                        float* bn_partial = nullptr, const BnRedFuse* red = nullptr);

// biu_c1.hip: bf16 first layer (Cin = 1) on the matrix cores through an im2col image in LDS
bool biu_c1m_ok(const biu_act* x, const biu_act* y, int kd, int kh, int kw, int dil, int dtype);
int biu_c1m_fwd_rows(const biu_act* y, int kd);
int biu_c1m_fwd(const biu_act* x, const biu_xform* xf, const float* w, const float* bias, int kd, const biu_act* y, float* bn_partial, hipStream_t st);
size_t biu_c1m_wgrad_workspace(int cout, int kd);
int biu_c1m_wgrad(const biu_act* x, const biu_xform* xf, const biu_act* da, const BnBwdFuse* bn, int kd, float* dw, void* ws, size_t ws_bytes,
                  hipStream_t st);
