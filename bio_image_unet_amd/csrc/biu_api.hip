// Public conv / conv-transpose entry points: validation + dispatch between the MFMA implicit-GEMM kernels
// (biu_conv_mfma.hip) and the shape-generic direct kernels (biu_direct.hip).
#include "biu_common.h"
#include "biu_internal.h"

// Debug switches (read once): BIU_DISABLE=conv_fwd,conv_dgrad,conv_wgrad,convt_fwd,convt_dgrad,convt_wgrad routes the named op family to the direct kernels.
#include <stdlib.h>
#include <string.h>
// The any-shape kernels are a correctness tier: a large layer landing on them (odd channel counts, dilation != 1, a sample
// beyond the 4 GB descriptor range, an unaligned slice) runs orders of magnitude below the MFMA path.  Say so, once per op.
static void warn_slow_path(const char* op, const biu_act* x, const biu_act* y, int taps) {
    const double macs = (double)nvox(y) * x->c * y->c * taps;
    if (macs < 4e9) return;
    static int shown = 0;
    if (shown++ < 8)
        fprintf(stderr, "[biu] %s: %d->%d channels on %lld voxels takes the any-shape kernel (%.1f GMAC): expect it to be slow; "
                        "the MFMA path needs channel counts that are multiples of 8 (>= 16), dilation 1, 16-byte aligned slices "
                        "and samples under 4 GB\n", op, x->c, y->c, (long long)nvox(y), macs * 1e-9);
}

static bool disabled(const char* what) {
    const char* env = getenv("BIU_DISABLE");       // re-read per call: tests flip it inside one process
    return env && strstr(env, what) != nullptr;
}

static bool conv_args_ok(const biu_act* x, const biu_act* y, int kd, int kh, int kw, int dil) {
    if (!valid_act(x) || !valid_act(y) || !same_space(x, y)) return false;
    if (!((kd == 1 || kd == 3) && kh == 3 && kw == 3) && !(kd == 1 && kh == 1 && kw == 1)) return false;
    return dil >= 1;
}

extern "C" size_t biu_conv_packed_bytes(int kind, int cin, int cout, int kd, int kh, int kw, int dilation, int dtype) {
    return biu_mfma_packed_bytes(kind, cin, cout, kd, kh, kw, dilation, dtype);
}

extern "C" int biu_conv_pack(int kind, const float* w, int cin, int cout, int kd, int kh, int kw, int dtype,
                             void* packed, biu_stream stream) {
    BIU_REQUIRE(w && packed, BIU_ERR_SHAPE, "conv_pack: null pointer");
    BIU_REQUIRE(biu_mfma_packed_bytes(kind, cin, cout, kd, kh, kw, 1, dtype) > 0, BIU_ERR_UNSUPPORTED,
                "conv_pack: shape (cin=%d, cout=%d, k=%dx%dx%d) is served by the direct kernels", cin, cout, kd, kh, kw);
    return biu_mfma_pack(kind, w, cin, cout, kd, kh, kw, dtype, packed, (hipStream_t)stream);
}

extern "C" int biu_pack_batch(const biu_pack_job* jobs, int n, int dtype, biu_stream stream) {
    BIU_REQUIRE(n >= 0 && (jobs || n == 0), BIU_ERR_SHAPE, "pack_batch: null job table");
    BIU_REQUIRE(n <= 65535, BIU_ERR_UNSUPPORTED, "pack_batch: more than 65535 jobs");
    return biu_mfma_pack_batch(jobs, n, dtype, (hipStream_t)stream);
}

// does a launch writing y (| y1) from cin channels take the input-channel split with the scratch the caller handed in?
static bool will_split(int cin, const biu_act* y, const biu_act* y1, int kd, int dtype, const void* ws, size_t ws_bytes) {
    const size_t need = biu_mfma_conv_split_bytes(cin, y, y1, kd, dtype);
    return need > 0 && ws && ws_bytes >= need;
}

extern "C" size_t biu_conv_split_workspace(int cin, const biu_act* y, const biu_act* y1, int kd, int kh, int kw, int dilation, int dtype) {
    if (!y || !valid_act(y) || (y1 && !valid_act(y1)) || kh != 3 || kw != 3 || (kd != 1 && kd != 3) || dilation != 1) return 0;
    biu_act xprobe = *y;                       // the input has y's extents; only its channel count and alignment matter here
    xprobe.c = xprobe.pitch = cin;
    biu_act yall = *y;
    if (y1) yall.c = y->c + y1->c;
    if (!biu_mfma_conv_ok(&xprobe, &yall, kd, kh, kw, dilation, dtype)) return 0;
    return biu_mfma_conv_split_bytes(cin, y, y1, kd, dtype);
}

extern "C" int biu_conv_fwd_stats(const biu_act* x, const biu_xform* xf, const float* w, const void* packed,
                                  const float* bias, int kd, int kh, int kw, int dilation, const biu_act* y,
                                  float* bn_partial, size_t bn_partial_floats, int* bn_nblk, void* ws, size_t ws_bytes, int dtype,
                                  biu_stream stream) {
    BIU_REQUIRE(conv_args_ok(x, y, kd, kh, kw, dilation), BIU_ERR_SHAPE,
                "conv_fwd: x/y extents differ or unsupported kernel %dx%dx%d dil %d", kd, kh, kw, dilation);
    BIU_REQUIRE(w && bn_partial && bn_nblk, BIU_ERR_SHAPE, "conv_fwd_stats: null pointer");
    *bn_nblk = 0;
    if (!disabled("c1") && !disabled("fused_stats") && biu_c1m_ok(x, y, kd, kh, kw, dilation, dtype)) {
        const int nb = biu_c1m_fwd_rows(y, kd);                    // first layer on the matrix cores, statistics from its epilogue
        if ((size_t)nb * y->c * 2 <= bn_partial_floats) {
            int rc = biu_c1m_fwd(x, xf, w, bias, kd, y, bn_partial, (hipStream_t)stream);
            if (rc == BIU_OK) *bn_nblk = nb;
            return rc;
        }
    }
    // (a launch split over the input channels cannot take its statistics from the epilogue: plain conv, then biu_bn_stats)
    if (packed && !disabled("conv_fwd") && !disabled("fused_stats") && biu_mfma_conv_ok(x, y, kd, kh, kw, dilation, dtype) &&
        !will_split(x->c, y, nullptr, kd, dtype, ws, ws_bytes)) {
        const int nb = biu_mfma_conv_stat_rows(y, kd, x, dtype);           // one partial row per workgroup column
        if ((size_t)nb * y->c * 2 <= bn_partial_floats) {
            int rc = biu_mfma_conv(x, xf, packed, bias, kd, kh, kw, y, 0, bn_partial, dtype, (hipStream_t)stream);
            if (rc == BIU_OK) *bn_nblk = nb;
            return rc;
        }
    }
    int rc = biu_conv_fwd(x, xf, w, packed, bias, kd, kh, kw, dilation, y, ws, ws_bytes, dtype, stream);
    if (rc != BIU_OK) return rc;
    BIU_REQUIRE(bn_partial_floats >= (size_t)BIU_BN_MAX_PARTIALS * y->c * 2, BIU_ERR_WORKSPACE, "conv_fwd_stats: partial buffer too small");
    return biu_bn_stats(y, bn_partial, bn_nblk, dtype, stream);
}

extern "C" size_t biu_conv_fwd_stats_floats(const biu_act* y, int kd) {
    size_t a = (size_t)BIU_BN_MAX_PARTIALS * y->c * 2, b = (size_t)biu_mfma_conv_bricks(y, kd == 3 ? 3 : 1) * y->c * 2;
    return a > b ? a : b;
}

extern "C" int biu_conv_fwd(const biu_act* x, const biu_xform* xf, const float* w, const void* packed,
                            const float* bias, int kd, int kh, int kw, int dilation, const biu_act* y, void* ws, size_t ws_bytes,
                            int dtype, biu_stream stream) {
    BIU_REQUIRE(conv_args_ok(x, y, kd, kh, kw, dilation), BIU_ERR_SHAPE,
                "conv_fwd: x/y extents differ or unsupported kernel %dx%dx%d dil %d", kd, kh, kw, dilation);
    BIU_REQUIRE(w, BIU_ERR_SHAPE, "conv_fwd: null weight");
    if (!disabled("c1") && biu_c1m_ok(x, y, kd, kh, kw, dilation, dtype))
        return biu_c1m_fwd(x, xf, w, bias, kd, y, nullptr, (hipStream_t)stream);
    if (!disabled("c1") && biu_c1_conv_ok(x, y, kd, kh, kw, dilation, dtype))
        return biu_c1_conv_fwd(x, xf, w, bias, kd, y, dtype, (hipStream_t)stream);
    if (packed && !disabled("conv_fwd") && biu_mfma_conv_ok(x, y, kd, kh, kw, dilation, dtype))
        return biu_mfma_conv(x, xf, packed, bias, kd, kh, kw, y, 0, nullptr, dtype, (hipStream_t)stream, nullptr, nullptr, ws, ws_bytes);
    warn_slow_path("conv_fwd", x, y, kd * kh * kw);
    return biu_conv_fwd_direct(x, xf, w, bias, kd, kh, kw, dilation, y, dtype, (hipStream_t)stream);
}

extern "C" int biu_conv_bwd_data(const biu_act* dy, const float* w, const void* packed, int kd, int kh, int kw,
                                 int dilation, const biu_act* dx, int accumulate, void* ws, size_t ws_bytes, int dtype, biu_stream stream) {
    BIU_REQUIRE(conv_args_ok(dy, dx, kd, kh, kw, dilation), BIU_ERR_SHAPE, "conv_bwd_data: dy/dx extents differ");
    BIU_REQUIRE(w, BIU_ERR_SHAPE, "conv_bwd_data: null weight");
    if (packed && !disabled("conv_dgrad") && biu_mfma_conv_ok(dy, dx, kd, kh, kw, dilation, dtype))
        return biu_mfma_conv(dy, nullptr, packed, nullptr, kd, kh, kw, dx, accumulate, nullptr, dtype, (hipStream_t)stream, nullptr, nullptr, ws, ws_bytes);
    warn_slow_path("conv_bwd_data", dx, dy, kd * kh * kw);
    return biu_conv_bwd_data_direct(dy, w, kd, kh, kw, dilation, dx, accumulate, dtype, (hipStream_t)stream);
}

// Data gradient + the BatchNorm-backward sums of the block that produced the tensor whose gradient is written (dx is that
// block's d loss / d a, y_up its raw conv output): saves the separate biu_bn_bwd_reduce pass.  Only valid when this call
// writes the COMPLETE gradient (accumulate == 0 and no later accumulation into dx).
extern "C" size_t biu_bwd_data_bnred_floats(const biu_act* dx, int kd, int transposed) {
    size_t a = (size_t)BIU_BN_MAX_PARTIALS * dx->c * 2;
    size_t b = (size_t)(transposed ? biu_mfma_convt_dgrad_bricks(dx, kd) : biu_mfma_conv_bricks(dx, kd == 3 ? 3 : 1)) * dx->c * 2;
    return a > b ? a : b;
}
extern "C" int biu_conv_bwd_data_bnred(const biu_act* dy, const float* w, const void* packed, int kd, int kh, int kw, int dilation,
                                       const biu_act* dx, const biu_act* y_up, const float* scale, const float* shift,
                                       const float* slope, const float* mean, const float* invstd, float* partial,
                                       size_t partial_floats, int* nblk, void* ws, size_t ws_bytes, int dtype, biu_stream stream) {
    BIU_REQUIRE(conv_args_ok(dy, dx, kd, kh, kw, dilation) && valid_act(y_up) && same_space(y_up, dx) && y_up->c == dx->c, BIU_ERR_SHAPE,
                "conv_bwd_data_bnred: extents differ");
    BIU_REQUIRE(w && scale && shift && mean && invstd && partial && nblk, BIU_ERR_SHAPE, "conv_bwd_data_bnred: null pointer");
    const size_t es = dsize(dtype);
    const bool yok = ((uintptr_t)y_up->p % 16) == 0 && ((size_t)y_up->pitch * es) % 16 == 0;
    if (packed && yok && !disabled("conv_dgrad") && !disabled("dgrad_bnred") && biu_mfma_conv_ok(dy, dx, kd, kh, kw, dilation, dtype) &&
        !will_split(dy->c, dx, nullptr, kd, dtype, ws, ws_bytes)) {
        const int nb = biu_mfma_conv_stat_rows(dx, kd, dy, dtype, true);     // one partial row per workgroup column (as the forward)
        if ((size_t)nb * dx->c * 2 <= partial_floats) {
            BnRedFuse red{y_up, scale, shift, slope, mean, invstd};
            int rc = biu_mfma_conv(dy, nullptr, packed, nullptr, kd, kh, kw, dx, 0, partial, dtype, (hipStream_t)stream, &red);
            if (rc == BIU_OK) *nblk = nb;
            return rc;
        }
    }
    int rc = biu_conv_bwd_data(dy, w, packed, kd, kh, kw, dilation, dx, 0, ws, ws_bytes, dtype, stream);
    if (rc != BIU_OK) return rc;
    BIU_REQUIRE(partial_floats >= (size_t)BIU_BN_MAX_PARTIALS * dx->c * 2, BIU_ERR_WORKSPACE, "conv_bwd_data_bnred: partial buffer too small");
    return biu_bn_bwd_reduce(dx, y_up, scale, shift, slope, mean, invstd, partial, nblk, dtype, stream);
}
extern "C" int biu_convt_bwd_data_bnred(const biu_act* dy, const float* w, const void* packed, int kd, const biu_act* dx,
                                        const biu_act* y_up, const float* scale, const float* shift, const float* slope,
                                        const float* mean, const float* invstd, float* partial, size_t partial_floats, int* nblk,
                                        int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(dx) && valid_act(dy) && w && biu_convt_shapes_ok(dx, dy, kd) && valid_act(y_up) && same_space(y_up, dx) &&
                    y_up->c == dx->c, BIU_ERR_SHAPE, "convt_bwd_data_bnred: extents differ");
    BIU_REQUIRE(scale && shift && mean && invstd && partial && nblk, BIU_ERR_SHAPE, "convt_bwd_data_bnred: null pointer");
    const size_t es = dsize(dtype);
    const bool yok = ((uintptr_t)y_up->p % 16) == 0 && ((size_t)y_up->pitch * es) % 16 == 0;
    if (packed && yok && !disabled("convt_dgrad") && !disabled("dgrad_bnred") && biu_mfma_convt_ok(1, dx, dy, kd, dtype) &&
        biu_convt_all_ok(dx, dy, kd, dtype)) {
        const int nb = biu_convt_all_dgrad_rows(dx, dy, kd);           // one partial row per persistent block
        if ((size_t)nb * dx->c * 2 <= partial_floats) {
            BnRedFuse red{y_up, scale, shift, slope, mean, invstd};
            int rc = biu_convt_all_dgrad(dy, packed, kd, dx, 0, (hipStream_t)stream, partial, &red);
            if (rc == BIU_OK) *nblk = nb;
            return rc;
        }
    }
    if (packed && yok && !disabled("convt_dgrad") && !disabled("dgrad_bnred") && biu_mfma_convt_ok(1, dx, dy, kd, dtype)) {
        const int nb = biu_mfma_convt_dgrad_rows(dx, kd);
        if ((size_t)nb * dx->c * 2 <= partial_floats) {
            BnRedFuse red{y_up, scale, shift, slope, mean, invstd};
            int rc = biu_mfma_convt_dgrad(dy, packed, kd, dx, 0, dtype, (hipStream_t)stream, partial, &red);
            if (rc == BIU_OK) *nblk = nb;
            return rc;
        }
    }
    int rc = biu_convt_bwd_data(dy, w, packed, kd, dx, 0, dtype, stream);
    if (rc != BIU_OK) return rc;
    BIU_REQUIRE(partial_floats >= (size_t)BIU_BN_MAX_PARTIALS * dx->c * 2, BIU_ERR_WORKSPACE, "convt_bwd_data_bnred: partial buffer too small");
    return biu_bn_bwd_reduce(dx, y_up, scale, shift, slope, mean, invstd, partial, nblk, dtype, stream);
}

extern "C" size_t biu_conv_bwd_weight_workspace(int cin, int cout, int kd, int kh, int kw, int dtype) {
    if (cin == 1 && kh == 3 && kw == 3) {
        const size_t a = biu_c1_wgrad_workspace(cout, kd), b = biu_c1m_wgrad_workspace(cout, kd);
        return a > b ? a : b;
    }
    return biu_mfma_wgrad_workspace(cin, cout, kd, kh, kw, dtype);
}

extern "C" int biu_conv_bwd_weight(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, int kh, int kw,
                                   int dilation, float* dw, float* dbias, void* ws, size_t ws_bytes, int dtype,
                                   biu_stream stream) {
    BIU_REQUIRE(conv_args_ok(x, dy, kd, kh, kw, dilation), BIU_ERR_SHAPE, "conv_bwd_weight: x/dy extents differ");
    BIU_REQUIRE(dw, BIU_ERR_SHAPE, "conv_bwd_weight: null dw");
    if (!disabled("c1") && biu_c1m_ok(x, dy, kd, kh, kw, dilation, dtype) && ws && ws_bytes >= biu_c1m_wgrad_workspace(dy->c, kd)) {
        int rc = biu_c1m_wgrad(x, xf, dy, nullptr, kd, dw, ws, ws_bytes, (hipStream_t)stream);
        if (rc == BIU_OK && dbias) rc = biu_chan_sum(dy, dbias, dtype, (hipStream_t)stream);
        return rc;
    }
    if (!disabled("c1") && biu_c1_conv_ok(x, dy, kd, kh, kw, dilation, dtype) && ws && ws_bytes >= biu_c1_wgrad_workspace(dy->c, kd)) {
        int rc = biu_c1_conv_wgrad(x, xf, dy, kd, dw, ws, ws_bytes, dtype, (hipStream_t)stream);
        if (rc == BIU_OK && dbias) rc = biu_chan_sum(dy, dbias, dtype, (hipStream_t)stream);
        return rc;
    }
    if (!disabled("conv_wgrad") && biu_mfma_wgrad_ok(x, dy, kd, kh, kw, dilation, dtype)) {
        BIU_REQUIRE(ws && ws_bytes >= biu_mfma_wgrad_workspace(x->c, dy->c, kd, kh, kw, dtype), BIU_ERR_WORKSPACE,
                    "conv_bwd_weight: workspace too small");
        return biu_mfma_wgrad(x, xf, dy, kd, kh, kw, dw, dbias, ws, ws_bytes, dtype, (hipStream_t)stream);
    }
    warn_slow_path("conv_bwd_weight", x, dy, kd * kh * kw);
    return biu_conv_bwd_weight_direct(x, xf, dy, kd, kh, kw, dilation, dw, dbias, dtype, (hipStream_t)stream);
}

// Weight gradient with the BatchNorm(+LeakyReLU) backward of the conv's own output fused in: `da` holds d loss / d a on entry
// and d loss / d y on return (in place), exactly what biu_bn_bwd_apply would have produced.
extern "C" int biu_conv_bwd_weight_bn(const biu_act* x, const biu_xform* xf, const biu_act* da, const biu_act* y,
                                      const float* scale, const float* shift, const float* slope, const float* coefA,
                                      const float* coefB, const float* coefC, int kd, int kh, int kw, int dilation,
                                      float* dw, void* ws, size_t ws_bytes, int dtype, biu_stream stream) {
    BIU_REQUIRE(conv_args_ok(x, da, kd, kh, kw, dilation) && valid_act(y) && same_space(y, da) && y->c == da->c, BIU_ERR_SHAPE,
                "conv_bwd_weight_bn: extents differ");
    BIU_REQUIRE(dw && scale && shift && coefA && coefB && coefC, BIU_ERR_SHAPE, "conv_bwd_weight_bn: null pointer");
    const size_t es = dsize(dtype);
    const bool yok = ((uintptr_t)y->p % 16) == 0 && ((size_t)y->pitch * es) % 16 == 0 &&
                     (i64)y->d * y->h * y->w * y->pitch * (i64)es < (1LL << 32) - 65536;
    if (!disabled("c1") && !disabled("wgrad_bn") && yok && biu_c1m_ok(x, da, kd, kh, kw, dilation, dtype) && ws &&
        ws_bytes >= biu_c1m_wgrad_workspace(da->c, kd)) {
        // first layer: BatchNorm backward in the dy staging of the im2col weight gradient (da -> dy written back in the same pass)
        BnBwdFuse bn{y, scale, shift, slope, coefA, coefB, coefC};
        return biu_c1m_wgrad(x, xf, da, &bn, kd, dw, ws, ws_bytes, (hipStream_t)stream);
    }
    if (!disabled("conv_wgrad") && !disabled("wgrad_bn") && yok && biu_mfma_wgrad_ok(x, da, kd, kh, kw, dilation, dtype)) {
        BIU_REQUIRE(ws && ws_bytes >= biu_mfma_wgrad_workspace(x->c, da->c, kd, kh, kw, dtype), BIU_ERR_WORKSPACE,
                    "conv_bwd_weight_bn: workspace too small");
        BnBwdFuse bn{y, scale, shift, slope, coefA, coefB, coefC};
        return biu_mfma_wgrad(x, xf, da, kd, kh, kw, dw, nullptr, ws, ws_bytes, dtype, (hipStream_t)stream, &bn);
    }
    int rc = biu_bn_bwd_apply(da, y, scale, shift, slope, coefA, coefB, coefC, da, dtype, stream);
    if (rc != BIU_OK) return rc;
    return biu_conv_bwd_weight(x, xf, da, kd, kh, kw, dilation, dw, nullptr, ws, ws_bytes, dtype, stream);
}

// ---- conv block on a channel concatenation (x0 | x1) held in two buffers ------------------------------------------------
// The decoder's torch.cat (unet/unet.py:62-67, unet3d/unet3d.py:60-61) without a concat buffer: both tensors stay dense, the
// K loop of the forward kernel walks x0's chunks then x1's, the weight gradient takes its input-channel tile from whichever
// holds it, the data gradient writes each output tile to its tensor.  MFMA shapes only (biu_conv_cat_ok says which).
extern "C" int biu_conv_cat_ok(const biu_act* x0, const biu_act* x1, const biu_act* y, int kd, int kh, int kw, int dilation, int dtype) {
    if (!x0 || !x1 || !y || !valid_act(x0) || !valid_act(x1) || !valid_act(y) || !same_space(x0, y) || !same_space(x1, y)) return 0;
    if (disabled("conv_fwd") || disabled("conv_dgrad") || disabled("conv_wgrad") || disabled("cat")) return 0;
    return biu_mfma_conv_cat_ok(x0, x1, y, kd, kh, kw, dilation, dtype) ? 1 : 0;
}
extern "C" int biu_conv_fwd_cat(const biu_act* x0, const biu_xform* xf0, const biu_act* x1, const biu_xform* xf1, const float* w,
                                const void* packed, const float* bias, int kd, int kh, int kw, int dilation, const biu_act* y,
                                float* bn_partial, size_t bn_partial_floats, int* bn_nblk, void* ws, size_t ws_bytes, int dtype,
                                biu_stream stream) {
    BIU_REQUIRE(biu_conv_cat_ok(x0, x1, y, kd, kh, kw, dilation, dtype) && w && packed, BIU_ERR_UNSUPPORTED,
                "conv_fwd_cat: shapes not served by the two-source kernels (ask biu_conv_cat_ok)");
    ConvCat cat{x1, xf1, nullptr, 0};
    if (bn_partial && will_split(x0->c + x1->c, y, nullptr, kd, dtype, ws, ws_bytes)) {       // split launch: statistics in a pass of their own
        BIU_REQUIRE(bn_nblk, BIU_ERR_SHAPE, "conv_fwd_cat: null bn_nblk");
        int rc = biu_mfma_conv(x0, xf0, packed, bias, kd, kh, kw, y, 0, nullptr, dtype, (hipStream_t)stream, nullptr, &cat, ws, ws_bytes);
        if (rc != BIU_OK) return rc;
        BIU_REQUIRE(bn_partial_floats >= (size_t)BIU_BN_MAX_PARTIALS * y->c * 2, BIU_ERR_WORKSPACE, "conv_fwd_cat: partial buffer too small");
        return biu_bn_stats(y, bn_partial, bn_nblk, dtype, stream);
    }
    if (bn_partial) {
        BIU_REQUIRE(bn_nblk, BIU_ERR_SHAPE, "conv_fwd_cat: null bn_nblk");
        const int nb = biu_mfma_conv_stat_rows(y, kd);
        BIU_REQUIRE((size_t)nb * y->c * 2 <= bn_partial_floats, BIU_ERR_WORKSPACE, "conv_fwd_cat: partial buffer too small");
        int rc = biu_mfma_conv(x0, xf0, packed, bias, kd, kh, kw, y, 0, bn_partial, dtype, (hipStream_t)stream, nullptr, &cat);
        if (rc == BIU_OK) *bn_nblk = nb;
        return rc;
    }
    return biu_mfma_conv(x0, xf0, packed, bias, kd, kh, kw, y, 0, nullptr, dtype, (hipStream_t)stream, nullptr, &cat, ws, ws_bytes);
}
extern "C" int biu_conv_bwd_data_cat(const biu_act* dy, const float* w, const void* packed, int kd, int kh, int kw, int dilation,
                                     const biu_act* dx0, int accumulate0, const biu_act* dx1, int accumulate1, void* ws, size_t ws_bytes,
                                     int dtype, biu_stream stream) {
    BIU_REQUIRE(biu_conv_cat_ok(dx0, dx1, dy, kd, kh, kw, dilation, dtype) && w && packed, BIU_ERR_UNSUPPORTED,
                "conv_bwd_data_cat: shapes not served by the two-source kernels (ask biu_conv_cat_ok)");
    ConvCat cat{nullptr, nullptr, dx1, accumulate1};
    return biu_mfma_conv(dy, nullptr, packed, nullptr, kd, kh, kw, dx0, accumulate0, nullptr, dtype, (hipStream_t)stream, nullptr, &cat, ws, ws_bytes);
}
// y == NULL: plain weight gradient (da is dy); y != NULL: BatchNorm backward fused as in biu_conv_bwd_weight_bn
extern "C" int biu_conv_bwd_weight_cat(const biu_act* x0, const biu_xform* xf0, const biu_act* x1, const biu_xform* xf1,
                                       const biu_act* da, const biu_act* y, const float* scale, const float* shift,
                                       const float* slope, const float* coefA, const float* coefB, const float* coefC, int kd,
                                       int kh, int kw, int dilation, float* dw, void* ws, size_t ws_bytes, int dtype,
                                       biu_stream stream) {
    BIU_REQUIRE(biu_conv_cat_ok(x0, x1, da, kd, kh, kw, dilation, dtype) && dw, BIU_ERR_UNSUPPORTED,
                "conv_bwd_weight_cat: shapes not served by the two-source kernels (ask biu_conv_cat_ok)");
    BIU_REQUIRE(ws && ws_bytes >= biu_mfma_wgrad_workspace(x0->c + x1->c, da->c, kd, kh, kw, dtype), BIU_ERR_WORKSPACE,
                "conv_bwd_weight_cat: workspace too small");
    if (y) {
        const size_t es = dsize(dtype);
        BIU_REQUIRE(valid_act(y) && same_space(y, da) && y->c == da->c && scale && shift && coefA && coefB && coefC, BIU_ERR_SHAPE,
                    "conv_bwd_weight_cat: BatchNorm operands missing");
        BIU_REQUIRE(((uintptr_t)y->p % 16) == 0 && ((size_t)y->pitch * es) % 16 == 0 && (i64)y->d * y->h * y->w * y->pitch * (i64)es < (1LL << 32) - 65536,
                    BIU_ERR_UNSUPPORTED, "conv_bwd_weight_cat: y alignment / size");
        BnBwdFuse bn{y, scale, shift, slope, coefA, coefB, coefC};
        return biu_mfma_wgrad(x0, xf0, da, kd, kh, kw, dw, nullptr, ws, ws_bytes, dtype, (hipStream_t)stream, &bn, x1, xf1);
    }
    return biu_mfma_wgrad(x0, xf0, da, kd, kh, kw, dw, nullptr, ws, ws_bytes, dtype, (hipStream_t)stream, nullptr, x1, xf1);
}

// ---- ConvTranspose k2 s2 ---------------------------------------------------------------------------
extern "C" size_t biu_convt_packed_bytes(int kind, int cin, int cout, int kd, int dtype) {
    return biu_mfma_convt_packed_bytes(kind, cin, cout, kd, dtype);
}
extern "C" int biu_convt_pack(int kind, const float* w, int cin, int cout, int kd, int dtype, void* packed, biu_stream stream) {
    BIU_REQUIRE(w && packed, BIU_ERR_SHAPE, "convt_pack: null pointer");
    return biu_mfma_convt_pack(kind, w, cin, cout, kd, dtype, packed, (hipStream_t)stream);
}
// ---- nearest-neighbour up-sampling folded into the 3x3x3 convolution behind it (forward) ----------------------------------
extern "C" int biu_upconv_ok(const biu_act* x, const biu_act* y, int dtype) {
    return (x && y && !disabled("upconv") && biu_mfma_upconv_ok(x, y, dtype)) ? 1 : 0;
}
extern "C" size_t biu_upconv_packed_bytes(int kind, int cin, int cout, int dtype) { return biu_mfma_upconv_packed_bytes(kind, cin, cout, dtype); }
extern "C" int biu_upconv_pack(int kind, const float* w, int cin, int cout, int dtype, void* packed, biu_stream stream) {
    BIU_REQUIRE(w && packed && biu_mfma_upconv_packed_bytes(kind, cin, cout, dtype) > 0, BIU_ERR_SHAPE,
                "upconv_pack: null pointer, kind %d or unsupported channels %d -> %d", kind, cin, cout);
    return biu_mfma_upconv_pack(kind, w, cin, cout, dtype, packed, (hipStream_t)stream);
}
extern "C" size_t biu_upconv_bwd_weight_workspace(int cin, int cout, int dtype) { return biu_mfma_upconv_wgrad_workspace(cin, cout, dtype); }
extern "C" int biu_upconv_bwd_weight_bn(const biu_act* x, const biu_xform* xf, const biu_act* da, const biu_act* y, const float* scale,
                                        const float* shift, const float* slope, const float* coefA, const float* coefB, const float* coefC,
                                        float* dw, void* ws, size_t ws_bytes, int dtype, biu_stream stream) {
    BIU_REQUIRE(x && da && dw && ws, BIU_ERR_SHAPE, "upconv_bwd_weight_bn: null pointer");
    BIU_REQUIRE(biu_mfma_upconv_ok(x, da, dtype) && biu_mfma_upconv_wgrad_workspace(x->c, da->c, dtype) > 0, BIU_ERR_UNSUPPORTED,
                "upconv_bwd_weight_bn: shape %dx%dx%dx%d c%d / %dx%dx%dx%d c%d is not served by the folded kernel", x->n, x->d, x->h, x->w, x->c,
                da->n, da->d, da->h, da->w, da->c);
    if (y) {
        BIU_REQUIRE(valid_act(y) && same_space(y, da) && y->c == da->c && scale && shift && coefA && coefB && coefC, BIU_ERR_SHAPE,
                    "upconv_bwd_weight_bn: y / coefficient vectors do not match da");
        const size_t es = dsize(dtype);
        BIU_REQUIRE(((uintptr_t)y->p % 16) == 0 && ((size_t)y->pitch * es) % 16 == 0 && (i64)y->d * y->h * y->w * y->pitch * (i64)es < (1LL << 32) - 65536,
                    BIU_ERR_UNSUPPORTED, "upconv_bwd_weight_bn: y is not 16-byte aligned or a sample exceeds 4 GB");
        BnBwdFuse bn{y, scale, shift, slope, coefA, coefB, coefC};
        return biu_mfma_upconv_wgrad(x, xf, da, dw, ws, ws_bytes, dtype, (hipStream_t)stream, &bn);
    }
    return biu_mfma_upconv_wgrad(x, xf, da, dw, ws, ws_bytes, dtype, (hipStream_t)stream, nullptr);
}
extern "C" int biu_upconv_bwd_data(const biu_act* dy, const void* packed, const biu_act* dx, int accumulate, int dtype, biu_stream stream) {
    BIU_REQUIRE(dy && dx && packed, BIU_ERR_SHAPE, "upconv_bwd_data: null pointer");
    BIU_REQUIRE(biu_mfma_upconv_ok(dx, dy, dtype) && biu_mfma_upconv_packed_bytes(1, dx->c, dy->c, dtype) > 0, BIU_ERR_UNSUPPORTED,
                "upconv_bwd_data: shape %dx%dx%dx%d c%d <- %dx%dx%dx%d c%d is not served by the folded kernel", dx->n, dx->d, dx->h, dx->w, dx->c,
                dy->n, dy->d, dy->h, dy->w, dy->c);
    return biu_mfma_upconv_dgrad(dy, packed, dx, accumulate, dtype, (hipStream_t)stream);
}
extern "C" size_t biu_upconv_fwd_stats_floats(const biu_act* x, const biu_act* y) {
    return (size_t)biu_mfma_upconv_stat_rows(x, y) * y->c * 2;
}
extern "C" int biu_upconv_fwd(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, const biu_act* y,
                              float* bn_partial, size_t bn_partial_floats, int* bn_nblk, int dtype, biu_stream stream) {
    BIU_REQUIRE(x && y && packed, BIU_ERR_SHAPE, "upconv_fwd: null pointer");
    BIU_REQUIRE(biu_mfma_upconv_ok(x, y, dtype), BIU_ERR_UNSUPPORTED, "upconv_fwd: shape %dx%dx%dx%d c%d -> %dx%dx%dx%d c%d is not served by the folded kernel",
                x->n, x->d, x->h, x->w, x->c, y->n, y->d, y->h, y->w, y->c);
    if (bn_nblk) *bn_nblk = 0;
    if (bn_partial) {
        BIU_REQUIRE(bn_nblk, BIU_ERR_SHAPE, "upconv_fwd: bn_partial without bn_nblk");
        const int nb = biu_mfma_upconv_stat_rows(x, y);
        BIU_REQUIRE((size_t)nb * y->c * 2 <= bn_partial_floats, BIU_ERR_WORKSPACE, "upconv_fwd: partial buffer too small (%zu floats, need %zu)",
                    bn_partial_floats, (size_t)nb * y->c * 2);
        int rc = biu_mfma_upconv_fwd(x, xf, packed, bias, y, bn_partial, dtype, (hipStream_t)stream);
        if (rc == BIU_OK) *bn_nblk = nb;
        return rc;
    }
    return biu_mfma_upconv_fwd(x, xf, packed, bias, y, nullptr, dtype, (hipStream_t)stream);
}

// ---- ConvTranspose(k2, s2) + concat + 3x3x3 conv of a decoder level, the up half folded onto the coarse tensor -------------------------
// 1: the folded kernels serve the level AND it is large enough for the fold to pay (BIU_FOLDT=always: wherever they serve it -- the tests)
extern "C" int biu_foldt_ok(const biu_act* x_low, const biu_act* skip, const biu_act* y, int dtype) {
    // BIU_FOLDT=always: wherever the kernels serve the level (the tests); BIU_FOLDT=cmax:N: exactly the levels whose coarse input has at
    // most N channels -- reproduces at test extents the pattern the size rule picks at a benchmark's extents (cfg4: cmax:128 = decode5 and
    // decode3 folded, decode1 through the 3-D ConvT + two-source kernels; tests/test_gpu_bench_dispatch.py)
    static int always = -1, cmax = 0;
    if (always < 0) {
        const char* e = getenv("BIU_FOLDT");
        const char* c = e ? strstr(e, "cmax:") : nullptr;
        cmax = c ? atoi(c + 5) : 0;
        always = (e && strstr(e, "always")) ? 1 : 0;
    }
    if (!(x_low && skip && y) || disabled("foldt") || !biu_mfma_foldt_ok(x_low, skip, y, dtype)) return 0;
    if (cmax > 0) return x_low->c <= cmax ? 1 : 0;
    return (always || biu_mfma_foldt_worth(x_low, y)) ? 1 : 0;
}
extern "C" size_t biu_foldt_packed_bytes(int cin_low, int cskip, int cout, int dtype) { return biu_mfma_foldt_packed_bytes(cin_low, cskip, cout, dtype); }
extern "C" int biu_foldt_pack(const float* w_conv, const float* b_conv, const float* w_t, const float* b_t, int cin_low, int cup, int cskip, int cout,
                              int dtype, void* packed, biu_stream stream) {
    BIU_REQUIRE(w_conv && w_t && packed && cin_low > 0 && cup > 0 && cskip > 0 && cout > 0, BIU_ERR_SHAPE, "foldt_pack: null pointer or empty shape");
    return biu_mfma_foldt_pack(w_conv, b_conv, w_t, b_t, cin_low, cup, cskip, cout, dtype, packed, (hipStream_t)stream);
}
// which form biu_foldt_fwd takes for these tensors: 0 = brick kernels (skip half + biases stored, border shell corrected, fold accumulated on top),
// 1 = rolling-window kernels (fold + border-state bias stored first, the skip half accumulated onto it): the two round intermediate results at
// different places, which a checker that models the storage roundings must know (tests/insitu.py)
extern "C" int biu_foldt_fwd_form(const biu_act* x_low, const biu_act* skip, const biu_act* y, int dtype) {
    if (!x_low || !skip || !y) return 0;
    return biu_mfma_foldt_form(x_low, skip, y, dtype);
}
extern "C" size_t biu_foldt_fwd_stats_floats(const biu_act* x_low, const biu_act* y) {
    // (an upper bound over the forms the launch can take: 8 rows per block of the brick form, at most one row per CU x 4 of the rolling one)
    const size_t a = (size_t)biu_mfma_foldt_stat_rows(x_low, y) * y->c * 2, b = (size_t)1024 * y->c * 2;
    return a > b ? a : b;
}
extern "C" int biu_foldt_fwd(const biu_act* x_low, const biu_xform* xf_low, const biu_act* skip, const biu_xform* xf_skip, const void* packed,
                             const biu_act* y, float* bn_partial, size_t bn_partial_floats, int* bn_nblk, int dtype, biu_stream stream) {
    BIU_REQUIRE(x_low && skip && y && packed, BIU_ERR_SHAPE, "foldt_fwd: null pointer");
    BIU_REQUIRE(biu_mfma_foldt_ok(x_low, skip, y, dtype), BIU_ERR_UNSUPPORTED, "foldt_fwd: shapes are not served by the folded kernels");
    if (bn_nblk) *bn_nblk = 0;
    if (bn_partial) {
        BIU_REQUIRE(bn_nblk, BIU_ERR_SHAPE, "foldt_fwd: bn_partial without bn_nblk");
        const int nb = biu_mfma_foldt_stat_rows(x_low, y, skip, dtype);            // (the rolling-window form: one row per block of the skip half's launch)
        BIU_REQUIRE((size_t)nb * y->c * 2 <= bn_partial_floats, BIU_ERR_WORKSPACE, "foldt_fwd: partial buffer too small");
        int rc = biu_mfma_foldt_fwd(x_low, xf_low, skip, xf_skip, packed, y, bn_partial, dtype, (hipStream_t)stream);
        if (rc == BIU_OK) *bn_nblk = nb;
        return rc;
    }
    return biu_mfma_foldt_fwd(x_low, xf_low, skip, xf_skip, packed, y, nullptr, dtype, (hipStream_t)stream);
}

extern "C" size_t biu_foldt_bwd_data_bnred_floats(const biu_act* dx_low) { return (size_t)biu_mfma_upconv_dgrad_rows(dx_low) * dx_low->c * 2; }
extern "C" int biu_foldt_bwd_data(const biu_act* dy, const void* packed, const biu_act* dx_low, int acc_low, const biu_act* dskip, int acc_skip,
                                  const biu_act* y_low, const float* scale, const float* shift, const float* slope, const float* mean,
                                  const float* invstd, float* partial, size_t partial_floats, int* nblk, void* ws, size_t ws_bytes, int dtype,
                                  biu_stream stream) {
    BIU_REQUIRE(dy && packed && dx_low && dskip, BIU_ERR_SHAPE, "foldt_bwd_data: null pointer");
    BIU_REQUIRE(biu_mfma_foldt_ok(dx_low, dskip, dy, dtype), BIU_ERR_UNSUPPORTED, "foldt_bwd_data: shapes are not served by the folded kernels");
    if (nblk) *nblk = 0;
    if (y_low) {                  // BatchNorm-backward sums of x_low's producer from the epilogue (x_low's gradient is complete after this write)
        BIU_REQUIRE(!acc_low && valid_act(y_low) && same_space(y_low, dx_low) && y_low->c == dx_low->c && scale && shift && mean && invstd && partial && nblk,
                    BIU_ERR_SHAPE, "foldt_bwd_data: fused reduction needs acc_low = 0, y_low like dx_low and its vectors");
        const int nb = biu_mfma_upconv_dgrad_rows(dx_low);
        BIU_REQUIRE((size_t)nb * dx_low->c * 2 <= partial_floats, BIU_ERR_WORKSPACE, "foldt_bwd_data: partial buffer too small");
        BnRedFuse red{y_low, scale, shift, slope, mean, invstd};
        int rc = biu_mfma_foldt_dgrad(dy, packed, dx_low, 0, dskip, acc_skip, dtype, (hipStream_t)stream, partial, &red, ws, ws_bytes);
        if (rc == BIU_OK) *nblk = nb;
        return rc;
    }
    return biu_mfma_foldt_dgrad(dy, packed, dx_low, acc_low, dskip, acc_skip, dtype, (hipStream_t)stream, nullptr, nullptr, ws, ws_bytes);
}
extern "C" size_t biu_foldt_bwd_weight_workspace(int cin_low, int cskip, int cout, int dtype) { return biu_mfma_foldt_wgrad_workspace(cin_low, cskip, cout, dtype); }
extern "C" int biu_foldt_bwd_weight_bn_phase(const biu_act* x_low, const biu_xform* xf_low, const biu_act* skip, const biu_xform* xf_skip, const biu_act* da,
                                             const biu_act* y, const float* scale, const float* shift, const float* slope, const float* coefA,
                                             const float* coefB, const float* coefC, const float* dy_sum, const float* w_conv, const float* w_t, const float* b_t,
                                             int cup, float* dw_conv, float* dw_t, float* db_t, void* ws, size_t ws_bytes, int dtype, int phases, biu_stream stream) {
    BIU_REQUIRE(phases >= 1 && phases <= 7, BIU_ERR_SHAPE, "foldt_bwd_weight_bn: phases is a mask of 1 (skip half, da -> dy), 4 (G), 2 (border sums + chain rule)");
    BIU_REQUIRE(x_low && skip && da && w_conv && w_t && dw_conv && dw_t && ws && cup > 0, BIU_ERR_SHAPE, "foldt_bwd_weight_bn: null pointer");
    BIU_REQUIRE(biu_mfma_foldt_ok(x_low, skip, da, dtype) && biu_mfma_wgrad_ok(skip, da, 3, 3, 3, 1, dtype), BIU_ERR_UNSUPPORTED,
                "foldt_bwd_weight_bn: shapes are not served by the folded kernels");
    // S_k (the sum of dy over the voxels whose tap k stays inside) = sum_v dy - (border sums): without dy_sum the total is taken as zero, which
    // holds only for the dy of a train-mode BatchNorm -- a plain dy (y = NULL) has no such guarantee
    BIU_REQUIRE(y || dy_sum || !(b_t || db_t), BIU_ERR_UNSUPPORTED,
                "foldt_bwd_weight_bn: a plain dy (y = NULL) needs dy_sum, its per-channel sum over all voxels (it is zero only behind a train-mode BatchNorm)");
    if (y) {
        BIU_REQUIRE(valid_act(y) && same_space(y, da) && y->c == da->c && scale && shift && coefA && coefB && coefC, BIU_ERR_SHAPE,
                    "foldt_bwd_weight_bn: y / coefficient vectors do not match da");
        const size_t es = dsize(dtype);
        const bool yok = ((uintptr_t)y->p % 16) == 0 && ((size_t)y->pitch * es) % 16 == 0 &&
                         (i64)y->d * y->h * y->w * y->pitch * (i64)es < (1LL << 32) - 65536;
        if (!yok) {                                      // the BatchNorm-fused loader reads y in 16-byte pieces through a 32-bit descriptor: apply first
            int rc = (phases & 1) ? biu_bn_bwd_apply(da, y, scale, shift, slope, coefA, coefB, coefC, da, dtype, stream) : BIU_OK;
            if (rc != BIU_OK) return rc;
            return biu_mfma_foldt_wgrad(x_low, xf_low, skip, xf_skip, da, nullptr, dy_sum, w_conv, w_t, b_t, cup, dw_conv, dw_t, db_t, ws, ws_bytes, dtype,
                                        (hipStream_t)stream, phases);
        }
        BnBwdFuse bn{y, scale, shift, slope, coefA, coefB, coefC};
        return biu_mfma_foldt_wgrad(x_low, xf_low, skip, xf_skip, da, &bn, dy_sum, w_conv, w_t, b_t, cup, dw_conv, dw_t, db_t, ws, ws_bytes, dtype, (hipStream_t)stream, phases);
    }
    return biu_mfma_foldt_wgrad(x_low, xf_low, skip, xf_skip, da, nullptr, dy_sum, w_conv, w_t, b_t, cup, dw_conv, dw_t, db_t, ws, ws_bytes, dtype, (hipStream_t)stream, phases);
}

extern "C" int biu_foldt_bwd_weight_bn(const biu_act* x_low, const biu_xform* xf_low, const biu_act* skip, const biu_xform* xf_skip, const biu_act* da,
                                       const biu_act* y, const float* scale, const float* shift, const float* slope, const float* coefA,
                                       const float* coefB, const float* coefC, const float* dy_sum, const float* w_conv, const float* w_t, const float* b_t,
                                       int cup, float* dw_conv, float* dw_t, float* db_t, void* ws, size_t ws_bytes, int dtype, biu_stream stream) {
    return biu_foldt_bwd_weight_bn_phase(x_low, xf_low, skip, xf_skip, da, y, scale, shift, slope, coefA, coefB, coefC, dy_sum, w_conv, w_t, b_t, cup, dw_conv, dw_t,
                                         db_t, ws, ws_bytes, dtype, 7, stream);
}

extern "C" int biu_convt_fwd(const biu_act* x, const biu_xform* xf, const float* w, const void* packed, const float* bias,
                             int kd, const biu_act* y, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(x) && valid_act(y) && w && biu_convt_shapes_ok(x, y, kd), BIU_ERR_SHAPE,
                "convt_fwd: output must be 2x the input extent (kd=%d)", kd);
    if (packed && !disabled("convt_fwd") && biu_mfma_convt_ok(0, x, y, kd, dtype)) {
        if (biu_convt_all_ok(x, y, kd, dtype)) return biu_convt_all_fwd(x, xf, packed, bias, kd, y, (hipStream_t)stream);     // all parities from one tile
        return biu_mfma_convt_fwd(x, xf, packed, bias, kd, y, dtype, (hipStream_t)stream);
    }
    return biu_convt_fwd_direct(x, xf, w, bias, kd, y, dtype, (hipStream_t)stream);
}
extern "C" int biu_convt_bwd_data(const biu_act* dy, const float* w, const void* packed, int kd, const biu_act* dx,
                                  int accumulate, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(dx) && valid_act(dy) && w && biu_convt_shapes_ok(dx, dy, kd), BIU_ERR_SHAPE,
                "convt_bwd_data: dy must be 2x the dx extent (kd=%d)", kd);
    if (packed && !disabled("convt_dgrad") && biu_mfma_convt_ok(1, dx, dy, kd, dtype)) {
        if (biu_convt_all_ok(dx, dy, kd, dtype)) return biu_convt_all_dgrad(dy, packed, kd, dx, accumulate, (hipStream_t)stream);
        return biu_mfma_convt_dgrad(dy, packed, kd, dx, accumulate, dtype, (hipStream_t)stream);
    }
    return biu_convt_bwd_data_direct(dy, w, kd, dx, accumulate, dtype, (hipStream_t)stream);
}
extern "C" size_t biu_convt_bwd_weight_workspace(int cin, int cout, int kd, int dtype) {
    return biu_mfma_wgrad_workspace(cin, cout, kd, 2, 2, dtype);
}
extern "C" int biu_convt_bwd_weight(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, float* dw,
                                    float* dbias, void* ws, size_t ws_bytes, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(x) && valid_act(dy) && dw && biu_convt_shapes_ok(x, dy, kd), BIU_ERR_SHAPE,
                "convt_bwd_weight: dy must be 2x the x extent (kd=%d)", kd);
    if (!disabled("convt_wgrad") && biu_mfma_convt_wgrad_ok(x, dy, kd, dtype)) {
        BIU_REQUIRE(ws && ws_bytes >= biu_mfma_wgrad_workspace(x->c, dy->c, kd, 2, 2, dtype), BIU_ERR_WORKSPACE,
                    "convt_bwd_weight: workspace too small");
        return biu_mfma_convt_wgrad(x, xf, dy, kd, dw, dbias, ws, ws_bytes, dtype, (hipStream_t)stream);
    }
    return biu_convt_bwd_weight_direct(x, xf, dy, kd, dw, dbias, dtype, (hipStream_t)stream);
}
