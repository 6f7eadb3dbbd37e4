// Two ops off the main stacks, any shape, scalar per (voxel, channel) with the channel fastest (coalesced rows):
//   * trilinear x2 up-sampling, align_corners = False  -- UNet3D(use_interpolation=True)   [unet3d/unet3d.py:82,89,96]
//   * depth-wise cross-correlation of two bottleneck maps -- Siam_UNet(mode='corr')        [siam_unet/siam_unet.py:75-83]
// Both sit at coarse resolutions; they are correctness-tier kernels (HBM/L2-bound gathers), not MFMA paths.
#include <hip/hip_runtime.h>

#include "biu_common.h"
#include "biu_internal.h"

namespace {

constexpr int TPB = 256;

// source taps of output index o for scale 2, align_corners = False (PyTorch area_pixel_compute_source_index):
//   src = max(0.5 * (o + 0.5) - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, n - 1), l1 = src - i0, l0 = 1 - l1
struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap tap_of(int o, int n_in, bool scaled) {
    if (!scaled) return Tap{o, o, 1.f, 0.f};
    float src = 0.5f * ((float)o + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    const int i0 = (int)src;
    const int i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
    const float l1 = src - (float)i0;
    return Tap{i0, i1, 1.f - l1, l1};
}
__device__ __forceinline__ void split(i64 v, int d, int h, int w, int& n, int& z, int& y, int& x) {
    x = (int)(v % w); v /= w;
    y = (int)(v % h); v /= h;
    z = (int)(v % d);
    n = (int)(v / d);
}

template <typename T>
__global__ void k_trilinear_up_fwd(DAct x, DXf xf, DAct out, int sd) {
    const i64 total = (i64)out.n * out.d * out.h * out.w * out.c;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        const int c = (int)(i % out.c);
        const i64 ov = i / out.c;
        int n, z, y, xx;
        split(ov, out.d, out.h, out.w, n, z, y, xx);
        const Tap td = tap_of(z, x.d, sd == 2), th = tap_of(y, x.h, true), tw = tap_of(xx, x.w, true);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float wgt = (a ? td.l1 : td.l0) * (b ? th.l1 : th.l0) * (e ? tw.l1 : tw.l0);
                    const i64 iv = (((i64)n * x.d + (a ? td.i1 : td.i0)) * x.h + (b ? th.i1 : th.i0)) * x.w + (e ? tw.i1 : tw.i0);
                    acc = fmaf(wgt, xf_apply(xf, c, ld_act<T>(x, iv, c)), acc);
                }
        st_act<T>(out, ov, c, acc);
    }
}

// gather form of the adjoint: dx[i] = sum over the (<= 4 per axis) outputs that tap i
template <typename T>
__global__ void k_trilinear_up_bwd(DAct dout, DAct dx, int sd, int accumulate) {
    const i64 total = (i64)dx.n * dx.d * dx.h * dx.w * dx.c;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        const int c = (int)(i % dx.c);
        const i64 v = i / dx.c;
        int n, z, y, xx;
        split(v, dx.d, dx.h, dx.w, n, z, y, xx);
        float acc = 0.f;
        const int zlo = sd == 2 ? 2 * z - 1 : z, zhi = sd == 2 ? 2 * z + 2 : z;
        for (int oz = zlo; oz <= zhi; ++oz) {
            if (oz < 0 || oz >= dout.d) continue;
            const Tap td = tap_of(oz, dx.d, sd == 2);
            const float wz = (td.i0 == z ? td.l0 : 0.f) + (td.i1 == z ? td.l1 : 0.f);
            if (wz == 0.f) continue;
            for (int oy = 2 * y - 1; oy <= 2 * y + 2; ++oy) {
                if (oy < 0 || oy >= dout.h) continue;
                const Tap th = tap_of(oy, dx.h, true);
                const float wy = (th.i0 == y ? th.l0 : 0.f) + (th.i1 == y ? th.l1 : 0.f);
                if (wy == 0.f) continue;
                for (int ox = 2 * xx - 1; ox <= 2 * xx + 2; ++ox) {
                    if (ox < 0 || ox >= dout.w) continue;
                    const Tap tw = tap_of(ox, dx.w, true);
                    const float wx = (tw.i0 == xx ? tw.l0 : 0.f) + (tw.i1 == xx ? tw.l1 : 0.f);
                    if (wx == 0.f) continue;
                    const i64 ov = (((i64)n * dout.d + oz) * dout.h + oy) * dout.w + ox;
                    acc = fmaf(wz * wy * wx, ld_act<T>(dout, ov, c), acc);
                }
            }
        }
        if (accumulate) acc += ld_act<T>(dx, v, c);
        st_act<T>(dx, v, c, acc);
    }
}

// out[n,y,x,c] = sum_{i,j} T(cur)[n, y + i - ph, x + j - pw, c] * T(prev)[n, i, j, c]   (zero outside), kernel = the whole map,
// padding='same': ph = (H - 1) / 2, pw = (W - 1) / 2 on the low side (PyTorch pads the remainder on the high side)
template <typename T>
__global__ void k_xcorr_fwd(DAct cur, DXf xc, DAct prev, DXf xp, DAct out) {
    const int H = cur.h, W = cur.w, ph = (H - 1) / 2, pw = (W - 1) / 2;
    const i64 total = (i64)out.n * H * W * out.c;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
        const int c = (int)(t % out.c);
        i64 v = t / out.c;
        const int x = (int)(v % W); v /= W;
        const int y = (int)(v % H);
        const int n = (int)(v / H);
        float acc = 0.f;
        for (int i = 0; i < H; ++i) {
            const int yy = y + i - ph;
            if (yy < 0 || yy >= H) continue;
            for (int j = 0; j < W; ++j) {
                const int xx = x + j - pw;
                if (xx < 0 || xx >= W) continue;
                const float a = xf_apply(xc, c, ld_act<T>(cur, ((i64)n * H + yy) * W + xx, c));
                const float b = xf_apply(xp, c, ld_act<T>(prev, ((i64)n * H + i) * W + j, c));
                acc = fmaf(a, b, acc);
            }
        }
        st_act<T>(out, ((i64)n * H + y) * W + x, c, acc);
    }
}
// which = 0: d cur[n,yy,xx,c] = sum_{i,j} dout[n, yy - i + ph, xx - j + pw, c] * prev[n,i,j,c]
// which = 1: d prev[n,i,j,c]  = sum_{y,x} dout[n,y,x,c] * cur[n, y + i - ph, x + j - pw, c]
template <typename T>
__global__ void k_xcorr_bwd(DAct other, DXf xo, DAct dout, DAct dst, int which, int accumulate) {
    const int H = dout.h, W = dout.w, ph = (H - 1) / 2, pw = (W - 1) / 2;
    const i64 total = (i64)dst.n * H * W * dst.c;
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
        const int c = (int)(t % dst.c);
        i64 v = t / dst.c;
        const int q = (int)(v % W); v /= W;
        const int p = (int)(v % H);
        const int n = (int)(v / H);
        float acc = 0.f;
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                // which 0: (p,q) = (yy,xx), (i,j) runs over prev ; which 1: (p,q) = kernel index, (i,j) runs over outputs (y,x)
                const int oy = which == 0 ? p - i + ph : i, ox = which == 0 ? q - j + pw : j;
                const int sy = which == 0 ? i : i + p - ph, sx = which == 0 ? j : j + q - pw;
                if (oy < 0 || oy >= H || ox < 0 || ox >= W || sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
                const float g = ld_act<T>(dout, ((i64)n * H + oy) * W + ox, c);
                const float o = xf_apply(xo, c, ld_act<T>(other, ((i64)n * H + sy) * W + sx, c));
                acc = fmaf(g, o, acc);
            }
        const i64 dv = ((i64)n * H + p) * W + q;
        if (accumulate) acc += ld_act<T>(dst, dv, c);
        st_act<T>(dst, dv, c, acc);
    }
}

int up_depth(const biu_act* lo, const biu_act* hi, const char* who) {
    if (lo->n != hi->n || lo->c != hi->c || hi->h != 2 * lo->h || hi->w != 2 * lo->w) {
        biu_fail(BIU_ERR_SHAPE, "%s: expected (h,w) = 2x of the coarse tensor and equal n,c", who);
        return 0;
    }
    if (hi->d == lo->d) return 1;
    if (hi->d == 2 * lo->d) return 2;
    biu_fail(BIU_ERR_SHAPE, "%s: depth %d vs %d is neither 1x nor 2x", who, hi->d, lo->d);
    return 0;
}

}  // namespace

extern "C" int biu_trilinear_up_fwd(const biu_act* x, const biu_xform* xf, const biu_act* out, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(x) && valid_act(out), BIU_ERR_SHAPE, "trilinear_up_fwd: bad tensor");
    const int sd = up_depth(x, out, "trilinear_up_fwd");
    if (!sd) return BIU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_trilinear_up_fwd<T>, dim3(grid_for(nvox(out) * out->c, TPB, 16384)), dim3(TPB), 0, st,
                                                 dact(x), dxf(xf), dact(out), sd));
    BIU_CHECK_LAUNCH("trilinear_up_fwd");
    return BIU_OK;
}
extern "C" int biu_trilinear_up_bwd(const biu_act* dout, const biu_act* dx, int accumulate, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(dout) && valid_act(dx), BIU_ERR_SHAPE, "trilinear_up_bwd: bad tensor");
    const int sd = up_depth(dx, dout, "trilinear_up_bwd");
    if (!sd) return BIU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_trilinear_up_bwd<T>, dim3(grid_for(nvox(dx) * dx->c, TPB, 16384)), dim3(TPB), 0, st,
                                                 dact(dout), dact(dx), sd, accumulate));
    BIU_CHECK_LAUNCH("trilinear_up_bwd");
    return BIU_OK;
}

extern "C" int biu_xcorr_fwd(const biu_act* cur, const biu_xform* xc, const biu_act* prev, const biu_xform* xp, const biu_act* out,
                             int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(cur) && valid_act(prev) && valid_act(out) && same_space(cur, prev) && same_space(cur, out) &&
                    cur->c == prev->c && cur->c == out->c && cur->d == 1, BIU_ERR_SHAPE, "xcorr_fwd: three equal 2-D tensors expected");
    hipStream_t st = (hipStream_t)stream;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_xcorr_fwd<T>, dim3(grid_for(nvox(out) * out->c, TPB, 16384)), dim3(TPB), 0, st,
                                                 dact(cur), dxf(xc), dact(prev), dxf(xp), dact(out)));
    BIU_CHECK_LAUNCH("xcorr_fwd");
    return BIU_OK;
}
extern "C" int biu_xcorr_bwd(const biu_act* cur, const biu_xform* xc, const biu_act* prev, const biu_xform* xp, const biu_act* dout,
                             const biu_act* dcur, const biu_act* dprev, int accumulate, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(cur) && valid_act(prev) && valid_act(dout) && valid_act(dcur) && valid_act(dprev) && same_space(cur, prev) &&
                    same_space(cur, dout) && same_space(cur, dcur) && same_space(cur, dprev) && cur->c == prev->c && cur->c == dout->c &&
                    cur->c == dcur->c && cur->c == dprev->c && cur->d == 1, BIU_ERR_SHAPE, "xcorr_bwd: equal 2-D tensors expected");
    BIU_REQUIRE(!(xc && (xc->scale || xc->shift || xc->slope)) && !(xp && (xp->scale || xp->shift || xp->slope)), BIU_ERR_UNSUPPORTED,
                "xcorr_bwd: operands must be materialised (identity transform); the gradient is taken w.r.t. the stored values");
    hipStream_t st = (hipStream_t)stream;
    const i64 tot = nvox(cur) * cur->c;
    BIU_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL(k_xcorr_bwd<T>, dim3(grid_for(tot, TPB, 16384)), dim3(TPB), 0, st, dact(prev), dxf(xp), dact(dout), dact(dcur), 0, accumulate);
        hipLaunchKernelGGL(k_xcorr_bwd<T>, dim3(grid_for(tot, TPB, 16384)), dim3(TPB), 0, st, dact(cur), dxf(xc), dact(dout), dact(dprev), 1, accumulate);
    });
    BIU_CHECK_LAUNCH("xcorr_bwd");
    return BIU_OK;
}

// =====================================================================================================================
// BCEDiceLoss in one pass over (logits, target), fp32 NC[D]HW as the heads emit them [unet/losses.py:78-112]:
//   fwd: per sample n the four sums  sum bce(l,t), sum p, sum t, sum p*t   (p = sigmoid(l)), as per-block partials
//   bwd: dl_i (+)= c_bce * (p_i - t_i) + (c_p[n] + c_pt[n] * t_i) * p_i * (1 - p_i)   with host-made coefficients
// The reference spends three full-tensor reductions with 4 output rows each (one workgroup per row) on this.
// =====================================================================================================================
namespace {
__global__ __launch_bounds__(256) void k_bce_dice_fwd(const float* __restrict__ lg, const float* __restrict__ tg, i64 per_sample,
                                                      float* __restrict__ partial /* [n][gridDim.x][4] */) {
    const int n = blockIdx.y;
    const float* l = lg + (i64)n * per_sample;
    const float* t = tg + (i64)n * per_sample;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (i64)gridDim.x * blockDim.x) {
        const float x = l[i], y = t[i];
        const float e = __expf(-fabsf(x));
        s0 += fmaxf(x, 0.f) - x * y + log1pf(e);
        const float p = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        s1 += p;
        s2 += y;
        s3 += p * y;
    }
    __shared__ float red[16];
    float* dst = partial + ((i64)n * gridDim.x + blockIdx.x) * 4;
    float r;
    r = block_sum(s0, red); if (threadIdx.x == 0) dst[0] = r;
    r = block_sum(s1, red); if (threadIdx.x == 0) dst[1] = r;
    r = block_sum(s2, red); if (threadIdx.x == 0) dst[2] = r;
    r = block_sum(s3, red); if (threadIdx.x == 0) dst[3] = r;
}
__global__ __launch_bounds__(256) void k_bce_dice_bwd(const float* __restrict__ lg, const float* __restrict__ tg, i64 per_sample,
                                                      const float* __restrict__ coef /* [n][3]: c_bce, c_p, c_pt */,
                                                      float* __restrict__ dl, int accumulate) {
    const int n = blockIdx.y;
    const float cb = coef[n * 3 + 0], cp = coef[n * 3 + 1], cpt = coef[n * 3 + 2];
    const i64 base = (i64)n * per_sample;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (i64)gridDim.x * blockDim.x) {
        const float x = lg[base + i], y = tg[base + i];
        const float e = __expf(-fabsf(x));
        const float p = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        float g = cb * (p - y) + (cp + cpt * y) * p * (1.f - p);
        if (accumulate) g += dl[base + i];
        dl[base + i] = g;
    }
}

// SmoothL1 (beta = 1, mean) between neighbouring BATCH entries of the logits -- the 3-D trainer's "time" term
// nn.SmoothL1Loss()(y_logits[1:], y_logits[:-1]) (unet3d/train.py:140-145).  d_i = l[i+1] - l[i] for i < (n-1)*per_sample.
__device__ __forceinline__ float sl1(float d) { const float a = fabsf(d); return a < 1.f ? 0.5f * d * d : a - 0.5f; }
__device__ __forceinline__ float dsl1(float d) { return d > 1.f ? 1.f : (d < -1.f ? -1.f : d); }
__global__ __launch_bounds__(256) void k_pair_sl1_fwd(const float* __restrict__ lg, i64 per_sample, i64 pairs, float* __restrict__ partial) {
    float s = 0.f;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (i64)gridDim.x * blockDim.x) s += sl1(lg[i + per_sample] - lg[i]);
    __shared__ float red[16];
    const float r = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// dl[j] (+)= c * (h'(l[j] - l[j - P]) [j >= P]  -  h'(l[j + P] - l[j]) [j < pairs]),  c = upstream gradient * weight / pairs
__global__ __launch_bounds__(256) void k_pair_sl1_bwd(const float* __restrict__ lg, i64 per_sample, i64 total, const float* __restrict__ coef,
                                                      float* __restrict__ dl, int accumulate) {
    const float c = coef[0];
    const i64 pairs = total - per_sample;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (i64)gridDim.x * blockDim.x) {
        const float x = lg[j];
        float g = 0.f;
        if (j >= per_sample) g += dsl1(x - lg[j - per_sample]);
        if (j < pairs) g -= dsl1(lg[j + per_sample] - x);
        g *= c;
        dl[j] = accumulate ? dl[j] + g : g;
    }
}

// d loss / d logits of one head from the caller's gradients w.r.t. (logits, activated output), written into channels
// [c0, c0 + ch) of a [n, ctot, S] buffer (the stacked operand of the multi-head backward): out = g_logits + g_act * f'(.)
__global__ __launch_bounds__(256) void k_head_dlogits(const float* __restrict__ gl, const float* __restrict__ ga, const float* __restrict__ av,
                                                      int act, int ch, i64 S, float* __restrict__ dst, int ctot, int c0, i64 total) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        const i64 s = i % S;
        const i64 t = i / S;
        const int c = (int)(t % ch);
        const i64 n = t / ch;
        float g = gl ? gl[i] : 0.f;
        if (ga) {
            const float a = av ? av[i] : 0.f;
            const float d = act == 1 ? a * (1.f - a) : (act == 2 ? 1.f - a * a : (act == 3 ? (a > 0.f ? 1.f : 0.f) : 1.f));
            g = fmaf(ga[i], d, g);
        }
        dst[(n * ctot + c0 + c) * S + s] = g;
    }
}
}  // namespace

extern "C" int biu_pair_smooth_l1_blocks(long long pairs) {
    long long b = (pairs + 256 * 8 - 1) / (256 * 8);
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
extern "C" int biu_pair_smooth_l1_fwd(const float* logits, int n, long long per_sample, float* partial, biu_stream stream) {
    BIU_REQUIRE(logits && partial && n > 1 && per_sample > 0, BIU_ERR_SHAPE, "pair_smooth_l1_fwd: needs at least two batch entries");
    const i64 pairs = (i64)(n - 1) * per_sample;
    hipLaunchKernelGGL(k_pair_sl1_fwd, dim3(biu_pair_smooth_l1_blocks(pairs)), dim3(256), 0, (hipStream_t)stream, logits, (i64)per_sample, pairs, partial);
    BIU_CHECK_LAUNCH("pair_smooth_l1_fwd");
    return BIU_OK;
}
extern "C" int biu_pair_smooth_l1_bwd(const float* logits, int n, long long per_sample, const float* coef, float* dlogits, int accumulate,
                                      biu_stream stream) {
    BIU_REQUIRE(logits && coef && dlogits && n > 1 && per_sample > 0, BIU_ERR_SHAPE, "pair_smooth_l1_bwd: bad arguments");
    const i64 total = (i64)n * per_sample;
    hipLaunchKernelGGL(k_pair_sl1_bwd, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, logits, (i64)per_sample, total, coef,
                       dlogits, accumulate);
    BIU_CHECK_LAUNCH("pair_smooth_l1_bwd");
    return BIU_OK;
}
extern "C" int biu_head_dlogits(const float* g_logits, const float* g_act, const float* activated, int act, int n, int ch, long long spatial,
                                float* dst, int dst_channels, int dst_c0, biu_stream stream) {
    BIU_REQUIRE((g_logits || g_act) && dst && n > 0 && ch > 0 && spatial > 0 && dst_c0 >= 0 && dst_c0 + ch <= dst_channels, BIU_ERR_SHAPE,
                "head_dlogits: bad arguments");
    BIU_REQUIRE(!g_act || act == 0 || activated, BIU_ERR_SHAPE, "head_dlogits: the activation's gradient needs the activated output");
    const i64 total = (i64)n * ch * spatial;
    hipLaunchKernelGGL(k_head_dlogits, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, g_logits, g_act, activated, act, ch,
                       (i64)spatial, dst, dst_channels, dst_c0, total);
    BIU_CHECK_LAUNCH("head_dlogits");
    return BIU_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The scalar part of the fused segmentation losses (bio_image_unet_amd/losses.py: _FusedSegLoss) in two one-block kernels instead of
// ~20 tiny torch launches per step: biu_seg_loss_finish merges the per-block partial sums and evaluates the loss, biu_seg_loss_coef
// turns the incoming gradient into the per-sample coefficients biu_bce_dice_bwd / biu_pair_smooth_l1_bwd take.
//   saved = { loss, sums[n][4] (BCE, P, T, P.T), den[n], tp, tden, tv }
// ---------------------------------------------------------------------------------------------------------------------
struct SegLossCfg { float a_bce, a_dice, smooth; int has_tv; float al, be, sm; int logcosh; float w_time; long long pairs; int nbt; };
namespace {
__global__ __launch_bounds__(256) void k_seg_loss_finish(const float* __restrict__ partial, int n, int nb, long long per,
                                                        const float* __restrict__ pt, SegLossCfg c, float* __restrict__ saved) {
    __shared__ double red[256];
    float* sums = saved + 1;
    // one wave per (sample, quantity) sum, lanes stride over the per-block partials, fixed-order butterfly in fp64 (16 threads walking
    // 1024 partials each took 0.1 ms at 4 x 128^3)
    for (int idx = threadIdx.x >> 6; idx < n * 4; idx += 4) {
        const int i = idx >> 2, k = idx & 3, lane = threadIdx.x & 63;
        double a = 0.0;
        for (int b = lane; b < nb; b += 64) a += (double)partial[((size_t)i * nb + b) * 4 + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if (lane == 0) sums[idx] = (float)a;
    }
    __syncthreads();
    double tsum = 0.0;
    if (pt)
        for (int b = threadIdx.x; b < c.nbt; b += 256) tsum += (double)pt[b];
    red[threadIdx.x] = tsum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double loss = 0.0, bce = 0.0, dice = 0.0, tp = 0.0, ps = 0.0, ts = 0.0;
        float* den = saved + 1 + 4 * n;
        for (int i = 0; i < n; ++i) {
            bce += (double)sums[4 * i];
            ps += (double)sums[4 * i + 1]; ts += (double)sums[4 * i + 2]; tp += (double)sums[4 * i + 3];
            const float d = sums[4 * i + 1] + sums[4 * i + 2] + c.smooth;
            den[i] = d;
            dice += 2.0 * ((double)sums[4 * i + 3] + c.smooth) / (double)d;
        }
        if (c.a_bce != 0.f) loss += (double)c.a_bce * bce / ((double)n * (double)per);
        if (c.a_dice != 0.f) loss += (double)c.a_dice * (1.0 - dice / n);
        float* tvs = den + n;
        tvs[0] = tvs[1] = tvs[2] = 0.f;
        if (c.has_tv) {
            const double tden = tp + c.al * (ps - tp) + c.be * (ts - tp) + c.sm;
            const double tv = (tp + c.sm) / tden;
            tvs[0] = (float)tp; tvs[1] = (float)tden; tvs[2] = (float)tv;
            loss += c.logcosh ? log(cosh(1.0 - tv)) : (1.0 - tv);
        }
        if (pt) loss += (double)c.w_time * red[0] / (double)c.pairs;
        saved[0] = (float)loss;
    }
}
// coef[n][3]: d loss / d logit_i = c0 (p_i - t_i) + (c1 + c2 t_i) p_i (1 - p_i) per sample; ctime[0]: the time term's factor
__global__ void k_seg_loss_coef(const float* __restrict__ g, const float* __restrict__ saved, int n, long long per, SegLossCfg c,
                                float* __restrict__ coef, float* __restrict__ ctime) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float gv = g[0];
    if (i == 0 && ctime) ctime[0] = (n > 1) ? gv * (c.w_time / ((float)(n - 1) * (float)per)) : 0.f;
    if (i >= n) return;
    const float* sums = saved + 1 + 4 * i;
    const float den = saved[1 + 4 * n + i];
    const float* tvs = saved + 1 + 5 * n;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (c.a_bce != 0.f) c0 = gv * (c.a_bce / ((float)n * (float)per));
    if (c.a_dice != 0.f) {
        c1 += gv * (c.a_dice / n) * 2.0f * (sums[3] + c.smooth) / (den * den);
        c2 += -gv * (c.a_dice / n) * 2.0f / den;
    }
    if (c.has_tv) {
        const float tp = tvs[0], tden = tvs[1], tv = tvs[2];
        const float outer = -gv * (c.logcosh ? tanhf(1.f - tv) : 1.f);                 // d loss / d Tversky
        const float d_tp = (tden - (tp + c.sm) * (1.f - c.al - c.be)) / (tden * tden);
        const float d_ps = -(tp + c.sm) * c.al / (tden * tden);
        c1 += outer * d_ps;
        c2 += outer * d_tp;
    }
    coef[3 * i] = c0; coef[3 * i + 1] = c1; coef[3 * i + 2] = c2;
}
}  // namespace

extern "C" int biu_seg_loss_finish(const float* partial, int n, int nb, long long per_sample, const float* time_partial, int nbt,
                                   float a_bce, float a_dice, float smooth, int has_tversky, float tv_alpha, float tv_beta, float tv_smooth,
                                   int logcosh, float w_time, float* saved, biu_stream stream) {
    BIU_REQUIRE(partial && saved && n > 0 && nb > 0 && per_sample > 0 && (!time_partial || (nbt > 0 && n > 1)), BIU_ERR_SHAPE, "seg_loss_finish: bad arguments");
    const SegLossCfg c{a_bce, a_dice, smooth, has_tversky, tv_alpha, tv_beta, tv_smooth, logcosh, w_time, (long long)(n - 1) * per_sample, nbt};
    hipLaunchKernelGGL(k_seg_loss_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, nb, per_sample, time_partial, c, saved);
    BIU_CHECK_LAUNCH("seg_loss_finish");
    return BIU_OK;
}
extern "C" int biu_seg_loss_coef(const float* g, const float* saved, int n, long long per_sample, float a_bce, float a_dice, float smooth,
                                 int has_tversky, float tv_alpha, float tv_beta, float tv_smooth, int logcosh, float w_time, float* coef,
                                 float* time_coef, biu_stream stream) {
    BIU_REQUIRE(g && saved && coef && n > 0 && per_sample > 0, BIU_ERR_SHAPE, "seg_loss_coef: bad arguments");
    const SegLossCfg c{a_bce, a_dice, smooth, has_tversky, tv_alpha, tv_beta, tv_smooth, logcosh, w_time, (long long)(n - 1) * per_sample, 0};
    hipLaunchKernelGGL(k_seg_loss_coef, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, g, saved, n, per_sample, c, coef, time_coef);
    BIU_CHECK_LAUNCH("seg_loss_coef");
    return BIU_OK;
}

extern "C" int biu_bce_dice_blocks(long long per_sample) {
    long long b = (per_sample + 256 * 8 - 1) / (256 * 8);
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
extern "C" int biu_bce_dice_fwd(const float* logits, const float* target, int n, long long per_sample, float* partial,
                                biu_stream stream) {
    BIU_REQUIRE(logits && target && partial && n > 0 && per_sample > 0, BIU_ERR_SHAPE, "bce_dice_fwd: bad arguments");
    hipLaunchKernelGGL(k_bce_dice_fwd, dim3(biu_bce_dice_blocks(per_sample), n), dim3(256), 0, (hipStream_t)stream, logits, target,
                       (i64)per_sample, partial);
    BIU_CHECK_LAUNCH("bce_dice_fwd");
    return BIU_OK;
}
extern "C" int biu_bce_dice_bwd(const float* logits, const float* target, int n, long long per_sample, const float* coef,
                                float* dlogits, int accumulate, biu_stream stream) {
    BIU_REQUIRE(logits && target && coef && dlogits && n > 0 && per_sample > 0, BIU_ERR_SHAPE, "bce_dice_bwd: bad arguments");
    hipLaunchKernelGGL(k_bce_dice_bwd, dim3(biu_bce_dice_blocks(per_sample), n), dim3(256), 0, (hipStream_t)stream, logits, target,
                       (i64)per_sample, coef, dlogits, accumulate);
    BIU_CHECK_LAUNCH("bce_dice_bwd");
    return BIU_OK;
}
