// Direct (shape-generic) kernels of libbiu_hip.so.
//
// Every op of the hot path has a kernel here that is correct for ANY channel count, dilation and extent.
// The MFMA implicit-GEMM kernels in biu_conv_mfma.hip take over the layers whose shapes they cover; the
// bandwidth-bound ops (BatchNorm, pooling, head, element-wise) live only here, vectorised to 16 B/lane
// where the slice allows it.
#include "biu_common.h"
#include "biu_internal.h"

thread_local char biu_errbuf[512] = {0};

int biu_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(biu_errbuf, sizeof(biu_errbuf), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* biu_last_error(void) { return biu_errbuf; }
extern "C" int biu_version(void) { return 100; }
extern "C" int biu_set_fp32_products(int mode) { return biu_mfma_set_fp32_products(mode); }

#define TPB 256

// decompose a linear voxel index
struct Vox {
    int n, d, h, w;
};
__device__ __forceinline__ Vox unvox(i64 v, int D, int H, int W) {
    Vox r;
    r.w = (int)(v % W); v /= W;
    r.h = (int)(v % H); v /= H;
    r.d = (int)(v % D); v /= D;
    r.n = (int)v;
    return r;
}
__device__ __forceinline__ i64 mkvox(int n, int d, int h, int w, int D, int H, int W) {
    return (((i64)n * D + d) * H + h) * W + w;
}

// =====================================================================================================
// 3x3(x3) convolution, direct form
// =====================================================================================================
template <typename T>
__global__ void k_conv_fwd_direct(DAct x, DXf xf, const float* __restrict__ w, const float* __restrict__ bias,
                                  int kd, int kh, int kw, int dil, DAct y) {
    const i64 total = (i64)y.n * y.d * y.h * y.w * y.c;
    const int taps = kd * kh * kw;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int co = (int)(i % y.c);
        i64 v = i / y.c;
        Vox p = unvox(v, y.d, y.h, y.w);
        float acc = bias ? bias[co] : 0.f;
        for (int a = 0; a < kd; ++a) {
            int id = p.d + (a - kd / 2) * dil;
            if (id < 0 || id >= x.d) continue;
            for (int b = 0; b < kh; ++b) {
                int ih = p.h + (b - kh / 2) * dil;
                if (ih < 0 || ih >= x.h) continue;
                for (int c = 0; c < kw; ++c) {
                    int iw = p.w + (c - kw / 2) * dil;
                    if (iw < 0 || iw >= x.w) continue;
                    i64 iv = mkvox(p.n, id, ih, iw, x.d, x.h, x.w);
                    int tap = (a * kh + b) * kw + c;
                    const float* wp = w + (i64)co * x.c * taps + tap;
                    for (int ci = 0; ci < x.c; ++ci)
                        acc = fmaf(xf_apply(xf, ci, ld_act<T>(x, iv, ci)), wp[(i64)ci * taps], acc);
                }
            }
        }
        st_act<T>(y, v, co, acc);
    }
}

template <typename T>
__global__ void k_conv_dgrad_direct(DAct dy, const float* __restrict__ w, int kd, int kh, int kw, int dil,
                                    DAct dx, int accumulate) {
    const i64 total = (i64)dx.n * dx.d * dx.h * dx.w * dx.c;
    const int taps = kd * kh * kw;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int ci = (int)(i % dx.c);
        i64 v = i / dx.c;
        Vox p = unvox(v, dx.d, dx.h, dx.w);
        float acc = 0.f;
        for (int a = 0; a < kd; ++a) {
            int od = p.d - (a - kd / 2) * dil;
            if (od < 0 || od >= dy.d) continue;
            for (int b = 0; b < kh; ++b) {
                int oh = p.h - (b - kh / 2) * dil;
                if (oh < 0 || oh >= dy.h) continue;
                for (int c = 0; c < kw; ++c) {
                    int ow = p.w - (c - kw / 2) * dil;
                    if (ow < 0 || ow >= dy.w) continue;
                    i64 ov = mkvox(p.n, od, oh, ow, dy.d, dy.h, dy.w);
                    int tap = (a * kh + b) * kw + c;
                    for (int co = 0; co < dy.c; ++co)
                        acc = fmaf(ld_act<T>(dy, ov, co), w[((i64)co * dx.c + ci) * taps + tap], acc);
                }
            }
        }
        if (accumulate) acc += ld_act<T>(dx, v, ci);
        st_act<T>(dx, v, ci, acc);
    }
}

// one block per (co, ci, tap): dw = sum_v T(x)[v+off, ci] * dy[v, co]
template <typename T>
__global__ void k_conv_wgrad_direct(DAct x, DXf xf, DAct dy, int kd, int kh, int kw, int dil, float* __restrict__ dw) {
    __shared__ float red[16];
    const int taps = kd * kh * kw;
    int o = blockIdx.x;
    int tap = o % taps;
    int ci = (o / taps) % x.c;
    int co = o / (taps * x.c);
    int c = tap % kw, b = (tap / kw) % kh, a = tap / (kw * kh);
    int od = (a - kd / 2) * dil, oh = (b - kh / 2) * dil, ow = (c - kw / 2) * dil;
    const i64 total = (i64)dy.n * dy.d * dy.h * dy.w;
    float acc = 0.f;
    for (i64 v = threadIdx.x; v < total; v += blockDim.x) {
        Vox p = unvox(v, dy.d, dy.h, dy.w);
        int id = p.d + od, ih = p.h + oh, iw = p.w + ow;
        if (id < 0 || id >= x.d || ih < 0 || ih >= x.h || iw < 0 || iw >= x.w) continue;
        i64 iv = mkvox(p.n, id, ih, iw, x.d, x.h, x.w);
        acc = fmaf(xf_apply(xf, ci, ld_act<T>(x, iv, ci)), ld_act<T>(dy, v, co), acc);
    }
    float r = block_sum(acc, red);
    if (threadIdx.x == 0) dw[o] = r;
}

// one block per channel: out[c] = sum_v a[v, c]
template <typename T> __global__ void k_chan_sum(DAct a, float* __restrict__ out) {
    __shared__ float red[16];
    int c = blockIdx.x;
    const i64 total = (i64)a.n * a.d * a.h * a.w;
    float acc = 0.f;
    for (i64 v = threadIdx.x; v < total; v += blockDim.x) acc += ld_act<T>(a, v, c);
    float r = block_sum(acc, red);
    if (threadIdx.x == 0) out[c] = r;
}

// =====================================================================================================
// per-channel two-term reductions with deterministic partials:  partial[blk][c][2]
// =====================================================================================================
// thread layout: cw channel lanes x (TPB / cw) voxel rows; each block owns a contiguous voxel range
template <typename F>
__global__ void k_chan_reduce2(F f, i64 total_vox, int C, int cw, i64 vox_per_block, float* __restrict__ partial) {
    __shared__ float s0[TPB], s1[TPB];
    const int rows = TPB / cw;
    const int cl = threadIdx.x % cw, row = threadIdx.x / cw;
    const i64 v0 = (i64)blockIdx.x * vox_per_block;
    i64 v1 = v0 + vox_per_block;
    if (v1 > total_vox) v1 = total_vox;
    for (int cb = 0; cb < C; cb += cw) {
        int c = cb + cl;
        float a0 = 0.f, a1 = 0.f;
        if (c < C)
            for (i64 v = v0 + row; v < v1; v += rows) f(v, c, a0, a1);
        s0[threadIdx.x] = a0;
        s1[threadIdx.x] = a1;
        __syncthreads();
        for (int r = rows >> 1; r > 0; r >>= 1) {
            if (row < r) {
                s0[threadIdx.x] += s0[threadIdx.x + r * cw];
                s1[threadIdx.x] += s1[threadIdx.x + r * cw];
            }
            __syncthreads();
        }
        if (row == 0 && c < C) {
            partial[((i64)blockIdx.x * C + c) * 2 + 0] = s0[threadIdx.x];
            partial[((i64)blockIdx.x * C + c) * 2 + 1] = s1[threadIdx.x];
        }
        __syncthreads();
    }
}

struct ReducePlan {
    int cw, nblk;
    i64 vpb;
};
static ReducePlan plan_reduce(i64 total_vox, int C, int max_blocks) {
    ReducePlan p;
    p.cw = 1;
    while (p.cw < C && p.cw < TPB) p.cw <<= 1;
    int rows = TPB / p.cw;
    i64 want = (total_vox + (i64)rows * 8 - 1) / ((i64)rows * 8);      // >= 8 voxels per thread
    if (want < 1) want = 1;
    if (want > max_blocks) want = max_blocks;
    p.vpb = (total_vox + want - 1) / want;
    p.nblk = (int)((total_vox + p.vpb - 1) / p.vpb);
    return p;
}

// Sum of (a0, a1) over the block's threads, left in d0[0] / d1[0] for thread 0: wave shuffles + one LDS exchange (these one-block-per-channel
// kernels sit between every two convolutions of a step: the former 8-barrier LDS tree was most of their run time).  Fixed order: reproducible.
__device__ __forceinline__ void bn_block_sum2(double a0, double a1, double* d0, double* d1) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
    if ((threadIdx.x & 63) == 0) { d0[threadIdx.x >> 6] = a0; d1[threadIdx.x >> 6] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = d0[0], t1 = d1[0];
        for (int w = 1; w < TPB / 64; ++w) { t0 += d0[w]; t1 += d1[w]; }
        d0[0] = t0; d1[0] = t1;
    }
}

// out0[c] = sum_b partial[b][c][0], out1[c] = sum_b partial[b][c][1]   (fp64 merge), one block per channel
__global__ void k_partial_sum(const float* __restrict__ partial, int nblk, int C, float* out0, float* out1) {
    __shared__ double d0[TPB / 64], d1[TPB / 64];
    int c = blockIdx.x;
    double a0 = 0, a1 = 0;
    for (int b = threadIdx.x; b < nblk; b += TPB) {
        a0 += partial[((i64)b * C + c) * 2 + 0];
        a1 += partial[((i64)b * C + c) * 2 + 1];
    }
    bn_block_sum2(a0, a1, d0, d1);                       // (totals in d0[0], d1[0], read by thread 0 below)
    if (threadIdx.x == 0) {
        if (out0) out0[c] = (float)d0[0];
        if (out1) out1[c] = (float)d1[0];
    }
}

// =====================================================================================================
// BatchNorm
// =====================================================================================================
template <typename T> struct StatsF {
    DAct y;
    __device__ void operator()(i64 v, int c, float& s, float& ss) const {
        float t = ld_act<T>(y, v, c);
        s += t;
        ss = fmaf(t, t, ss);
    }
};

__global__ void k_bn_finalize(const float* __restrict__ partial, int nblk, int C, double count,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              float momentum, float eps, float* scale, float* shift, float* save_mean,
                              float* save_invstd) {
    __shared__ double d0[TPB / 64], d1[TPB / 64];
    int c = blockIdx.x;
    double a0 = 0, a1 = 0;
    for (int b = threadIdx.x; b < nblk; b += TPB) {
        a0 += partial[((i64)b * C + c) * 2 + 0];
        a1 += partial[((i64)b * C + c) * 2 + 1];
    }
    bn_block_sum2(a0, a1, d0, d1);                       // (totals in d0[0], d1[0], read by thread 0 below)
    if (threadIdx.x == 0) {
        double mean = d0[0] / count;
        double var = d1[0] / count - mean * mean;
        if (var < 0) var = 0;
        double invstd = 1.0 / sqrt(var + (double)eps);
        float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        float sc = (float)(g * invstd);
        scale[c] = sc;
        shift[c] = (float)(b - mean * g * invstd);
        if (save_mean) save_mean[c] = (float)mean;
        if (save_invstd) save_invstd[c] = (float)invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        if (running_var) {
            double unbiased = count > 1 ? var * count / (count - 1) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

__global__ void k_bn_eval_affine(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                 float eps, float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float invstd = 1.f / sqrtf(rv[c] + eps);
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = b - rm[c] * g * invstd;
}

// element-wise T(x) with G-wide vectors
template <typename T, int G>
__global__ void k_xform_apply(DAct x, DXf xf, DAct out) {
    const int cg = x.c / G;
    const i64 total = (i64)x.n * x.d * x.h * x.w * cg;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * G;
        i64 v = i / cg;
        Pack<T, G> in = *(const Pack<T, G>*)((const T*)x.p + v * x.pitch + c0);
        Pack<T, G> o;
#pragma unroll
        for (int j = 0; j < G; ++j) o.v[j] = from_f<T>(xf_apply(xf, c0 + j, to_f(in.v[j])));
        *(Pack<T, G>*)((T*)out.p + v * out.pitch + c0) = o;
    }
}

// backward of a = T(y):  dz = da * T'(scale*y+shift)
template <typename T> struct BnBwdF {
    DAct da, y;
    DXf xf;
    const float* mean;
    const float* invstd;
    __device__ void operator()(i64 v, int c, float& s1, float& s2) const {
        float yv = ld_act<T>(y, v, c);
        float t = xf_pre(xf, c, yv);
        float dz = ld_act<T>(da, v, c) * xf_dact(xf, c, t);
        s1 += dz;
        s2 = fmaf(dz, (yv - mean[c]) * invstd[c], s2);
    }
};

__global__ void k_bn_bwd_finalize(const float* __restrict__ partial, int nblk, int C, double count,
                                  const float* scale, const float* mean, const float* invstd, float* dgamma,
                                  float* dbeta, float* A, float* B, float* Cc) {
    __shared__ double d0[TPB / 64], d1[TPB / 64];
    int c = blockIdx.x;
    double a0 = 0, a1 = 0;
    for (int b = threadIdx.x; b < nblk; b += TPB) {
        a0 += partial[((i64)b * C + c) * 2 + 0];
        a1 += partial[((i64)b * C + c) * 2 + 1];
    }
    bn_block_sum2(a0, a1, d0, d1);                       // (totals in d0[0], d1[0], read by thread 0 below)
    if (threadIdx.x == 0) {
        double S1 = d0[0], S2 = d1[0];
        if (dbeta) dbeta[c] = (float)S1;
        if (dgamma) dgamma[c] = (float)S2;
        // dy = scale * (dz - S1/M - yhat * S2/M),  yhat = (y - mean) * invstd
        double sc = scale[c], r = invstd[c], mu = mean[c];
        double kb = -sc * r * S2 / count;
        A[c] = (float)sc;
        B[c] = (float)kb;
        Cc[c] = (float)(-sc * S1 / count - kb * mu);
    }
}

// eval-mode BatchNorm: the statistics are constants, dy = scale * dz
__global__ void k_bn_bwd_finalize_eval(const float* __restrict__ partial, int nblk, int C, const float* scale, float* dgamma, float* dbeta,
                                       float* dbias, float* A, float* B, float* Cc) {
    __shared__ double d0[TPB / 64], d1[TPB / 64];
    int c = blockIdx.x;
    double a0 = 0, a1 = 0;
    for (int b = threadIdx.x; b < nblk; b += TPB) {
        a0 += partial[((i64)b * C + c) * 2 + 0];
        a1 += partial[((i64)b * C + c) * 2 + 1];
    }
    bn_block_sum2(a0, a1, d0, d1);                       // (totals in d0[0], d1[0], read by thread 0 below)
    if (threadIdx.x == 0) {
        if (dbeta) dbeta[c] = (float)d0[0];
        if (dgamma) dgamma[c] = (float)d1[0];
        if (dbias) dbias[c] = (float)((double)scale[c] * d0[0]);
        A[c] = scale[c];
        B[c] = 0.f;
        Cc[c] = 0.f;
    }
}

template <typename T, int G>
__global__ void k_bn_bwd_apply(DAct da, DAct y, DXf xf, const float* __restrict__ A, const float* __restrict__ B,
                               const float* __restrict__ Cc, DAct dy) {
    const int cg = y.c / G;
    const i64 total = (i64)y.n * y.d * y.h * y.w * cg;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * G;
        i64 v = i / cg;
        Pack<T, G> g = *(const Pack<T, G>*)((const T*)da.p + v * da.pitch + c0);
        Pack<T, G> yy = *(const Pack<T, G>*)((const T*)y.p + v * y.pitch + c0);
        Pack<T, G> o;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            int c = c0 + j;
            float yv = to_f(yy.v[j]);
            float dz = to_f(g.v[j]) * xf_dact(xf, c, xf_pre(xf, c, yv));
            o.v[j] = from_f<T>(fmaf(A[c], dz, fmaf(B[c], yv, Cc[c])));
        }
        *(Pack<T, G>*)((T*)dy.p + v * dy.pitch + c0) = o;
    }
}

// =====================================================================================================
// pooling / nearest resampling.  Window along d is 2 when in.d == 2*out.d, 1 when equal (2-D tensors).
// =====================================================================================================
template <typename T, int G>
__global__ void k_maxpool_fwd(DAct x, DXf xf, DAct out, int pd) {
    const int cg = x.c / G;
    const i64 total = (i64)out.n * out.d * out.h * out.w * cg;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * G;
        i64 ov = i / cg;
        Vox p = unvox(ov, out.d, out.h, out.w);
        float best[G];
#pragma unroll
        for (int j = 0; j < G; ++j) best[j] = -INFINITY;
        for (int a = 0; a < pd; ++a)
            for (int b = 0; b < 2; ++b)
                for (int c = 0; c < 2; ++c) {
                    i64 iv = mkvox(p.n, p.d * pd + a, p.h * 2 + b, p.w * 2 + c, x.d, x.h, x.w);
                    Pack<T, G> in = *(const Pack<T, G>*)((const T*)x.p + iv * x.pitch + c0);
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        float t = xf_apply(xf, c0 + j, to_f(in.v[j]));
                        if (t > best[j] || t != t) best[j] = t;
                    }
                }
        Pack<T, G> o;
#pragma unroll
        for (int j = 0; j < G; ++j) o.v[j] = from_f<T>(best[j]);
        *(Pack<T, G>*)((T*)out.p + ov * out.pitch + c0) = o;
    }
}

template <typename T, int G>
__global__ void k_maxpool_bwd(DAct x, DXf xf, DAct dout, DAct dx, int pd, int accumulate) {
    const int cg = x.c / G;
    const i64 total = (i64)dout.n * dout.d * dout.h * dout.w * cg;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * G;
        i64 ov = i / cg;
        Vox p = unvox(ov, dout.d, dout.h, dout.w);
        float best[G];
        int arg[G];
#pragma unroll
        for (int j = 0; j < G; ++j) { best[j] = -INFINITY; arg[j] = 0; }
        for (int a = 0; a < pd; ++a)
            for (int b = 0; b < 2; ++b)
                for (int c = 0; c < 2; ++c) {
                    i64 iv = mkvox(p.n, p.d * pd + a, p.h * 2 + b, p.w * 2 + c, x.d, x.h, x.w);
                    Pack<T, G> in = *(const Pack<T, G>*)((const T*)x.p + iv * x.pitch + c0);
                    int code = (a * 2 + b) * 2 + c;
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        float t = xf_apply(xf, c0 + j, to_f(in.v[j]));
                        if (t > best[j] || t != t) { best[j] = t; arg[j] = code; }
                    }
                }
        Pack<T, G> g = *(const Pack<T, G>*)((const T*)dout.p + ov * dout.pitch + c0);
        for (int a = 0; a < pd; ++a)
            for (int b = 0; b < 2; ++b)
                for (int c = 0; c < 2; ++c) {
                    i64 iv = mkvox(p.n, p.d * pd + a, p.h * 2 + b, p.w * 2 + c, x.d, x.h, x.w);
                    int code = (a * 2 + b) * 2 + c;
                    T* dst = (T*)dx.p + iv * dx.pitch + c0;
                    Pack<T, G> o;
                    if (accumulate) o = *(Pack<T, G>*)dst;
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        float r = (arg[j] == code) ? to_f(g.v[j]) : 0.f;
                        o.v[j] = from_f<T>(accumulate ? to_f(o.v[j]) + r : r);
                    }
                    *(Pack<T, G>*)dst = o;
                }
    }
}

// nearest x0.5: out[o] = T(x[2o]);  bwd: dx[2o] (+)= dout[o], other positions (+)= 0
template <typename T, int G>
__global__ void k_nearest_down_fwd(DAct x, DXf xf, DAct out, int pd) {
    const int cg = x.c / G;
    const i64 total = (i64)out.n * out.d * out.h * out.w * cg;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * G;
        i64 ov = i / cg;
        Vox p = unvox(ov, out.d, out.h, out.w);
        i64 iv = mkvox(p.n, p.d * pd, p.h * 2, p.w * 2, x.d, x.h, x.w);
        Pack<T, G> in = *(const Pack<T, G>*)((const T*)x.p + iv * x.pitch + c0);
        Pack<T, G> o;
#pragma unroll
        for (int j = 0; j < G; ++j) o.v[j] = from_f<T>(xf_apply(xf, c0 + j, to_f(in.v[j])));
        *(Pack<T, G>*)((T*)out.p + ov * out.pitch + c0) = o;
    }
}
template <typename T, int G>
__global__ void k_nearest_down_bwd(DAct dout, DAct dx, int pd, int accumulate) {
    const int cg = dx.c / G;
    const i64 total = (i64)dx.n * dx.d * dx.h * dx.w * cg;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * G;
        i64 v = i / cg;
        Vox p = unvox(v, dx.d, dx.h, dx.w);
        bool hit = (p.d % pd == 0) && (p.h % 2 == 0) && (p.w % 2 == 0);
        T* dst = (T*)dx.p + v * dx.pitch + c0;
        Pack<T, G> o;
        if (accumulate) {
            if (!hit) continue;
            o = *(Pack<T, G>*)dst;
        }
        Pack<T, G> g;
        if (hit) {
            i64 ov = mkvox(p.n, p.d / pd, p.h / 2, p.w / 2, dout.d, dout.h, dout.w);
            g = *(const Pack<T, G>*)((const T*)dout.p + ov * dout.pitch + c0);
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            float r = hit ? to_f(g.v[j]) : 0.f;
            o.v[j] = from_f<T>(accumulate ? to_f(o.v[j]) + r : r);
        }
        *(Pack<T, G>*)dst = o;
    }
}
// nearest x2: out[o] = T(x[o/2]);  bwd: dx[i] (+)= sum of the 2^k children
template <typename T, int G>
__global__ void k_nearest_up_fwd(DAct x, DXf xf, DAct out, int pd) {
    const int cg = x.c / G;
    const i64 total = (i64)out.n * out.d * out.h * out.w * cg;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * G;
        i64 ov = i / cg;
        Vox p = unvox(ov, out.d, out.h, out.w);
        i64 iv = mkvox(p.n, p.d / pd, p.h / 2, p.w / 2, x.d, x.h, x.w);
        Pack<T, G> in = *(const Pack<T, G>*)((const T*)x.p + iv * x.pitch + c0);
        Pack<T, G> o;
#pragma unroll
        for (int j = 0; j < G; ++j) o.v[j] = from_f<T>(xf_apply(xf, c0 + j, to_f(in.v[j])));
        *(Pack<T, G>*)((T*)out.p + ov * out.pitch + c0) = o;
    }
}
template <typename T, int G>
__global__ void k_nearest_up_bwd(DAct dout, DAct dx, int pd, int accumulate) {
    const int cg = dx.c / G;
    const i64 total = (i64)dx.n * dx.d * dx.h * dx.w * cg;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * G;
        i64 v = i / cg;
        Vox p = unvox(v, dx.d, dx.h, dx.w);
        float acc[G];
#pragma unroll
        for (int j = 0; j < G; ++j) acc[j] = 0.f;
        for (int a = 0; a < pd; ++a)
            for (int b = 0; b < 2; ++b)
                for (int c = 0; c < 2; ++c) {
                    i64 ov = mkvox(p.n, p.d * pd + a, p.h * 2 + b, p.w * 2 + c, dout.d, dout.h, dout.w);
                    Pack<T, G> g = *(const Pack<T, G>*)((const T*)dout.p + ov * dout.pitch + c0);
#pragma unroll
                    for (int j = 0; j < G; ++j) acc[j] += to_f(g.v[j]);
                }
        T* dst = (T*)dx.p + v * dx.pitch + c0;
        Pack<T, G> o;
        if (accumulate) o = *(Pack<T, G>*)dst;
#pragma unroll
        for (int j = 0; j < G; ++j) o.v[j] = from_f<T>(accumulate ? to_f(o.v[j]) + acc[j] : acc[j]);
        *(Pack<T, G>*)dst = o;
    }
}

// =====================================================================================================
// ConvTranspose k=2 s=2, direct form.  w: (Cin, Cout, kd, 2, 2)
// =====================================================================================================
template <typename T>
__global__ void k_convt_fwd_direct(DAct x, DXf xf, const float* __restrict__ w, const float* __restrict__ bias,
                                   int kd, DAct y) {
    const i64 total = (i64)y.n * y.d * y.h * y.w * y.c;
    const int taps = kd * 4;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int co = (int)(i % y.c);
        i64 ov = i / y.c;
        Vox p = unvox(ov, y.d, y.h, y.w);
        int a = kd == 2 ? (p.d & 1) : 0, b = p.h & 1, c = p.w & 1;
        i64 iv = mkvox(p.n, kd == 2 ? p.d >> 1 : p.d, p.h >> 1, p.w >> 1, x.d, x.h, x.w);
        int tap = (a * 2 + b) * 2 + c;
        float acc = bias ? bias[co] : 0.f;
        for (int ci = 0; ci < x.c; ++ci)
            acc = fmaf(xf_apply(xf, ci, ld_act<T>(x, iv, ci)), w[((i64)ci * y.c + co) * taps + tap], acc);
        st_act<T>(y, ov, co, acc);
    }
}
template <typename T>
__global__ void k_convt_dgrad_direct(DAct dy, const float* __restrict__ w, int kd, DAct dx, int accumulate) {
    const i64 total = (i64)dx.n * dx.d * dx.h * dx.w * dx.c;
    const int taps = kd * 4;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int ci = (int)(i % dx.c);
        i64 v = i / dx.c;
        Vox p = unvox(v, dx.d, dx.h, dx.w);
        float acc = 0.f;
        for (int a = 0; a < kd; ++a)
            for (int b = 0; b < 2; ++b)
                for (int c = 0; c < 2; ++c) {
                    i64 ov = mkvox(p.n, p.d * kd + a, p.h * 2 + b, p.w * 2 + c, dy.d, dy.h, dy.w);
                    int tap = (a * 2 + b) * 2 + c;
                    for (int co = 0; co < dy.c; ++co)
                        acc = fmaf(ld_act<T>(dy, ov, co), w[((i64)ci * dy.c + co) * taps + tap], acc);
                }
        if (accumulate) acc += ld_act<T>(dx, v, ci);
        st_act<T>(dx, v, ci, acc);
    }
}
// one block per (ci, co, tap)
template <typename T>
__global__ void k_convt_wgrad_direct(DAct x, DXf xf, DAct dy, int kd, float* __restrict__ dw) {
    __shared__ float red[16];
    const int taps = kd * 4;
    int o = blockIdx.x;
    int tap = o % taps;
    int co = (o / taps) % dy.c;
    int ci = o / (taps * dy.c);
    int c = tap & 1, b = (tap >> 1) & 1, a = tap >> 2;
    const i64 total = (i64)x.n * x.d * x.h * x.w;
    float acc = 0.f;
    for (i64 v = threadIdx.x; v < total; v += blockDim.x) {
        Vox p = unvox(v, x.d, x.h, x.w);
        i64 ov = mkvox(p.n, p.d * kd + a, p.h * 2 + b, p.w * 2 + c, dy.d, dy.h, dy.w);
        acc = fmaf(xf_apply(xf, ci, ld_act<T>(x, v, ci)), ld_act<T>(dy, ov, co), acc);
    }
    float r = block_sum(acc, red);
    if (threadIdx.x == 0) dw[o] = r;
}

// =====================================================================================================
// 1x1 head: fp32 NCDHW outputs
// =====================================================================================================
#define HEAD_MAX_COUT 8
__device__ __forceinline__ float head_act(int act, float v) {
    switch (act) {
        case 1: return 1.f / (1.f + __expf(-v));
        case 2: return tanhf(v);
        case 3: return v > 0.f ? v : 0.f;
        default: return v;
    }
}
template <typename T>
__global__ void k_head_fwd(DAct x, DXf xf, const float* __restrict__ w, const float* __restrict__ bias, int cout,
                           int act, float* __restrict__ logits, float* __restrict__ activated) {
    const i64 S = (i64)x.d * x.h * x.w;
    const i64 total = (i64)x.n * S;
    for (i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (i64)gridDim.x * blockDim.x) {
        float acc[HEAD_MAX_COUT];
#pragma unroll
        for (int o = 0; o < HEAD_MAX_COUT; ++o) acc[o] = (o < cout && bias) ? bias[o] : 0.f;
        for (int c = 0; c < x.c; ++c) {
            float t = xf_apply(xf, c, ld_act<T>(x, v, c));
#pragma unroll
            for (int o = 0; o < HEAD_MAX_COUT; ++o)
                if (o < cout) acc[o] = fmaf(t, w[o * x.c + c], acc[o]);
        }
        i64 n = v / S, s = v % S;
#pragma unroll
        for (int o = 0; o < HEAD_MAX_COUT; ++o)
            if (o < cout) {
                i64 idx = (n * cout + o) * S + s;
                if (logits) logits[idx] = acc[o];
                if (activated) activated[idx] = head_act(act, acc[o]);
            }
    }
}
// vectorised form: one voxel per thread, its C channels as 16-byte packs; transform vectors and weights sit in LDS
// (broadcast reads).  C is a compile-time 8/16/32/64.
template <typename T, int C, int O>
__global__ __launch_bounds__(TPB) void k_head_fwd_vec(DAct x, DXf xf, const float* __restrict__ w, const float* __restrict__ bias,
                                                      int cout, int act, float* __restrict__ logits, float* __restrict__ activated) {
    constexpr int G = 16 / sizeof(T);
    __shared__ float sm[3 * C + O * C + O];
    float* sc = sm; float* sh = sm + C; float* sl = sm + 2 * C; float* ws = sm + 3 * C; float* bs = ws + O * C;
    for (int i = threadIdx.x; i < C; i += TPB) {
        sc[i] = xf.scale ? xf.scale[i] : 1.f;
        sh[i] = xf.shift ? xf.shift[i] : 0.f;
        sl[i] = xf.slope ? xf.slope[i] : 1.f;
    }
    for (int i = threadIdx.x; i < O * C; i += TPB) ws[i] = (i / C) < cout ? w[i] : 0.f;
    for (int i = threadIdx.x; i < O; i += TPB) bs[i] = (bias && i < cout) ? bias[i] : 0.f;
    __syncthreads();
    const i64 S = (i64)x.d * x.h * x.w;
    const i64 total = (i64)x.n * S;
    for (i64 v = (i64)blockIdx.x * TPB + threadIdx.x; v < total; v += (i64)gridDim.x * TPB) {
        const T* src = (const T*)x.p + v * x.pitch;
        Pack<T, G> in[C / G];
#pragma unroll
        for (int q = 0; q < C / G; ++q) in[q] = *(const Pack<T, G>*)(src + q * G);
        float acc[O];
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] = bs[o];
#pragma unroll
        for (int q = 0; q < C / G; ++q)
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int c = q * G + j;
                float t = fmaf(sc[c], to_f(in[q].v[j]), sh[c]);
                t = t > 0.f ? t : sl[c] * t;
#pragma unroll
                for (int o = 0; o < O; ++o) acc[o] = fmaf(t, ws[o * C + c], acc[o]);
            }
        const i64 n = v / S, sidx = v % S;
#pragma unroll
        for (int o = 0; o < O; ++o)
            if (o < cout) {
                const i64 idx = (n * cout + o) * S + sidx;
                if (logits) logits[idx] = acc[o];
                if (activated) activated[idx] = head_act(act, acc[o]);
            }
    }
}
template <typename T, int C>
static bool head_fwd_vec_launch(const biu_act* x, const biu_xform* xf, const float* w, const float* bias, int cout, int act, float* logits,
                                float* activated, hipStream_t st) {
    const int grid = grid_for(nvox(x), TPB, 8192);
    if (cout == 1) hipLaunchKernelGGL((k_head_fwd_vec<T, C, 1>), dim3(grid), dim3(TPB), 0, st, dact(x), dxf(xf), w, bias, cout, act, logits, activated);
    else if (cout == 2) hipLaunchKernelGGL((k_head_fwd_vec<T, C, 2>), dim3(grid), dim3(TPB), 0, st, dact(x), dxf(xf), w, bias, cout, act, logits, activated);
    else if (cout <= 4) hipLaunchKernelGGL((k_head_fwd_vec<T, C, 4>), dim3(grid), dim3(TPB), 0, st, dact(x), dxf(xf), w, bias, cout, act, logits, activated);
    else return false;
    return true;
}

template <typename T>
__global__ void k_head_bwd_data(int cin, const float* __restrict__ w, int cout, const float* __restrict__ dl, DAct dx) {
    const i64 S = (i64)dx.d * dx.h * dx.w;
    const i64 total = (i64)dx.n * S * cin;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c = (int)(i % cin);
        i64 v = i / cin;
        i64 n = v / S, s = v % S;
        float acc = 0.f;
        for (int o = 0; o < cout; ++o) acc = fmaf(dl[(n * cout + o) * S + s], w[o * cin + c], acc);
        st_act<T>(dx, v, c, acc);
    }
}
template <typename T> struct HeadWgradF {
    DAct x;
    DXf xf;
    const float* dl;
    int cout, o0;
    i64 S;
    __device__ void operator()(i64 v, int c, float& a0, float& a1) const {
        float t = xf_apply(xf, c, ld_act<T>(x, v, c));
        i64 n = v / S, s = v % S;
        a0 = fmaf(t, dl[(n * cout + o0) * S + s], a0);
        if (o0 + 1 < cout) a1 = fmaf(t, dl[(n * cout + o0 + 1) * S + s], a1);
    }
};
// dbias[o] = sum over n,s of dl[n,o,s]; one block per o
__global__ void k_head_dbias(const float* __restrict__ dl, int N, int cout, i64 S, float* dbias) {
    __shared__ float red[16];
    int o = blockIdx.x;
    float acc = 0.f;
    for (int n = 0; n < N; ++n)
        for (i64 s = threadIdx.x; s < S; s += blockDim.x) acc += dl[((i64)n * cout + o) * S + s];
    float r = block_sum(acc, red);
    if (threadIdx.x == 0) dbias[o] = r;
}

// =====================================================================================================
// element-wise helpers
// =====================================================================================================
template <typename T, int G>
__global__ void k_max_join_fwd(DAct a, DXf xa, DAct b, DXf xb, DAct out) {
    const int cg = a.c / G;
    const i64 total = (i64)a.n * a.d * a.h * a.w * cg;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * G;
        i64 v = i / cg;
        Pack<T, G> pa = *(const Pack<T, G>*)((const T*)a.p + v * a.pitch + c0);
        Pack<T, G> pb = *(const Pack<T, G>*)((const T*)b.p + v * b.pitch + c0);
        Pack<T, G> o;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            float ta = xf_apply(xa, c0 + j, to_f(pa.v[j])), tb = xf_apply(xb, c0 + j, to_f(pb.v[j]));
            o.v[j] = from_f<T>((ta != ta || tb != tb) ? NAN : fmaxf(ta, tb));
        }
        *(Pack<T, G>*)((T*)out.p + v * out.pitch + c0) = o;
    }
}
// torch.maximum backward: grad to a where a > b, to b where b > a, split in half on ties
template <typename T, int G>
__global__ void k_max_join_bwd(DAct a, DXf xa, DAct b, DXf xb, DAct dout, DAct da, DAct db, int accumulate) {
    const int cg = a.c / G;
    const i64 total = (i64)a.n * a.d * a.h * a.w * cg;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * G;
        i64 v = i / cg;
        Pack<T, G> pa = *(const Pack<T, G>*)((const T*)a.p + v * a.pitch + c0);
        Pack<T, G> pb = *(const Pack<T, G>*)((const T*)b.p + v * b.pitch + c0);
        Pack<T, G> g = *(const Pack<T, G>*)((const T*)dout.p + v * dout.pitch + c0);
        T* pda = (T*)da.p + v * da.pitch + c0;
        T* pdb = (T*)db.p + v * db.pitch + c0;
        Pack<T, G> oa, ob;
        if (accumulate) { oa = *(Pack<T, G>*)pda; ob = *(Pack<T, G>*)pdb; }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            float ta = xf_apply(xa, c0 + j, to_f(pa.v[j])), tb = xf_apply(xb, c0 + j, to_f(pb.v[j]));
            float gg = to_f(g.v[j]);
            float ga = ta > tb ? gg : (ta == tb ? 0.5f * gg : 0.f);
            float gb = tb > ta ? gg : (ta == tb ? 0.5f * gg : 0.f);
            oa.v[j] = from_f<T>(accumulate ? to_f(oa.v[j]) + ga : ga);
            ob.v[j] = from_f<T>(accumulate ? to_f(ob.v[j]) + gb : gb);
        }
        *(Pack<T, G>*)pda = oa;
        *(Pack<T, G>*)pdb = ob;
    }
}
template <typename T, int G>
__global__ void k_act_add(DAct src, DAct dst, int accumulate) {
    const int cg = src.c / G;
    const i64 total = (i64)src.n * src.d * src.h * src.w * cg;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c0 = (int)(i % cg) * G;
        i64 v = i / cg;
        Pack<T, G> s = *(const Pack<T, G>*)((const T*)src.p + v * src.pitch + c0);
        T* pd = (T*)dst.p + v * dst.pitch + c0;
        if (accumulate) {
            Pack<T, G> d = *(Pack<T, G>*)pd;
#pragma unroll
            for (int j = 0; j < G; ++j) s.v[j] = from_f<T>(to_f(s.v[j]) + to_f(d.v[j]));
        }
        *(Pack<T, G>*)pd = s;
    }
}
template <typename T>
__global__ void k_from_nchw(const float* __restrict__ src, DAct dst) {
    const i64 S = (i64)dst.d * dst.h * dst.w;
    const i64 total = (i64)dst.n * S * dst.c;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        int c = (int)(i % dst.c);
        i64 v = i / dst.c;
        i64 n = v / S, s = v % S;
        st_act<T>(dst, v, c, src[(n * dst.c + c) * S + s]);
    }
}
template <typename T>
__global__ void k_to_nchw(DAct src, DXf xf, float* __restrict__ dst) {
    const i64 S = (i64)src.d * src.h * src.w;
    const i64 total = (i64)src.n * S * src.c;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        i64 s = i % S;
        i64 r = i / S;
        int c = (int)(r % src.c);
        i64 n = r / src.c;
        dst[i] = xf_apply(xf, c, ld_act<T>(src, n * S + s, c));
    }
}

// =====================================================================================================
// fused multi-tensor Adam
// =====================================================================================================
// hyper == nullptr: the step's scalars come by value.  Otherwise they are read from device memory -- {lr, beta1, beta2, eps, grad_scale,
// step} as six floats, written by biu_adam_set_hyper -- so that the launch can sit in a captured hipGraph and still follow the
// learning-rate schedule and the bias correction of the step it is replayed for.
__global__ void k_adam(int n, float* const* params, const float* const* grads, float* const* m, float* const* v,
                       const int64_t* numel, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                       float gscale, const float* __restrict__ hyper) {
    int t = blockIdx.y;
    if (t >= n) return;
    if (hyper) {
        lr = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; gscale = hyper[4];
        const float step = hyper[5];
        bc1 = 1.f - powf(b1, step);
        bc2_sqrt = sqrtf(1.f - powf(b2, step));
    }
    float* p = params[t];
    const float* g = grads[t];
    float* mm = m[t];
    float* vv = v[t];
    const i64 cnt = numel[t];
    const float step_lr = lr / bc1;
    auto upd = [&](float pv, float gv, float& m1, float& v1) -> float {
        const float gr = gv * gscale;
        m1 = b1 * m1 + (1.f - b1) * gr;                // torch: exp_avg.lerp_(grad, 1 - beta1)
        v1 = b2 * v1 + (1.f - b2) * gr * gr;
        const float denom = sqrtf(v1) / bc2_sqrt + eps;
        return pv - step_lr * (m1 / denom);
    };
    // 16-byte accesses when the four arrays allow it (torch allocations are 256-byte aligned; views inside a flat bucket may not be)
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)mm | (uintptr_t)vv) & 15) == 0;
    const i64 nv = vec ? cnt / 4 : 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (i64)gridDim.x * blockDim.x) {
        float4 pv = ((float4*)p)[i], mv = ((float4*)mm)[i], vq = ((float4*)vv)[i];
        const float4 gv = ((const float4*)g)[i];
        pv.x = upd(pv.x, gv.x, mv.x, vq.x); pv.y = upd(pv.y, gv.y, mv.y, vq.y);
        pv.z = upd(pv.z, gv.z, mv.z, vq.z); pv.w = upd(pv.w, gv.w, mv.w, vq.w);
        ((float4*)mm)[i] = mv; ((float4*)vv)[i] = vq; ((float4*)p)[i] = pv;
    }
    for (i64 i = nv * 4 + (i64)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (i64)gridDim.x * blockDim.x) {
        float m1 = mm[i], v1 = vv[i];
        p[i] = upd(p[i], g[i], m1, v1);
        mm[i] = m1; vv[i] = v1;
    }
}

// =====================================================================================================
// host entry points
// =====================================================================================================
#define VEC_DISPATCH(T, cond_vec, KERNEL_CALL_G, KERNEL_CALL_1) \
    do {                                                        \
        if (cond_vec) { KERNEL_CALL_G; } else { KERNEL_CALL_1; } \
    } while (0)

template <typename T> constexpr int vecg() { return 16 / sizeof(T); }

extern "C" BIU_HIDDEN int biu_conv_fwd_direct(const biu_act* x, const biu_xform* xf, const float* w, const float* bias,
                                   int kd, int kh, int kw, int dil, const biu_act* y, int dtype, hipStream_t st) {
    i64 total = nvox(y) * y->c;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_conv_fwd_direct<T>, dim3(grid_for(total, TPB)), dim3(TPB), 0, st,
                                                 dact(x), dxf(xf), w, bias, kd, kh, kw, dil, dact(y)));
    BIU_CHECK_LAUNCH("conv_fwd_direct");
    return BIU_OK;
}
extern "C" BIU_HIDDEN int biu_conv_bwd_data_direct(const biu_act* dy, const float* w, int kd, int kh, int kw, int dil,
                                        const biu_act* dx, int accumulate, int dtype, hipStream_t st) {
    i64 total = nvox(dx) * dx->c;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_conv_dgrad_direct<T>, dim3(grid_for(total, TPB)), dim3(TPB), 0, st,
                                                 dact(dy), w, kd, kh, kw, dil, dact(dx), accumulate));
    BIU_CHECK_LAUNCH("conv_dgrad_direct");
    return BIU_OK;
}
extern "C" BIU_HIDDEN int biu_conv_bwd_weight_direct(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, int kh,
                                          int kw, int dil, float* dw, float* dbias, int dtype, hipStream_t st) {
    int outs = dy->c * x->c * kd * kh * kw;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_conv_wgrad_direct<T>, dim3(outs), dim3(TPB), 0, st, dact(x), dxf(xf),
                                                 dact(dy), kd, kh, kw, dil, dw));
    BIU_CHECK_LAUNCH("conv_wgrad_direct");
    if (dbias) {
        BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_chan_sum<T>, dim3(dy->c), dim3(TPB), 0, st, dact(dy), dbias));
        BIU_CHECK_LAUNCH("chan_sum");
    }
    return BIU_OK;
}

int biu_chan_sum(const biu_act* a, float* out, int dtype, hipStream_t st) {
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_chan_sum<T>, dim3(a->c), dim3(TPB), 0, st, dact(a), out));
    BIU_CHECK_LAUNCH("chan_sum");
    return BIU_OK;
}

extern "C" int biu_bn_stats(const biu_act* y, float* partial, int* nblk_out, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(y) && partial && nblk_out, BIU_ERR_SHAPE, "bn_stats: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (biu_vec_reduce_ok(y, dtype)) return biu_bn_stats_vec(y, partial, nblk_out, dtype, st);
    ReducePlan p = plan_reduce(nvox(y), y->c, BIU_BN_MAX_PARTIALS);
    BIU_DISPATCH_DTYPE(dtype, {
        StatsF<T> f{dact(y)};
        hipLaunchKernelGGL(k_chan_reduce2<StatsF<T>>, dim3(p.nblk), dim3(TPB), 0, st, f, nvox(y), y->c, p.cw, p.vpb, partial);
    });
    BIU_CHECK_LAUNCH("bn_stats");
    *nblk_out = p.nblk;
    return BIU_OK;
}

extern "C" int biu_bn_finalize(const float* partial, int nblk, int c, double count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* scale, float* shift, float* save_mean, float* save_invstd, biu_stream stream) {
    BIU_REQUIRE(partial && nblk > 0 && c > 0 && count > 0 && scale && shift, BIU_ERR_SHAPE, "bn_finalize: bad arguments");
    hipLaunchKernelGGL(k_bn_finalize, dim3(c), dim3(TPB), 0, (hipStream_t)stream, partial, nblk, c, count, gamma, beta,
                       running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd);
    BIU_CHECK_LAUNCH("bn_finalize");
    return BIU_OK;
}

extern "C" int biu_bn_eval_affine(int c, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift, biu_stream stream) {
    BIU_REQUIRE(c > 0 && running_mean && running_var && scale && shift, BIU_ERR_SHAPE, "bn_eval_affine: bad arguments");
    hipLaunchKernelGGL(k_bn_eval_affine, dim3((c + 63) / 64), dim3(64), 0, (hipStream_t)stream, c, gamma, beta,
                       running_mean, running_var, eps, scale, shift);
    BIU_CHECK_LAUNCH("bn_eval_affine");
    return BIU_OK;
}

#define EW_LAUNCH(KERNEL, total_groups_expr, okvec, ...)                                                           \
    BIU_DISPATCH_DTYPE(dtype, {                                                                                    \
        constexpr int G = vecg<T>();                                                                               \
        if (okvec) {                                                                                               \
            i64 tot = (total_groups_expr) / G;                                                                     \
            hipLaunchKernelGGL((KERNEL<T, G>), dim3(grid_for(tot, TPB, 8192)), dim3(TPB), 0, st, __VA_ARGS__);     \
        } else {                                                                                                   \
            i64 tot = (total_groups_expr);                                                                         \
            hipLaunchKernelGGL((KERNEL<T, 1>), dim3(grid_for(tot, TPB, 8192)), dim3(TPB), 0, st, __VA_ARGS__);     \
        }                                                                                                          \
    })

extern "C" int biu_xform_apply(const biu_act* x, const biu_xform* xf, const biu_act* out, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(x) && valid_act(out) && same_space(x, out) && x->c == out->c, BIU_ERR_SHAPE,
                "xform_apply: shape mismatch");
    hipStream_t st = (hipStream_t)stream;
    if (biu_rowvec_ok(x, dtype) && biu_rowvec_ok(out, dtype)) return biu_xform_apply_rv(x, xf, out, dtype, st);
    const int g = 16 / (int)dsize(dtype);
    bool ok = vec_ok(x, g, dtype) && vec_ok(out, g, dtype);
    EW_LAUNCH(k_xform_apply, nvox(x) * x->c, ok, dact(x), dxf(xf), dact(out));
    BIU_CHECK_LAUNCH("xform_apply");
    return BIU_OK;
}

extern "C" int biu_bn_bwd_reduce(const biu_act* da, const biu_act* y, const float* scale, const float* shift,
                                 const float* slope, const float* save_mean, const float* save_invstd, float* partial,
                                 int* nblk_out, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(da) && valid_act(y) && same_space(da, y) && da->c == y->c, BIU_ERR_SHAPE,
                "bn_bwd_reduce: shape mismatch");
    BIU_REQUIRE(scale && shift && save_mean && save_invstd && partial && nblk_out, BIU_ERR_SHAPE, "bn_bwd_reduce: null vector");
    hipStream_t st = (hipStream_t)stream;
    if (biu_vec_reduce_ok(y, dtype) && biu_vec_reduce_ok(da, dtype))
        return biu_bn_bwd_reduce_vec(da, y, scale, shift, slope, save_mean, save_invstd, partial, nblk_out, dtype, st);
    ReducePlan p = plan_reduce(nvox(y), y->c, BIU_BN_MAX_PARTIALS);
    BIU_DISPATCH_DTYPE(dtype, {
        BnBwdF<T> f{dact(da), dact(y), DXf{scale, shift, slope}, save_mean, save_invstd};
        hipLaunchKernelGGL(k_chan_reduce2<BnBwdF<T>>, dim3(p.nblk), dim3(TPB), 0, st, f, nvox(y), y->c, p.cw, p.vpb, partial);
    });
    BIU_CHECK_LAUNCH("bn_bwd_reduce");
    *nblk_out = p.nblk;
    return BIU_OK;
}

extern "C" int biu_bn_bwd_finalize(const float* partial, int nblk, int c, double count, const float* scale,
                                   const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                                   float* coefA, float* coefB, float* coefC, biu_stream stream) {
    BIU_REQUIRE(partial && nblk > 0 && c > 0 && count > 0 && scale && save_mean && save_invstd && coefA && coefB && coefC,
                BIU_ERR_SHAPE, "bn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(c), dim3(TPB), 0, (hipStream_t)stream, partial, nblk, c, count, scale,
                       save_mean, save_invstd, dgamma, dbeta, coefA, coefB, coefC);
    BIU_CHECK_LAUNCH("bn_bwd_finalize");
    return BIU_OK;
}

extern "C" int biu_bn_bwd_finalize_eval(const float* partial, int nblk, int c, const float* scale, float* dgamma, float* dbeta, float* dbias,
                                        float* coefA, float* coefB, float* coefC, biu_stream stream) {
    BIU_REQUIRE(partial && nblk > 0 && c > 0 && scale && coefA && coefB && coefC, BIU_ERR_SHAPE, "bn_bwd_finalize_eval: bad arguments");
    hipLaunchKernelGGL(k_bn_bwd_finalize_eval, dim3(c), dim3(TPB), 0, (hipStream_t)stream, partial, nblk, c, scale, dgamma, dbeta, dbias, coefA, coefB,
                       coefC);
    BIU_CHECK_LAUNCH("bn_bwd_finalize_eval");
    return BIU_OK;
}

extern "C" int biu_bn_bwd_apply(const biu_act* da, const biu_act* y, const float* scale, const float* shift,
                                const float* slope, const float* coefA, const float* coefB, const float* coefC,
                                const biu_act* dy, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(da) && valid_act(y) && valid_act(dy) && same_space(da, y) && same_space(da, dy) &&
                    da->c == y->c && dy->c == y->c, BIU_ERR_SHAPE, "bn_bwd_apply: shape mismatch");
    hipStream_t st = (hipStream_t)stream;
    if (biu_rowvec_ok(da, dtype) && biu_rowvec_ok(y, dtype) && biu_rowvec_ok(dy, dtype))
        return biu_bn_bwd_apply_rv(da, y, scale, shift, slope, coefA, coefB, coefC, dy, dtype, st);
    const int g = 16 / (int)dsize(dtype);
    bool ok = vec_ok(da, g, dtype) && vec_ok(y, g, dtype) && vec_ok(dy, g, dtype);
    EW_LAUNCH(k_bn_bwd_apply, nvox(y) * y->c, ok, dact(da), dact(y), DXf{scale, shift, slope}, coefA, coefB, coefC, dact(dy));
    BIU_CHECK_LAUNCH("bn_bwd_apply");
    return BIU_OK;
}

// ---- pooling ---------------------------------------------------------------------------------------
static int pool_window(const biu_act* big, const biu_act* small, const char* who) {
    if (big->n != small->n || big->c != small->c || big->h != 2 * small->h || big->w != 2 * small->w) {
        biu_fail(BIU_ERR_SHAPE, "%s: expected (h,w) = 2x of the pooled tensor and equal n,c", who);
        return 0;
    }
    if (big->d == small->d) return 1;
    if (big->d == 2 * small->d) return 2;
    biu_fail(BIU_ERR_SHAPE, "%s: depth %d vs %d is neither 1x nor 2x", who, big->d, small->d);
    return 0;
}

extern "C" int biu_maxpool_fwd(const biu_act* x, const biu_xform* xf, const biu_act* out, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(x) && valid_act(out), BIU_ERR_SHAPE, "maxpool_fwd: bad tensor");
    int pd = pool_window(x, out, "maxpool_fwd");
    if (!pd) return BIU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (biu_rowvec_ok(x, dtype) && biu_rowvec_ok(out, dtype)) return biu_maxpool_fwd_rv(x, xf, out, pd, dtype, st);
    const int g = 16 / (int)dsize(dtype);
    bool ok = vec_ok(x, g, dtype) && vec_ok(out, g, dtype);
    EW_LAUNCH(k_maxpool_fwd, nvox(out) * out->c, ok, dact(x), dxf(xf), dact(out), pd);
    BIU_CHECK_LAUNCH("maxpool_fwd");
    return BIU_OK;
}
extern "C" int biu_maxpool_bwd(const biu_act* x, const biu_xform* xf, const biu_act* dout, const biu_act* dx,
                               int accumulate, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(x) && valid_act(dout) && valid_act(dx) && same_space(x, dx) && x->c == dx->c, BIU_ERR_SHAPE,
                "maxpool_bwd: bad tensor");
    int pd = pool_window(x, dout, "maxpool_bwd");
    if (!pd) return BIU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (biu_rowvec_ok(x, dtype) && biu_rowvec_ok(dout, dtype) && biu_rowvec_ok(dx, dtype))
        return biu_maxpool_bwd_rv(x, xf, dout, dx, pd, accumulate, dtype, st);
    const int g = 16 / (int)dsize(dtype);
    bool ok = vec_ok(x, g, dtype) && vec_ok(dout, g, dtype) && vec_ok(dx, g, dtype);
    EW_LAUNCH(k_maxpool_bwd, nvox(dout) * dout->c, ok, dact(x), dxf(xf), dact(dout), dact(dx), pd, accumulate);
    BIU_CHECK_LAUNCH("maxpool_bwd");
    return BIU_OK;
}
// max-pool backward + the BatchNorm-backward sums of the block that produced x (valid when this call completes x's gradient)
extern "C" int biu_maxpool_bwd_bnred(const biu_act* x, const biu_xform* xf, const biu_act* dout, const biu_act* dx, int accumulate,
                                     const float* mean, const float* invstd, float* partial, size_t partial_floats, int* nblk,
                                     int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(x) && valid_act(dout) && valid_act(dx) && same_space(x, dx) && x->c == dx->c, BIU_ERR_SHAPE,
                "maxpool_bwd_bnred: bad tensor");
    BIU_REQUIRE(xf && xf->scale && xf->shift && mean && invstd && partial && nblk, BIU_ERR_SHAPE, "maxpool_bwd_bnred: null vector");
    BIU_REQUIRE(partial_floats >= (size_t)BIU_BN_MAX_PARTIALS * x->c * 2, BIU_ERR_WORKSPACE, "maxpool_bwd_bnred: partial buffer too small");
    int pd = pool_window(x, dout, "maxpool_bwd_bnred");
    if (!pd) return BIU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (!getenv("BIU_NO_POOL_BNRED") && biu_rowvec_ok(x, dtype) && biu_rowvec_ok(dout, dtype) && biu_rowvec_ok(dx, dtype) &&
        x->c / 4 <= 256 && (size_t)(256 / (x->c / 4)) * x->c * 2 * sizeof(float) <= 64 * 1024)
        return biu_maxpool_bwd_bnred_rv(x, xf, dout, dx, pd, accumulate, mean, invstd, partial, partial_floats, nblk, dtype, st);
    int rc = biu_maxpool_bwd(x, xf, dout, dx, accumulate, dtype, stream);
    if (rc != BIU_OK) return rc;
    return biu_bn_bwd_reduce(dx, x, xf->scale, xf->shift, xf->slope, mean, invstd, partial, nblk, dtype, stream);
}
extern "C" int biu_nearest_down_fwd(const biu_act* x, const biu_xform* xf, const biu_act* out, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(x) && valid_act(out), BIU_ERR_SHAPE, "nearest_down_fwd: bad tensor");
    int pd = pool_window(x, out, "nearest_down_fwd");
    if (!pd) return BIU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (biu_rowvec_ok(x, dtype) && biu_rowvec_ok(out, dtype)) return biu_nearest_rv(0, x, xf, out, pd, 0, dtype, st);
    const int g = 16 / (int)dsize(dtype);
    bool ok = vec_ok(x, g, dtype) && vec_ok(out, g, dtype);
    EW_LAUNCH(k_nearest_down_fwd, nvox(out) * out->c, ok, dact(x), dxf(xf), dact(out), pd);
    BIU_CHECK_LAUNCH("nearest_down_fwd");
    return BIU_OK;
}
extern "C" int biu_nearest_down_bwd(const biu_act* dout, const biu_act* dx, int accumulate, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(dout) && valid_act(dx), BIU_ERR_SHAPE, "nearest_down_bwd: bad tensor");
    int pd = pool_window(dx, dout, "nearest_down_bwd");
    if (!pd) return BIU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (biu_rowvec_ok(dout, dtype) && biu_rowvec_ok(dx, dtype)) return biu_nearest_rv(3, dout, nullptr, dx, pd, accumulate, dtype, st);
    const int g = 16 / (int)dsize(dtype);
    bool ok = vec_ok(dout, g, dtype) && vec_ok(dx, g, dtype);
    EW_LAUNCH(k_nearest_down_bwd, nvox(dx) * dx->c, ok, dact(dout), dact(dx), pd, accumulate);
    BIU_CHECK_LAUNCH("nearest_down_bwd");
    return BIU_OK;
}
extern "C" int biu_nearest_up_fwd(const biu_act* x, const biu_xform* xf, const biu_act* out, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(x) && valid_act(out), BIU_ERR_SHAPE, "nearest_up_fwd: bad tensor");
    int pd = pool_window(out, x, "nearest_up_fwd");
    if (!pd) return BIU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (biu_rowvec_ok(x, dtype) && biu_rowvec_ok(out, dtype)) return biu_nearest_rv(1, x, xf, out, pd, 0, dtype, st);
    const int g = 16 / (int)dsize(dtype);
    bool ok = vec_ok(x, g, dtype) && vec_ok(out, g, dtype);
    EW_LAUNCH(k_nearest_up_fwd, nvox(out) * out->c, ok, dact(x), dxf(xf), dact(out), pd);
    BIU_CHECK_LAUNCH("nearest_up_fwd");
    return BIU_OK;
}
extern "C" int biu_nearest_up_bwd(const biu_act* dout, const biu_act* dx, int accumulate, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(dout) && valid_act(dx), BIU_ERR_SHAPE, "nearest_up_bwd: bad tensor");
    int pd = pool_window(dout, dx, "nearest_up_bwd");
    if (!pd) return BIU_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (biu_rowvec_ok(dout, dtype) && biu_rowvec_ok(dx, dtype)) return biu_nearest_rv(2, dout, nullptr, dx, pd, accumulate, dtype, st);
    const int g = 16 / (int)dsize(dtype);
    bool ok = vec_ok(dout, g, dtype) && vec_ok(dx, g, dtype);
    EW_LAUNCH(k_nearest_up_bwd, nvox(dx) * dx->c, ok, dact(dout), dact(dx), pd, accumulate);
    BIU_CHECK_LAUNCH("nearest_up_bwd");
    return BIU_OK;
}

// ---- ConvTranspose ---------------------------------------------------------------------------------
static bool convt_shapes(const biu_act* lo, const biu_act* hi, int kd) {
    return lo->n == hi->n && hi->h == 2 * lo->h && hi->w == 2 * lo->w && hi->d == kd * lo->d && (kd == 1 || kd == 2);
}
extern "C" BIU_HIDDEN int biu_convt_fwd_direct(const biu_act* x, const biu_xform* xf, const float* w, const float* bias, int kd,
                                    const biu_act* y, int dtype, hipStream_t st) {
    i64 total = nvox(y) * y->c;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_convt_fwd_direct<T>, dim3(grid_for(total, TPB)), dim3(TPB), 0, st,
                                                 dact(x), dxf(xf), w, bias, kd, dact(y)));
    BIU_CHECK_LAUNCH("convt_fwd_direct");
    return BIU_OK;
}
extern "C" BIU_HIDDEN int biu_convt_bwd_data_direct(const biu_act* dy, const float* w, int kd, const biu_act* dx, int accumulate,
                                         int dtype, hipStream_t st) {
    i64 total = nvox(dx) * dx->c;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_convt_dgrad_direct<T>, dim3(grid_for(total, TPB)), dim3(TPB), 0, st,
                                                 dact(dy), w, kd, dact(dx), accumulate));
    BIU_CHECK_LAUNCH("convt_dgrad_direct");
    return BIU_OK;
}
extern "C" BIU_HIDDEN int biu_convt_bwd_weight_direct(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, float* dw,
                                           float* dbias, int dtype, hipStream_t st) {
    int outs = x->c * dy->c * kd * 4;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_convt_wgrad_direct<T>, dim3(outs), dim3(TPB), 0, st, dact(x), dxf(xf),
                                                 dact(dy), kd, dw));
    BIU_CHECK_LAUNCH("convt_wgrad_direct");
    if (dbias) {
        BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_chan_sum<T>, dim3(dy->c), dim3(TPB), 0, st, dact(dy), dbias));
        BIU_CHECK_LAUNCH("chan_sum");
    }
    return BIU_OK;
}
bool biu_convt_shapes_ok(const biu_act* lo, const biu_act* hi, int kd) { return convt_shapes(lo, hi, kd); }

// ---- head ------------------------------------------------------------------------------------------
extern "C" int biu_head_fwd(const biu_act* x, const biu_xform* xf, const float* w, const float* bias, int cout, int act,
                            float* logits, float* activated, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(x) && w && cout > 0 && cout <= HEAD_MAX_COUT, BIU_ERR_UNSUPPORTED,
                "head_fwd: cout must be in 1..%d", HEAD_MAX_COUT);
    BIU_REQUIRE(act >= 0 && act <= 3, BIU_ERR_UNSUPPORTED, "head_fwd: unknown activation %d", act);
    hipStream_t st = (hipStream_t)stream;
    if (vec_ok(x, 16 / (int)dsize(dtype), dtype)) {
        bool done = false;
        BIU_DISPATCH_DTYPE(dtype, {
            switch (x->c) {
                case 8: done = head_fwd_vec_launch<T, 8>(x, xf, w, bias, cout, act, logits, activated, st); break;
                case 16: done = head_fwd_vec_launch<T, 16>(x, xf, w, bias, cout, act, logits, activated, st); break;
                case 32: done = head_fwd_vec_launch<T, 32>(x, xf, w, bias, cout, act, logits, activated, st); break;
                case 64: done = head_fwd_vec_launch<T, 64>(x, xf, w, bias, cout, act, logits, activated, st); break;
                default: break;
            }
        });
        if (done) {
            BIU_CHECK_LAUNCH("head_fwd_vec");
            return BIU_OK;
        }
    }
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_head_fwd<T>, dim3(grid_for(nvox(x), TPB, 8192)), dim3(TPB), 0, st,
                                                 dact(x), dxf(xf), w, bias, cout, act, logits, activated));
    BIU_CHECK_LAUNCH("head_fwd");
    return BIU_OK;
}
extern "C" size_t biu_head_bwd_workspace(int cin) {
    size_t a = (size_t)BIU_BN_MAX_PARTIALS * cin * 2 * sizeof(float), b = biu_head_bwd_fused_workspace(cin);
    return a > b ? a : b;
}
extern "C" int biu_head_bwd(const biu_act* x, const biu_xform* xf, const float* w, int cout, const float* dlogits,
                            const biu_act* dx, float* dw, float* dbias, void* ws, size_t ws_bytes, int dtype,
                            biu_stream stream) {
    BIU_REQUIRE(valid_act(x) && w && dlogits && cout > 0 && cout <= HEAD_MAX_COUT, BIU_ERR_UNSUPPORTED, "head_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dx) BIU_REQUIRE(valid_act(dx) && same_space(x, dx) && dx->c == x->c, BIU_ERR_SHAPE, "head_bwd: dx shape mismatch");
    if (biu_head_bwd_fused_ok(x, dx, cout, dtype) && ws && ws_bytes >= biu_head_bwd_fused_workspace(x->c))
        return biu_head_bwd_fused(x, xf, w, cout, dlogits, dx, dw, dbias, ws, ws_bytes, dtype, st);
    if (dx) {
        i64 total = nvox(dx) * dx->c;
        BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_head_bwd_data<T>, dim3(grid_for(total, TPB, 16384)), dim3(TPB), 0,
                                                     st, x->c, w, cout, dlogits, dact(dx)));
        BIU_CHECK_LAUNCH("head_bwd_data");
    }
    if (dw) {
        BIU_REQUIRE(ws && ws_bytes >= biu_head_bwd_workspace(x->c), BIU_ERR_WORKSPACE, "head_bwd: workspace too small");
        ReducePlan p = plan_reduce(nvox(x), x->c, BIU_BN_MAX_PARTIALS);
        i64 S = (i64)x->d * x->h * x->w;
        for (int o0 = 0; o0 < cout; o0 += 2) {
            BIU_DISPATCH_DTYPE(dtype, {
                HeadWgradF<T> f{dact(x), dxf(xf), dlogits, cout, o0, S};
                hipLaunchKernelGGL(k_chan_reduce2<HeadWgradF<T>>, dim3(p.nblk), dim3(TPB), 0, st, f, nvox(x), x->c, p.cw,
                                   p.vpb, (float*)ws);
            });
            BIU_CHECK_LAUNCH("head_wgrad");
            hipLaunchKernelGGL(k_partial_sum, dim3(x->c), dim3(TPB), 0, st, (const float*)ws, p.nblk, x->c,
                               dw + (i64)o0 * x->c, (o0 + 1 < cout) ? dw + (i64)(o0 + 1) * x->c : (float*)nullptr);
            BIU_CHECK_LAUNCH("partial_sum");
        }
    }
    if (dbias) {
        i64 S = (i64)x->d * x->h * x->w;
        hipLaunchKernelGGL(k_head_dbias, dim3(cout), dim3(TPB), 0, st, dlogits, x->n, cout, S, dbias);
        BIU_CHECK_LAUNCH("head_dbias");
    }
    return BIU_OK;
}

// head backward that also emits biu_bn_bwd_reduce's partial sums for the block that produced x (the head being its only reader)
extern "C" int biu_head_bwd_bnred(const biu_act* x, const biu_xform* xf, const float* w, int cout, const float* dlogits,
                                  const biu_act* dx, float* dw, float* dbias, void* ws, size_t ws_bytes, const float* mean,
                                  const float* invstd, float* partial, size_t partial_floats, int* nblk, int dtype,
                                  biu_stream stream) {
    BIU_REQUIRE(valid_act(x) && w && dlogits && dx && valid_act(dx) && same_space(x, dx) && dx->c == x->c && cout > 0 &&
                    cout <= HEAD_MAX_COUT, BIU_ERR_SHAPE, "head_bwd_bnred: bad arguments");
    BIU_REQUIRE(xf && xf->scale && xf->shift && mean && invstd && partial && nblk, BIU_ERR_SHAPE, "head_bwd_bnred: null vector");
    BIU_REQUIRE(partial_floats >= (size_t)BIU_BN_MAX_PARTIALS * x->c * 2, BIU_ERR_WORKSPACE, "head_bwd_bnred: partial buffer too small");
    if (!getenv("BIU_NO_HEAD_BNRED") && biu_head_bwd_bnred_ok(x, dx, cout, dtype) && ws && ws_bytes >= biu_head_bwd_fused_workspace(x->c))
        return biu_head_bwd_bnred_fused(x, xf, w, cout, dlogits, dx, dw, dbias, ws, mean, invstd, partial, nblk, dtype, (hipStream_t)stream);
    int rc = biu_head_bwd(x, xf, w, cout, dlogits, dx, dw, dbias, ws, ws_bytes, dtype, stream);
    if (rc != BIU_OK) return rc;
    return biu_bn_bwd_reduce(dx, x, xf->scale, xf->shift, xf->slope, mean, invstd, partial, nblk, dtype, stream);
}

// ---- element-wise ----------------------------------------------------------------------------------
extern "C" int biu_max_join_fwd(const biu_act* a, const biu_xform* xa, const biu_act* b, const biu_xform* xb,
                                const biu_act* out, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(a) && valid_act(b) && valid_act(out) && same_space(a, b) && same_space(a, out) &&
                    a->c == b->c && a->c == out->c, BIU_ERR_SHAPE, "max_join_fwd: shape mismatch");
    hipStream_t st = (hipStream_t)stream;
    const int g = 16 / (int)dsize(dtype);
    bool ok = vec_ok(a, g, dtype) && vec_ok(b, g, dtype) && vec_ok(out, g, dtype);
    EW_LAUNCH(k_max_join_fwd, nvox(a) * a->c, ok, dact(a), dxf(xa), dact(b), dxf(xb), dact(out));
    BIU_CHECK_LAUNCH("max_join_fwd");
    return BIU_OK;
}
extern "C" int biu_max_join_bwd(const biu_act* a, const biu_xform* xa, const biu_act* b, const biu_xform* xb,
                                const biu_act* dout, const biu_act* da, const biu_act* db, int accumulate, int dtype,
                                biu_stream stream) {
    BIU_REQUIRE(valid_act(a) && valid_act(b) && valid_act(dout) && valid_act(da) && valid_act(db) && same_space(a, b) &&
                    same_space(a, dout) && same_space(a, da) && same_space(a, db) && a->c == b->c && a->c == dout->c &&
                    a->c == da->c && a->c == db->c, BIU_ERR_SHAPE, "max_join_bwd: shape mismatch");
    hipStream_t st = (hipStream_t)stream;
    const int g = 16 / (int)dsize(dtype);
    bool ok = vec_ok(a, g, dtype) && vec_ok(b, g, dtype) && vec_ok(dout, g, dtype) && vec_ok(da, g, dtype) && vec_ok(db, g, dtype);
    EW_LAUNCH(k_max_join_bwd, nvox(a) * a->c, ok, dact(a), dxf(xa), dact(b), dxf(xb), dact(dout), dact(da), dact(db), accumulate);
    BIU_CHECK_LAUNCH("max_join_bwd");
    return BIU_OK;
}
extern "C" int biu_act_add(const biu_act* src, const biu_act* dst, int accumulate, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(src) && valid_act(dst) && same_space(src, dst) && src->c == dst->c, BIU_ERR_SHAPE,
                "act_add: shape mismatch");
    hipStream_t st = (hipStream_t)stream;
    const int g = 16 / (int)dsize(dtype);
    bool ok = vec_ok(src, g, dtype) && vec_ok(dst, g, dtype);
    EW_LAUNCH(k_act_add, nvox(src) * src->c, ok, dact(src), dact(dst), accumulate);
    BIU_CHECK_LAUNCH("act_add");
    return BIU_OK;
}
extern "C" int biu_from_nchw(const float* src, const biu_act* dst, int dtype, biu_stream stream) {
    BIU_REQUIRE(src && valid_act(dst), BIU_ERR_SHAPE, "from_nchw: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    i64 total = nvox(dst) * dst->c;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_from_nchw<T>, dim3(grid_for(total, TPB, 16384)), dim3(TPB), 0, st, src, dact(dst)));
    BIU_CHECK_LAUNCH("from_nchw");
    return BIU_OK;
}
extern "C" int biu_to_nchw(const biu_act* src, const biu_xform* xf, float* dst, int dtype, biu_stream stream) {
    BIU_REQUIRE(dst && valid_act(src), BIU_ERR_SHAPE, "to_nchw: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    i64 total = nvox(src) * src->c;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_to_nchw<T>, dim3(grid_for(total, TPB, 16384)), dim3(TPB), 0, st, dact(src), dxf(xf), dst));
    BIU_CHECK_LAUNCH("to_nchw");
    return BIU_OK;
}

extern "C" int biu_adam_step(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2,
                             float eps, int step, float grad_scale, biu_stream stream) {
    BIU_REQUIRE(n > 0 && params && grads && exp_avg && exp_avg_sq && numel && step >= 1, BIU_ERR_SHAPE, "adam_step: bad arguments");
    float bc1 = 1.f - powf(beta1, (float)step);
    float bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(k_adam, dim3(128, n), dim3(TPB), 0, (hipStream_t)stream, n, params, grads, exp_avg, exp_avg_sq, numel,
                       lr, beta1, beta2, eps, bc1, sqrtf(bc2), grad_scale, (const float*)nullptr);
    BIU_CHECK_LAUNCH("adam_step");
    return BIU_OK;
}

__global__ void k_adam_set_hyper(float* hyper, float lr, float b1, float b2, float eps, float gscale, float step) {
    hyper[0] = lr; hyper[1] = b1; hyper[2] = b2; hyper[3] = eps; hyper[4] = gscale; hyper[5] = step;
}

extern "C" int biu_adam_set_hyper(float* hyper, float lr, float beta1, float beta2, float eps, int step, float grad_scale, biu_stream stream) {
    BIU_REQUIRE(hyper && step >= 1, BIU_ERR_SHAPE, "adam_set_hyper: bad arguments");
    hipLaunchKernelGGL(k_adam_set_hyper, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper, lr, beta1, beta2, eps, grad_scale, (float)step);
    BIU_CHECK_LAUNCH("adam_set_hyper");
    return BIU_OK;
}

extern "C" int biu_adam_step_hyper(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                   const int64_t* numel, const float* hyper, biu_stream stream) {
    BIU_REQUIRE(n > 0 && params && grads && exp_avg && exp_avg_sq && numel && hyper, BIU_ERR_SHAPE, "adam_step_hyper: bad arguments");
    hipLaunchKernelGGL(k_adam, dim3(128, n), dim3(TPB), 0, (hipStream_t)stream, n, params, grads, exp_avg, exp_avg_sq, numel,
                       0.f, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, hyper);
    BIU_CHECK_LAUNCH("adam_step_hyper");
    return BIU_OK;
}

// =====================================================================================================
// gradient-norm clipping over a table of tensors (torch.nn.utils.clip_grad_norm_(params, max_norm), multi_output_unet3d/train.py:201):
//   total = sqrt(sum_i |g_i|^2),  coef = min(1, max_norm / (total + 1e-6)),  g_i *= coef in place.
// Three launches whatever the number of tensors, every sum in a fixed order (no atomics): per-(tensor, block) partial sums of squares,
// one block folding them (double accumulators) into {total, coef}, and the in-place scaling -- which returns at once when coef == 1
// (torch multiplies by 1.0 then: the same values).
// =====================================================================================================
constexpr int CLIP_BX = 32;
__global__ __launch_bounds__(256) void k_grad_sumsq(int n, const float* const* grads, const int64_t* numel, float* __restrict__ partial) {
    __shared__ float red[256];
    const int t = blockIdx.y;
    const float* g = grads[t];
    const i64 cnt = numel[t];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const bool vec = ((uintptr_t)g & 15) == 0;
    const i64 nv = vec ? cnt / 4 : 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (i64)gridDim.x * blockDim.x) {
        const float4 v = ((const float4*)g)[i];
        s0 = fmaf(v.x, v.x, s0); s1 = fmaf(v.y, v.y, s1); s2 = fmaf(v.z, v.z, s2); s3 = fmaf(v.w, v.w, s3);
    }
    for (i64 i = nv * 4 + (i64)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (i64)gridDim.x * blockDim.x) s0 = fmaf(g[i], g[i], s0);
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)t * CLIP_BX + blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void k_grad_clip_coef(int nparts, const float* __restrict__ partial, float max_norm, float* __restrict__ out2,
                                                        float* __restrict__ total_norm) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(red[0]);
        const float coef = max_norm / (total + 1e-6f);
        out2[0] = total;
        out2[1] = coef < 1.f ? coef : 1.f;
        if (total_norm) *total_norm = total;
    }
}
__global__ __launch_bounds__(256) void k_grad_scale(int n, float* const* grads, const int64_t* numel, const float* __restrict__ coef) {
    const float c = *coef;
    if (c >= 1.f) return;
    const int t = blockIdx.y;
    float* g = grads[t];
    const i64 cnt = numel[t];
    const bool vec = ((uintptr_t)g & 15) == 0;
    const i64 nv = vec ? cnt / 4 : 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (i64)gridDim.x * blockDim.x) {
        float4 v = ((float4*)g)[i];
        v.x *= c; v.y *= c; v.z *= c; v.w *= c;
        ((float4*)g)[i] = v;
    }
    for (i64 i = nv * 4 + (i64)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (i64)gridDim.x * blockDim.x) g[i] *= c;
}
extern "C" size_t biu_grad_clip_scratch_floats(int n) { return n > 0 ? (size_t)n * CLIP_BX + 2 : 0; }
extern "C" int biu_grad_clip(int n, float* const* grads, const int64_t* numel, float max_norm, float* scratch, size_t scratch_floats, float* total_norm,
                             biu_stream stream) {
    BIU_REQUIRE(n > 0 && grads && numel && scratch && max_norm > 0.f, BIU_ERR_SHAPE, "grad_clip: bad arguments");
    BIU_REQUIRE(scratch_floats >= biu_grad_clip_scratch_floats(n), BIU_ERR_WORKSPACE, "grad_clip: scratch %zu floats too small (need %zu)", scratch_floats,
                biu_grad_clip_scratch_floats(n));
    hipStream_t st = (hipStream_t)stream;
    float* out2 = scratch + (size_t)n * CLIP_BX;
    hipLaunchKernelGGL(k_grad_sumsq, dim3(CLIP_BX, n), dim3(256), 0, st, n, (const float* const*)grads, numel, scratch);
    hipLaunchKernelGGL(k_grad_clip_coef, dim3(1), dim3(256), 0, st, n * CLIP_BX, (const float*)scratch, max_norm, out2, total_norm);
    hipLaunchKernelGGL(k_grad_scale, dim3(CLIP_BX, n), dim3(256), 0, st, n, grads, numel, (const float*)(out2 + 1));
    BIU_CHECK_LAUNCH("grad_clip");
    return BIU_OK;
}
