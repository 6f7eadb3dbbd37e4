// Bandwidth-bound special cases of the hot path, written for the shapes the real networks hit:
//   * first layer (Cin = 1): 3x3(x3) conv forward and weight gradient on the vector ALUs (no MFMA: K = 27)
//   * 1x1 head backward fused: dx, dW and dbias in ONE pass over (x, dlogits)
//   * per-channel two-term reductions (BatchNorm statistics, BatchNorm backward sums, channel sums) with 16-byte
//     loads and deterministic per-block partials
#include "biu_internal.h"
#include <cstring>
#include <cstdlib>
#include <stdlib.h>

#define TPB 256

struct Vox4 { int n, d, h, w; };
__device__ __forceinline__ Vox4 unvox4(i64 v, int D, int H, int W) {
    Vox4 r;
    if ((unsigned long long)v < (1ull << 32)) {      // the common case: three 32-bit divisions instead of three 64-bit ones (~4x fewer instructions)
        unsigned u = (unsigned)v;
        r.w = (int)(u % (unsigned)W); u /= (unsigned)W;
        r.h = (int)(u % (unsigned)H); u /= (unsigned)H;
        r.d = (int)(u % (unsigned)D); u /= (unsigned)D;
        r.n = (int)u;
        return r;
    }
    r.w = (int)(v % W); v /= W;
    r.h = (int)(v % H); v /= H;
    r.d = (int)(v % D); v /= D;
    r.n = (int)v;
    return r;
}

// =====================================================================================================================
// Cin = 1 convolution forward:  y[v, co] = b[co] + sum_tap T(x)[v + off(tap)] * w[co][tap]
// One thread per voxel, all COUT channels in registers, weights broadcast from LDS.  Writes COUT*sizeof(T) contiguous
// bytes per lane.  VALU-bound (27*COUT FMA per voxel) at ~0.2 ms for 8.4 M voxels x 16 channels.
// =====================================================================================================================
template <typename T, int COUT, int KD>
__global__ __launch_bounds__(TPB) void k_conv_c1_fwd(DAct x, DXf xf, const float* __restrict__ w, const float* __restrict__ bias,
                                                     DAct y, int co0) {
    constexpr int TAPS = KD * 9;
    __shared__ float ws[TAPS * COUT + COUT];
    for (int i = threadIdx.x; i < TAPS * COUT; i += TPB) {
        const int tap = i / COUT, co = i % COUT;
        ws[i] = w[(i64)(co0 + co) * TAPS + tap];                   // PyTorch (Cout, 1, taps)
    }
    for (int i = threadIdx.x; i < COUT; i += TPB) ws[TAPS * COUT + i] = bias ? bias[co0 + i] : 0.f;
    __syncthreads();
    const float s = xf.scale ? xf.scale[0] : 1.f, b = xf.shift ? xf.shift[0] : 0.f, sl = xf.slope ? xf.slope[0] : 1.f;
    const i64 total = (i64)y.n * y.d * y.h * y.w;
    for (i64 v = (i64)blockIdx.x * TPB + threadIdx.x; v < total; v += (i64)gridDim.x * TPB) {
        const Vox4 p = unvox4(v, y.d, y.h, y.w);
        float acc[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = ws[TAPS * COUT + c];
#pragma unroll 1
        for (int a = 0; a < KD; ++a) {
            const int id = p.d + a - KD / 2;
#pragma unroll 1
            for (int bb = 0; bb < 3; ++bb) {
                const int ih = p.h + bb - 1;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int iw = p.w + c - 1;
                    const bool ok = id >= 0 && id < x.d && ih >= 0 && ih < x.h && iw >= 0 && iw < x.w;
                    const i64 iv = ok ? (((i64)p.n * x.d + id) * x.h + ih) * x.w + iw : 0;
                    const float t = fmaf(s, to_f(((const T*)x.p)[iv * x.pitch]), b);
                    const float xv = ok ? (t > 0.f ? t : sl * t) : 0.f;
                    const float* wr = ws + ((a * 3 + bb) * 3 + c) * COUT;
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv, wr[co], acc[co]);
                }
            }
        }
        T* dst = (T*)y.p + v * y.pitch + co0;
        constexpr int G = 16 / sizeof(T);
#pragma unroll
        for (int c = 0; c < COUT; c += G) {
            Pack<T, G> o;
#pragma unroll
            for (int j = 0; j < G; ++j) o.v[j] = from_f<T>(acc[c + j]);
            *(Pack<T, G>*)(dst + c) = o;
        }
    }
}

// =====================================================================================================================
// Cin = 1 weight gradient:  dw[co][tap] = sum_v T(x)[v + off(tap)] * dy[v][co]
// Block = NW waves; wave q owns taps [q*TPW, q*TPW + TPW); a lane walks voxels and keeps TPW x COUT partial sums.
// Per-block partials -> fp64 merge (deterministic).
// =====================================================================================================================
template <typename T, int COUT, int KD, int TPW>
__global__ __launch_bounds__(576) void k_conv_c1_wgrad(DAct x, DXf xf, DAct dy, int co0, float* __restrict__ partial /* [nblk][TAPS*COUT] */) {
    constexpr int TAPS = KD * 9;
    constexpr int G = 16 / sizeof(T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tap0 = wave * TPW;
    float acc[TPW][COUT];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[t][c] = 0.f;
    const float s = xf.scale ? xf.scale[0] : 1.f, b = xf.shift ? xf.shift[0] : 0.f, sl = xf.slope ? xf.slope[0] : 1.f;
    const i64 total = (i64)dy.n * dy.d * dy.h * dy.w;
    for (i64 v = (i64)blockIdx.x * 64 + lane; v < total; v += (i64)gridDim.x * 64) {
        const Vox4 p = unvox4(v, dy.d, dy.h, dy.w);
        float g[COUT];
        const T* src = (const T*)dy.p + v * dy.pitch + co0;
#pragma unroll
        for (int c = 0; c < COUT; c += G) {
            Pack<T, G> in = *(const Pack<T, G>*)(src + c);
#pragma unroll
            for (int j = 0; j < G; ++j) g[c + j] = to_f(in.v[j]);
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tap = tap0 + t;
            if (tap < TAPS) {
                const int a = tap / 9, bb = (tap / 3) % 3, c = tap % 3;
                const int id = p.d + a - KD / 2, ih = p.h + bb - 1, iw = p.w + c - 1;
                const bool ok = id >= 0 && id < x.d && ih >= 0 && ih < x.h && iw >= 0 && iw < x.w;
                const i64 iv = ok ? (((i64)p.n * x.d + id) * x.h + ih) * x.w + iw : 0;
                const float tt = fmaf(s, to_f(((const T*)x.p)[iv * x.pitch]), b);
                const float xv = ok ? (tt > 0.f ? tt : sl * tt) : 0.f;
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[t][co] = fmaf(xv, g[co], acc[t][co]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = tap0 + t;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float r = wave_sum(acc[t][co]);
            if (lane == 0 && tap < TAPS) partial[(i64)blockIdx.x * (TAPS * COUT) + tap * COUT + co] = r;
        }
    }
}

// out[co0 + co][tap] (+ layout cout x taps) = sum_b partial[b][tap*COUT + co]
__global__ void k_c1_wgrad_finalize(const float* __restrict__ partial, int nblk, int taps, int cout_chunk, int co0, float* __restrict__ dw) {
    __shared__ double red[TPB];
    const int o = blockIdx.x;             // tap * cout_chunk + co
    double a = 0;
    for (int b = threadIdx.x; b < nblk; b += TPB) a += partial[(i64)b * (taps * cout_chunk) + o];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = TPB / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int tap = o / cout_chunk, co = o % cout_chunk;
        dw[(i64)(co0 + co) * taps + tap] = (float)red[0];
    }
}

// ---- 4 voxels per thread (consecutive along W): every LDS weight read (4 channels, 16 B, broadcast) feeds 16 FMAs,
//      and each input row of 6 values serves 3 taps x 4 voxels.  Needs W % 4 == 0.
template <typename T, int COUT, int KD>
__global__ __launch_bounds__(TPB) void k_conv_c1_fwd4(DAct x, DXf xf, const float* __restrict__ w, const float* __restrict__ bias,
                                                      DAct y, int co0) {
    constexpr int TAPS = KD * 9;
    __shared__ __attribute__((aligned(16))) float ws[TAPS * COUT + COUT];
    for (int i = threadIdx.x; i < TAPS * COUT; i += TPB) {
        const int tap = i / COUT, co = i % COUT;
        ws[i] = w[(i64)(co0 + co) * TAPS + tap];
    }
    for (int i = threadIdx.x; i < COUT; i += TPB) ws[TAPS * COUT + i] = bias ? bias[co0 + i] : 0.f;
    __syncthreads();
    const float s = xf.scale ? xf.scale[0] : 1.f, b = xf.shift ? xf.shift[0] : 0.f, sl = xf.slope ? xf.slope[0] : 1.f;
    const unsigned W4 = (unsigned)y.w / 4u;          // 32-bit index arithmetic (nvox < 2^31, checked on the host)
    const unsigned total = (unsigned)y.n * (unsigned)y.d * (unsigned)y.h * W4;
    for (unsigned gidx = blockIdx.x * (unsigned)TPB + threadIdx.x; gidx < total; gidx += gridDim.x * (unsigned)TPB) {
        unsigned t = gidx;
        const int w0 = (int)(t % W4) * 4; t /= W4;
        const int ph = (int)(t % (unsigned)y.h); t /= (unsigned)y.h;
        const int pdd = (int)(t % (unsigned)y.d);
        const int pn = (int)(t / (unsigned)y.d);
        float acc[4][COUT];
#pragma unroll
        for (int vx = 0; vx < 4; ++vx)
#pragma unroll
            for (int c = 0; c < COUT; ++c) acc[vx][c] = ws[TAPS * COUT + c];
#pragma unroll 1
        for (int a = 0; a < KD; ++a) {        // rolled: a fully unrolled 27-tap body spills ~2 KB per lane
            const int id = pdd + a - KD / 2;
#pragma unroll 1
            for (int bb = 0; bb < 3; ++bb) {
                const int ih = ph + bb - 1;
                const bool rowok = id >= 0 && id < x.d && ih >= 0 && ih < x.h;
                float xr[6];
                const unsigned rowbase = (((unsigned)pn * (unsigned)x.d + (unsigned)id) * (unsigned)x.h + (unsigned)ih) * (unsigned)x.w;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int iw = w0 - 1 + k;
                    // branch-free: a clamped address and a select keep the six loads of a row in flight together
                    const bool ok = rowok && iw >= 0 && iw < x.w;
                    const float tt = fmaf(s, to_f(((const T*)x.p)[ok ? (i64)(rowbase + (unsigned)iw) * x.pitch : 0]), b);
                    xr[k] = ok ? (tt > 0.f ? tt : sl * tt) : 0.f;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float4* wr = (const float4*)(ws + ((a * 3 + bb) * 3 + c) * COUT);
#pragma unroll
                    for (int c4 = 0; c4 < COUT / 4; ++c4) {
                        const float4 wv = wr[c4];
#pragma unroll
                        for (int vx = 0; vx < 4; ++vx) {
                            const float xv = xr[vx + c];
                            acc[vx][c4 * 4 + 0] = fmaf(xv, wv.x, acc[vx][c4 * 4 + 0]);
                            acc[vx][c4 * 4 + 1] = fmaf(xv, wv.y, acc[vx][c4 * 4 + 1]);
                            acc[vx][c4 * 4 + 2] = fmaf(xv, wv.z, acc[vx][c4 * 4 + 2]);
                            acc[vx][c4 * 4 + 3] = fmaf(xv, wv.w, acc[vx][c4 * 4 + 3]);
                        }
                    }
                }
            }
        }
        const i64 v0 = (i64)((((unsigned)pn * (unsigned)y.d + pdd) * (unsigned)y.h + ph) * (unsigned)y.w + w0);
        constexpr int G = 16 / sizeof(T);
#pragma unroll
        for (int vx = 0; vx < 4; ++vx) {
            T* dst = (T*)y.p + (v0 + vx) * y.pitch + co0;
#pragma unroll
            for (int c = 0; c < COUT; c += G) {
                Pack<T, G> o;
#pragma unroll
                for (int j = 0; j < G; ++j) o.v[j] = from_f<T>(acc[vx][c + j]);
                *(Pack<T, G>*)(dst + c) = o;
            }
        }
    }
}

// weight gradient, 4 voxels per lane: wave q owns taps [q*TPW, ...); acc[t][co] += sum_vx x[vx + tap] * dy[vx][co]
template <typename T, int COUT, int KD, int TPW>
__global__ __launch_bounds__(64 * ((KD * 9 + TPW - 1) / TPW)) void k_conv_c1_wgrad4(DAct x, DXf xf, DAct dy, int co0, float* __restrict__ partial) {
    constexpr int TAPS = KD * 9;
    constexpr int G = 16 / sizeof(T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tap0 = wave * TPW;
    float acc[TPW][COUT];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[t][c] = 0.f;
    const float s = xf.scale ? xf.scale[0] : 1.f, b = xf.shift ? xf.shift[0] : 0.f, sl = xf.slope ? xf.slope[0] : 1.f;
    // 32-bit index arithmetic throughout (the host checks nvox < 2^31): the 64-bit divisions of a linear voxel index cost as
    // many VALU instructions per iteration as the 448 FMAs they fed
    const unsigned W4 = (unsigned)dy.w / 4u;
    const unsigned total = (unsigned)dy.n * (unsigned)dy.d * (unsigned)dy.h * W4;
    for (unsigned gidx = blockIdx.x * 64u + lane; gidx < total; gidx += gridDim.x * 64u) {
        unsigned tt_ = gidx;
        const int w0 = (int)(tt_ % W4) * 4; tt_ /= W4;
        const int ph = (int)(tt_ % (unsigned)dy.h); tt_ /= (unsigned)dy.h;
        const int pdd = (int)(tt_ % (unsigned)dy.d);
        const int pn = (int)(tt_ / (unsigned)dy.d);
        const i64 v0 = (i64)((((unsigned)pn * (unsigned)dy.d + pdd) * (unsigned)dy.h + ph) * (unsigned)dy.w + w0);
        float g[4][COUT];
#pragma unroll
        for (int vx = 0; vx < 4; ++vx) {
            const T* src = (const T*)dy.p + (v0 + vx) * dy.pitch + co0;
#pragma unroll
            for (int c = 0; c < COUT; c += G) {
                Pack<T, G> in = *(const Pack<T, G>*)(src + c);
#pragma unroll
                for (int j = 0; j < G; ++j) g[vx][c + j] = to_f(in.v[j]);
            }
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tap = tap0 + t;
            if (tap < TAPS) {
                const int a = tap / 9, bb = (tap / 3) % 3, c = tap % 3;
                const int id = pdd + a - KD / 2, ih = ph + bb - 1;
                const bool rowok = id >= 0 && id < x.d && ih >= 0 && ih < x.h;
                const unsigned rowbase = (((unsigned)pn * (unsigned)x.d + (unsigned)id) * (unsigned)x.h + (unsigned)ih) * (unsigned)x.w;
#pragma unroll
                for (int vx = 0; vx < 4; ++vx) {
                    const int iw = w0 + vx + c - 1;
                    const bool ok = rowok && iw >= 0 && iw < x.w;          // branch-free (loads stay in flight together)
                    const float q = fmaf(s, to_f(((const T*)x.p)[ok ? (i64)(rowbase + (unsigned)iw) * x.pitch : 0]), b);
                    const float xv = ok ? (q > 0.f ? q : sl * q) : 0.f;
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[t][co] = fmaf(xv, g[vx][co], acc[t][co]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = tap0 + t;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float r = wave_sum(acc[t][co]);
            if (lane == 0 && tap < TAPS) partial[(i64)blockIdx.x * (TAPS * COUT) + tap * COUT + co] = r;
        }
    }
}

#define C1_MAX_BLOCKS 2048

static bool c1_ok(const biu_act* x, const biu_act* y, int kd, int kh, int kw, int dil, int dtype) {
    if (x->c != 1 || dil != 1 || kh != 3 || kw != 3 || (kd != 1 && kd != 3)) return false;
    if (y->c % 8 != 0 || y->c < 8) return false;
    if (nvox(y) >= (1LL << 31) || nvox(x) >= (1LL << 31)) return false;          // 32-bit voxel indices in the 4-voxel kernels
    const size_t es = dsize(dtype);
    return ((uintptr_t)y->p % 16) == 0 && ((size_t)y->pitch * es) % 16 == 0;
}
bool biu_c1_conv_ok(const biu_act* x, const biu_act* y, int kd, int kh, int kw, int dil, int dtype) { return c1_ok(x, y, kd, kh, kw, dil, dtype); }

template <typename T, int KD>
static int c1_fwd_t(const biu_act* x, const biu_xform* xf, const float* w, const float* bias, const biu_act* y, hipStream_t st) {
    const int grid = grid_for(nvox(y), TPB, 16384);
    int co0 = 0;
    while (co0 < y->c) {
        const int rem = y->c - co0;
        if (rem >= 32 && y->w % 4 != 0) { hipLaunchKernelGGL((k_conv_c1_fwd<T, 32, KD>), dim3(grid), dim3(TPB), 0, st, dact(x), dxf(xf), w, bias, dact(y), co0); co0 += 32; }
        else if (rem >= 16) {
            if (y->w % 4 == 0) hipLaunchKernelGGL((k_conv_c1_fwd4<T, 16, KD>), dim3(grid_for(nvox(y) / 4, TPB, 16384)), dim3(TPB), 0, st, dact(x), dxf(xf), w, bias, dact(y), co0);
            else hipLaunchKernelGGL((k_conv_c1_fwd<T, 16, KD>), dim3(grid), dim3(TPB), 0, st, dact(x), dxf(xf), w, bias, dact(y), co0);
            co0 += 16;
        } else {
            if (y->w % 4 == 0) hipLaunchKernelGGL((k_conv_c1_fwd4<T, 8, KD>), dim3(grid_for(nvox(y) / 4, TPB, 16384)), dim3(TPB), 0, st, dact(x), dxf(xf), w, bias, dact(y), co0);
            else hipLaunchKernelGGL((k_conv_c1_fwd<T, 8, KD>), dim3(grid), dim3(TPB), 0, st, dact(x), dxf(xf), w, bias, dact(y), co0);
            co0 += 8;
        }
    }
    BIU_CHECK_LAUNCH("conv_c1_fwd");
    return BIU_OK;
}
int biu_c1_conv_fwd(const biu_act* x, const biu_xform* xf, const float* w, const float* bias, int kd, const biu_act* y, int dtype, hipStream_t st) {
    if (dtype == BIU_BF16) return kd == 3 ? c1_fwd_t<bf16_t, 3>(x, xf, w, bias, y, st) : c1_fwd_t<bf16_t, 1>(x, xf, w, bias, y, st);
    return kd == 3 ? c1_fwd_t<float, 3>(x, xf, w, bias, y, st) : c1_fwd_t<float, 1>(x, xf, w, bias, y, st);
}

size_t biu_c1_wgrad_workspace(int cout, int kd) { return (size_t)C1_MAX_BLOCKS * kd * 9 * 32 * sizeof(float); }

template <typename T, int KD>
static int c1_wgrad_t(const biu_act* x, const biu_xform* xf, const biu_act* dy, float* dw, void* ws, hipStream_t st) {
    constexpr int TAPS = KD * 9;
    const i64 total = nvox(dy);
    int nblk = (int)((total + 64 * 64 - 1) / (64 * 64));          // >= 64 voxels per lane
    if (nblk > C1_MAX_BLOCKS) nblk = C1_MAX_BLOCKS;
    if (nblk < 1) nblk = 1;
    int co0 = 0;
    while (co0 < dy->c) {
        const int rem = dy->c - co0;
        int chunk;
        if (rem >= 32 && dy->w % 4 != 0) {
            constexpr int TPW = 3; chunk = 32;
            hipLaunchKernelGGL((k_conv_c1_wgrad<T, 32, KD, TPW>), dim3(nblk), dim3(64 * ((TAPS + TPW - 1) / TPW)), 0, st, dact(x), dxf(xf), dact(dy), co0, (float*)ws);
        } else if (rem >= 16) {
            constexpr int TPW = 7; chunk = 16;
            static const bool tpw4 = [] { const char* e = getenv("BIU_C1_TPW"); return e && e[0] == '4'; }();       // experiment switch
            if (dy->w % 4 == 0 && tpw4) hipLaunchKernelGGL((k_conv_c1_wgrad4<T, 16, KD, 4>), dim3(nblk), dim3(64 * ((TAPS + 3) / 4)), 0, st, dact(x), dxf(xf), dact(dy), co0, (float*)ws);
            else if (dy->w % 4 == 0) hipLaunchKernelGGL((k_conv_c1_wgrad4<T, 16, KD, TPW>), dim3(nblk), dim3(64 * ((TAPS + TPW - 1) / TPW)), 0, st, dact(x), dxf(xf), dact(dy), co0, (float*)ws);
            else hipLaunchKernelGGL((k_conv_c1_wgrad<T, 16, KD, TPW>), dim3(nblk), dim3(64 * ((TAPS + TPW - 1) / TPW)), 0, st, dact(x), dxf(xf), dact(dy), co0, (float*)ws);
        } else {
            constexpr int TPW = 9; chunk = 8;
            if (dy->w % 4 == 0) hipLaunchKernelGGL((k_conv_c1_wgrad4<T, 8, KD, TPW>), dim3(nblk), dim3(64 * ((TAPS + TPW - 1) / TPW)), 0, st, dact(x), dxf(xf), dact(dy), co0, (float*)ws);
            else hipLaunchKernelGGL((k_conv_c1_wgrad<T, 8, KD, TPW>), dim3(nblk), dim3(64 * ((TAPS + TPW - 1) / TPW)), 0, st, dact(x), dxf(xf), dact(dy), co0, (float*)ws);
        }
        BIU_CHECK_LAUNCH("conv_c1_wgrad");
        hipLaunchKernelGGL(k_c1_wgrad_finalize, dim3(TAPS * chunk), dim3(TPB), 0, st, (const float*)ws, nblk, TAPS, chunk, co0, dw);
        BIU_CHECK_LAUNCH("conv_c1_wgrad_finalize");
        co0 += chunk;
    }
    return BIU_OK;
}
int biu_c1_conv_wgrad(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, float* dw, void* ws, size_t ws_bytes, int dtype, hipStream_t st) {
    BIU_REQUIRE(ws && ws_bytes >= biu_c1_wgrad_workspace(dy->c, kd), BIU_ERR_WORKSPACE, "conv_c1_wgrad: workspace too small");
    if (dtype == BIU_BF16) return kd == 3 ? c1_wgrad_t<bf16_t, 3>(x, xf, dy, dw, ws, st) : c1_wgrad_t<bf16_t, 1>(x, xf, dy, dw, ws, st);
    return kd == 3 ? c1_wgrad_t<float, 3>(x, xf, dy, dw, ws, st) : c1_wgrad_t<float, 1>(x, xf, dy, dw, ws, st);
}

// =====================================================================================================================
// vectorised per-channel two-term reductions:  partial[blk][c][2]
// thread = (piece of PE channels) x (voxel row); 16-byte loads; LDS tree over the rows of the block
// =====================================================================================================================
template <typename T> struct VStats {        // (sum y, sum y^2)
    DAct y;
    static constexpr int PE = 16 / sizeof(T);
    __device__ __forceinline__ void operator()(i64 v, int c0, float* a0, float* a1) const {
        Pack<T, PE> in = *(const Pack<T, PE>*)((const T*)y.p + v * y.pitch + c0);
#pragma unroll
        for (int j = 0; j < PE; ++j) {
            const float t = to_f(in.v[j]);
            a0[j] += t;
            a1[j] = fmaf(t, t, a1[j]);
        }
    }
};
template <typename T> struct VBnBwd {        // (sum dz, sum dz * yhat)
    DAct da, y;
    const float *scale, *shift, *slope, *mean, *invstd;
    static constexpr int PE = 16 / sizeof(T);
    __device__ __forceinline__ void operator()(i64 v, int c0, float* a0, float* a1) const {
        Pack<T, PE> g = *(const Pack<T, PE>*)((const T*)da.p + v * da.pitch + c0);
        Pack<T, PE> yy = *(const Pack<T, PE>*)((const T*)y.p + v * y.pitch + c0);
#pragma unroll
        for (int j = 0; j < PE; ++j) {
            const int c = c0 + j;
            const float yv = to_f(yy.v[j]);
            const float t = fmaf(scale[c], yv, shift[c]);
            const float dz = to_f(g.v[j]) * (t > 0.f ? 1.f : (slope ? slope[c] : 1.f));
            a0[j] += dz;
            a1[j] = fmaf(dz, (yv - mean[c]) * invstd[c], a1[j]);
        }
    }
};
template <typename T> struct VSum {          // (sum a, 0)
    DAct a;
    static constexpr int PE = 16 / sizeof(T);
    __device__ __forceinline__ void operator()(i64 v, int c0, float* a0, float* a1) const {
        Pack<T, PE> in = *(const Pack<T, PE>*)((const T*)a.p + v * a.pitch + c0);
#pragma unroll
        for (int j = 0; j < PE; ++j) a0[j] += to_f(in.v[j]);
    }
};

template <typename F, int PE>
__global__ __launch_bounds__(TPB) void k_chan_reduce2_vec(F f, i64 total_vox, int C, int ppv /* pieces per voxel, pow2-padded */,
                                                          i64 vox_per_block, float* __restrict__ partial) {
    extern __shared__ float sm[];                 // [rows][ppv*PE][2]
    const int rows = TPB / ppv;
    const int piece = threadIdx.x % ppv, row = threadIdx.x / ppv;
    const int c0 = piece * PE;
    const i64 v0 = (i64)blockIdx.x * vox_per_block;
    i64 v1 = v0 + vox_per_block;
    if (v1 > total_vox) v1 = total_vox;
    float a0[PE], a1[PE];
#pragma unroll
    for (int j = 0; j < PE; ++j) a0[j] = a1[j] = 0.f;
    if (c0 < C)
        for (i64 v = v0 + row; v < v1; v += rows) f(v, c0, a0, a1);
    const int W = ppv * PE;
#pragma unroll
    for (int j = 0; j < PE; ++j) {
        sm[(row * W + c0 + j) * 2 + 0] = a0[j];
        sm[(row * W + c0 + j) * 2 + 1] = a1[j];
    }
    __syncthreads();
    for (int r = rows >> 1; r > 0; r >>= 1) {
        if (row < r) {
#pragma unroll
            for (int j = 0; j < PE; ++j) {
                sm[(row * W + c0 + j) * 2 + 0] += sm[((row + r) * W + c0 + j) * 2 + 0];
                sm[(row * W + c0 + j) * 2 + 1] += sm[((row + r) * W + c0 + j) * 2 + 1];
            }
        }
        __syncthreads();
    }
    if (row == 0 && c0 < C) {
#pragma unroll
        for (int j = 0; j < PE; ++j) {
            partial[((i64)blockIdx.x * C + c0 + j) * 2 + 0] = sm[(c0 + j) * 2 + 0];
            partial[((i64)blockIdx.x * C + c0 + j) * 2 + 1] = sm[(c0 + j) * 2 + 1];
        }
    }
}

struct VPlan { int ppv, nblk; i64 vpb; size_t smem; };
static VPlan vplan(i64 total_vox, int C, int PE) {
    VPlan p;
    const int pieces = C / PE;
    p.ppv = 1;
    while (p.ppv < pieces) p.ppv <<= 1;
    const int rows = TPB / p.ppv;
    i64 want = (total_vox + (i64)rows * 16 - 1) / ((i64)rows * 16);           // >= 16 voxels per thread
    if (want < 1) want = 1;
    if (want > BIU_BN_MAX_PARTIALS) want = BIU_BN_MAX_PARTIALS;
    p.vpb = (total_vox + want - 1) / want;
    p.nblk = (int)((total_vox + p.vpb - 1) / p.vpb);
    p.smem = (size_t)rows * p.ppv * PE * 2 * sizeof(float);
    return p;
}
bool biu_vec_reduce_ok(const biu_act* a, int dtype) {
    const int pe = 16 / (int)dsize(dtype);
    return a->c % pe == 0 && a->c / pe <= TPB && vec_ok(a, pe, dtype);
}

int biu_bn_stats_vec(const biu_act* y, float* partial, int* nblk_out, int dtype, hipStream_t st) {
    BIU_DISPATCH_DTYPE(dtype, {
        constexpr int PE = 16 / sizeof(T);
        VPlan p = vplan(nvox(y), y->c, PE);
        VStats<T> f{dact(y)};
        hipLaunchKernelGGL((k_chan_reduce2_vec<VStats<T>, PE>), dim3(p.nblk), dim3(TPB), p.smem, st, f, nvox(y), y->c, p.ppv, p.vpb, partial);
        *nblk_out = p.nblk;
    });
    BIU_CHECK_LAUNCH("bn_stats_vec");
    return BIU_OK;
}
int biu_bn_bwd_reduce_vec(const biu_act* da, const biu_act* y, const float* scale, const float* shift, const float* slope,
                          const float* mean, const float* invstd, float* partial, int* nblk_out, int dtype, hipStream_t st) {
    BIU_DISPATCH_DTYPE(dtype, {
        constexpr int PE = 16 / sizeof(T);
        VPlan p = vplan(nvox(y), y->c, PE);
        VBnBwd<T> f{dact(da), dact(y), scale, shift, slope, mean, invstd};
        hipLaunchKernelGGL((k_chan_reduce2_vec<VBnBwd<T>, PE>), dim3(p.nblk), dim3(TPB), p.smem, st, f, nvox(y), y->c, p.ppv, p.vpb, partial);
        *nblk_out = p.nblk;
    });
    BIU_CHECK_LAUNCH("bn_bwd_reduce_vec");
    return BIU_OK;
}

// out[c] = sum_b partial[b][c][0]  (fp64 merge)
__global__ void k_partial_sum0(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out) {
    __shared__ double red[TPB];
    const int c = blockIdx.x;
    double a = 0;
    for (int b = threadIdx.x; b < nblk; b += TPB) a += partial[((i64)b * C + c) * 2];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = TPB / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = (float)red[0];
}
size_t biu_chan_sum_workspace(int c) { return (size_t)BIU_BN_MAX_PARTIALS * c * 2 * sizeof(float); }
// out[c] = sum_v a[v, c]; ws >= biu_chan_sum_workspace(c)
int biu_chan_sum_vec(const biu_act* a, float* out, void* ws, int dtype, hipStream_t st) {
    int nblk = 0;
    BIU_DISPATCH_DTYPE(dtype, {
        constexpr int PE = 16 / sizeof(T);
        VPlan p = vplan(nvox(a), a->c, PE);
        VSum<T> f{dact(a)};
        hipLaunchKernelGGL((k_chan_reduce2_vec<VSum<T>, PE>), dim3(p.nblk), dim3(TPB), p.smem, st, f, nvox(a), a->c, p.ppv, p.vpb, (float*)ws);
        nblk = p.nblk;
    });
    BIU_CHECK_LAUNCH("chan_sum_vec");
    hipLaunchKernelGGL(k_partial_sum0, dim3(a->c), dim3(TPB), 0, st, (const float*)ws, nblk, a->c, out);
    BIU_CHECK_LAUNCH("partial_sum0");
    return BIU_OK;
}

// =====================================================================================================================
// fused 1x1 head backward: one pass over (x, dlogits) produces dx, and per-block partials of dW and dbias
//   dx[v, c] = sum_o dl[n, o, s] * w[o, c];  dW[o, c] = sum_v dl * T(x)[v, c];  db[o] = sum_v dl
// =====================================================================================================================
// RED: also the BatchNorm-backward sums (sum dz, sum dz * yhat) of the block that produced x, reduced from the dx values this
//      kernel writes (the head is that tensor's only reader): bn_partial[block][C][2]
template <typename T, int C, int O, bool RED>
__global__ __launch_bounds__(TPB) void k_head_bwd_fused(DAct x, DXf xf, const float* __restrict__ w, int cout,
                                                        const float* __restrict__ dl, DAct dx, int want_dx,
                                                        float* __restrict__ partial /* [nblk][O*C + O] */,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        float* __restrict__ bn_partial) {
    constexpr int G = 16 / sizeof(T);
    __shared__ float wsm[O * C];
    __shared__ float xsc[C], xsh[C], xsl[C];
    __shared__ float red[4][O * C + O];
    for (int i = threadIdx.x; i < O * C; i += TPB) wsm[i] = (i / C) < cout ? w[i] : 0.f;
    for (int i = threadIdx.x; i < C; i += TPB) {
        xsc[i] = xf.scale ? xf.scale[i] : 1.f;
        xsh[i] = xf.shift ? xf.shift[i] : 0.f;
        xsl[i] = xf.slope ? xf.slope[i] : 1.f;
    }
    __syncthreads();
    const i64 S = (i64)x.d * x.h * x.w;
    const i64 total = (i64)x.n * S;
    float aw[O][C], ab[O];
    float r1[RED ? C : 1], r2[RED ? C : 1];
#pragma unroll
    for (int o = 0; o < O; ++o) {
        ab[o] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) aw[o][c] = 0.f;
    }
    if (RED) {
#pragma unroll
        for (int c = 0; c < C; ++c) r1[c] = r2[c] = 0.f;
    }
    for (i64 v = (i64)blockIdx.x * TPB + threadIdx.x; v < total; v += (i64)gridDim.x * TPB) {
        const i64 n = v / S, s = v % S;
        float g[O];
#pragma unroll
        for (int o = 0; o < O; ++o) {
            g[o] = (o < cout) ? dl[(n * cout + o) * S + s] : 0.f;
            ab[o] += g[o];
        }
        const T* src = (const T*)x.p + v * x.pitch;
        T* dst = want_dx ? (T*)dx.p + v * dx.pitch : nullptr;
#pragma unroll
        for (int c0 = 0; c0 < C; c0 += G) {
            Pack<T, G> in = *(const Pack<T, G>*)(src + c0);
            Pack<T, G> od;
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int c = c0 + j;
                float t = fmaf(xsc[c], to_f(in.v[j]), xsh[c]);
                t = t > 0.f ? t : xsl[c] * t;
                float d = 0.f;
#pragma unroll
                for (int o = 0; o < O; ++o) {
                    aw[o][c] = fmaf(g[o], t, aw[o][c]);
                    d = fmaf(g[o], wsm[o * C + c], d);
                }
                od.v[j] = from_f<T>(d);
                if (RED) {
                    const float yv = to_f(in.v[j]);
                    const float dz = to_f(od.v[j]) * (fmaf(xsc[c], yv, xsh[c]) > 0.f ? 1.f : xsl[c]);
                    r1[c] += dz;
                    r2[c] = fmaf(dz, yv, r2[c]);          // raw; centred below
                }
            }
            if (want_dx) *(Pack<T, G>*)(dst + c0) = od;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 0; o < O; ++o) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float r = wave_sum(aw[o][c]);
            if (lane == 0) red[wave][o * C + c] = r;
        }
        const float rb = wave_sum(ab[o]);
        if (lane == 0) red[wave][O * C + o] = rb;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < O * C + O; i += TPB)
        partial[(i64)blockIdx.x * (O * C + O) + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (RED) {
        __shared__ float redb[4][2 * C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float a1 = wave_sum(r1[c]), a2 = wave_sum(r2[c]);
            if (lane == 0) { redb[wave][2 * c] = a1; redb[wave][2 * c + 1] = a2; }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += TPB) {
            const float s1v = redb[0][2 * c] + redb[1][2 * c] + redb[2][2 * c] + redb[3][2 * c];
            const float s2v = redb[0][2 * c + 1] + redb[1][2 * c + 1] + redb[2][2 * c + 1] + redb[3][2 * c + 1];
            bn_partial[((i64)blockIdx.x * C + c) * 2 + 0] = s1v;
            bn_partial[((i64)blockIdx.x * C + c) * 2 + 1] = invstd[c] * (s2v - mean[c] * s1v);
        }
    }
}

__global__ void k_head_bwd_finalize(const float* __restrict__ partial, int nblk, int C, int O, int cout, float* dw, float* db) {
    __shared__ double red[TPB];
    const int i = blockIdx.x;              // 0 .. O*C + O
    double a = 0;
    for (int b = threadIdx.x; b < nblk; b += TPB) a += partial[(i64)b * (O * C + O) + i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = TPB / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (i < O * C) {
            const int o = i / C, c = i % C;
            if (o < cout && dw) dw[o * C + c] = (float)red[0];
        } else {
            const int o = i - O * C;
            if (o < cout && db) db[o] = (float)red[0];
        }
    }
}

#define HEAD_FUSED_BLOCKS 1024
bool biu_head_bwd_fused_ok(const biu_act* x, const biu_act* dx, int cout, int dtype) {
    const int c = x->c;
    if (!(c == 8 || c == 16 || c == 32 || c == 64)) return false;
    if (cout < 1 || cout > 4 || c * (cout == 3 ? 4 : cout) > 128) return false;
    const int g = 16 / (int)dsize(dtype);
    if (!vec_ok(x, g, dtype)) return false;
    if (dx && !vec_ok(dx, g, dtype)) return false;
    return true;
}
size_t biu_head_bwd_fused_workspace(int cin) { return (size_t)HEAD_FUSED_BLOCKS * (4 * cin + 4) * sizeof(float); }

template <typename T, int C>
static int head_fused_t(const biu_act* x, const biu_xform* xf, const float* w, int cout, const float* dl, const biu_act* dx, float* dw,
                        float* db, void* ws, hipStream_t st, const float* mean = nullptr, const float* invstd = nullptr,
                        float* bn_partial = nullptr, int* bn_nblk = nullptr) {
    const i64 total = nvox(x);
    int nblk = (int)((total + TPB * 8 - 1) / (TPB * 8));
    if (nblk > HEAD_FUSED_BLOCKS) nblk = HEAD_FUSED_BLOCKS;
    if (nblk < 1) nblk = 1;
    DAct dxa = dx ? dact(dx) : dact(x);
    const int O = cout == 3 ? 4 : cout;
    if (bn_partial) {
        if constexpr (C <= 32) {
            if (O == 2) hipLaunchKernelGGL((k_head_bwd_fused<T, C, 2, true>), dim3(nblk), dim3(TPB), 0, st, dact(x), dxf(xf), w, cout, dl, dxa, 1, (float*)ws, mean, invstd, bn_partial);
            else hipLaunchKernelGGL((k_head_bwd_fused<T, C, 1, true>), dim3(nblk), dim3(TPB), 0, st, dact(x), dxf(xf), w, cout, dl, dxa, 1, (float*)ws, mean, invstd, bn_partial);
            *bn_nblk = nblk;
        }
    } else {
        if constexpr (C <= 32) {
            if (O == 4) hipLaunchKernelGGL((k_head_bwd_fused<T, C, 4, false>), dim3(nblk), dim3(TPB), 0, st, dact(x), dxf(xf), w, cout, dl, dxa, dx ? 1 : 0, (float*)ws, nullptr, nullptr, nullptr);
        }
        if (O == 2) hipLaunchKernelGGL((k_head_bwd_fused<T, C, 2, false>), dim3(nblk), dim3(TPB), 0, st, dact(x), dxf(xf), w, cout, dl, dxa, dx ? 1 : 0, (float*)ws, nullptr, nullptr, nullptr);
        if (O == 1) hipLaunchKernelGGL((k_head_bwd_fused<T, C, 1, false>), dim3(nblk), dim3(TPB), 0, st, dact(x), dxf(xf), w, cout, dl, dxa, dx ? 1 : 0, (float*)ws, nullptr, nullptr, nullptr);
    }
    BIU_CHECK_LAUNCH("head_bwd_fused");
    hipLaunchKernelGGL(k_head_bwd_finalize, dim3(O * C + O), dim3(TPB), 0, st, (const float*)ws, nblk, C, O, cout, dw, db);
    BIU_CHECK_LAUNCH("head_bwd_finalize");
    return BIU_OK;
}
// fused head backward + BatchNorm-backward sums of x's producer: C <= 32, cout <= 2, dx required; partial rows <= 1024
bool biu_head_bwd_bnred_ok(const biu_act* x, const biu_act* dx, int cout, int dtype) {
    return dx && x->c <= 32 && cout <= 2 && biu_head_bwd_fused_ok(x, dx, cout, dtype);
}
int biu_head_bwd_bnred_fused(const biu_act* x, const biu_xform* xf, const float* w, int cout, const float* dl, const biu_act* dx, float* dw,
                             float* db, void* ws, const float* mean, const float* invstd, float* bn_partial, int* bn_nblk, int dtype,
                             hipStream_t st) {
    BIU_DISPATCH_DTYPE(dtype, {
        switch (x->c) {
            case 8: return head_fused_t<T, 8>(x, xf, w, cout, dl, dx, dw, db, ws, st, mean, invstd, bn_partial, bn_nblk);
            case 16: return head_fused_t<T, 16>(x, xf, w, cout, dl, dx, dw, db, ws, st, mean, invstd, bn_partial, bn_nblk);
            default: return head_fused_t<T, 32>(x, xf, w, cout, dl, dx, dw, db, ws, st, mean, invstd, bn_partial, bn_nblk);
        }
    });
    return BIU_OK;
}
int biu_head_bwd_fused(const biu_act* x, const biu_xform* xf, const float* w, int cout, const float* dl, const biu_act* dx, float* dw,
                       float* db, void* ws, size_t ws_bytes, int dtype, hipStream_t st) {
    BIU_REQUIRE(ws && ws_bytes >= biu_head_bwd_fused_workspace(x->c), BIU_ERR_WORKSPACE, "head_bwd: workspace too small");
    BIU_DISPATCH_DTYPE(dtype, {
        switch (x->c) {
            case 8: return head_fused_t<T, 8>(x, xf, w, cout, dl, dx, dw, db, ws, st);
            case 16: return head_fused_t<T, 16>(x, xf, w, cout, dl, dx, dw, db, ws, st);
            case 32: return head_fused_t<T, 32>(x, xf, w, cout, dl, dx, dw, db, ws, st);
            default: return head_fused_t<T, 64>(x, xf, w, cout, dl, dx, dw, db, ws, st);
        }
    });
    return BIU_OK;
}

// =====================================================================================================================
// element-wise passes with a FIXED channel group per thread: the per-channel vectors (BatchNorm coefficients, producer
// transform) are loaded once into registers instead of once per element; every access is a 16-byte vector.
// thread t -> (channel group g = t % cg, voxel row = t / cg); rows stride over the voxels.
// =====================================================================================================================
struct RowPlan { int cg, rows, grid; };
static RowPlan row_plan(i64 total_vox, int C, int PE) {
    RowPlan p;
    p.cg = C / PE;
    p.rows = TPB / p.cg;
    i64 want = (total_vox + (i64)p.rows * 8 - 1) / ((i64)p.rows * 8);        // ~8 voxels per thread
    if (want < 1) want = 1;
    if (want > 4096) want = 4096;
    p.grid = (int)want;
    return p;
}
bool biu_rowvec_ok(const biu_act* a, int dtype) {
    const int pe = 16 / (int)dsize(dtype);
    return a->c % pe == 0 && a->c / pe <= TPB && vec_ok(a, pe, dtype);
}

template <typename T>
__global__ __launch_bounds__(TPB) void k_bn_bwd_apply_rv(DAct da, DAct y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ slope, const float* __restrict__ A,
                                                         const float* __restrict__ B, const float* __restrict__ Cc, DAct dy, int cg, int rows) {
    constexpr int PE = 16 / sizeof(T);
    const int g = threadIdx.x % cg, row = threadIdx.x / cg;
    if (row >= rows) return;
    const int c0 = g * PE;
    float sc[PE], sh[PE], sl[PE], ka[PE], kb[PE], kc[PE];
#pragma unroll
    for (int j = 0; j < PE; ++j) {
        sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; sl[j] = slope ? slope[c0 + j] : 1.f;
        ka[j] = A[c0 + j]; kb[j] = B[c0 + j]; kc[j] = Cc[c0 + j];
    }
    const i64 total = (i64)y.n * y.d * y.h * y.w;
    for (i64 v = (i64)blockIdx.x * rows + row; v < total; v += (i64)gridDim.x * rows) {
        Pack<T, PE> gq = *(const Pack<T, PE>*)((const T*)da.p + v * da.pitch + c0);
        Pack<T, PE> yy = *(const Pack<T, PE>*)((const T*)y.p + v * y.pitch + c0);
        Pack<T, PE> o;
#pragma unroll
        for (int j = 0; j < PE; ++j) {
            const float yv = to_f(yy.v[j]);
            const float t = fmaf(sc[j], yv, sh[j]);
            const float dz = to_f(gq.v[j]) * (t > 0.f ? 1.f : sl[j]);
            o.v[j] = from_f<T>(fmaf(ka[j], dz, fmaf(kb[j], yv, kc[j])));
        }
        *(Pack<T, PE>*)((T*)dy.p + v * dy.pitch + c0) = o;
    }
}
int biu_bn_bwd_apply_rv(const biu_act* da, const biu_act* y, const float* scale, const float* shift, const float* slope,
                        const float* A, const float* B, const float* Cc, const biu_act* dy, int dtype, hipStream_t st) {
    BIU_DISPATCH_DTYPE(dtype, {
        constexpr int PE = 16 / sizeof(T);
        RowPlan p = row_plan(nvox(y), y->c, PE);
        hipLaunchKernelGGL(k_bn_bwd_apply_rv<T>, dim3(p.grid), dim3(TPB), 0, st, dact(da), dact(y), scale, shift, slope, A, B, Cc, dact(dy), p.cg, p.rows);
    });
    BIU_CHECK_LAUNCH("bn_bwd_apply_rv");
    return BIU_OK;
}

template <typename T, int MODE>   // MODE 0: out = T(x) ; MODE 1: max-pool fwd ; MODE 2: max-pool bwd
__global__ __launch_bounds__(TPB) void k_pool_rv(DAct x, DXf xf, DAct small, DAct dx, int pd, int accumulate, int cg, int rows) {
    constexpr int PE = 16 / sizeof(T);
    const int g = threadIdx.x % cg, row = threadIdx.x / cg;
    if (row >= rows) return;
    const int c0 = g * PE;
    float sc[PE], sh[PE], sl[PE];
#pragma unroll
    for (int j = 0; j < PE; ++j) {
        sc[j] = xf.scale ? xf.scale[c0 + j] : 1.f;
        sh[j] = xf.shift ? xf.shift[c0 + j] : 0.f;
        sl[j] = xf.slope ? xf.slope[c0 + j] : 1.f;
    }
    const i64 total = (i64)small.n * small.d * small.h * small.w;
    for (i64 ov = (i64)blockIdx.x * rows + row; ov < total; ov += (i64)gridDim.x * rows) {
        if (MODE == 0) {
            Pack<T, PE> in = *(const Pack<T, PE>*)((const T*)x.p + ov * x.pitch + c0);
            Pack<T, PE> o;
#pragma unroll
            for (int j = 0; j < PE; ++j) {
                const float t = fmaf(sc[j], to_f(in.v[j]), sh[j]);
                o.v[j] = from_f<T>(t > 0.f ? t : sl[j] * t);
            }
            *(Pack<T, PE>*)((T*)small.p + ov * small.pitch + c0) = o;
            continue;
        }
        const Vox4 p = unvox4(ov, small.d, small.h, small.w);
        float best[PE];
        int arg[PE];
#pragma unroll
        for (int j = 0; j < PE; ++j) { best[j] = -INFINITY; arg[j] = 0; }
        for (int a = 0; a < pd; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const i64 iv = (((i64)p.n * x.d + p.d * pd + a) * x.h + p.h * 2 + b) * x.w + p.w * 2 + c;
                    Pack<T, PE> in = *(const Pack<T, PE>*)((const T*)x.p + iv * x.pitch + c0);
                    const int code = (a * 2 + b) * 2 + c;
#pragma unroll
                    for (int j = 0; j < PE; ++j) {
                        float t = fmaf(sc[j], to_f(in.v[j]), sh[j]);
                        t = t > 0.f ? t : sl[j] * t;
                        if (t > best[j] || t != t) { best[j] = t; arg[j] = code; }
                    }
                }
        if (MODE == 1) {
            Pack<T, PE> o;
#pragma unroll
            for (int j = 0; j < PE; ++j) o.v[j] = from_f<T>(best[j]);
            *(Pack<T, PE>*)((T*)small.p + ov * small.pitch + c0) = o;
        } else {
            Pack<T, PE> gq = *(const Pack<T, PE>*)((const T*)small.p + ov * small.pitch + c0);     // small = dout here
            for (int a = 0; a < pd; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const i64 iv = (((i64)p.n * x.d + p.d * pd + a) * x.h + p.h * 2 + b) * x.w + p.w * 2 + c;
                        const int code = (a * 2 + b) * 2 + c;
                        T* dst = (T*)dx.p + iv * dx.pitch + c0;
                        Pack<T, PE> o;
                        if (accumulate) o = *(Pack<T, PE>*)dst;
#pragma unroll
                        for (int j = 0; j < PE; ++j) {
                            const float r = (arg[j] == code) ? to_f(gq.v[j]) : 0.f;
                            o.v[j] = from_f<T>(accumulate ? to_f(o.v[j]) + r : r);
                        }
                        *(Pack<T, PE>*)dst = o;
                    }
        }
    }
}
// Max-pool forward, software-pipelined like k_maxpool_bwd_pair: the 8 (4) loads of the NEXT output voxel are in flight while this one is
// reduced and stored.  Same arithmetic and NaN rule as k_pool_rv<T, 1>.
template <typename T>
__global__ __launch_bounds__(TPB, 2) void k_maxpool_fwd_pipe(DAct x, DXf xf, DAct out, int pd, int cg, int rows) {
    constexpr int PE = 16 / sizeof(T);
    const int g = threadIdx.x % cg, row = threadIdx.x / cg;
    if (row >= rows) return;
    const int c0 = g * PE;
    float sc[PE], sh[PE], sl[PE];
#pragma unroll
    for (int j = 0; j < PE; ++j) {
        sc[j] = xf.scale ? xf.scale[c0 + j] : 1.f;
        sh[j] = xf.shift ? xf.shift[c0 + j] : 0.f;
        sl[j] = xf.slope ? xf.slope[c0 + j] : 1.f;
    }
    const i64 total = (i64)out.n * out.d * out.h * out.w;
    const i64 step_b = x.w, step_a = (i64)x.h * x.w;
    struct Win { Pack<T, PE> in[8]; };
    auto issue = [&](i64 ov, Win& w) {
        const Vox4 p = unvox4(ov, out.d, out.h, out.w);
        const i64 base = (((i64)p.n * x.d + p.d * pd) * x.h + p.h * 2) * x.w + p.w * 2;
        const T* xp = (const T*)x.p + base * x.pitch + c0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((k >> 2) < pd) w.in[k] = *(const Pack<T, PE>*)(xp + ((k >> 2) * step_a + ((k >> 1) & 1) * step_b + (k & 1)) * x.pitch);
    };
    auto finish = [&](i64 ov, const Win& w) {
        Pack<T, PE> o;
#pragma unroll
        for (int j = 0; j < PE; ++j) {
            float best = -INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((k >> 2) >= pd) continue;
                float t = fmaf(sc[j], to_f(w.in[k].v[j]), sh[j]);
                t = t > 0.f ? t : sl[j] * t;
                if (t > best || t != t) best = t;
            }
            o.v[j] = from_f<T>(best);
        }
        *(Pack<T, PE>*)((T*)out.p + ov * out.pitch + c0) = o;
    };
    const i64 stride = (i64)gridDim.x * rows;
    i64 ov = (i64)blockIdx.x * rows + row;
    if (ov < total) {
        Win cur, nxt;
        issue(ov, cur);
        while (true) {
            const i64 nov = ov + stride;
            const bool more = nov < total;
            if (more) issue(nov, nxt);
            finish(ov, cur);
            if (!more) break;
            cur = nxt;
            ov = nov;
        }
    }
}

// Max-pool backward with every load of a window in flight together (x, and dx when accumulating), and -- optionally -- the
// BatchNorm-backward sums of the layer that produced x reduced from the finished gradient in the same pass:
//   (sum dz, sum dz * yhat),  dz = dx_final * T'(scale*x + shift),  yhat = (x - mean) * invstd      -> partial[block][C][2]
template <typename T, bool RED, int PE>       // PE channels per thread: 4 keeps the fully unrolled window inside 128 registers
__global__ __launch_bounds__(TPB, RED ? 3 : 4) void k_maxpool_bwd_rv(DAct x, DXf xf, DAct dout, DAct dx, int pd, int accumulate, int cg, int rows,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           float* __restrict__ partial) {
    extern __shared__ float sm[];                 // RED: [rows][cg*PE][2]
    const int g = threadIdx.x % cg, row = threadIdx.x / cg;
    const int c0 = g * PE;
    const bool live = row < rows;
    float sc[PE], sh[PE], sl[PE], s1[PE], s2[PE];  // s2 accumulates sum dz * x raw; centred with (mean, invstd) at the end
#pragma unroll
    for (int j = 0; j < PE; ++j) {
        sc[j] = xf.scale ? xf.scale[c0 + j] : 1.f;
        sh[j] = xf.shift ? xf.shift[c0 + j] : 0.f;
        sl[j] = xf.slope ? xf.slope[c0 + j] : 1.f;
        s1[j] = s2[j] = 0.f;
    }
    const i64 total = (i64)dout.n * dout.d * dout.h * dout.w;
    const i64 step_b = x.w, step_a = (i64)x.h * x.w;     // voxel strides of the window's h / d offsets (x and dx share extents)
    if (live)
        for (i64 ov = (i64)blockIdx.x * rows + row; ov < total; ov += (i64)gridDim.x * rows) {
            const Vox4 p = unvox4(ov, dout.d, dout.h, dout.w);
            const i64 base = (((i64)p.n * x.d + p.d * pd) * x.h + p.h * 2) * x.w + p.w * 2;
            const T* xp = (const T*)x.p + base * x.pitch + c0;
            T* dp = (T*)dx.p + base * dx.pitch + c0;
            Pack<T, PE> in[8], old[8];
            const Pack<T, PE> gq = *(const Pack<T, PE>*)((const T*)dout.p + ov * dout.pitch + c0);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if ((k >> 2) < pd) in[k] = *(const Pack<T, PE>*)(xp + ((k >> 2) * step_a + ((k >> 1) & 1) * step_b + (k & 1)) * x.pitch);
            if (accumulate) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((k >> 2) < pd) old[k] = *(const Pack<T, PE>*)(dp + ((k >> 2) * step_a + ((k >> 1) & 1) * step_b + (k & 1)) * dx.pitch);
            }
            unsigned argp = 0;                     // 3-bit argmax per channel
#pragma unroll
            for (int j = 0; j < PE; ++j) {
                float best = -INFINITY;
                unsigned arg = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if ((k >> 2) >= pd) continue;
                    float t = fmaf(sc[j], to_f(in[k].v[j]), sh[j]);
                    t = t > 0.f ? t : sl[j] * t;
                    if (t > best || t != t) { best = t; arg = k; }
                }
                argp |= arg << (3 * j);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((k >> 2) >= pd) continue;
                Pack<T, PE> o;
#pragma unroll
                for (int j = 0; j < PE; ++j) {
                    const float r = (((argp >> (3 * j)) & 7u) == (unsigned)k) ? to_f(gq.v[j]) : 0.f;
                    o.v[j] = from_f<T>(accumulate ? to_f(old[k].v[j]) + r : r);
                    if (RED) {
                        const float yv = to_f(in[k].v[j]);
                        const float dz = to_f(o.v[j]) * (fmaf(sc[j], yv, sh[j]) > 0.f ? 1.f : sl[j]);
                        s1[j] += dz;
                        s2[j] = fmaf(dz, yv, s2[j]);
                    }
                }
                *(Pack<T, PE>*)(dp + ((k >> 2) * step_a + ((k >> 1) & 1) * step_b + (k & 1)) * dx.pitch) = o;
            }
        }
    if (RED) {
        const int C = cg * PE;
        if (live) {
#pragma unroll
            for (int j = 0; j < PE; ++j) {
                sm[(row * C + c0 + j) * 2 + 0] = s1[j];
                sm[(row * C + c0 + j) * 2 + 1] = s2[j];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += TPB) {
            float a0 = 0.f, a1 = 0.f;
            for (int r = 0; r < rows; ++r) { a0 += sm[(r * C + c) * 2]; a1 += sm[(r * C + c) * 2 + 1]; }
            partial[((i64)blockIdx.x * C + c) * 2 + 0] = a0;
            partial[((i64)blockIdx.x * C + c) * 2 + 1] = invstd[c] * (a1 - mean[c] * a0);        // sum dz * yhat
        }
    }
}

// The same pass with the window's two w-neighbours on two lanes: lane (c, g) owns fine voxels (.., 2w + c) and channels g*PE .. +PE-1, so the
// 2*cg lanes of a pooled voxel read / write the two voxels' channel rows as ONE contiguous run (128 B at 32 bf16 channels with 16-byte lanes;
// k_maxpool_bwd_rv's lanes touch 64-byte halves of two different lines per instruction) and hold 4 + 4 window pieces instead of 8 + 8.
// The argmax of a channel is settled between the two lanes with one exchange (lane ^ cg): larger value wins, equal values go to the
// smaller window index, a NaN wins over any number and the later NaN over the earlier -- exactly the scan order of the one-lane kernel.
template <typename T, bool RED, int PE>
__global__ __launch_bounds__(TPB, sizeof(T) * PE >= 16 ? 2 : 4) void k_maxpool_bwd_pair(DAct x, DXf xf, DAct dout, DAct dx, int pd, int accumulate, int cg, int rows,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             float* __restrict__ partial) {
    extern __shared__ float sm[];                 // RED: [rows * 2][cg*PE][2]
    const int q = threadIdx.x % (2 * cg), row = threadIdx.x / (2 * cg);
    const int g = q % cg, c = q / cg;
    const int c0 = g * PE;
    const bool live = row < rows;
    float sc[PE], sh[PE], sl[PE], s1[PE], s2[PE];
#pragma unroll
    for (int j = 0; j < PE; ++j) {
        sc[j] = xf.scale ? xf.scale[c0 + j] : 1.f;
        sh[j] = xf.shift ? xf.shift[c0 + j] : 0.f;
        sl[j] = xf.slope ? xf.slope[c0 + j] : 1.f;
        s1[j] = s2[j] = 0.f;
    }
    const i64 total = (i64)dout.n * dout.d * dout.h * dout.w;
    const i64 step_b = x.w, step_a = (i64)x.h * x.w;
    // software-pipelined: the 9 loads of the NEXT pooled voxel are in flight while this one is reduced and stored (a lane that issues,
    // waits, computes and stores in turn leaves the memory system idle for half of its time; 18 more registers buy the overlap)
    struct Win { Pack<T, PE> in[4], old[4], gq; const T* xp; T* dp; };
    auto issue = [&](i64 ov, Win& w) {
        const Vox4 p = unvox4(ov, dout.d, dout.h, dout.w);
        const i64 base = (((i64)p.n * x.d + p.d * pd) * x.h + p.h * 2) * x.w + p.w * 2 + c;
        w.xp = (const T*)x.p + base * x.pitch + c0;
        w.dp = (T*)dx.p + base * dx.pitch + c0;
        w.gq = *(const Pack<T, PE>*)((const T*)dout.p + ov * dout.pitch + c0);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2)
            if ((k2 >> 1) < pd) w.in[k2] = *(const Pack<T, PE>*)(w.xp + ((k2 >> 1) * step_a + (k2 & 1) * step_b) * x.pitch);
        if (accumulate) {
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2)
                if ((k2 >> 1) < pd) w.old[k2] = *(const Pack<T, PE>*)(w.dp + ((k2 >> 1) * step_a + (k2 & 1) * step_b) * dx.pitch);
        }
    };
    auto finish = [&](const Win& w) {
        unsigned argp = 0;                     // 3-bit window index of the maximum per channel
#pragma unroll
        for (int j = 0; j < PE; ++j) {
            float best = -INFINITY;
            int arg = c;                       // (all -inf: index 0 wins, as in the one-lane scan)
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                if ((k2 >> 1) >= pd) continue;
                float t = fmaf(sc[j], to_f(w.in[k2].v[j]), sh[j]);
                t = t > 0.f ? t : sl[j] * t;
                if (t > best || t != t) { best = t; arg = 2 * k2 + c; }
            }
            const float pb = __shfl_xor(best, cg);
            const int pa = __shfl_xor(arg, cg);
            const bool own_nan = best != best, p_nan = pb != pb;
            const bool take_p = p_nan ? (!own_nan || pa > arg) : (!own_nan && (pb > best || (pb == best && pa < arg)));
            argp |= (unsigned)(take_p ? pa : arg) << (3 * j);
        }
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            if ((k2 >> 1) >= pd) continue;
            Pack<T, PE> o;
#pragma unroll
            for (int j = 0; j < PE; ++j) {
                const float r = (((argp >> (3 * j)) & 7u) == (unsigned)(2 * k2 + c)) ? to_f(w.gq.v[j]) : 0.f;
                o.v[j] = from_f<T>(accumulate ? to_f(w.old[k2].v[j]) + r : r);
                if (RED) {
                    const float yv = to_f(w.in[k2].v[j]);
                    const float dz = to_f(o.v[j]) * (fmaf(sc[j], yv, sh[j]) > 0.f ? 1.f : sl[j]);
                    s1[j] += dz;
                    s2[j] = fmaf(dz, yv, s2[j]);
                }
            }
            *(Pack<T, PE>*)(w.dp + ((k2 >> 1) * step_a + (k2 & 1) * step_b) * dx.pitch) = o;
        }
    };
    const i64 stride = (i64)gridDim.x * rows;
    i64 ov = (i64)blockIdx.x * rows + row;
    if (live && ov < total) {
        Win cur, nxt;
        issue(ov, cur);
        while (true) {
            const i64 nov = ov + stride;
            const bool more = nov < total;
            if (more) issue(nov, nxt);
            finish(cur);
            if (!more) break;
            cur = nxt;
            ov = nov;
        }
    }
    if (RED) {
        const int C = cg * PE;
        if (live) {
#pragma unroll
            for (int j = 0; j < PE; ++j) {
                sm[((row * 2 + c) * C + c0 + j) * 2 + 0] = s1[j];
                sm[((row * 2 + c) * C + c0 + j) * 2 + 1] = s2[j];
            }
        }
        __syncthreads();
        for (int ch = threadIdx.x; ch < C; ch += TPB) {
            float a0 = 0.f, a1 = 0.f;
            for (int r = 0; r < rows * 2; ++r) { a0 += sm[(r * C + ch) * 2]; a1 += sm[(r * C + ch) * 2 + 1]; }
            partial[((i64)blockIdx.x * C + ch) * 2 + 0] = a0;
            partial[((i64)blockIdx.x * C + ch) * 2 + 1] = invstd[ch] * (a1 - mean[ch] * a0);        // sum dz * yhat
        }
    }
}
// lanes per pooled voxel = 2 * C / PE must fit a wave and divide the block: C / PE a power of two <= 32
static bool pool_pair_ok(int C, int PE) {
    static const bool off = [] { const char* e = getenv("BIU_DISABLE"); return e && strstr(e, "pool_pair") != nullptr; }();
    const int cg = C / PE;
    return !off && C % PE == 0 && cg >= 1 && cg <= 32 && (cg & (cg - 1)) == 0;      // (callers: bf16 only -- the fp32 one-lane kernel already runs at 5 TB/s)
}
static RowPlan pool_pair_plan(const biu_act* x, const biu_act* dout, int PE, i64 max_blocks) {
    RowPlan p;
    p.cg = x->c / PE;
    p.rows = TPB / (2 * p.cg);
    i64 want = (nvox(dout) + (i64)p.rows * 4 - 1) / ((i64)p.rows * 4);          // ~4 pooled voxels per lane pair
    p.grid = (int)(want < 1 ? 1 : (want > max_blocks ? max_blocks : want));
    return p;
}

template <int MODE>
static int pool_rv_launch(const biu_act* x, const biu_xform* xf, const biu_act* small, const biu_act* dx, int pd, int accumulate, int dtype, hipStream_t st) {
    BIU_DISPATCH_DTYPE(dtype, {
        constexpr int PE = 16 / sizeof(T);
        RowPlan p = row_plan(nvox(small), x->c, PE);
        hipLaunchKernelGGL((k_pool_rv<T, MODE>), dim3(p.grid), dim3(TPB), 0, st, dact(x), dxf(xf), dact(small), dx ? dact(dx) : dact(x), pd, accumulate, p.cg, p.rows);
    });
    BIU_CHECK_LAUNCH("pool_rv");
    return BIU_OK;
}
int biu_xform_apply_rv(const biu_act* x, const biu_xform* xf, const biu_act* out, int dtype, hipStream_t st) { return pool_rv_launch<0>(x, xf, out, nullptr, 1, 0, dtype, st); }
int biu_maxpool_fwd_rv(const biu_act* x, const biu_xform* xf, const biu_act* out, int pd, int dtype, hipStream_t st) {
    static const bool off = [] { const char* e = getenv("BIU_DISABLE"); return e && strstr(e, "pool_pipe") != nullptr; }();
    if (off) return pool_rv_launch<1>(x, xf, out, nullptr, pd, 0, dtype, st);
    BIU_DISPATCH_DTYPE(dtype, {
        constexpr int PE = 16 / sizeof(T);
        RowPlan p = row_plan(nvox(out), x->c, PE);
        i64 want = (nvox(out) + (i64)p.rows * 4 - 1) / ((i64)p.rows * 4);          // ~4 output voxels per thread: the pipeline needs a loop
        p.grid = (int)(want < 1 ? 1 : (want > 16384 ? 16384 : want));
        hipLaunchKernelGGL(k_maxpool_fwd_pipe<T>, dim3(p.grid), dim3(TPB), 0, st, dact(x), dxf(xf), dact(out), pd, p.cg, p.rows);
    });
    BIU_CHECK_LAUNCH("maxpool_fwd_pipe");
    return BIU_OK;
}
static RowPlan pool_bwd_plan(const biu_act* x, const biu_act* dout, int PE, i64 max_blocks) {
    RowPlan p = row_plan(nvox(dout), x->c, PE);
    i64 want = (nvox(dout) + (i64)p.rows * 4 - 1) / ((i64)p.rows * 4);          // ~4 pooled voxels (32 full-res) per thread
    p.grid = (int)(want < 1 ? 1 : (want > max_blocks ? max_blocks : want));
    return p;
}
int biu_maxpool_bwd_rv(const biu_act* x, const biu_xform* xf, const biu_act* dout, const biu_act* dx, int pd, int accumulate, int dtype, hipStream_t st) {
    BIU_DISPATCH_DTYPE(dtype, {
        constexpr int PW = 4;
        if (sizeof(T) == 2 && pool_pair_ok(x->c, PW)) {
            RowPlan p = pool_pair_plan(x, dout, PW, 16384);
            hipLaunchKernelGGL((k_maxpool_bwd_pair<T, false, PW>), dim3(p.grid), dim3(TPB), 0, st, dact(x), dxf(xf), dact(dout), dact(dx), pd, accumulate,
                               p.cg, p.rows, nullptr, nullptr, nullptr);
        } else {
            constexpr int PE = 4;
            RowPlan p = pool_bwd_plan(x, dout, PE, 16384);
            hipLaunchKernelGGL((k_maxpool_bwd_rv<T, false, PE>), dim3(p.grid), dim3(TPB), 0, st, dact(x), dxf(xf), dact(dout), dact(dx), pd, accumulate,
                               p.cg, p.rows, nullptr, nullptr, nullptr);
        }
    });
    BIU_CHECK_LAUNCH("maxpool_bwd_rv");
    return BIU_OK;
}
// same pass + BatchNorm-backward sums of x's producer; *nblk rows of [C][2] in partial (<= BIU_BN_MAX_PARTIALS)
int biu_maxpool_bwd_bnred_rv(const biu_act* x, const biu_xform* xf, const biu_act* dout, const biu_act* dx, int pd, int accumulate,
                             const float* mean, const float* invstd, float* partial, size_t partial_floats, int* nblk, int dtype,
                             hipStream_t st) {
    i64 cap = (i64)(partial_floats / ((size_t)x->c * 2));
    if (cap > 8192) cap = 8192;
    BIU_DISPATCH_DTYPE(dtype, {
        constexpr int PW = 4;
        if (sizeof(T) == 2 && pool_pair_ok(x->c, PW)) {
            RowPlan p = pool_pair_plan(x, dout, PW, cap);
            const size_t shm = (size_t)p.rows * 2 * x->c * 2 * sizeof(float);
            hipLaunchKernelGGL((k_maxpool_bwd_pair<T, true, PW>), dim3(p.grid), dim3(TPB), shm, st, dact(x), dxf(xf), dact(dout), dact(dx), pd, accumulate,
                               p.cg, p.rows, mean, invstd, partial);
            *nblk = p.grid;
        } else {
            constexpr int PE = 4;
            RowPlan p = pool_bwd_plan(x, dout, PE, cap);
            const size_t shm = (size_t)p.rows * x->c * 2 * sizeof(float);
            hipLaunchKernelGGL((k_maxpool_bwd_rv<T, true, PE>), dim3(p.grid), dim3(TPB), shm, st, dact(x), dxf(xf), dact(dout), dact(dx), pd, accumulate,
                               p.cg, p.rows, mean, invstd, partial);
            *nblk = p.grid;
        }
    });
    BIU_CHECK_LAUNCH("maxpool_bwd_bnred_rv");
    return BIU_OK;
}

// nearest-neighbour resampling with register-resident transform (MultiOutputUnet3D's interpolation path)
// MODE 0: down fwd  out[o] = T(x[2o])        (iterates over out)
// MODE 1: up fwd    out[o] = T(x[o/2])       (iterates over out)
// MODE 2: up bwd    dx[i] (+)= sum of the children of i in dout (iterates over dx)
// MODE 3: down bwd  dx[i] (+)= dout[i/2] if every coordinate of i is even else 0   (iterates over dx)
template <typename T, int MODE>
__global__ __launch_bounds__(TPB) void k_nearest_rv(DAct src, DXf xf, DAct dst, int pd, int accumulate, int cg, int rows) {
    constexpr int PE = 16 / sizeof(T);
    const int g = threadIdx.x % cg, row = threadIdx.x / cg;
    if (row >= rows) return;
    const int c0 = g * PE;
    float sc[PE], sh[PE], sl[PE];
#pragma unroll
    for (int j = 0; j < PE; ++j) {
        sc[j] = xf.scale ? xf.scale[c0 + j] : 1.f;
        sh[j] = xf.shift ? xf.shift[c0 + j] : 0.f;
        sl[j] = xf.slope ? xf.slope[c0 + j] : 1.f;
    }
    const i64 total = (i64)dst.n * dst.d * dst.h * dst.w;
    for (i64 ov = (i64)blockIdx.x * rows + row; ov < total; ov += (i64)gridDim.x * rows) {
        const Vox4 p = unvox4(ov, dst.d, dst.h, dst.w);
        T* dp = (T*)dst.p + ov * dst.pitch + c0;
        Pack<T, PE> o;
        if (MODE == 0 || MODE == 1) {
            const i64 iv = (MODE == 0) ? ((((i64)p.n * src.d + p.d * pd) * src.h + p.h * 2) * src.w + p.w * 2)
                                       : ((((i64)p.n * src.d + p.d / pd) * src.h + p.h / 2) * src.w + p.w / 2);
            Pack<T, PE> in = *(const Pack<T, PE>*)((const T*)src.p + iv * src.pitch + c0);
#pragma unroll
            for (int j = 0; j < PE; ++j) {
                const float t = fmaf(sc[j], to_f(in.v[j]), sh[j]);
                o.v[j] = from_f<T>(t > 0.f ? t : sl[j] * t);
            }
        } else {
            float acc[PE];
#pragma unroll
            for (int j = 0; j < PE; ++j) acc[j] = 0.f;
            if (MODE == 2) {
                for (int a = 0; a < pd; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const i64 iv = (((i64)p.n * src.d + p.d * pd + a) * src.h + p.h * 2 + b) * src.w + p.w * 2 + c;
                            Pack<T, PE> in = *(const Pack<T, PE>*)((const T*)src.p + iv * src.pitch + c0);
#pragma unroll
                            for (int j = 0; j < PE; ++j) acc[j] += to_f(in.v[j]);
                        }
            } else {
                const bool hit = (p.d % pd == 0) && (p.h % 2 == 0) && (p.w % 2 == 0);
                if (accumulate && !hit) continue;
                if (hit) {
                    const i64 iv = (((i64)p.n * src.d + p.d / pd) * src.h + p.h / 2) * src.w + p.w / 2;
                    Pack<T, PE> in = *(const Pack<T, PE>*)((const T*)src.p + iv * src.pitch + c0);
#pragma unroll
                    for (int j = 0; j < PE; ++j) acc[j] = to_f(in.v[j]);
                }
            }
            if (accumulate) {
                Pack<T, PE> old = *(Pack<T, PE>*)dp;
#pragma unroll
                for (int j = 0; j < PE; ++j) acc[j] += to_f(old.v[j]);
            }
#pragma unroll
            for (int j = 0; j < PE; ++j) o.v[j] = from_f<T>(acc[j]);
        }
        *(Pack<T, PE>*)dp = o;
    }
}
template <int MODE>
static int nearest_rv_launch(const biu_act* src, const biu_xform* xf, const biu_act* dst, int pd, int accumulate, int dtype, hipStream_t st) {
    BIU_DISPATCH_DTYPE(dtype, {
        constexpr int PE = 16 / sizeof(T);
        RowPlan p = row_plan(nvox(dst), dst->c, PE);
        hipLaunchKernelGGL((k_nearest_rv<T, MODE>), dim3(p.grid), dim3(TPB), 0, st, dact(src), dxf(xf), dact(dst), pd, accumulate, p.cg, p.rows);
    });
    BIU_CHECK_LAUNCH("nearest_rv");
    return BIU_OK;
}
int biu_nearest_rv(int mode, const biu_act* src, const biu_xform* xf, const biu_act* dst, int pd, int accumulate, int dtype, hipStream_t st) {
    switch (mode) {
        case 0: return nearest_rv_launch<0>(src, xf, dst, pd, accumulate, dtype, st);
        case 1: return nearest_rv_launch<1>(src, xf, dst, pd, accumulate, dtype, st);
        case 2: return nearest_rv_launch<2>(src, xf, dst, pd, accumulate, dtype, st);
        default: return nearest_rv_launch<3>(src, xf, dst, pd, accumulate, dtype, st);
    }
}
