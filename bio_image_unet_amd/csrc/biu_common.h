// Shared host/device helpers for libbiu_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "biu.h"

typedef __bf16 bf16_t;
typedef long long i64;

// ---------------------------------------------------------------------------------------------------
// host-side error plumbing
// ---------------------------------------------------------------------------------------------------
extern __attribute__((visibility("hidden"))) thread_local char biu_errbuf[512];
int biu_fail(int code, const char* fmt, ...);

#define BIU_REQUIRE(cond, code, ...)                       \
    do {                                                   \
        if (!(cond)) return biu_fail((code), __VA_ARGS__); \
    } while (0)

#define BIU_CHECK_LAUNCH(name)                                                                  \
    do {                                                                                        \
        hipError_t e__ = hipGetLastError();                                                     \
        if (e__ != hipSuccess) return biu_fail(BIU_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline bool same_space(const biu_act* a, const biu_act* b) {
    return a->n == b->n && a->d == b->d && a->h == b->h && a->w == b->w;
}
static inline bool valid_act(const biu_act* a) {
    return a && a->p && a->n > 0 && a->d > 0 && a->h > 0 && a->w > 0 && a->c > 0 && a->pitch >= a->c;
}
static inline i64 nvox(const biu_act* a) { return (i64)a->n * a->d * a->h * a->w; }
static inline size_t dsize(int dtype) { return dtype == BIU_BF16 ? 2 : 4; }

// Is the slice addressable with vectors of `g` elements (16 B when g*sizeof(T) == 16)?
static inline bool vec_ok(const biu_act* a, int g, int dtype) {
    size_t bytes = (size_t)g * dsize(dtype);
    return a->c % g == 0 && a->pitch % g == 0 && ((uintptr_t)a->p % bytes) == 0;
}

// ---------------------------------------------------------------------------------------------------
// device-side views
// ---------------------------------------------------------------------------------------------------
struct DAct {
    char* p;
    int n, d, h, w, c, pitch;
};
static inline DAct dact(const biu_act* a) { return DAct{(char*)a->p, a->n, a->d, a->h, a->w, a->c, a->pitch}; }

struct DXf {
    const float* scale;
    const float* shift;
    const float* slope;
};
static inline DXf dxf(const biu_xform* x) {
    if (!x) return DXf{nullptr, nullptr, nullptr};
    return DXf{x->scale, x->shift, x->slope};
}

__device__ __forceinline__ float xf_pre(const DXf& xf, int c, float v) {   // affine part: t
    float s = xf.scale ? xf.scale[c] : 1.f;
    float b = xf.shift ? xf.shift[c] : 0.f;
    return (xf.scale || xf.shift) ? fmaf(s, v, b) : v;
}
__device__ __forceinline__ float xf_act(const DXf& xf, int c, float t) {   // leaky part
    if (!xf.slope) return t;
    float sl = xf.slope[c];
    return t > 0.f ? t : sl * t;
}
__device__ __forceinline__ float xf_apply(const DXf& xf, int c, float v) { return xf_act(xf, c, xf_pre(xf, c, v)); }
// d T / d t at t (PyTorch leaky_relu backward: x > 0 ? 1 : slope)
__device__ __forceinline__ float xf_dact(const DXf& xf, int c, float t) {
    if (!xf.slope) return 1.f;
    return t > 0.f ? 1.f : xf.slope[c];
}

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

template <typename T> __device__ __forceinline__ float ld_act(const DAct& a, i64 vox, int c) {
    return to_f(((const T*)a.p)[vox * a.pitch + c]);
}
template <typename T> __device__ __forceinline__ void st_act(const DAct& a, i64 vox, int c, float v) {
    ((T*)a.p)[vox * a.pitch + c] = from_f<T>(v);
}

template <typename T, int G> struct alignas(sizeof(T) * G) Pack {
    T v[G];
};

// block-wide sum of `val` (blockDim.x threads, multiple of 64, <= 1024); result valid in thread 0
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 16 floats */) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
        int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += smem[i];
    }
    return r;
}

#define BIU_DISPATCH_DTYPE(dtype, ...)                   \
    do {                                                 \
        if ((dtype) == BIU_BF16) {                       \
            typedef bf16_t T;                            \
            __VA_ARGS__;                                 \
        } else if ((dtype) == BIU_F32) {                 \
            typedef float T;                             \
            __VA_ARGS__;                                 \
        } else {                                         \
            return biu_fail(BIU_ERR_UNSUPPORTED, "unknown dtype %d", (int)(dtype)); \
        }                                                \
    } while (0)

static inline int grid_for(i64 total, int block, int cap = 1 << 20) {
    i64 g = (total + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
