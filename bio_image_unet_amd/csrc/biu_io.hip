// Data formats on either side of the network: the reference feeds uint8 tiles scaled by 1/255
// (unet/data.py:253-266, unet/predict.py:192-196) and re-quantises probabilities to uint8 before stitching overlapping
// tiles (unet/predict.py:199-229; linear-ramp blending in multi_output_unet3d/predict.py:203-307).  These kernels keep
// both ends on the device: uint8 batches are uploaded as they are and scaled while they are laid out channels-last, results
// are quantised, accumulated into the stitched volume and normalised without leaving HBM.  All of it is HBM-bound byte work.
#include <hip/hip_runtime.h>

#include "biu_common.h"
#include "biu_internal.h"

namespace {
constexpr int TPB = 256;

template <typename T>
__global__ void k_from_nchw_u8(const uint8_t* __restrict__ src, float scale, DAct dst) {
    const i64 S = (i64)dst.d * dst.h * dst.w;
    const i64 total = (i64)dst.n * S * dst.c;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        const int c = (int)(i % dst.c);
        const i64 v = i / dst.c;
        const i64 n = v / S, s = v % S;
        st_act<T>(dst, v, c, (float)src[(n * dst.c + c) * S + s] * scale);
    }
}
__global__ void k_u8_to_f32(const uint8_t* __restrict__ src, float scale, float* __restrict__ dst, i64 total) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) dst[i] = (float)src[i] * scale;
}
// (p * 255).astype('uint8'): truncation toward zero of a value in [0, 255]
__global__ void k_quantize_u8(const float* __restrict__ src, float scale, uint8_t* __restrict__ dst, i64 total) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        float v = src[i] * scale;
        v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
        dst[i] = (uint8_t)v;
    }
}

struct StitchGeo {
    int C, pd, ph, pw;       // patch extent (channels, depth, height, width)
    int D, H, W;             // volume extent
    int z0, y0, x0;          // patch origin inside the volume (parts beyond the volume are dropped)
};
// acc[c, z, y, x] (+)= patch[c, ..] * w[..] ; wsum[z, y, x] (+)= w[..]   (set != 0: overwrite instead of accumulate)
template <typename P>
__global__ void k_stitch_add(const P* __restrict__ patch, const float* __restrict__ weight, float* __restrict__ acc,
                             float* __restrict__ wsum, StitchGeo g, int set) {
    const i64 pv = (i64)g.pd * g.ph * g.pw;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < pv; i += (i64)gridDim.x * blockDim.x) {
        const int x = (int)(i % g.pw);
        const int y = (int)((i / g.pw) % g.ph);
        const int z = (int)(i / ((i64)g.pw * g.ph));
        const int vz = g.z0 + z, vy = g.y0 + y, vx = g.x0 + x;
        if (vz >= g.D || vy >= g.H || vx >= g.W) continue;
        const float w = weight ? weight[i] : 1.f;
        const i64 o = ((i64)vz * g.H + vy) * g.W + vx;
        wsum[o] = set ? w : wsum[o] + w;
        for (int c = 0; c < g.C; ++c) {
            const float v = (float)patch[c * pv + i] * w;
            float* a = acc + (i64)c * g.D * g.H * g.W + o;
            *a = set ? v : *a + v;
        }
    }
}
// mode 0: out_f32 = wsum > 0 ? acc / wsum : 0            (weighted blend, multi_output_unet3d/predict.py:300-303)
// mode 1: out_u8  = floor(sum_l acc_l / sum_l wsum_l)     (nan-mean of uint8 tiles cast to uint8; nl layers, unet3d/predict.py:173-195)
__global__ void k_stitch_finish(const float* __restrict__ acc, const float* __restrict__ wsum, int nl, int C, i64 S, void* __restrict__ out,
                                int mode) {
    const i64 total = (i64)C * S;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        const i64 s = i % S;
        float a = 0.f, w = 0.f;
        for (int l = 0; l < nl; ++l) { a += acc[(i64)l * total + i]; w += wsum[(i64)l * S + s]; }
        if (mode == 0) ((float*)out)[i] = w > 0.f ? a / w : 0.f;
        else ((uint8_t*)out)[i] = w > 0.f ? (uint8_t)(int)floorf(a / w) : 0;      // exact: sums of <= 2^16 uint8 values
    }
}
}  // namespace

extern "C" int biu_from_nchw_u8(const uint8_t* src, float scale, const biu_act* dst, int dtype, biu_stream stream) {
    BIU_REQUIRE(src && valid_act(dst), BIU_ERR_SHAPE, "from_nchw_u8: bad arguments");
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_from_nchw_u8<T>, dim3(grid_for(nvox(dst) * dst->c, TPB, 16384)), dim3(TPB), 0,
                                                 (hipStream_t)stream, src, scale, dact(dst)));
    BIU_CHECK_LAUNCH("from_nchw_u8");
    return BIU_OK;
}
extern "C" int biu_u8_to_f32(const uint8_t* src, float scale, float* dst, long long n, biu_stream stream) {
    BIU_REQUIRE(src && dst && n > 0, BIU_ERR_SHAPE, "u8_to_f32: bad arguments");
    hipLaunchKernelGGL(k_u8_to_f32, dim3(grid_for(n, TPB, 8192)), dim3(TPB), 0, (hipStream_t)stream, src, scale, dst, (i64)n);
    BIU_CHECK_LAUNCH("u8_to_f32");
    return BIU_OK;
}
extern "C" int biu_quantize_u8(const float* src, float scale, uint8_t* dst, long long n, biu_stream stream) {
    BIU_REQUIRE(src && dst && n > 0, BIU_ERR_SHAPE, "quantize_u8: bad arguments");
    hipLaunchKernelGGL(k_quantize_u8, dim3(grid_for(n, TPB, 8192)), dim3(TPB), 0, (hipStream_t)stream, src, scale, dst, (i64)n);
    BIU_CHECK_LAUNCH("quantize_u8");
    return BIU_OK;
}
extern "C" int biu_stitch_add(const void* patch, int patch_is_u8, const float* weight, int channels, int pd, int ph, int pw, float* acc,
                              float* wsum, int D, int H, int W, int z0, int y0, int x0, int set, biu_stream stream) {
    BIU_REQUIRE(patch && acc && wsum && channels > 0 && pd > 0 && ph > 0 && pw > 0 && D > 0 && H > 0 && W > 0 && z0 >= 0 && y0 >= 0 && x0 >= 0,
                BIU_ERR_SHAPE, "stitch_add: bad arguments");
    const StitchGeo g{channels, pd, ph, pw, D, H, W, z0, y0, x0};
    const int grid = grid_for((i64)pd * ph * pw, TPB, 4096);
    if (patch_is_u8) hipLaunchKernelGGL(k_stitch_add<uint8_t>, dim3(grid), dim3(TPB), 0, (hipStream_t)stream, (const uint8_t*)patch, weight, acc, wsum, g, set);
    else hipLaunchKernelGGL(k_stitch_add<float>, dim3(grid), dim3(TPB), 0, (hipStream_t)stream, (const float*)patch, weight, acc, wsum, g, set);
    BIU_CHECK_LAUNCH("stitch_add");
    return BIU_OK;
}
extern "C" int biu_stitch_finish(const float* acc, const float* wsum, int layers, int channels, long long spatial, void* out, int out_is_u8,
                                 biu_stream stream) {
    BIU_REQUIRE(acc && wsum && out && layers > 0 && channels > 0 && spatial > 0, BIU_ERR_SHAPE, "stitch_finish: bad arguments");
    hipLaunchKernelGGL(k_stitch_finish, dim3(grid_for((i64)channels * spatial, TPB, 8192)), dim3(TPB), 0, (hipStream_t)stream, acc, wsum, layers,
                       channels, (i64)spatial, out, out_is_u8 ? 1 : 0);
    BIU_CHECK_LAUNCH("stitch_finish");
    return BIU_OK;
}
