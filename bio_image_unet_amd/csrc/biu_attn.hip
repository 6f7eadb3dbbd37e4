// Attention gate of AttentionUnet (unet/attention_unet.py:112-181), the two element-wise stages between its 1x1 conv blocks:
//   psi_in = relu(BN(W_gate * gate) + BN(W_x * skip))            -> biu_add_relu_fwd / _bwd
//   out    = skip * sigmoid(BN(psi * psi_in))   (1 channel gate)  -> biu_gate_fwd / _bwd
// The 1x1 convolutions and their BatchNorms are ordinary conv blocks of the engine (slope 1 = no activation); these kernels
// read their raw outputs through the consumer transform like every other kernel.  Correctness tier: scalar per (voxel, channel)
// with the channel fastest (coalesced rows); the gates sit on decoder levels whose 3x3 convs dominate.
#include <hip/hip_runtime.h>

#include "biu_common.h"
#include "biu_internal.h"

namespace {
constexpr int TPB = 256;

template <typename T>
__global__ void k_add_relu_fwd(DAct a, DXf xa, DAct b, DXf xb, DAct out) {
    const i64 total = (i64)a.n * a.d * a.h * a.w * a.c;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        const int c = (int)(i % a.c);
        const i64 v = i / a.c;
        const float s = xf_apply(xa, c, ld_act<T>(a, v, c)) + xf_apply(xb, c, ld_act<T>(b, v, c));
        st_act<T>(out, v, c, s > 0.f ? s : 0.f);
    }
}
// d(sum) = dout where the stored output is positive (nn.ReLU(inplace=True) backward uses the output)
template <typename T>
__global__ void k_add_relu_bwd(DAct out, DAct dout, DAct da, DAct db, int accumulate) {
    const i64 total = (i64)out.n * out.d * out.h * out.w * out.c;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        const int c = (int)(i % out.c);
        const i64 v = i / out.c;
        const float g = ld_act<T>(out, v, c) > 0.f ? ld_act<T>(dout, v, c) : 0.f;
        st_act<T>(da, v, c, accumulate ? g + ld_act<T>(da, v, c) : g);
        st_act<T>(db, v, c, accumulate ? g + ld_act<T>(db, v, c) : g);
    }
}

__device__ __forceinline__ float sigm(float x) {
    const float e = __expf(-fabsf(x));
    return x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}

template <typename T>
__global__ void k_gate_fwd(DAct e, DXf xe, DAct psi, DXf xpsi, DAct out) {
    const i64 total = (i64)e.n * e.d * e.h * e.w * e.c;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        const int c = (int)(i % e.c);
        const i64 v = i / e.c;
        const float s = sigm(xf_apply(xpsi, 0, ld_act<T>(psi, v, 0)));
        st_act<T>(out, v, c, xf_apply(xe, c, ld_act<T>(e, v, c)) * s);
    }
}
// one wave per voxel: de[v, c] (+)= dout[v, c] * s ;  dpsi[v] = s (1 - s) * sum_c dout[v, c] * T(e)[v, c]
template <typename T>
__global__ __launch_bounds__(TPB) void k_gate_bwd(DAct e, DXf xe, DAct psi, DXf xpsi, DAct dout, DAct de, int acc_e, DAct dpsi) {
    const int lane = threadIdx.x & 63;
    const i64 nv = (i64)e.n * e.d * e.h * e.w;
    for (i64 v = (i64)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6); v < nv; v += (i64)gridDim.x * (TPB / 64)) {
        const float s = sigm(xf_apply(xpsi, 0, ld_act<T>(psi, v, 0)));
        float acc = 0.f;
        for (int c = lane; c < e.c; c += 64) {
            const float g = ld_act<T>(dout, v, c);
            acc = fmaf(g, xf_apply(xe, c, ld_act<T>(e, v, c)), acc);
            if (de.p) st_act<T>(de, v, c, acc_e ? fmaf(g, s, ld_act<T>(de, v, c)) : g * s);
        }
        acc = wave_sum(acc);
        if (lane == 0) st_act<T>(dpsi, v, 0, acc * s * (1.f - s));
    }
}
}  // namespace

extern "C" int biu_add_relu_fwd(const biu_act* a, const biu_xform* xa, const biu_act* b, const biu_xform* xb, const biu_act* out,
                                int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(a) && valid_act(b) && valid_act(out) && same_space(a, b) && same_space(a, out) && a->c == b->c && a->c == out->c,
                BIU_ERR_SHAPE, "add_relu_fwd: shape mismatch");
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_add_relu_fwd<T>, dim3(grid_for(nvox(a) * a->c, TPB, 8192)), dim3(TPB), 0,
                                                 (hipStream_t)stream, dact(a), dxf(xa), dact(b), dxf(xb), dact(out)));
    BIU_CHECK_LAUNCH("add_relu_fwd");
    return BIU_OK;
}
extern "C" int biu_add_relu_bwd(const biu_act* out, const biu_act* dout, const biu_act* da, const biu_act* db, int accumulate, int dtype,
                                biu_stream stream) {
    BIU_REQUIRE(valid_act(out) && valid_act(dout) && valid_act(da) && valid_act(db) && same_space(out, dout) && same_space(out, da) &&
                    same_space(out, db) && out->c == dout->c && out->c == da->c && out->c == db->c,
                BIU_ERR_SHAPE, "add_relu_bwd: shape mismatch");
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_add_relu_bwd<T>, dim3(grid_for(nvox(out) * out->c, TPB, 8192)), dim3(TPB), 0,
                                                 (hipStream_t)stream, dact(out), dact(dout), dact(da), dact(db), accumulate));
    BIU_CHECK_LAUNCH("add_relu_bwd");
    return BIU_OK;
}
extern "C" int biu_gate_fwd(const biu_act* e, const biu_xform* xe, const biu_act* psi, const biu_xform* xpsi, const biu_act* out, int dtype,
                            biu_stream stream) {
    BIU_REQUIRE(valid_act(e) && valid_act(psi) && valid_act(out) && same_space(e, psi) && same_space(e, out) && psi->c == 1 && e->c == out->c,
                BIU_ERR_SHAPE, "gate_fwd: shape mismatch (psi must have one channel)");
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_gate_fwd<T>, dim3(grid_for(nvox(e) * e->c, TPB, 8192)), dim3(TPB), 0, (hipStream_t)stream,
                                                 dact(e), dxf(xe), dact(psi), dxf(xpsi), dact(out)));
    BIU_CHECK_LAUNCH("gate_fwd");
    return BIU_OK;
}
extern "C" int biu_gate_bwd(const biu_act* e, const biu_xform* xe, const biu_act* psi, const biu_xform* xpsi, const biu_act* dout,
                            const biu_act* de, int accumulate_e, const biu_act* dpsi, int dtype, biu_stream stream) {
    BIU_REQUIRE(valid_act(e) && valid_act(psi) && valid_act(dout) && valid_act(dpsi) && same_space(e, psi) && same_space(e, dout) &&
                    same_space(e, dpsi) && psi->c == 1 && dpsi->c == 1 && e->c == dout->c && (!de || (valid_act(de) && same_space(e, de) && de->c == e->c)),
                BIU_ERR_SHAPE, "gate_bwd: shape mismatch");
    const DAct dde = de ? dact(de) : DAct{nullptr, 0, 0, 0, 0, 0, 0};
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_gate_bwd<T>, dim3(grid_for(nvox(e), TPB / 64, 8192)), dim3(TPB), 0, (hipStream_t)stream,
                                                 dact(e), dxf(xe), dact(psi), dxf(xpsi), dact(dout), dde, accumulate_e, dact(dpsi)));
    BIU_CHECK_LAUNCH("gate_bwd");
    return BIU_OK;
}
