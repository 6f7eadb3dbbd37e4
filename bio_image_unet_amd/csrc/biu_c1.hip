// First layer (Cin = 1) of the 3-D / 2-D networks in bf16 on the matrix cores.
//
//   forward : y[v, co]     = b[co] + sum_tap T(x)[v + off(tap)] * w[co, tap]           (unet3d/unet3d.py:24, unet/unet.py:20)
//   wgrad   : dw[co, tap]  = sum_v  T(x)[v + off(tap)] * dy[v, co]
//
// With one input channel the 27 (9) taps ARE the reduction dimension: a brick's input is expanded once into an im2col image
// xp[voxel][32 taps] in LDS (27 gathers from a 2.7 KB halo tile per voxel), and the layer becomes a plain GEMM
//   forward : D[co][voxel] = W[co][tap] * xp^T[tap][voxel]      (2 x v_mfma_f32_32x32x16_bf16 per 32 voxels, B = 16-byte row reads)
//   wgrad   : D[tap][co]   = xp^T[tap][voxel] * dy[voxel][co]   (K = voxels: both operands through ds_read_b64_tr_b16)
// so the kernels are bound by streaming the 16-channel tensor (268 MB at 4 x 128^3), not by 432 scalar FMAs per voxel: the
// vector-ALU versions (biu_special.hip) took 0.23 / 0.45 ms at that extent against 0.05 ms of HBM time.
// The forward kernel also emits the BatchNorm statistics of the stored output (one partial row per block); the weight gradient
// applies the BatchNorm + LeakyReLU backward of the block's own output while it stages dy (dy = cA * da * T'(.) + cB * y + cC,
// written back over da like biu_conv_bwd_weight_bn promises).  fp32 and other channel counts stay on biu_special.hip.
#include "biu_internal.h"
#include <stdlib.h>
#include <string.h>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int ROWB = 80;            // LDS row of 32 bf16 + 16 bytes pad: conflict-free 16-byte and transposed reads

struct C1Args {
    const char* x; int xpitch;      // input (1 channel), elements between voxels
    char* y; int ypitch;            // forward: output ; wgrad: da (in) / dy (out, in place)
    const char* yraw; int yrawpitch;   // wgrad with fused BatchNorm backward: the block's raw conv output (else null)
    const float* w;                 // forward: weights (Cout, 1, taps) fp32
    const float* bias;
    const float* xs; const float* xb; const float* xl;   // consumer transform of the single input channel (all null: identity)
    int N, D, H, W, co0, cout;      // cout = channels handled by this launch (16 or 32), starting at co0
    int nbd, nbh, nbw, nbricks;
    float* partial;                 // forward: BatchNorm statistics rows [gridDim.x][Ctot][2] (or null); wgrad: [gridDim.x][taps * cout]
    int ctot;                       // channel count of the whole tensor (row length of the statistics partials)
    const float* bn_scale; const float* bn_shift; const float* bn_slope; const float* bn_cA; const float* bn_cB; const float* bn_cC;
};

__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 p = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(u[i] << 16); f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
}
__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }

template <int KD, int HALF = 0> struct Geo {                      // HALF: 256-voxel bricks (the weight gradient runs three 256-thread blocks per CU)
    static constexpr int TD = (KD == 3) ? (HALF ? 1 : 2) : 1, TH = (KD == 3) ? 8 : (HALF ? 8 : 16), TW = 32;
    static constexpr int BV = TD * TH * TW;                      // 512 (256) voxels per brick
    static constexpr int HD = TD + KD - 1, HH = TH + 2, HW = TW + 2;
    static constexpr int HV = HD * HH * HW;
    static constexpr int TAPS = KD * 9;
};

// halo tile of T(x) (zero outside the volume): xp_request loads this thread's halo voxels of a brick into registers (issued a brick ahead,
// so their latency hides behind the previous brick's work), xp_commit writes them to lx and builds the im2col image xp[v][32] (taps
// beyond TAPS are zero)
template <int KD, int NTHR, int HALF = 0>
struct XpRegs { unsigned short v[(Geo<KD, HALF>::HV + NTHR - 1) / NTHR]; unsigned inside; };   // inside: bit k = voxel k lies in the volume

template <int KD, int NTHR, int HALF = 0>
__device__ __forceinline__ void xp_request(const C1Args& a, int n, int d0, int h0, int w0, XpRegs<KD, NTHR, HALF>& xr) {
    using G = Geo<KD, HALF>;
    constexpr int NX = (G::HV + NTHR - 1) / NTHR;
    const int tid = threadIdx.x;
    xr.inside = 0;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int i = tid + NTHR * k;
        const int hw = i % G::HW, t = i / G::HW;
        const int hh = t % G::HH, hd = t / G::HH;
        const int gd = d0 + hd - (KD == 3 ? 1 : 0), gh = h0 + hh - 1, gw = w0 + hw - 1;
        const bool ok = i < G::HV && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        xr.v[k] = ok ? *(const unsigned short*)(a.x + ((size_t)(((size_t)n * a.D + gd) * a.H + gh) * a.W + gw) * a.xpitch * 2) : (unsigned short)0;
        xr.inside |= ok ? (1u << k) : 0u;
    }
}
template <int KD, int NTHR, int HALF = 0>
__device__ __forceinline__ void xp_commit(const C1Args& a, const XpRegs<KD, NTHR, HALF>& xr, unsigned short* lx, char* lxp) {
    using G = Geo<KD, HALF>;
    constexpr int NX = (G::HV + NTHR - 1) / NTHR;
    const int tid = threadIdx.x;
    const bool has_xf = a.xs != nullptr;
    const float xs = has_xf ? a.xs[0] : 1.f, xb = has_xf ? a.xb[0] : 0.f, xl = has_xf ? a.xl[0] : 1.f;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int i = tid + NTHR * k;
        if (i < G::HV) {
            unsigned short v = xr.v[k];
            if (has_xf && ((xr.inside >> k) & 1u)) {                 // padding stays zero: it is applied after the transform
                const float tt = fmaf(xs, bf2f(v), xb);
                const float o = tt > 0.f ? tt : xl * tt;
                v = (unsigned short)(pack2(o, 0.f) & 0xffffu);
            }
            lx[i] = v;
        }
    }
    __syncthreads();
    for (int v = tid; v < G::BV; v += NTHR) {
        const int lw = v % G::TW, t = v / G::TW;
        const int lh = t % G::TH, ld = t / G::TH;
        unsigned short val[32];
#pragma unroll
        for (int tap = 0; tap < 32; ++tap) {
            if (tap < G::TAPS) {
                const int ta = tap / 9, tb = (tap / 3) % 3, tc = tap % 3;
                val[tap] = lx[((ld + ta) * G::HH + (lh + tb)) * G::HW + (lw + tc)];
            } else {
                val[tap] = 0;
            }
        }
        uint4* dst = (uint4*)(lxp + v * ROWB);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dst[q] = make_uint4(val[8 * q] | ((unsigned)val[8 * q + 1] << 16), val[8 * q + 2] | ((unsigned)val[8 * q + 3] << 16),
                                val[8 * q + 4] | ((unsigned)val[8 * q + 5] << 16), val[8 * q + 6] | ((unsigned)val[8 * q + 7] << 16));
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// forward (+ BatchNorm statistics of the stored output)
// ---------------------------------------------------------------------------------------------------------------------
template <int KD, int NQ /* 16-channel pieces pairs: 1 -> cout 16, 2 -> cout 32 */>
__global__ __launch_bounds__(256) void k_c1_fwd_mfma(C1Args a) {
    using G = Geo<KD>;
    __shared__ __attribute__((aligned(16))) char smem[G::BV * ROWB + G::HV * 2 + 16 + 4 * 64 * 4 + 32 * 4];
    char* lxp = smem;
    unsigned short* lx = (unsigned short*)(smem + G::BV * ROWB);
    float* lred = (float*)(smem + G::BV * ROWB + ((G::HV * 2 + 15) & ~15));       // [4 waves][32][2]: every wave its own row, summed in a fixed order
    float* lbias = lred + 4 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;
    lred[tid] = 0.f;
    if (tid < 32) lbias[tid] = (a.bias && tid < a.cout) ? a.bias[a.co0 + tid] : 0.f;
    // A operand: W[co = r][tap = 16 ks + 8 hf + e], two k-steps, built once
    uint4 wf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int tap = 16 * ks + 8 * hf + e;
            f[e] = (r < a.cout && tap < G::TAPS) ? a.w[(size_t)(a.co0 + r) * G::TAPS + tap] : 0.f;
        }
        wf[ks] = make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
    }
    float s1[NQ][8], s2[NQ][8];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[q][e] = s2[q][e] = 0.f;
    __syncthreads();

    auto origin_of = [&](int brick, int& n, int& d0, int& h0, int& w0) {
        int b = brick;
        w0 = (b % a.nbw) * G::TW; b /= a.nbw;
        h0 = (b % a.nbh) * G::TH; b /= a.nbh;
        d0 = (b % a.nbd) * G::TD;
        n = b / a.nbd;
    };
    XpRegs<KD, 256> xr;
    if ((int)blockIdx.x < a.nbricks) { int n, d0, h0, w0; origin_of(blockIdx.x, n, d0, h0, w0); xp_request<KD, 256>(a, n, d0, h0, w0, xr); }
    for (int brick = blockIdx.x; brick < a.nbricks; brick += gridDim.x) {
        int n, d0, h0, w0;
        origin_of(brick, n, d0, h0, w0);
        xp_commit<KD, 256>(a, xr, lx, lxp);
        if (brick + (int)gridDim.x < a.nbricks) { int n2, d2, h2, w2; origin_of(brick + gridDim.x, n2, d2, h2, w2); xp_request<KD, 256>(a, n2, d2, h2, w2, xr); }
        for (int tile = wave; tile < G::BV / 32; tile += 4) {
            const int v = tile * 32 + r;
            floatx16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = lbias[8 * (e >> 2) + 4 * hf + (e & 3)];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 bx = *(const uint4*)(lxp + v * ROWB + (16 * ks + 8 * hf) * 2);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[ks]), __builtin_bit_cast(bf16x8, bx), acc, 0, 0, 0);
            }
            const int lw = v % G::TW, t = v / G::TW;
            const int gd = d0 + t / G::TH, gh = h0 + t % G::TH, gw = w0 + lw;
            const bool vok = gd < a.D && gh < a.H && gw < a.W;
            char* orow = a.y + ((size_t)(((size_t)n * a.D + gd) * a.H + gh) * a.W + gw) * a.ypitch * 2;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                unsigned a0 = pack2(acc[8 * q + 0], acc[8 * q + 1]), a1 = pack2(acc[8 * q + 2], acc[8 * q + 3]);
                unsigned b0 = pack2(acc[8 * q + 4], acc[8 * q + 5]), b1 = pack2(acc[8 * q + 6], acc[8 * q + 7]);
                const u32x2 r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const u32x2 r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                const uint4 piece = make_uint4(r0[0], r1[0], r0[1], r1[1]);       // channels 16 q + 8 hf .. + 7 of voxel v
                const int c = 16 * q + 8 * hf;
                if (vok && c < a.cout) {
                    *(uint4*)(orow + (size_t)(a.co0 + c) * 2) = piece;
                    float f[8];
                    unpack8(piece, f);                                           // statistics of the values as stored
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s1[q][e] += f[e]; s2[q][e] = fmaf(f[e], f[e], s2[q][e]); }
                }
            }
        }
        __syncthreads();                                         // the tiles are rebuilt for the next brick
    }
    if (a.partial) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float u = s1[q][e], w_ = s2[q][e];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { u += __shfl_xor(u, o, 64); w_ += __shfl_xor(w_, o, 64); }
                if (r == 0) { lred[wave * 64 + 2 * (16 * q + 8 * hf + e)] = u; lred[wave * 64 + 2 * (16 * q + 8 * hf + e) + 1] = w_; }   // (no atomics: the statistics are bit-reproducible)
            }
        __syncthreads();
        if (tid < a.cout) {
            float* dst = a.partial + ((size_t)blockIdx.x * a.ctot + a.co0 + tid) * 2;
            dst[0] = (lred[2 * tid] + lred[64 + 2 * tid]) + (lred[128 + 2 * tid] + lred[192 + 2 * tid]);
            dst[1] = (lred[2 * tid + 1] + lred[64 + 2 * tid + 1]) + (lred[128 + 2 * tid + 1] + lred[192 + 2 * tid + 1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient (optionally with the BatchNorm + LeakyReLU backward of the block's own output fused into the dy staging)
// ---------------------------------------------------------------------------------------------------------------------
// A streaming kernel (4 MFMAs per wave and brick): three 256-thread blocks per CU keep other bricks' loads in flight while one block
// stages or multiplies.
template <int KD>
__global__ __launch_bounds__(256, 3) void k_c1_wgrad_mfma(C1Args a) {
    using G = Geo<KD, 1>;
    constexpr int NTHR = 256, NWV = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lxp = smem;                                            // [BV][ROWB]  xp[v][tap]
    char* ldy = smem + G::BV * ROWB;                             // [BV][ROWB]  dy[v][co] (32 columns, zero beyond cout)
    unsigned short* lx = (unsigned short*)(ldy + G::BV * ROWB);
    float* lacc = (float*)((char*)lx + ((G::HV * 2 + 15) & ~15)); // [32 taps][32 co] cross-wave sum
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool fused = a.yraw != nullptr;
    const int ppv = a.cout / 8;                                  // 16-byte pieces per voxel (2 or 4)
    // this thread's channel piece is the same for every piece it stages: hoist the BatchNorm-backward vectors
    const int mypiece = tid % 4;                                 // (i = tid + NTHR k in the staging loop: i % 4 is constant)
    float ks[8], kh[8], kl[8], ka[8], kb[8], kc[8];
    if (fused && mypiece < ppv) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = a.co0 + mypiece * 8 + e;
            ks[e] = a.bn_scale[c]; kh[e] = a.bn_shift[c]; kl[e] = a.bn_slope ? a.bn_slope[c] : 1.f;
            ka[e] = a.bn_cA[c]; kb[e] = a.bn_cB[c]; kc[e] = a.bn_cC[c];
        }
    }
    for (int i = tid; i < 32 * 32; i += NTHR) lacc[i] = 0.f;
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // transposed-read lane offsets (same block geometry as k_wgrad_pipe: 16 lanes = 4 voxels x 16 columns)
    const int g = lane >> 4, li = lane & 15, qrow = li >> 2, p = li & 3, cg = g & 1, h = g >> 1;
    const int tr_lane = (8 * h + qrow) * ROWB + (16 * cg + 4 * p) * 2;

    // The (da, y) pieces of the NEXT brick are requested before this brick's im2col staging and MFMAs (the loop has nothing else to hide
    // their latency behind); a thread's pieces: voxels tid / 4 + 64 k, channel piece tid % 4.
    constexpr int NPC = G::BV * 4 / NTHR;
    uint4 pda[NPC], pyr[NPC];
    size_t pvox[NPC];
    unsigned pmask = 0;
    auto origin = [&](int brick, int& n, int& d0, int& h0, int& w0) {
        int b = brick;
        w0 = (b % a.nbw) * G::TW; b /= a.nbw;
        h0 = (b % a.nbh) * G::TH; b /= a.nbh;
        d0 = (b % a.nbd) * G::TD;
        n = b / a.nbd;
    };
    auto request = [&](int brick) {
        int n, d0, h0, w0;
        origin(brick, n, d0, h0, w0);
        pmask = 0;
#pragma unroll
        for (int k = 0; k < NPC; ++k) {
            const int v = (tid + NTHR * k) / 4;
            const int lw = v % G::TW, t = v / G::TW;
            const int gd = d0 + t / G::TH, gh = h0 + t % G::TH, gw = w0 + lw;
            pda[k] = make_uint4(0, 0, 0, 0);
            pyr[k] = make_uint4(0, 0, 0, 0);
            if (mypiece < ppv && gd < a.D && gh < a.H && gw < a.W) {
                const size_t vox = (((size_t)n * a.D + gd) * a.H + gh) * a.W + gw;
                pvox[k] = vox;
                pda[k] = *(const uint4*)(a.y + (vox * a.ypitch + a.co0 + mypiece * 8) * 2);
                if (fused) pyr[k] = *(const uint4*)(a.yraw + (vox * a.yrawpitch + a.co0 + mypiece * 8) * 2);
                pmask |= 1u << k;
            }
        }
    };
    XpRegs<KD, NTHR, 1> xr;
    if ((int)blockIdx.x < a.nbricks) {
        request(blockIdx.x);
        int n0, d00, h00, w00;
        origin(blockIdx.x, n0, d00, h00, w00);
        xp_request<KD, NTHR, 1>(a, n0, d00, h00, w00, xr);
    }
    for (int brick = blockIdx.x; brick < a.nbricks; brick += gridDim.x) {
        int n, d0, h0, w0;
        origin(brick, n, d0, h0, w0);
        // dy tile (+ fused BatchNorm backward, + write-back), zero rows for voxels outside the volume
#pragma unroll
        for (int k = 0; k < NPC; ++k) {
            const int v = (tid + NTHR * k) / 4;
            uint4 out = pda[k];
            if (fused && ((pmask >> k) & 1u)) {
                float gg[8], yy[8];
                unpack8(out, gg);
                unpack8(pyr[k], yy);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dz = gg[e] * (fmaf(ks[e], yy[e], kh[e]) > 0.f ? 1.f : kl[e]);
                    gg[e] = fmaf(ka[e], dz, fmaf(kb[e], yy[e], kc[e]));
                }
                out = make_uint4(pack2(gg[0], gg[1]), pack2(gg[2], gg[3]), pack2(gg[4], gg[5]), pack2(gg[6], gg[7]));
                *(uint4*)(a.y + (pvox[k] * a.ypitch + a.co0 + mypiece * 8) * 2) = out;          // dy replaces da
            }
            *(uint4*)(ldy + v * ROWB + mypiece * 16) = out;
        }
        xp_commit<KD, NTHR, 1>(a, xr, lx, lxp);                         // (its barriers also publish the dy tile)
        if (brick + (int)gridDim.x < a.nbricks) {
            request(brick + gridDim.x);
            int n2, d2, h2, w2;
            origin(brick + gridDim.x, n2, d2, h2, w2);
            xp_request<KD, NTHR, 1>(a, n2, d2, h2, w2, xr);
        }
        typedef bf16x4 __attribute__((address_space(3))) * lp;
        for (int ksx = wave; ksx < G::BV / 16; ksx += NWV) {
            const char* ap = lxp + ksx * 16 * ROWB + tr_lane;
            const char* bp = ldy + ksx * 16 * ROWB + tr_lane;
            const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(ap));
            const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(ap + 4 * ROWB));
            const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(bp));
            const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(bp + 4 * ROWB));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7),
                                                          __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7), acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // D[i = tap][j = co]: lane j = lane & 31, rows (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    {
        const int j = lane & 31, hf = lane >> 5;
#pragma unroll
        for (int e = 0; e < 16; ++e) atomicAdd(&lacc[((e & 3) + 8 * (e >> 2) + 4 * hf) * 32 + j], acc[e]);
    }
    __syncthreads();
    for (int i = tid; i < G::TAPS * a.cout; i += NTHR) {
        const int tap = i / a.cout, co = i % a.cout;
        a.partial[(size_t)blockIdx.x * (G::TAPS * a.cout) + i] = lacc[tap * 32 + co];
    }
}

// out[co0 + co][tap] = sum_b partial[b][tap * cout + co]   (fp64 merge, one block per output)
__global__ void k_c1m_finalize(const float* __restrict__ partial, int nblk, int taps, int cout, int co0, float* __restrict__ dw) {
    __shared__ double red[256];
    const int o = blockIdx.x;
    double s = 0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += partial[(size_t)b * (taps * cout) + o];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) dw[(size_t)(co0 + o % cout) * taps + o / cout] = (float)red[0];
}

int cus() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int KD, int HALF = 0> void bricks_of(C1Args& a) {
    using G = Geo<KD, HALF>;
    a.nbd = (a.D + G::TD - 1) / G::TD; a.nbh = (a.H + G::TH - 1) / G::TH; a.nbw = (a.W + G::TW - 1) / G::TW;
    a.nbricks = a.N * a.nbd * a.nbh * a.nbw;
}

bool off() {
    static const bool v = [] { const char* e = getenv("BIU_DISABLE"); return e && strstr(e, "c1_mfma") != nullptr; }();
    return v;
}

}  // namespace

bool biu_c1m_ok(const biu_act* x, const biu_act* y, int kd, int kh, int kw, int dil, int dtype) {
    if (off() || dtype != BIU_BF16 || x->c != 1 || dil != 1 || kh != 3 || kw != 3 || (kd != 1 && kd != 3)) return false;
    if (y->c % 16 != 0 || y->c < 16) return false;
    if ((uintptr_t)y->p % 16 || (y->pitch * 2) % 16 || (uintptr_t)x->p % 2) return false;
    return nvox(y) < (1LL << 31);
}

// rows of BatchNorm-statistics partials the forward writes
int biu_c1m_fwd_rows(const biu_act* y, int kd) {
    C1Args a{};
    a.N = y->n; a.D = y->d; a.H = y->h; a.W = y->w;
    if (kd == 3) bricks_of<3>(a); else bricks_of<1>(a);
    const int g = 3 * cus();
    return g < a.nbricks ? g : a.nbricks;
}

int biu_c1m_fwd(const biu_act* x, const biu_xform* xf, const float* w, const float* bias, int kd, const biu_act* y, float* bn_partial, hipStream_t st) {
    C1Args a{};
    a.x = (const char*)x->p; a.xpitch = x->pitch; a.y = (char*)y->p; a.ypitch = y->pitch; a.w = w; a.bias = bias;
    if (xf && (xf->scale || xf->shift || xf->slope)) {
        BIU_REQUIRE(xf->scale && xf->shift && xf->slope, BIU_ERR_UNSUPPORTED, "c1 conv: partial biu_xform");
        a.xs = xf->scale; a.xb = xf->shift; a.xl = xf->slope;
    }
    a.N = y->n; a.D = y->d; a.H = y->h; a.W = y->w; a.ctot = y->c; a.partial = bn_partial;
    const int rows = biu_c1m_fwd_rows(y, kd);
    if (kd == 3) bricks_of<3>(a); else bricks_of<1>(a);
    for (int co0 = 0; co0 < y->c;) {
        const int chunk = (y->c - co0) >= 32 ? 32 : 16;
        a.co0 = co0; a.cout = chunk;
        if (kd == 3) { if (chunk == 32) hipLaunchKernelGGL((k_c1_fwd_mfma<3, 2>), dim3(rows), dim3(256), 0, st, a); else hipLaunchKernelGGL((k_c1_fwd_mfma<3, 1>), dim3(rows), dim3(256), 0, st, a); }
        else { if (chunk == 32) hipLaunchKernelGGL((k_c1_fwd_mfma<1, 2>), dim3(rows), dim3(256), 0, st, a); else hipLaunchKernelGGL((k_c1_fwd_mfma<1, 1>), dim3(rows), dim3(256), 0, st, a); }
        co0 += chunk;
    }
    BIU_CHECK_LAUNCH("c1_fwd_mfma");
    return BIU_OK;
}

size_t biu_c1m_wgrad_workspace(int cout, int kd) { return (size_t)3 * cus() * kd * 9 * 32 * sizeof(float); }

int biu_c1m_wgrad(const biu_act* x, const biu_xform* xf, const biu_act* da, const BnBwdFuse* bn, int kd, float* dw, void* ws, size_t ws_bytes,
                  hipStream_t st) {
    BIU_REQUIRE(ws && ws_bytes >= biu_c1m_wgrad_workspace(da->c, kd), BIU_ERR_WORKSPACE, "c1_wgrad_mfma: workspace too small");
    C1Args a{};
    if (xf && (xf->scale || xf->shift || xf->slope)) {
        BIU_REQUIRE(xf->scale && xf->shift && xf->slope, BIU_ERR_UNSUPPORTED, "c1 conv: partial biu_xform");
        a.xs = xf->scale; a.xb = xf->shift; a.xl = xf->slope;
    }
    a.x = (const char*)x->p; a.xpitch = x->pitch; a.y = (char*)da->p; a.ypitch = da->pitch;
    a.N = da->n; a.D = da->d; a.H = da->h; a.W = da->w; a.partial = (float*)ws; a.ctot = da->c;
    if (bn) {
        a.yraw = (const char*)bn->y->p; a.yrawpitch = bn->y->pitch;
        a.bn_scale = bn->scale; a.bn_shift = bn->shift; a.bn_slope = bn->slope; a.bn_cA = bn->cA; a.bn_cB = bn->cB; a.bn_cC = bn->cC;
    }
    const int taps = kd * 9;
    if (kd == 3) bricks_of<3, 1>(a); else bricks_of<1, 1>(a);
    int grid = 3 * cus();
    if (grid > a.nbricks) grid = a.nbricks;
    const size_t hv = kd == 3 ? (size_t)Geo<3, 1>::HV : (size_t)Geo<1, 1>::HV;
    const size_t lds = (size_t)2 * 256 * ROWB + ((hv * 2 + 15) & ~(size_t)15) + 32 * 32 * sizeof(float);
    for (int co0 = 0; co0 < da->c;) {
        const int chunk = (da->c - co0) >= 32 ? 32 : 16;
        a.co0 = co0; a.cout = chunk;
        if (kd == 3) hipLaunchKernelGGL(k_c1_wgrad_mfma<3>, dim3(grid), dim3(256), lds, st, a);
        else hipLaunchKernelGGL(k_c1_wgrad_mfma<1>, dim3(grid), dim3(256), lds, st, a);
        BIU_CHECK_LAUNCH("c1_wgrad_mfma");
        hipLaunchKernelGGL(k_c1m_finalize, dim3(taps * chunk), dim3(256), 0, st, (const float*)ws, grid, taps, chunk, co0, dw);
        BIU_CHECK_LAUNCH("c1_wgrad_mfma_finalize");
        co0 += chunk;
    }
    return BIU_OK;
}
