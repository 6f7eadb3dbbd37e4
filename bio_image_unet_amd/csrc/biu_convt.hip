// ConvTranspose k2 s2 (non-overlapping) forward and data gradient for bf16, all 2^d parities from ONE resident tile.
//
//   forward : y[2v + a, co] = b[co] + sum_ci T(x)[v, ci] * W[ci, co, a]        (unet/unet.py:38-47, unet3d/unet3d.py:40-42)
//   dgrad   : dx[v, ci]     = sum_{a, co} dy[2v + a, co] * W[ci, co, a]
//
// The layer is byte work: 2 * Cin * Cout * 2^d FLOP per coarse voxel against (Cin + 2^d Cout) * 2 bytes (AI ~ 57 F/B at
// 64 -> 64 channels in 3-D, ridge 312) -- the fine tensor (8x the coarse one) has to cross HBM exactly once and nothing else
// matters.  The generic conv kernel ran it as 2^d separate one-tap launches (forward: the coarse tensor read 8 times) or as a
// stride-2 conv over 16-channel chunks (dgrad: every 128-byte fine row touched 4 times); here
//   * the packed weights of ALL parities sit in LDS for the life of a persistent block (64 - 128 KB),
//   * a wave owns 32 consecutive coarse voxels: their MFMA B operands are 16-byte pieces loaded straight from global memory
//     (lane = voxel, 8 consecutive channels per lane and k-step -- exactly the 32x32x16 B-fragment), no LDS round trip,
//   * per parity the 32 x 32 accumulators are packed to bf16, the two lane halves exchange half of their channel groups
//     (v_permlane32_swap, guide T21) and every lane stores 16 bytes -- 8 consecutive channels of its voxel; the two width
//     parities of a row interleave in memory, so a tile's 2 x 64 fine voxels per (depth, height) parity are one contiguous run,
//   * dgrad walks the parities the same way on the load side and optionally accumulates the BatchNorm-backward sums of the
//     block that produced the coarse tensor (biu_convt_bwd_data_bnred) in registers, reduced once per block.
// fp32 and channel counts that are not multiples of 32 stay on k_conv_pipe / the direct kernels.
#include "biu_internal.h"
#include <stdlib.h>
#include <string.h>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int NTHR = 256;          // 4 waves: one per SIMD, two blocks per CU when the weight slab is <= 64 KB

struct CtArgs {
    const char* lo;                // coarse tensor (forward: x, dgrad: dx)
    char* hi;                      // fine tensor   (forward: y, dgrad: dy)
    const uint4* wpk;              // packed weights (biu_mfma_convt_pack kind 0 / 1)
    const float* bias;             // forward only
    const float* xs; const float* xb; const float* xl;       // forward: consumer transform of x (or null)
    int lopitch, hipitch;          // elements
    int N, D, H, W;                // coarse extent (2-D: D = 1)
    int Clo, Chi;                  // channels of the coarse / fine tensor
    int ntiles_total;              // 32-wide tiles of the OUTPUT channel count (forward: Chi, dgrad: Clo)
    int nKS;                       // 16-channel k-steps of the reduction (forward: Clo / 16, dgrad: Chi / 16)
    int accumulate;                // dgrad: dx += result
    // dgrad: BatchNorm-backward sums of the block that produced lo (red_y = its raw output)
    float* red_partial; const char* red_y; int red_ypitch;
    const float* red_scale; const float* red_shift; const float* red_slope; const float* red_mean; const float* red_invstd;
};

__device__ __forceinline__ uint4 xform8(uint4 v, const float* sc, const float* sh, const float* sl) {
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = __uint_as_float(u[i] << 16), b = __uint_as_float(u[i] & 0xffff0000u);
        a = fmaf(sc[2 * i], a, sh[2 * i]);
        b = fmaf(sc[2 * i + 1], b, sh[2 * i + 1]);
        a = fmaxf(a, sl[2 * i] * a);
        b = fmaxf(b, sl[2 * i + 1] * b);
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const bf2 p = {(__bf16)a, (__bf16)b};
        o[i] = __builtin_bit_cast(unsigned, p);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

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

// 16 accumulators of one 32-channel tile (lane: voxel r, channels (e & 3) + 8 (e >> 2) + 4 h) -> two 16-byte pieces:
// lanes 0-31 get channels [16 q, 16 q + 8), lanes 32-63 channels [16 q + 8, 16 q + 16) of their voxel (q = 0, 1)
__device__ __forceinline__ void pieces_of(const floatx16& acc, uint4 out[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        unsigned a0 = pack2(acc[8 * q + 0], acc[8 * q + 1]), a1 = pack2(acc[8 * q + 2], acc[8 * q + 3]);     // group 2q
        unsigned b0 = pack2(acc[8 * q + 4], acc[8 * q + 5]), b1 = pack2(acc[8 * q + 6], acc[8 * q + 7]);     // group 2q + 1
        const u32x2 r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const u32x2 r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        out[q] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}

// KD2 = 2: 3-D (8 parities), 1: 2-D (4 parities).  NTB = 32-wide output-channel tiles per block.
template <int KD2, int NTB, int MAXKS, bool DGRAD, bool RED>
__global__ __launch_bounds__(NTHR) void k_convt_all(CtArgs a) {
    constexpr int P = KD2 * 4;
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int nKS = a.nKS;
    uint4* lw = lds;                                           // forward: [P][NTB][nKS][64] ; dgrad: [NTB][nKS][P][64]
    float* lxf = (float*)(lw + (size_t)P * NTB * nKS * 64);    // forward: [3][Clo] transform ; then [NTB*32] bias
    float* lbias = lxf + 3 * a.Clo;
    float* lred = lbias + NTB * 32;                            // RED: [4 waves][NTB*32][2] -- every wave its own row, summed in a fixed order (no atomics)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int nt0 = blockIdx.y * NTB;                           // first output-channel tile of this block
    const bool has_xf = !DGRAD && a.xs != nullptr;

    // ---- weights of this block's output tiles, all parities, into LDS (once) ----------------------------------------------
    {
        const int total = P * NTB * nKS * 64;       // (the host picks NTB as a divisor of the tile count: every tile exists)
        for (int i = tid; i < total; i += NTHR) {
            size_t src;
            if (!DGRAD) {          // lw index = ((p * NTB + nt) * nKS + ks) * 64 + lane ; packed = ((p * ntiles + nt0 + nt) * nKS + ks) * 64 + lane
                const int ln = i & 63;
                int t = i >> 6;
                const int ks = t % nKS; t /= nKS;
                const int nt = t % NTB, p = t / NTB;
                src = ((size_t)(p * a.ntiles_total + nt0 + nt) * nKS + ks) * 64 + ln;
            } else {               // lw index = ((nt * nKS + ks) * P + p) * 64 + lane    ; packed = (((nt0 + nt) * nKS + ks) * P + p) * 64 + lane
                src = (size_t)nt0 * nKS * P * 64 + i;
            }
            lw[i] = a.wpk[src];
        }
        if (has_xf)
            for (int i = tid; i < a.Clo; i += NTHR) { lxf[i] = a.xs[i]; lxf[a.Clo + i] = a.xb[i]; lxf[2 * a.Clo + i] = a.xl[i]; }
        if (tid < NTB * 32) {
            const int co = nt0 * 32 + tid;
            lbias[tid] = (!DGRAD && a.bias && co < a.Chi) ? a.bias[co] : 0.f;
            if (RED) {
                for (int w_ = 0; w_ < NTHR / 64; ++w_) { lred[w_ * NTB * 64 + 2 * tid] = 0.f; lred[w_ * NTB * 64 + 2 * tid + 1] = 0.f; }
                // the upstream block's transform vectors, read from LDS in the epilogue (lxf is free in the data-gradient kernels)
                const bool okc = co < a.Clo;
                lxf[tid] = okc ? a.red_scale[co] : 0.f;
                lxf[NTB * 32 + tid] = okc ? a.red_shift[co] : 0.f;
                lxf[2 * NTB * 32 + tid] = (okc && a.red_slope) ? a.red_slope[co] : 1.f;
            }
        }
    }
    __syncthreads();

    const unsigned nvox = (unsigned)a.N * a.D * a.H * a.W;
    const unsigned ntile = (nvox + 31u) / 32u;
    const int FD = a.D * KD2, FH = a.H * 2, FW = a.W * 2;
    const int Cout_t = DGRAD ? a.Clo : a.Chi;                    // channel count of the tensor this kernel writes

    // RED: per-lane partial sums of this lane's channels (8 per tile and pair), kept across all tiles of the block
    float s1[RED ? NTB : 1][2][8], s2[RED ? NTB : 1][2][8];
    if (RED) {
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) s1[nt][q][e] = s2[nt][q][e] = 0.f;
    }

    for (unsigned tile = blockIdx.x * 4u + wave; tile < ntile; tile += gridDim.x * 4u) {
        const unsigned v = tile * 32u + r;
        const bool vok = v < nvox;
        const unsigned vc = vok ? v : nvox - 1;
        const unsigned cw = vc % (unsigned)a.W, t1 = vc / (unsigned)a.W;
        const unsigned ch = t1 % (unsigned)a.H, t2 = t1 / (unsigned)a.H;
        const unsigned cd = t2 % (unsigned)a.D, cn = t2 / (unsigned)a.D;
        const char* lorow = a.lo + (size_t)vc * a.lopitch * 2;
        auto fine_row = [&](int p) -> size_t {                   // element offset of fine voxel 2v + parity p
            const int pd = (KD2 == 2) ? (p >> 2) : 0, ph = (p >> 1) & 1, pw = p & 1;
            const size_t fv = (((size_t)cn * FD + (cd * KD2 + pd)) * FH + (ch * 2 + ph)) * FW + (cw * 2 + pw);
            return fv * (size_t)a.hipitch;
        };

        if constexpr (!DGRAD) {
            // ---- forward: B operand = T(x) of the coarse tile, resident in registers for all parities --------------------
            uint4 bx[MAXKS];
#pragma unroll
            for (int ks = 0; ks < MAXKS; ++ks)
                if (ks < nKS) bx[ks] = *(const uint4*)(lorow + (16 * ks + 8 * hf) * 2);
            if (has_xf) {
#pragma unroll
                for (int ks = 0; ks < MAXKS; ++ks)
                    if (ks < nKS) {
                        const int c0 = 16 * ks + 8 * hf;
                        bx[ks] = xform8(bx[ks], lxf + c0, lxf + a.Clo + c0, lxf + 2 * a.Clo + c0);
                    }
            }
#pragma unroll 1
            for (int p = 0; p < P; ++p) {
                floatx16 acc[NTB];
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[nt][e] = lbias[nt * 32 + 8 * (e >> 2) + 4 * hf + (e & 3)];
#pragma unroll
                for (int ks = 0; ks < MAXKS; ++ks)
                    if (ks < nKS) {
#pragma unroll
                        for (int nt = 0; nt < NTB; ++nt) {
                            const uint4 wf = lw[((size_t)(p * NTB + nt) * nKS + ks) * 64 + lane];
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, bx[ks]), acc[nt], 0, 0, 0);
                        }
                    }
                char* orow = a.hi + fine_row(p) * 2;
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt) {
                    uint4 pc[2];
                    pieces_of(acc[nt], pc);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int co = (nt0 + nt) * 32 + 16 * q + 8 * hf;
                        if (vok && co < a.Chi) *(uint4*)(orow + (size_t)co * 2) = pc[q];
                    }
                }
            }
        } else {
            // ---- data gradient: walk the parities on the load side, one accumulator set for the coarse tile ---------------
            floatx16 acc[NTB];
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
            uint4 bcur[MAXKS], bnxt[MAXKS];
            uint4 yq[RED ? NTB : 1][2];                          // RED: the upstream block's raw output at this lane's voxel, requested before
            if constexpr (RED) {                                 // the eight parities are multiplied (it was a round trip per piece in the epilogue)
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int ci = (nt0 + nt) * 32 + 16 * q + 8 * hf;
                        yq[nt][q] = (vok && ci < a.Clo) ? *(const uint4*)(a.red_y + ((size_t)vc * a.red_ypitch + ci) * 2) : make_uint4(0, 0, 0, 0);
                    }
            }
            {
                const char* row = a.hi + fine_row(0) * 2;
#pragma unroll
                for (int ks = 0; ks < MAXKS; ++ks)
                    if (ks < nKS) bcur[ks] = *(const uint4*)(row + (16 * ks + 8 * hf) * 2);
            }
#pragma unroll 1
            for (int p = 0; p < P; ++p) {
                if (p + 1 < P) {
                    const char* row = a.hi + fine_row(p + 1) * 2;
#pragma unroll
                    for (int ks = 0; ks < MAXKS; ++ks)
                        if (ks < nKS) bnxt[ks] = *(const uint4*)(row + (16 * ks + 8 * hf) * 2);
                }
#pragma unroll
                for (int ks = 0; ks < MAXKS; ++ks)
                    if (ks < nKS) {
#pragma unroll
                        for (int nt = 0; nt < NTB; ++nt) {
                            const uint4 wf = lw[((size_t)(nt * nKS + ks) * P + p) * 64 + lane];
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, bcur[ks]), acc[nt], 0, 0, 0);
                        }
                    }
#pragma unroll
                for (int ks = 0; ks < MAXKS; ++ks) bcur[ks] = bnxt[ks];
            }
            char* orow = (char*)a.lo + (size_t)vc * a.lopitch * 2;
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt) {
                uint4 pc[2];
                pieces_of(acc[nt], pc);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int ci = (nt0 + nt) * 32 + 16 * q + 8 * hf;
                    if (!(vok && ci < a.Clo)) continue;
                    uint4 piece = pc[q];
                    if (a.accumulate) {
                        float f[8], g[8];
                        unpack8(piece, f);
                        unpack8(*(const uint4*)(orow + (size_t)ci * 2), g);
                        piece = make_uint4(pack2(f[0] + g[0], f[1] + g[1]), pack2(f[2] + g[2], f[3] + g[3]), pack2(f[4] + g[4], f[5] + g[5]),
                                           pack2(f[6] + g[6], f[7] + g[7]));
                    }
                    *(uint4*)(orow + (size_t)ci * 2) = piece;
                    if constexpr (RED) {
                        float f[8], yv[8];
                        unpack8(piece, f);                              // sums of the values as stored
                        unpack8(yq[nt][q], yv);
                        const int cl = nt * 32 + 16 * q + 8 * hf;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float tt = fmaf(lxf[cl + e], yv[e], lxf[NTB * 32 + cl + e]);
                            const float dz = f[e] * (tt > 0.f ? 1.f : lxf[2 * NTB * 32 + cl + e]);
                            s1[nt][q][e] += dz;
                            s2[nt][q][e] = fmaf(dz, yv[e], s2[nt][q][e]);
                        }
                    }
                }
            }
        }
    }

    if constexpr (RED) {
        // lanes of one half hold the same channels: butterfly over the 32 lanes of the half, then one LDS add per wave
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float u = s1[nt][q][e], w = s2[nt][q][e];
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) { u += __shfl_xor(u, o, 64); w += __shfl_xor(w, o, 64); }
                    if (r == 0) {
                        const int cc = nt * 32 + 16 * q + 8 * hf + e;
                        lred[wave * NTB * 64 + 2 * cc] = u;
                        lred[wave * NTB * 64 + 2 * cc + 1] = w;
                    }
                }
        __syncthreads();
        if (tid < NTB * 32) {
            const int ci = nt0 * 32 + tid;
            if (ci < a.Clo) {
                float* dst = a.red_partial + ((size_t)blockIdx.x * a.Clo + ci) * 2;
                float l0 = 0.f, l1 = 0.f;
#pragma unroll
                for (int w_ = 0; w_ < NTHR / 64; ++w_) { l0 += lred[w_ * NTB * 64 + 2 * tid]; l1 += lred[w_ * NTB * 64 + 2 * tid + 1]; }
                dst[0] = l0;
                dst[1] = a.red_invstd[ci] * (l1 - a.red_mean[ci] * l0);       // sum dz * yhat
            }
        }
    }
}

int num_cus_() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// output-channel tiles per block and the weight slab they need
static int pick_ntb(int ntiles, int nKS, int P) {
    const size_t per_tile = (size_t)P * nKS * 1024;
    if (ntiles % 2 == 0 && 2 * per_tile <= 128 * 1024) return 2;
    return 1;
}

template <int KD2, bool DGRAD, bool RED>
int launch(const CtArgs& a, int ntb, int grid_x, size_t lds_bytes, hipStream_t st) {
    const int gy = (a.ntiles_total + ntb - 1) / ntb;
#define CT_LAUNCH(NTB_, MAXKS_)                                                                                         \
    do {                                                                                                                \
        auto kern = k_convt_all<KD2, NTB_, MAXKS_, DGRAD, RED>;                                                          \
        static size_t attr = 0;                                                                                         \
        if (attr < lds_bytes) {                                                                                         \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) \
                return biu_fail(BIU_ERR_LAUNCH, "convt_all: cannot reserve %zu bytes of LDS", lds_bytes);                \
            attr = lds_bytes;                                                                                           \
        }                                                                                                               \
        hipLaunchKernelGGL(kern, dim3(grid_x, gy), dim3(NTHR), lds_bytes, st, a);                                        \
    } while (0)
    if (a.nKS <= 4) { if (ntb == 2) CT_LAUNCH(2, 4); else CT_LAUNCH(1, 4); }
    else if (a.nKS <= 8) { if (ntb == 2) CT_LAUNCH(2, 8); else CT_LAUNCH(1, 8); }
    else { if (ntb == 2) CT_LAUNCH(2, 16); else CT_LAUNCH(1, 16); }
#undef CT_LAUNCH
    BIU_CHECK_LAUNCH("convt_all");
    return BIU_OK;
}

bool shapes_ok(const biu_act* lo, const biu_act* hi, int kd, int dtype) {
    if (dtype != BIU_BF16 || (kd != 1 && kd != 2)) return false;
    if (lo->c % 32 || hi->c % 32 || lo->c < 32 || hi->c < 32 || lo->c > 256 || hi->c > 256) return false;
    // measured (cfg4, r02): 64 -> 64 channels at 64^3 -> 128^3 runs 0.41 -> 0.33 ms forward and 0.68 -> 0.41 ms data gradient against the
    // per-parity launches of k_conv_pipe; at 128 / 256 channels the weight slab (128 KB) leaves one 4-wave block per CU and the
    // small coarse tensors no longer hide it (up2 0.15 -> 0.16, 0.16 -> 0.21 ms): those stay on k_conv_pipe
    if (lo->c > 64 || hi->c > 64) return false;
    if ((uintptr_t)lo->p % 16 || (uintptr_t)hi->p % 16 || (lo->pitch * 2) % 16 || (hi->pitch * 2) % 16) return false;
    if (nvox(hi) >= (1LL << 31)) return false;
    return true;
}

}  // namespace

bool biu_convt_all_ok(const biu_act* lo, const biu_act* hi, int kd, int dtype) {
    static const bool off = [] { const char* e = getenv("BIU_DISABLE"); return e && strstr(e, "convt_all") != nullptr; }();
    return !off && shapes_ok(lo, hi, kd, dtype);
}

int biu_convt_all_fwd(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, int kd, const biu_act* y, hipStream_t st) {
    CtArgs a{};
    a.lo = (const char*)x->p; a.hi = (char*)y->p; a.wpk = (const uint4*)packed; a.bias = bias;
    const bool has = xf && (xf->scale || xf->shift || xf->slope);
    if (has) BIU_REQUIRE(xf->scale && xf->shift && xf->slope, BIU_ERR_UNSUPPORTED, "convt_all: partial biu_xform");
    a.xs = has ? xf->scale : nullptr; a.xb = has ? xf->shift : nullptr; a.xl = has ? xf->slope : nullptr;
    a.lopitch = x->pitch; a.hipitch = y->pitch;
    a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w; a.Clo = x->c; a.Chi = y->c;
    a.ntiles_total = y->c / 32; a.nKS = x->c / 16;
    const int P = kd * 4, ntb = pick_ntb(a.ntiles_total, a.nKS, P);
    const size_t lds = (size_t)P * ntb * a.nKS * 1024 + (3 * (size_t)a.Clo + ntb * 32 * 9) * sizeof(float);
    const int per_cu = lds <= 72 * 1024 ? 2 : 1;
    const int gy = a.ntiles_total / ntb;
    int gx = per_cu * num_cus_() / gy;
    if (gx < 1) gx = 1;
    const long long ntile = (nvox(x) + 31) / 32;
    if ((long long)gx * 4 > ntile) gx = (int)((ntile + 3) / 4);
    return kd == 2 ? launch<2, false, false>(a, ntb, gx, lds, st) : launch<1, false, false>(a, ntb, gx, lds, st);
}

// rows of BatchNorm-backward partials the fused data gradient writes (= its grid.x)
int biu_convt_all_dgrad_rows(const biu_act* dx, const biu_act* dy, int kd) {
    const int P = kd * 4, ntiles = dx->c / 32, nKS = dy->c / 16, ntb = pick_ntb(ntiles, nKS, P);
    const size_t lds = (size_t)P * ntb * nKS * 1024 + (3 * (size_t)dx->c + ntb * 32 * 9) * sizeof(float);
    const int per_cu = lds <= 72 * 1024 ? 2 : 1;
    int gx = per_cu * num_cus_() / (ntiles / ntb);
    if (gx < 1) gx = 1;
    const long long ntile = (nvox(dx) + 31) / 32;
    if ((long long)gx * 4 > ntile) gx = (int)((ntile + 3) / 4);
    return gx;
}

int biu_convt_all_dgrad(const biu_act* dy, const void* packed, int kd, const biu_act* dx, int accumulate, hipStream_t st, float* bn_partial,
                        const BnRedFuse* red) {
    CtArgs a{};
    a.lo = (const char*)dx->p; a.hi = (char*)dy->p; a.wpk = (const uint4*)packed;
    a.lopitch = dx->pitch; a.hipitch = dy->pitch;
    a.N = dx->n; a.D = dx->d; a.H = dx->h; a.W = dx->w; a.Clo = dx->c; a.Chi = dy->c;
    a.ntiles_total = dx->c / 32; a.nKS = dy->c / 16;
    a.accumulate = accumulate;
    const int P = kd * 4, ntb = pick_ntb(a.ntiles_total, a.nKS, P);
    const size_t lds = (size_t)P * ntb * a.nKS * 1024 + (3 * (size_t)a.Clo + ntb * 32 * 9) * sizeof(float);
    const int gx = biu_convt_all_dgrad_rows(dx, dy, kd);
    if (red) {
        a.red_partial = bn_partial; a.red_y = (const char*)red->y->p; a.red_ypitch = red->y->pitch;
        a.red_scale = red->scale; a.red_shift = red->shift; a.red_slope = red->slope; a.red_mean = red->mean; a.red_invstd = red->invstd;
        return kd == 2 ? launch<2, true, true>(a, ntb, gx, lds, st) : launch<1, true, true>(a, ntb, gx, lds, st);
    }
    return kd == 2 ? launch<2, true, false>(a, ntb, gx, lds, st) : launch<1, true, false>(a, ntb, gx, lds, st);
}
