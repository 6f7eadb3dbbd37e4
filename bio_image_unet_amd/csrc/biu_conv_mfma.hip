// MFMA implicit-GEMM 3x3 / 3x3x3 convolution for gfx950 (CDNA4), forward and data-gradient.
//
//   out[v, i] = sum_{tap, k} T(in)[v + off(tap), k] * Wp[i][k][tap]          (stride 1, "same", dilation 1)
//
// GEMM view per tap: D[i = out channel][j = voxel] += A[i][k] * B[k][j] with
//   A = weights  (fragment-ordered in HBM by biu_mfma_pack, streamed L2 -> VGPR, 1 KiB per wave-load)
//   B = activations, read from an LDS halo tile that is staged ONCE per input-channel chunk and re-used by all
//       27 (9) taps: a tap is just a constant LDS offset.
// Voxels sit on the MFMA lane axis, so every lane ends up with 16 output channels of ONE voxel: the epilogue
// stores 8-16 contiguous bytes per lane into the channels-last output and needs no LDS transpose.
//
// Block = 256 threads = 4 waves, each wave owns MT voxel tiles (32 voxels) x NT channel tiles (32 channels).
// LDS image: planes of 16-byte "pieces" (8 bf16 / 4 f32 channels): [piece][halo voxel] so that the 32 lanes of
// a half-wave read 512 contiguous bytes (bank-conflict free for TW = 32).  The producer's BatchNorm-affine +
// LeakyReLU (biu_xform) is applied while staging, zero padding after it.
//
// fp32 uses v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains), bf16 uses v_mfma_f32_32x32x16_bf16 (fp32 accumulate).
#include "biu_internal.h"
#include <type_traits>
#include <cstring>
#include <mutex>
#include <cstdlib>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
    static constexpr int PE = 8;     // elements per 16-byte piece
    __device__ static __forceinline__ void mma(const uint4& a, const uint4& b, floatx16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(u[i] << 16);
            f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)f[i];
        return __builtin_bit_cast(uint4, o);
    }
};
template <> struct Frag<float> {
    static constexpr int PE = 4;
    __device__ static __forceinline__ void mma(const uint4& a, const uint4& b, floatx16& c) {
        floatx4 av = __builtin_bit_cast(floatx4, a), bv = __builtin_bit_cast(floatx4, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], c, 0, 0, 0);
    }
    __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};

// fp32 tensors multiplied on the bf16 matrix pipe ("bf16x3"): every operand value v is split into hi = bf16(v) and lo = bf16(v - hi)
// (round to nearest even; v - hi is exact in fp32), and a product is taken as hi*hi' + hi*lo' + lo*hi' with fp32 accumulation -- the
// dropped lo*lo' term and the rounding of lo are each <= 2^-16 of |v v'|.  Three 32x32x16 bf16 MFMAs (96 cycles) cover the 16 reduction
// channels that take eight 32x32x2 fp32 MFMAs (512 cycles).  Global memory stays fp32 on both sides; the split happens while staging.
//
// "bf16x6" (round 3, the DEFAULT of the fp32 2-D kernels): three parts v = hi + mid + lo (lo = bf16(v - hi - mid): 24 significant bits, the split
// is exact up to 2^-24 |v|) and the six products of order >= 2^-16: lo*hi', hi*lo', mid*mid', mid*hi', hi*mid', hi*hi' (small terms first).
// What is dropped (mid*lo', lo*mid', lo*lo') is <= 2^-23 |v v'|: the products are as good as fp32's own rounding, so every test written
// for the exact fp32 MFMA holds unchanged, at 192 matrix-pipe cycles per 16 channels instead of 512.
struct f32x3_t { float v; };
struct f32x6_t { float v; };
template <> struct Frag<f32x3_t> : Frag<float> {};
template <> struct Frag<f32x6_t> : Frag<float> {};
// parts per value (0: not a split type), number of product terms, and the (a part, b part) of term t, small terms first
template <typename T> struct SplitOf { static constexpr int parts = 0, terms = 0; };
template <> struct SplitOf<f32x3_t> { static constexpr int parts = 2, terms = 3; };
template <> struct SplitOf<f32x6_t> { static constexpr int parts = 3, terms = 6; };
template <int XP> __device__ __forceinline__ constexpr int term_a(int t) { return XP == 2 ? (t == 0 ? 1 : 0) : (t == 0 ? 2 : t == 1 ? 0 : t == 2 ? 1 : t == 3 ? 1 : 0); }
template <int XP> __device__ __forceinline__ constexpr int term_b(int t) { return XP == 2 ? (t == 1 ? 1 : 0) : (t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 1 : t == 4 ? 1 : 0); }
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <int XP>
__device__ __forceinline__ void split_bf16(const float (&f)[4], uint2 (&out)[XP]) {
    bf16x4 p[XP];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float r = f[e];
#pragma unroll
        for (int k = 0; k < XP; ++k) {
            p[k][e] = (__bf16)r;
            r -= (float)p[k][e];          // exact in fp32
        }
    }
#pragma unroll
    for (int k = 0; k < XP; ++k) out[k] = __builtin_bit_cast(uint2, p[k]);
}
__device__ __forceinline__ void split_bf16x3(const float (&f)[4], uint2& hi, uint2& lo) {
    uint2 o[2];
    split_bf16<2>(f, o);
    hi = o[0]; lo = o[1];
}
__device__ __forceinline__ void mma_bf16(const uint4& a, const uint4& b, floatx16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// T(v) = max(t, slope * t), t = scale * v + shift on PE values: the affine part and the slope product as packed fp32 pairs
// (v_pk_fma_f32 / v_pk_mul_f32: half the VALU instructions of the scalar form; there is no packed max)
typedef float floatx2 __attribute__((ext_vector_type(2)));
template <int PE>
__device__ __forceinline__ void lrelu_affine(float (&f)[PE], const float* sc, const float* sh, const float* sl) {
#pragma unroll
    for (int e = 0; e < PE; e += 2) {
        const floatx2 v = {f[e], f[e + 1]}, s = {sc[e], sc[e + 1]}, b = {sh[e], sh[e + 1]}, l = {sl[e], sl[e + 1]};
        const floatx2 t = __builtin_elementwise_fma(s, v, b);
        const floatx2 u = l * t;
        f[e] = fmaxf(t[0], u[0]);
        f[e + 1] = fmaxf(t[1], u[1]);
    }
}

struct ConvArgs {
    const char* x;
    char* y;
    const uint4* wpk;
    const float* bias;
    const float* xs;     // input transform (all three non-null or all null)
    const float* xb;
    const float* xl;
    int xpitch, ypitch;  // elements
    int N, Cin, Cout;
    int GD, GH, GW;      // extent of the brick grid (= output extent / output stride)
    int ID, IH, IW;      // input extent
    int OD, OH, OW;      // output extent
    int osd, osh, osw;   // output stride: output voxel = grid voxel * os + offset(blockIdx.z)   (ConvTranspose scatter)
    int nbd, nbh, nbw;
    int nKS;             // total k-steps = Cin / (2 * PE)
    int wz_stride;       // packed-weight stride (uint4) between blockIdx.z slices
    int accumulate;
    unsigned long long* diag;   // BIU_DIAG builds only: per-phase cycle sums
    float* bn_partial;          // optional [nbricks][Cout][2] per-brick partial sums of the epilogue reduction
    // red_mode 0: (sum y, sum y^2) of the stored output (BatchNorm statistics of a forward conv)
    // red_mode 1: the output is d loss / d a of an upstream conv block whose raw output is red_y: emit
    //             (sum dz, sum dz * yhat), dz = da * T'(scale*y + shift), yhat = (y - mean) * invstd  (BatchNorm backward sums)
    int red_mode;
    const char* red_y; int red_ypitch;
    // Channel concatenation without a concat buffer.  Input: channels [0, csplit) come from x (xpitch, transform xs/xb/xl),
    // channels [csplit, Cin) from x1 (xpitch1, transform xs1/xb1/xl1); csplit is a multiple of the channel chunk.  Output (data
    // gradient of such a conv): channels [0, osplit) go to y, the rest to y1; osplit is a multiple of the block's channel tile.
    const char* x1; int xpitch1; int csplit;
    const float* xs1; const float* xb1; const float* xl1;
    char* y1; int ypitch1; int osplit; int accumulate1;
    const float* red_scale; const float* red_shift; const float* red_slope; const float* red_mean; const float* red_invstd;
    // split over the input channels (fp32 only): blockIdx.z = split index takes the chunks [z, z+1) * nchunks / ksplit and writes its partial
    // result (split 0 carries the bias) to slice z of a workspace laid out like y (y/y1 point at slice 0, slices y_zstride/y1_zstride bytes
    // apart); k_split_reduce then sums the slices into the real output.  For launches whose bricks x channel tiles would fill only a few
    // CUs (the 16 x 16 images of a U-Net bottleneck).
    int ksplit;
    size_t y_zstride, y1_zstride;
    // k_conv_pipe<T, 2, 2, 1, ...> ("fold"): 0 / 1 = forward of up-sampling + conv on the coarse tensor (parity = blockIdx.z, scattered
    // store); 2 = its data gradient: the reduction runs over (output parity p, channel chunk) -- 8 x Cin / CK items per brick --, item
    // (p, chunk) stages the parity-p sub-lattice of the fine tensor x (voxel 2u + p, a stride-2 gather), input voxel = grid voxel + tap - p
    int fold;
};

#ifdef BIU_DIAG
extern "C" unsigned long long* biu_diag_buffer = nullptr;
#define DIAG_STAMP(k) do { if (a.diag && tid == 0) { unsigned long long now_ = __builtin_readcyclecounter(); dsum_[k] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define DIAG_STAMP(k) do { } while (0)
#endif

constexpr int cpad_planes(int hv, int ckp) {     // plane stride in 16-B units: == 8/ckp (mod 8) -> conflict-free staging writes
    int want = 8 / ckp;
    int v = hv;
    while (v % 8 != want % 8) ++v;
    return v;
}

// KD x KHW x KHW taps; input voxel = grid voxel * S + tap - pad, pad = 1 for 3-tap axes, 0 otherwise; the D axis has
// stride 1 when KD == 1 (2-D tensors).  (KD,KHW,S) = (3|1,3,1): 3x3(x3) conv and its data gradient;
// (2|1,2,2): data gradient of ConvTranspose k2 s2; (1,1,1) + output scatter: ConvTranspose k2 s2 forward;
// (2,2,1) + output scatter ("fold", round 3): nearest-neighbour up-sampling followed by a 3x3x3 convolution, computed on the COARSE tensor --
// for output parity p (blockIdx.z) per axis the three fine taps collapse to two coarse ones, input voxel = grid voxel + tap - (1 - p).

// ===============================================================================================================
// Pipelined persistent variant (the one the launchers use).
//
// 512 threads = 8 waves, one block per CU, blocks walk bricks (XCD-grouped so halo neighbours share an L2).
// Work item = (brick, channel chunk).  While item i is being multiplied out of LDS, the global loads of item i+1
// -- its activation halo tile AND its weight slab -- are already in flight into registers; they are committed to LDS
// after the MFMA phase (register-staged double buffering: issue early, write late), so HBM/L2 latency hides under
// the MFMAs and LDS holds only one copy.  Weight fragments are read from LDS (1 KiB contiguous per wave-read).
// Epilogue per brick: bias, optional accumulate, store; optional per-channel (sum, sum^2) partials for BatchNorm.
// ===============================================================================================================
// LDS-DMA (global -> LDS, 16 B per lane, lane l lands at lds_base + 16 l) issued through inline asm: hipcc would
// otherwise put an s_waitcnt vmcnt(0) in front of every LDS read that follows a __builtin_amdgcn_global_load_lds,
// which serialises the whole prefetch.  The caller drains it with an explicit s_waitcnt vmcnt(0) before the barrier
// that precedes the first read of that buffer (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_base_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base_uniform)
                 : "memory");
}

// LDS a block may use for the activation tile + weight slabs (the transform vectors, partial sums come on top)
#ifndef BIU_CONV_ILV
#define BIU_CONV_ILV 1
#endif
// Prefetch placement: the next item's n global loads go out in the first BIU_PF_SPAN-th of the ng MFMA slices (1 = spread over all
// of them, 2 = over the first half, ...): pf_lo(g, n, ng) is the first piece slice g issues.
#ifndef BIU_PF_SPAN
#define BIU_PF_SPAN 1
#endif
constexpr int pf_lo(int g, int n, int ng) {
    const int span = (ng + BIU_PF_SPAN - 1) / BIU_PF_SPAN;
    return g >= span ? n : (g * n) / span;
}
#ifndef BIU_TAIL_BRANCH
#define BIU_TAIL_BRANCH 1
#endif
#ifndef BIU_TWO_PHASE_DIV
#define BIU_TWO_PHASE_DIV 8
#endif
#ifndef BIU_WGRAD_RR
#define BIU_WGRAD_RR 1
#endif
constexpr size_t conv_lds_budget(int nw) { return nw == 8 ? 142 * 1024 : 70 * 1024; }

template <typename T, int KD, int KHW, int S, int TD, int TH, int TW, int NT, int CKP, bool RED, int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_conv_pipe(ConvArgs a) {
    using F = Frag<T>;
    constexpr int NTHR = NW * 64, NWAVE = NW;         // NW = 8: one block per CU; NW = 4: two (half the LDS each)
    constexpr int XP = SplitOf<T>::parts, XT = SplitOf<T>::terms;
    constexpr bool X3 = XP > 0;                               // fp32 tensors, split bf16 products: a chunk is 16 channels = planes [part][hf] of 8 bf16
    static_assert(!X3 || CKP == 4, "split-product chunks are 16 channels (4 fp32 pieces)");
    constexpr int PE = F::PE;
    constexpr int PD = (KD == 3) ? 1 : 0;
    constexpr int PHW = (KHW == 3) ? 1 : 0;
    constexpr bool FOLD = (KD == 2 && KHW == 2 && S == 1);       // up-sampling folded into the conv: the padding depends on the output parity
    constexpr int SD = (KD == 1) ? 1 : S;
    constexpr int HD = (TD - 1) * SD + KD, HH = (TH - 1) * S + KHW, HW = (TW - 1) * S + KHW;
    constexpr int HV = HD * HH * HW;
    constexpr int PSV = cpad_planes(HV, CKP);
    constexpr int TILES = TD * TH * TW / 32;
    static_assert(TILES % NWAVE == 0, "brick must give a multiple of NW voxel tiles");
    constexpr int MT = TILES / NWAVE;
#ifndef BIU_CONV_RH
#define BIU_CONV_RH 1
#endif
    constexpr bool RH = BIU_CONV_RH && TW == 32 && S == 1 && KHW == 3 && MT >= 2 && TH % MT == 0;   // row-stacked fragment reuse
    constexpr int TAPS = KD * KHW * KHW;
    constexpr int SPC = X3 ? XP : CKP / 2;                       // k-steps (weight fragments) per tap and chunk
    constexpr int ACT16 = X3 ? 2 * XP * PSV : CKP * PSV;         // activation tile, 16-byte units
    constexpr int NSTEP = TAPS * SPC;
    constexpr int CK = CKP * PE;
    constexpr int NPA = (HV * CKP + NTHR - 1) / NTHR;            // activation pieces per thread
    constexpr int WN = NSTEP * NT * 64;                          // weight fragments (16 B) per chunk
    constexpr int NPW = (WN + NTHR - 1) / NTHR;
    // weight slab path: async global->LDS copies into a double buffer (no VGPRs) when two slabs fit; otherwise
    // register-staged like the activations
    constexpr size_t BUD = conv_lds_budget(NW);
    constexpr bool W2 = (size_t)(ACT16 + 2 * WN) * 16 <= BUD;                // double-buffered slab, DMA issued an item ahead
    constexpr bool W1 = !W2 && (size_t)(ACT16 + WN) * 16 <= BUD;             // one slab, DMA issued between the two barriers
    constexpr bool WGLDS = W2 || W1;
    constexpr int NWB = W2 ? 2 : 1;

    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    uint4* lact = lds;                       // [CKP][PSV]   (split products: [part][2][PSV])
    uint4* lw = lds + ACT16;                 // [NWB][NSTEP][NT][64]
    float* lxf = (float*)(lw + NWB * WN);    // [3][Cin] transform vectors (if any)
    float* lred = lxf + 3 * a.Cin;           // [NWAVE][NT*32][2] per-wave partial sums of the epilogue reduction
    float* lbias = lred + NWAVE * NT * 32 * 2;   // [NT*32] bias of this block's output channels (zero when there is none)
    float* lrs = lbias + NT * 32;            // [3][NT*32] RED: scale / shift / slope of the upstream block's transform

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const bool has_xf = a.xs != nullptr || a.xs1 != nullptr;
    const size_t esz = sizeof(T);
    const bool fold_dg = FOLD && a.fold == 2;                  // folded data gradient: 8 parity classes x channel chunks form the reduction
    const int nchunks_real = a.Cin / CK;
    const int nchunks = fold_dg ? 8 * nchunks_real : nchunks_real;
    const int ksz = (a.ksplit > 1) ? (int)blockIdx.z : 0;
    const int c_begin = (a.ksplit > 1) ? (ksz * nchunks) / a.ksplit : 0;
    const int c_end = (a.ksplit > 1) ? ((ksz + 1) * nchunks) / a.ksplit : nchunks;
    const bool w_static = WGLDS && (c_end - c_begin == 1);        // one chunk per brick: the weight slab in LDS never changes after the first item
    const int nbricks = a.N * a.nbd * a.nbh * a.nbw;
    const int p_mine = tid % CKP;

    for (int i = tid; i < NWAVE * NT * 32 * 2; i += NTHR) lred[i] = 0.f;
    if (tid < NT * 32) {
        const int co = blockIdx.y * NT * 32 + tid;
        lbias[tid] = (a.bias && co < a.Cout && (a.ksplit <= 1 || blockIdx.z == 0)) ? a.bias[co] : 0.f;
        if constexpr (RED) {
            const bool okc = co < a.Cout && a.red_scale != nullptr;
            lrs[tid] = okc ? a.red_scale[co] : 0.f;
            lrs[NT * 32 + tid] = okc ? a.red_shift[co] : 0.f;
            lrs[2 * NT * 32 + tid] = (okc && a.red_slope) ? a.red_slope[co] : 1.f;
        }
    }
    if (has_xf) {
        for (int i = tid; i < a.Cin; i += NTHR) {      // logical channel order; a source without a transform reads as identity
            const bool s1 = a.x1 && i >= a.csplit;
            const float* ps = s1 ? a.xs1 : a.xs;
            const float* pb = s1 ? a.xb1 : a.xb;
            const float* pl = s1 ? a.xl1 : a.xl;
            const int k = s1 ? i - a.csplit : i;
            lxf[i] = ps ? ps[k] : 1.f;
            lxf[a.Cin + i] = ps ? pb[k] : 0.f;
            lxf[2 * a.Cin + i] = ps ? pl[k] : 1.f;
        }
    }

    // brick walk: block b of XCD group (b % 8) takes consecutive bricks of that group's contiguous range
    const int G = gridDim.x;
    auto brick_of = [&](int k) -> int {       // k-th brick of this block, or >= nbricks when exhausted
        if ((G & 7) == 0) {
            const int per = G >> 3;
            return k * G + (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        }
        return k * G + blockIdx.x;
    };

    int hvb[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (wave * MT + mt) * 32 + r;
        const int lw_ = q % TW;
        const int t = q / TW;
        const int lh = t % TH;
        const int ld = t / TH;
        hvb[mt] = hf * PSV + (ld * SD * HH + lh * S) * HW + lw_ * S;
    }

    floatx16 acc[NT][MT];
    uint4 pa[NPA];
    uint4 pw[WGLDS ? 1 : NPW];
    int wcur = 0;                            // weight buffer the current item reads
    const uint4* wgrp = a.wpk + (size_t)blockIdx.z * a.wz_stride + ((size_t)blockIdx.y * NT * a.nKS * TAPS) * 64;

    struct Org { int n, d0, h0, w0; };
    auto origin = [&](int brick) -> Org {
        int b = brick;
        Org o;
        o.w0 = (b % a.nbw) * TW; b /= a.nbw;
        o.h0 = (b % a.nbh) * TH; b /= a.nbh;
        o.d0 = (b % a.nbd) * TD;
        o.n = b / a.nbd;
        return o;
    };
    // Per-thread constants of the activation pieces this thread stages (they do not depend on the brick): the halo
    // coordinate of piece j packed in 10-bit fields (hd | hh << 10 | hw << 20) and its byte offset inside the input
    // sample relative to the tile's first voxel.  Per brick only a field-wise range test (two adds, guard bits) and one
    // add remain; out-of-volume pieces read through the buffer descriptor's range check and come back as zeros.
    constexpr unsigned GBITS = (1u << 9) | (1u << 19) | (1u << 29);
    unsigned xs_[NPA];
    unsigned lvox[NPA];                      // voxel offset of the piece relative to the tile's first voxel (< 2^24)
    const unsigned cpb = (unsigned)(p_mine * PE * (int)esz);
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int i = tid + NTHR * j;
        const int hv = i / CKP;
        const int hw = hv % HW;
        const int t = hv / HW;
        const int hh = t % HH;
        const int hd = t / HH;
        xs_[j] = (hv < HV) ? (unsigned)(hd | (hh << 10) | (hw << 20)) : 511u;
        lvox[j] = fold_dg ? (unsigned)((hd * 2 * a.IH + hh) * 2 * a.IW + hw)      // (x 2 below: voxel 2u + p of the fine tensor, extents 2 ID x 2 IH x 2 IW)
                          : (unsigned)((hd * a.IH + hh) * a.IW + hw);
    }
    // weight slab of chunk ch: async global->LDS copy into the buffer the NEXT item reads (or into registers)
    auto issue_wpiece = [&](int ch, bool live, int j) {
        const uint4* wch = wgrp + (size_t)(ch * SPC) * TAPS * 64;
        const int q = tid + NTHR * j;
        if (q < WN && live) {                          // wave-uniform: WN is a multiple of 64
            const int ln = q & 63, nt = (q >> 6) % NT, step = q / (64 * NT);
            const int tap = step / SPC, sidx = step % SPC;
            const uint4* src = wch + ((size_t)nt * a.nKS * TAPS + (size_t)sidx * TAPS + tap) * 64 + ln;
            if constexpr (WGLDS) {
                uint4* dstw = lw + (W2 ? (wcur ^ 1) * WN : 0) + (q - ln);     // wave-uniform LDS base; lane l lands at +16*l
                const unsigned lbase = __builtin_amdgcn_readfirstlane(
                    (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)dstw));
                glds16(src, lbase);
            } else {
                pw[j] = *src;
            }
        }
    };
    auto issue_w = [&](int ch, bool live) {
#pragma unroll
        for (int j = 0; j < NPW; ++j) issue_wpiece(ch, live, j);
    };
    // issue the global loads of item (brick, ch) into pa / pw
    unsigned inb_mask = 0;
    // `live` = false issues nothing but still (re)defines pa: the prefetch registers must not be loop-carried PHIs,
    // or the register allocator copies them -- and waits for the loads -- in front of the MFMA phase
    unsigned c_lo = 0, c_hi = 0, rowb = 0;
    int brb = 0;
    __amdgpu_buffer_rsrc_t rs;
    auto issue_prep = [&](int brick, int ch, bool live) {
        const Org o = origin(brick);
        // (fold, forward: parity 0 of an axis reads coarse voxels v - 1, v; parity 1 reads v, v + 1.  Data gradient: the item's parity class p
        //  reads sub-lattice voxels u - p, u - p + 1)
        const int fpar = fold_dg ? ch / nchunks_real : (int)blockIdx.z;
        if (fold_dg) ch -= fpar * nchunks_real;
        const int fpd = (fpar >> 2) & 1, fph = (fpar >> 1) & 1, fpw = fpar & 1;
        const int gd0 = o.d0 * SD - PD - (FOLD ? (fold_dg ? fpd : 1 - fpd) : 0);
        const int gh0 = o.h0 * S - PHW - (FOLD ? (fold_dg ? fph : 1 - fph) : 0);
        const int gw0 = o.w0 * S - PHW - (FOLD ? (fold_dg ? fpw : 1 - fpw) : 0);
        const int lod = max(0, -gd0), loh = max(0, -gh0), low = max(0, -gw0);
        const int hid = min(HD - 1, a.ID - 1 - gd0), hih = min(HH - 1, a.IH - 1 - gh0), hiw = min(HW - 1, a.IW - 1 - gw0);
        c_lo = GBITS - (unsigned)(lod | (loh << 10) | (low << 20));     // x + c_lo keeps a guard bit iff x >= lo
        c_hi = GBITS + (unsigned)(hid | (hih << 10) | (hiw << 20));     // c_hi - x keeps a guard bit iff x <= hi
        // byte offsets inside a sample are taken mod 2^32 (samples up to 4 GB): the base may be negative (halo above the
        // volume) or beyond 2^31, the sum for an in-volume piece is the true offset
        const bool s1 = a.x1 && ch * CK >= a.csplit;          // which source tensor this channel chunk lives in
        const char* xbase = s1 ? a.x1 : a.x;
        rowb = (unsigned)((s1 ? a.xpitch1 : a.xpitch) * (int)esz);
        const int cch = ch * CK - (s1 ? a.csplit : 0);
        size_t sample_bytes = (size_t)a.ID * a.IH * a.IW * rowb;              // < 2^32 - 64 Ki (checked on the host)
        if (fold_dg) {
            // fine voxel (2 (gd0 + hd) + pd, ...) = 2 * lvox + the brick's base: the row pitch doubles, the base carries the parity
            brb = (int)(unsigned)((long long)(((2 * gd0 + fpd) * 2 * a.IH + 2 * gh0 + fph) * 2 * a.IW + 2 * gw0 + fpw) * rowb + (long long)cch * (int)esz);
            sample_bytes *= 8;
            rowb *= 2;
        } else {
            brb = (int)(unsigned)((long long)((gd0 * a.IH + gh0) * a.IW + gw0) * rowb + (long long)cch * (int)esz);
        }
        // a dead prefetch (nothing follows) reads through an empty descriptor: every piece is zero, nothing is fetched
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xbase + (size_t)o.n * sample_bytes), 0, live ? (int)(unsigned)sample_bytes : 0, 0x00020000);
        inb_mask = 0;
    };
    auto issue_piece = [&](int j) {
        const unsigned in_lo = xs_[j] + c_lo, in_hi = c_hi - xs_[j];
        const bool ok = ((in_lo & in_hi) & GBITS) == GBITS;
        const int off = ok ? (int)(__umul24(lvox[j], rowb) + cpb + (unsigned)brb) : -1;
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        pa[j] = make_uint4(v[0], v[1], v[2], v[3]);
        inb_mask |= ok ? (1u << j) : 0u;
    };
    auto issue = [&](int brick, int ch, bool live) {
        issue_prep(brick, ch, live);
#pragma unroll
        for (int j = 0; j < NPA; ++j) issue_piece(j);
        if constexpr (!W1) issue_w(ch, live);
    };
    // commit pa / pw to LDS (producer transform + zero padding applied here)
    auto commit = [&](int ch) {
        float sc[PE], sh[PE], sl[PE];
        // a chunk lies in ONE source tensor; when that source has no transform (the up-sampled half of a decoder concat) the pieces
        // go to LDS as they are -- no unpack / fma / max / pack for two thirds of decode5's chunks
        const bool xf_here = has_xf && ((a.x1 && ch * CK >= a.csplit) ? (a.xs1 != nullptr) : (a.xs != nullptr));
        if (xf_here) {
            const int c0 = ch * CK + p_mine * PE;
#pragma unroll
            for (int e = 0; e < PE; ++e) { sc[e] = lxf[c0 + e]; sh[e] = lxf[a.Cin + c0 + e]; sl[e] = lxf[2 * a.Cin + c0 + e]; }
        }
        if constexpr (X3) {
            // piece p = 4 channels 4p .. 4p+3 of the chunk; lane half hf = p & 1 multiplies pieces hf and hf + 2 as ONE 8-element fragment:
            // plane (hi: hf, lo: 2 + hf), slot of the voxel, 8-byte half p >> 1
            uint2* l2 = (uint2*)lact;
#pragma unroll
            for (int j = 0; j < NPA; ++j) {
                const int i = tid + NTHR * j;
                if (i < HV * CKP) {
                    float f[4];
                    F::unpack(pa[j], f);
                    if (xf_here && ((inb_mask >> j) & 1u)) lrelu_affine<4>(f, sc, sh, sl);
                    uint2 parts[XP];
                    split_bf16<XP>(f, parts);
                    const int slot = ((p_mine & 1) * PSV + i / CKP) * 2 + (p_mine >> 1);
#pragma unroll
                    for (int k2 = 0; k2 < XP; ++k2) l2[slot + k2 * 4 * PSV] = parts[k2];
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            const int i = tid + NTHR * j;
            if (i < HV * CKP) {
                uint4 v = pa[j];
                if (xf_here && ((inb_mask >> j) & 1u)) {
                    float f[PE];
                    F::unpack(v, f);
                    lrelu_affine<PE>(f, sc, sh, sl);
                    v = F::pack(f);
                }
                lact[p_mine * PSV + i / CKP] = v;
            }
        }
        }
        if constexpr (!WGLDS) {
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                const int q = tid + NTHR * j;
                if (q < WN) lw[q] = pw[j];
            }
        }
    };

#ifdef BIU_DIAG
    unsigned long long tprev_ = __builtin_readcyclecounter();
    unsigned long long dsum_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0c_ = tprev_, t0r_ = __builtin_amdgcn_s_memrealtime();
#endif
    // Epilogue partial sums.  After the MFMAs a lane holds 16 of the 32 channels of ONE voxel per channel tile (rows (e & 3) +
    // 8 (e >> 2) + 4 hf of D), i.e. NP 16-byte pieces of CPP channels once the bf16 halves are exchanged (see the epilogue).  Its
    // per-channel sums live in s1/s2[nt][piece][e].  Forward kernels keep them in registers across ALL bricks of the block and
    // reduce once after the loop (one partial row per block); the data-gradient kernels (RED) reduce every brick, piece by piece.
    constexpr int CPP = 16 / (int)sizeof(T);     // channels per 16-byte piece: 8 (bf16) / 4 (fp32)
    constexpr int NP = 16 / CPP;                 // pieces per lane and channel tile: 2 / 4
    constexpr int PSTEP = 32 / NP;               // channel distance between a lane's pieces: 16 / 8
    constexpr bool ACCB = !RED;
    const bool want_stats = a.bn_partial != nullptr;
    float s1[ACCB ? NT : 1][ACCB ? NP : 1][CPP], s2[ACCB ? NT : 1][ACCB ? NP : 1][CPP];
#pragma unroll
    for (int nt = 0; nt < (ACCB ? NT : 1); ++nt)
#pragma unroll
        for (int p = 0; p < (ACCB ? NP : 1); ++p)
#pragma unroll
            for (int e = 0; e < CPP; ++e) s1[nt][p][e] = s2[nt][p][e] = 0.f;
    // the 32 lanes of a half-wave hold the same channels: four DPP rotate-adds sum each row of 16 lanes, row_bcast:15 adds the
    // even rows into the odd ones (VALU only), lanes 16 / 48 add the half-wave sums into the wave's OWN LDS slots -- lred[wave][channel][2],
    // zeroed at kernel start, no atomics, no barrier -- and flush_stats adds the NWAVE rows ONCE, after the block's last brick: one
    // partial row per block for the forward statistics and for the data-gradient sums alike.  (channel = nt * 32 + p * PSTEP + hf * CPP + e)
    auto reduce_piece = [&](float (&u_)[CPP], float (&v_)[CPP], int nt, int p) {
        float* slot = lred + ((size_t)(wave * (NT * 32) + nt * 32 + p * PSTEP + hf * CPP)) * 2;
#pragma unroll
        for (int e = 0; e < CPP; ++e) {
            float u = u_[e], v = v_[e];
            u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x128, 0xf, 0xf, false));   // row_ror:8
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
            u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x124, 0xf, 0xf, false));   // row_ror:4
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
            u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x122, 0xf, 0xf, false));   // row_ror:2
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));
            u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x121, 0xf, 0xf, false));   // row_ror:1
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
            u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
            if ((lane & 31) == 16) { float2 o = *(float2*)(slot + 2 * e); o.x += u; o.y += v; *(float2*)(slot + 2 * e) = o; }   // this wave's own slot
            u_[e] = v_[e] = 0.f;
        }
    };
    auto flush_stats = [&](int row) {
        __syncthreads();
        if (tid < NT * 32) {
            const int co = blockIdx.y * NT * 32 + tid;
            float l0 = 0.f, l1 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < NWAVE; ++w2) {
                const float2 t = *(const float2*)(lred + ((size_t)w2 * (NT * 32) + tid) * 2);
                l0 += t.x; l1 += t.y;
            }
            if (co < a.Cout) {
                float* dstp = a.bn_partial + ((size_t)row * a.Cout + co) * 2;
                dstp[0] = l0;
                if constexpr (RED) dstp[1] = a.red_invstd[co] * (l1 - a.red_mean[co] * l0);   // sum dz * yhat
                else dstp[1] = l1;
            }
        }
    };
    int k = 0;
    int brick = brick_of(0);
    if (brick >= nbricks) return;            // uniform per block
    int ch = c_begin;
    issue(brick, c_begin, true);
    if constexpr (W1) issue_w(c_begin, true);
    __syncthreads();                         // lxf visible
    commit(c_begin);
    if constexpr (WGLDS) { if constexpr (W2) wcur ^= 1; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __syncthreads();

    while (true) {
        if (ch == c_begin) {
            // accumulators start at the bias (lane holds channels 8*qq + 4*hf + i of each 32-channel tile)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float b0 = lbias[nt * 32 + 8 * (e >> 2) + 4 * hf + (e & 3)];     // LDS: a global load here is exposed latency
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[nt][mt][e] = b0;
                }
        }
        // next item
        int nbrick = brick, nch = ch + 1, nk = k;
        if (nch == c_end) { nch = c_begin; nk = k + 1; nbrick = brick_of(nk); }
        const bool have_next = nbrick < nbricks;
        DIAG_STAMP(0);
        const uint4* lwc = lw + (W2 ? wcur * WN : 0) + lane;
        auto window = [&](int ta, int tb) {
            const uint4* lwp = lwc + ((ta * KHW + tb) * KHW) * (SPC * NT * 64);
            const uint4* lap = lact + (ta * HH + tb) * HW;
            if constexpr (X3) {
#pragma unroll
                for (int tc = 0; tc < KHW; ++tc) {
                    // one activation part at a time (lo first), against the weight parts it pairs with (part a of the weights with part b of
                    // the activations when a + b < XP): only MT + XP * NT fragments are live; MT * NT independent accumulators lie between
                    // dependent MFMAs
                    uint4 wv[XP][NT];
#pragma unroll
                    for (int k2 = 0; k2 < XP; ++k2)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) wv[k2][nt] = lwp[((tc * XP + k2) * NT + nt) * 64];
#pragma unroll
                    for (int kb = XP - 1; kb >= 0; --kb) {
                        uint4 bv[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) bv[mt] = lap[hvb[mt] + 2 * kb * PSV + tc];
#pragma unroll
                        for (int ka = XP - 1 - kb; ka >= 0; --ka)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) mma_bf16(wv[ka][nt], bv[mt], acc[nt][mt]);
                    }
                }
                return;
            }
#pragma unroll
            for (int st2 = 0; st2 < KHW * SPC; ++st2) {
                const int tc = st2 / SPC, sidx = st2 % SPC;
                uint4 wf[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wf[nt] = lwp[(st2 * NT + nt) * 64];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const uint4 bf = lap[hvb[mt] + 2 * sidx * PSV + tc];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) F::mma(wf[nt], bf, acc[nt][mt]);
                }
            }
        };
        // Row-stacked variant (RH): a wave's MT voxel tiles are MT consecutive rows of 32 voxels, so the fragment tile mt needs for
        // tap row tb is the row (mt + tb) of the halo tile: MT + 2 fragment reads serve 3 * MT MFMAs per (ta, tc) -- 0.5 KiB of LDS
        // activation traffic per MFMA instead of 1 KiB, which at NT = 1 is what bounded the kernel (LDS: 128 B/clk per CU against
        // 4 SIMDs * 1.25 KiB per 32-cycle MFMA).
        auto window_h = [&](int ta, int tc) {
            if constexpr (X3) {
                // one activation part at a time (lo first): its MT + 2 row fragments stay live across the three kh taps, whose weight parts
                // (part a pairs with activation part b when a + b < XP) are read per tap
                const uint4* lap = lact + hvb[0] + ta * HH * HW + tc;
#pragma unroll
                for (int kb = XP - 1; kb >= 0; --kb) {
                    uint4 rv[MT + 2];
#pragma unroll
                    for (int j = 0; j < MT + 2; ++j) rv[j] = lap[2 * kb * PSV + j * HW];
#pragma unroll
                    for (int tb = 0; tb < KHW; ++tb) {
                        uint4 wv[XP][NT];
#pragma unroll
                        for (int ka = 0; ka < XP - kb; ++ka)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) wv[ka][nt] = lwc[((((ta * KHW + tb) * KHW + tc) * XP + ka) * NT + nt) * 64];
#pragma unroll
                        for (int ka = XP - 1 - kb; ka >= 0; --ka)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) mma_bf16(wv[ka][nt], rv[mt + tb], acc[nt][mt]);
                    }
                }
                return;
            }
#pragma unroll
            for (int sidx = 0; sidx < SPC; ++sidx) {
                uint4 rows[MT + 2];
                const uint4* lap = lact + hvb[0] + 2 * sidx * PSV + ta * HH * HW + tc;
#pragma unroll
                for (int j = 0; j < MT + 2; ++j) rows[j] = lap[j * HW];
#pragma unroll
                for (int tb = 0; tb < KHW; ++tb) {
                    uint4 wf[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) wf[nt] = lwc[(((((ta * KHW + tb) * KHW + tc) * SPC) + sidx) * NT + nt) * 64];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) F::mma(wf[nt], rows[mt + tb], acc[nt][mt]);
                }
            }
        };
        auto tapgroup = [&](int g0, int g1) {
            if constexpr (RH) window_h(g0, g1); else window(g0, g1);
        };
        if constexpr (BIU_CONV_ILV) {
            // ---- MFMA phase with the next item's global loads spread over its (kd, kh) tap groups: the address
            //      unit takes the pieces one by one while the matrix cores run, instead of in a burst in front of them.
            //      sched_barrier keeps the groups apart (a scheduler free to hoist LDS fragments across them spills).
            constexpr int NG = KD * KHW;
            issue_prep(have_next ? nbrick : brick, nch, have_next);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int j = pf_lo(g, NPA, NG); j < pf_lo(g + 1, NPA, NG); ++j) issue_piece(j);
                if constexpr (!W1) {
#pragma unroll
                    for (int j = pf_lo(g, NPW, NG); j < pf_lo(g + 1, NPW, NG); ++j) issue_wpiece(nch, have_next && !w_static, j);
                }
                tapgroup(g / KHW, g % KHW);
                __builtin_amdgcn_sched_barrier(0);
            }
            DIAG_STAMP(1);
        } else {
            issue(have_next ? nbrick : brick, nch, have_next);
            DIAG_STAMP(1);
            // Only the innermost (kw, k-step) window is unrolled: a fully unrolled tap loop lets the scheduler hoist dozens
            // of LDS fragments and spill -- and every spill reload carries an s_waitcnt vmcnt(0) that would serialise the
            // prefetch loads issued above.
#pragma unroll 1
            for (int ta = 0; ta < KD; ++ta) {
#pragma unroll 1
                for (int tb = 0; tb < KHW; ++tb) tapgroup(ta, tb);
            }
        }

        DIAG_STAMP(2);
        // ---- brick finished: epilogue ----------------------------------------------------------------------------------
        if (ch == c_end - 1) {
            // No LDS staging: a lane holds, for ITS voxel, channel groups 8 g + 4 hf .. + 3 (g = 0..3) of each tile.  fp32: every
            // group is a 16-byte piece already.  bf16: a group is 8 bytes; v_permlane32_swap exchanges the upper half-wave's group
            // 2p with the lower one's group 2p+1, after which lanes 0-31 hold channels 16 p .. + 7 and lanes 32-63 channels
            // 16 p + 8 .. + 15 of their voxel: one 16-byte store per lane and pair (cdna_hip_programming.md T21).
            const Org o = origin(brick);
            int vo[MT];                                  // output voxel index of this lane's voxel in tile mt, -1 outside the tensor
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int q = (wave * MT + mt) * 32 + r;
                const int lw_ = q % TW;
                const int t = q / TW;
                const int lh = t % TH;
                const int ld = t / TH;
                const int gd = o.d0 + ld, gh = o.h0 + lh, gw = o.w0 + lw_;
                const int od = gd * a.osd + ((a.osd == 2) ? (int)(blockIdx.z >> 2) : 0);
                const int oh = gh * a.osh + ((a.osh == 2) ? (int)((blockIdx.z >> 1) & 1) : 0);
                const int ow = gw * a.osw + ((a.osw == 2) ? (int)(blockIdx.z & 1) : 0);
                vo[mt] = (gd < a.GD && gh < a.GH && gw < a.GW) ? ((o.n * a.OD + od) * a.OH + oh) * a.OW + ow : -1;
            }
            // destination of this block's channel tile (the second tensor of a split output when the tile lies beyond osplit)
            const bool o1 = a.y1 && (int)(blockIdx.y * NT * 32) >= a.osplit;
            char* ybase = (o1 ? a.y1 : a.y) + (a.ksplit > 1 ? (size_t)blockIdx.z * (o1 ? a.y1_zstride : a.y_zstride) : (size_t)0);
            const int ypitch_o = o1 ? a.ypitch1 : a.ypitch;
            const int coff = o1 ? a.osplit : 0;
            const int accum_o = o1 ? a.accumulate1 : a.accumulate;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int cl = (blockIdx.y * NT + nt) * 32 + p * PSTEP + hf * CPP;      // this lane's channels of piece p
                    const bool c_ok = cl < a.Cout;
                    float rsc[CPP], rsh[CPP], rsl[CPP], t1[CPP], t2[CPP];
                    if constexpr (RED) {
#pragma unroll
                        for (int e = 0; e < CPP; ++e) {
                            const int ci = nt * 32 + p * PSTEP + hf * CPP + e;
                            rsc[e] = lrs[ci]; rsh[e] = lrs[NT * 32 + ci]; rsl[e] = lrs[2 * NT * 32 + ci];
                            t1[e] = t2[e] = 0.f;
                        }
                    }
                    // RED: the upstream block's raw output at this lane's MT voxels -- all MT loads of the piece in flight together
                    // (inside the loop below they were MT serial round trips per piece, NT * NP * MT per brick)
                    constexpr int YB = (MT > 2) ? 2 : MT;           // loads per batch (MT = 4: two batches of two -- all four spill)
                    uint4 yq[RED ? YB : 1];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        if constexpr (RED) {
                            if (want_stats && mt % YB == 0) {
#pragma unroll
                                for (int m2 = 0; m2 < YB; ++m2)
                                    yq[m2] = (c_ok && vo[mt + m2] >= 0) ? *(const uint4*)((const T*)a.red_y + (size_t)vo[mt + m2] * a.red_ypitch + cl)
                                                                        : make_uint4(0, 0, 0, 0);
                            }
                        }
                        uint4 piece;
                        if constexpr (sizeof(T) == 2) {
                            Pack<T, 4> g0, g1;
#pragma unroll
                            for (int i = 0; i < 4; ++i) { g0.v[i] = (T)acc[nt][mt][8 * p + i]; g1.v[i] = (T)acc[nt][mt][8 * p + 4 + i]; }
                            uint2 ua = __builtin_bit_cast(uint2, g0), ub = __builtin_bit_cast(uint2, g1);
                            const auto sx = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
                            const auto sy = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
                            piece = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                        } else {
                            piece = make_uint4(__float_as_uint(acc[nt][mt][4 * p]), __float_as_uint(acc[nt][mt][4 * p + 1]),
                                               __float_as_uint(acc[nt][mt][4 * p + 2]), __float_as_uint(acc[nt][mt][4 * p + 3]));
                        }
                        if (!(c_ok && vo[mt] >= 0)) continue;
                        const size_t vox = (size_t)vo[mt];
                        uint4* dst = (uint4*)((T*)ybase + vox * ypitch_o + (cl - coff));
                        float f[CPP];
                        if (accum_o) {
                            float g[CPP];
                            F::unpack(piece, f);
                            F::unpack(*dst, g);
#pragma unroll
                            for (int e = 0; e < CPP; ++e) f[e] += g[e];
                            piece = F::pack(f);
                        }
                        *dst = piece;
                        if (want_stats) {
                            F::unpack(piece, f);          // statistics of the values as stored
                            if constexpr (!RED) {
#pragma unroll
                                for (int e = 0; e < CPP; ++e) {
                                    s1[nt][p][e] += f[e];
                                    s2[nt][p][e] = fmaf(f[e], f[e], s2[nt][p][e]);
                                }
                            } else {
                                float yv[CPP];
                                F::unpack(yq[mt % YB], yv);
#pragma unroll
                                for (int e = 0; e < CPP; ++e) {
                                    const float tt = fmaf(rsc[e], yv[e], rsh[e]);
                                    const float dz = f[e] * (tt > 0.f ? 1.f : rsl[e]);
                                    t1[e] += dz;
                                    t2[e] = fmaf(dz, yv[e], t2[e]);     // raw; centred when the partial is written
                                }
                            }
                        }
                    }
                    if constexpr (RED) {
                        if (want_stats) reduce_piece(t1, t2, nt, p);
                    }
                }
            }

        }
        DIAG_STAMP(3);
#ifdef BIU_DIAG
        dsum_[7] += 1;
        if (!have_next && a.diag && tid == 0) {
            for (int q_ = 0; q_ < 8; ++q_) atomicAdd(a.diag + q_, dsum_[q_]);
            atomicAdd(a.diag + 8, __builtin_readcyclecounter() - t0c_);            // shader cycles ...
            atomicAdd(a.diag + 9, __builtin_amdgcn_s_memrealtime() - t0r_);        // ... per 100 MHz tick = clock
        }
#endif
        if (!have_next) break;
        __syncthreads();                     // everyone is done reading the tile
        DIAG_STAMP(4);
        if constexpr (W1) { if (!w_static) issue_w(nch, true); }   // the slab is free now; the copy flies while the tile is committed
        commit(nch);
        if constexpr (WGLDS) { if constexpr (W2) { if (!w_static) wcur ^= 1; } asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // weight DMA landed
        DIAG_STAMP(5);
        __syncthreads();
        DIAG_STAMP(6);
        brick = nbrick; ch = nch; k = nk;
    }
    if (want_stats) {
        if constexpr (ACCB) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int p = 0; p < NP; ++p) reduce_piece(s1[nt][p], s2[nt][p], nt, p);
        }
        flush_stats(FOLD ? (int)(blockIdx.z * gridDim.x + blockIdx.x) : (int)blockIdx.x);      // (fold: one row per block and output parity)
    }
}

// ===============================================================================================================
// 16-output-channel variant (bf16): v_mfma_f32_16x16x32_bf16, so a 16-channel layer (decode6 forward, the data gradient
// of encode2 in UNet3D(n_filter=16..32)) does not multiply 16 rows of zeros on a 32-row tile.
//
//   D[cout 0..15][voxel 0..15] += A[cout][k 0..31] * B[k][voxel]       one MFMA = one tap, 32 input channels, 16 voxels
//
// Lane l = (n = l & 15, q = l >> 4): A = W[cout n][k = 8q..8q+7], B = in[voxel n][channel piece q], D rows 4q..4q+3 of column n.
// The LDS image is the one k_conv_pipe uses -- planes of 16-byte pieces [piece][halo voxel], 4 planes = 32 channels per
// chunk -- so a B fragment is lact[q * PSV + voxel].  A wave owns R consecutive rows of 16 voxels (brick TD x TH x 16): per
// (kd tap, kw tap) it reads R + 2 row fragments and 3 weight fragments for 3 R MFMAs (the fragment of tile row j and kh tap b is
// row j + b).  Staging, prefetch, weight DMA, brick walk and the epilogue reductions follow k_conv_pipe.
// ===============================================================================================================
typedef float floatx4m __attribute__((ext_vector_type(4)));

// MTL = 16-row output tiles per block column (16 * MTL output channels: 1 for a 16-channel layer; 2 -- round 3 -- for 32-channel tiles
// of a layer whose reduction is ONE 32-channel chunk, so that its 27 x MTL KiB weight slab stays in LDS for the life of the block).
template <int KD, int TD, int TH, int MTL, bool RED>
__global__ __launch_bounds__(512, 2) void k_conv16_pipe(ConvArgs a) {
    using T = bf16_t;
    using F = Frag<T>;
    constexpr int TW = 16, NTHR = 512, NWAVE = 8, PE = 8, CKP = 4, CK = 32;
    constexpr int PD = (KD == 3) ? 1 : 0;
    constexpr int HD = TD + KD - 1, HH = TH + 2, HW = TW + 2, HV = HD * HH * HW;
    constexpr int PSV = (HV + 15) & ~15;              // plane stride = 0 (mod 16 pieces): the four planes of a read hit disjoint banks
    constexpr int R = TD * TH / NWAVE;                // 16-voxel rows per wave
    static_assert(TH % R == 0 && R >= 2 && (TD * TH) % NWAVE == 0, "rows of a wave must stay inside one plane");
    constexpr int TAPS = KD * 9;
    constexpr int NPA = (HV * CKP + NTHR - 1) / NTHR;
    constexpr int WN = TAPS * MTL * 64;               // weight fragments (16 B) per chunk: [tap][m][lane]
    constexpr int NWB = MTL == 1 ? 2 : 1;             // MTL > 1: single-chunk layers only, one resident slab
    constexpr int CB = 16 * MTL;                      // output channels of a block column
    constexpr int NPW = (WN + NTHR - 1) / NTHR;
    constexpr int NG = KD * 3;                        // MFMA groups per item: (kd tap, kw tap)

    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    uint4* lact = lds;                       // [CKP][PSV]
    uint4* lw = lds + CKP * PSV;             // [NWB][TAPS][MTL][64]
    float* lxf = (float*)(lw + NWB * WN);    // [3][Cin]
    float* lred = lxf + 3 * a.Cin;           // [NWAVE][CB][2]
    float* lbias = lred + NWAVE * CB * 2;    // [CB]
    float* lrs = lbias + CB;                 // [3][CB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, q4 = lane >> 4;
    const bool has_xf = a.xs != nullptr;
    const int nchunks = a.Cin / CK;
    const int nbricks = a.N * a.nbd * a.nbh * a.nbw;
    const int p_mine = tid % CKP;

    const int co0 = (int)blockIdx.y * CB;                 // first output channel of this block column
    for (int i = tid; i < NWAVE * CB * 2; i += NTHR) lred[i] = 0.f;
    if (tid < CB) {
        const int co = co0 + tid;
        lbias[tid] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
        if constexpr (RED) {
            const bool okc = co < a.Cout && a.red_scale != nullptr;
            lrs[tid] = okc ? a.red_scale[co] : 0.f;
            lrs[CB + tid] = okc ? a.red_shift[co] : 0.f;
            lrs[2 * CB + tid] = (okc && a.red_slope) ? a.red_slope[co] : 1.f;
        }
    }
    if (has_xf) {
        for (int i = tid; i < a.Cin; i += NTHR) {
            lxf[i] = a.xs[i];
            lxf[a.Cin + i] = a.xb[i];
            lxf[2 * a.Cin + i] = a.xl[i];
        }
    }
    const int G = gridDim.x;
    auto brick_of = [&](int k) -> int {
        if ((G & 7) == 0) return k * G + (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
        return k * G + (int)blockIdx.x;
    };
    struct Org { int n, d0, h0, w0; };
    auto origin = [&](int brick) -> Org {
        int b = brick;
        Org o;
        o.w0 = (b % a.nbw) * TW; b /= a.nbw;
        o.h0 = (b % a.nbh) * TH; b /= a.nbh;
        o.d0 = (b % a.nbd) * TD;
        o.n = b / a.nbd;
        return o;
    };
    // this wave's rows: flattened (d, h) rows wave * R .. + R - 1 of the brick
    const int ld_w = (wave * R) / TH, lh_w = (wave * R) % TH;
    const int hvb = q4 * PSV + (ld_w * HH + lh_w) * HW + n16;      // fragment address of row 0, tap (0, 0, 0)

    floatx4m acc[MTL][R];
    uint2 yrp[RED ? MTL : 1][RED ? R : 1];
    uint4 pa[NPA];
    int wcur = 0;
    constexpr unsigned GBITS = (1u << 9) | (1u << 19) | (1u << 29);
    unsigned xs_[NPA], lvox[NPA];
    const unsigned cpb = (unsigned)(p_mine * PE * 2);
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int i = tid + NTHR * j;
        const int hv = i / CKP;
        const int hw = hv % HW;
        const int t = hv / HW;
        const int hh = t % HH;
        const int hd = t / HH;
        xs_[j] = (hv < HV) ? (unsigned)(hd | (hh << 10) | (hw << 20)) : 511u;
        lvox[j] = (unsigned)((hd * a.IH + hh) * a.IW + hw);
    }
    auto issue_wpiece = [&](int ch, bool live, int j) {
        const int q = tid + NTHR * j;
        if (q < WN && live) {                          // wave-uniform: WN is a multiple of 64
            const uint4* src = a.wpk + ((size_t)blockIdx.y * nchunks + ch) * WN + q;
            uint4* dstw = lw + (NWB == 2 ? (wcur ^ 1) * WN : 0) + (q - lane);
            const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)dstw));
            glds16(src, lbase);
        }
    };
    unsigned inb_mask = 0, c_lo = 0, c_hi = 0, rowb = 0;
    int brb = 0;
    __amdgpu_buffer_rsrc_t rs;
    auto issue_prep = [&](int brick, int ch, bool live) {
        const Org o = origin(brick);
        const int gd0 = o.d0 - PD, gh0 = o.h0 - 1, gw0 = o.w0 - 1;
        const int lod = max(0, -gd0), loh = max(0, -gh0), low = max(0, -gw0);
        const int hid = min(HD - 1, a.ID - 1 - gd0), hih = min(HH - 1, a.IH - 1 - gh0), hiw = min(HW - 1, a.IW - 1 - gw0);
        c_lo = GBITS - (unsigned)(lod | (loh << 10) | (low << 20));
        c_hi = GBITS + (unsigned)(hid | (hih << 10) | (hiw << 20));
        rowb = (unsigned)(a.xpitch * 2);
        brb = (int)(unsigned)((long long)((gd0 * a.IH + gh0) * a.IW + gw0) * rowb + (long long)ch * CK * 2);
        const size_t sample_bytes = (size_t)a.ID * a.IH * a.IW * rowb;
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)o.n * sample_bytes), 0, live ? (int)(unsigned)sample_bytes : 0, 0x00020000);
        inb_mask = 0;
    };
    auto issue_piece = [&](int j) {
        const unsigned in_lo = xs_[j] + c_lo, in_hi = c_hi - xs_[j];
        const bool ok = ((in_lo & in_hi) & GBITS) == GBITS;
        const int off = ok ? (int)(__umul24(lvox[j], rowb) + cpb + (unsigned)brb) : -1;
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        pa[j] = make_uint4(v[0], v[1], v[2], v[3]);
        inb_mask |= ok ? (1u << j) : 0u;
    };
    auto commit = [&](int ch) {
        float sc[PE], sh[PE], sl[PE];
        if (has_xf) {
            const int c0 = ch * CK + p_mine * PE;
#pragma unroll
            for (int e = 0; e < PE; ++e) { sc[e] = lxf[c0 + e]; sh[e] = lxf[a.Cin + c0 + e]; sl[e] = lxf[2 * a.Cin + c0 + e]; }
        }
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            const int i = tid + NTHR * j;
            if (i < HV * CKP) {
                uint4 v = pa[j];
                if (has_xf && ((inb_mask >> j) & 1u)) {
                    float f[PE];
                    F::unpack(v, f);
                    lrelu_affine<PE>(f, sc, sh, sl);
                    v = F::pack(f);
                }
                lact[p_mine * PSV + i / CKP] = v;
            }
        }
    };
    // per-channel sums: a lane holds channels 4 q4 .. + 3 of its voxel; the 16 lanes of a DPP row share them
    const bool want_stats = a.bn_partial != nullptr;
    float s1[MTL][4], s2[MTL][4];
#pragma unroll
    for (int m = 0; m < MTL; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[m][e] = s2[m][e] = 0.f;
    auto reduce4 = [&](float (&u_)[4], float (&v_)[4], int m) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float u = u_[e], v = v_[e];
            u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x128, 0xf, 0xf, false));
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
            u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x124, 0xf, 0xf, false));
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
            u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x122, 0xf, 0xf, false));
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));
            u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x121, 0xf, 0xf, false));
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
            if (n16 == 0) { float2* sl_ = (float2*)(lred + ((wave * CB) + 16 * m + 4 * q4 + e) * 2); float2 o = *sl_; o.x += u; o.y += v; *sl_ = o; }
            u_[e] = v_[e] = 0.f;
        }
    };
    auto flush_stats = [&](int row) {
        __syncthreads();
        if (tid < CB) {
            float l0 = 0.f, l1 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < NWAVE; ++w2) {
                const float2 t = *(const float2*)(lred + (w2 * CB + tid) * 2);
                l0 += t.x; l1 += t.y;
            }
            const int co = co0 + tid;
            if (co < a.Cout) {
                float* dstp = a.bn_partial + ((size_t)row * a.Cout + co) * 2;
                dstp[0] = l0;
                if constexpr (RED) dstp[1] = a.red_invstd[co] * (l1 - a.red_mean[co] * l0);
                else dstp[1] = l1;
            }
        }
    };

#ifdef BIU_DIAG
    unsigned long long tprev_ = __builtin_readcyclecounter();
    unsigned long long dsum_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0c_ = tprev_, t0r_ = __builtin_amdgcn_s_memrealtime();
#endif
    int k = 0;
    int brick = brick_of(0);
    if (brick >= nbricks) return;
    int ch = 0;
    issue_prep(brick, 0, true);
#pragma unroll
    for (int j = 0; j < NPA; ++j) issue_piece(j);
#pragma unroll
    for (int j = 0; j < NPW; ++j) issue_wpiece(0, true, j);
    __syncthreads();                         // lxf visible
    commit(0);
    if (NWB == 2) wcur ^= 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    while (true) {
        if (ch == 0) {
#pragma unroll
            for (int m = 0; m < MTL; ++m)
#pragma unroll
                for (int mt = 0; mt < R; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[m][mt][e] = lbias[16 * m + 4 * q4 + e];
        }
        int nbrick = brick, nch = ch + 1, nk = k;
        if (nch == nchunks) { nch = 0; nk = k + 1; nbrick = brick_of(nk); }
        const bool have_next = nbrick < nbricks;
        DIAG_STAMP(0);
        const uint4* lwc = lw + (NWB == 2 ? wcur * WN : 0) + lane;
        issue_prep(have_next ? nbrick : brick, nch, have_next);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ta = g / 3, tc = g % 3;
            if constexpr (RED) {
                // the upstream block's raw output at this lane's voxels, needed by the epilogue's BatchNorm-backward sums: requested at
                // the start of the brick's last MFMA phase instead of inside the epilogue (a global round trip per brick otherwise)
                if (g == 0 && ch == nchunks - 1 && want_stats) {
                    const Org o = origin(brick);
#pragma unroll
                    for (int mt = 0; mt < R; ++mt) {
                        const int gd = o.d0 + ld_w, gh = o.h0 + lh_w + mt, gw = o.w0 + n16;
                        const size_t vox = ((size_t)(o.n * a.OD + gd) * a.OH + gh) * a.OW + gw;
#pragma unroll
                        for (int m = 0; m < MTL; ++m) {
                            const int cl = co0 + 16 * m + 4 * q4;
                            const bool ok = cl < a.Cout && gd < a.GD && gh < a.GH && gw < a.GW;
                            yrp[m][mt] = ok ? *(const uint2*)((const T*)a.red_y + vox * a.red_ypitch + cl) : make_uint2(0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = pf_lo(g, NPA, NG); j < pf_lo(g + 1, NPA, NG); ++j) issue_piece(j);
#pragma unroll
            for (int j = pf_lo(g, NPW, NG); j < pf_lo(g + 1, NPW, NG); ++j) issue_wpiece(nch, have_next && nchunks > 1 && NWB == 2, j);
            uint4 rows[R + 2];
            const uint4* lap = lact + hvb + ta * HH * HW + tc;
#pragma unroll
            for (int j = 0; j < R + 2; ++j) rows[j] = lap[j * HW];
#pragma unroll
            for (int tb = 0; tb < 3; ++tb) {
                uint4 wf[MTL];
#pragma unroll
                for (int m = 0; m < MTL; ++m) wf[m] = lwc[(((ta * 3 + tb) * 3 + tc) * MTL + m) * 64];
#pragma unroll
                for (int mt = 0; mt < R; ++mt)
#pragma unroll
                    for (int m = 0; m < MTL; ++m)
                        acc[m][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[m]), __builtin_bit_cast(bf16x8, rows[mt + tb]), acc[m][mt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        DIAG_STAMP(1);
        DIAG_STAMP(2);
        if (ch == nchunks - 1) {
            // epilogue: lane = (voxel n16 of row mt, channels 16 m + 4 q4 .. + 3 of the block column): 8-byte stores.  The column's
            // destination follows k_conv_pipe: the second tensor of a split output when the column lies beyond osplit.
            const Org o = origin(brick);
            const bool o1 = a.y1 && co0 >= a.osplit;
            char* ybase = o1 ? a.y1 : a.y;
            const int ypitch_o = o1 ? a.ypitch1 : a.ypitch;
            const int coff = o1 ? a.osplit : 0;
            const int accum_o = o1 ? a.accumulate1 : a.accumulate;
#pragma unroll
            for (int m = 0; m < MTL; ++m) {
                const int cl = co0 + 16 * m + 4 * q4;
                const bool c_ok = cl < a.Cout;
                float rsc[4], rsh[4], rsl[4], t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (RED) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { rsc[e] = lrs[16 * m + 4 * q4 + e]; rsh[e] = lrs[CB + 16 * m + 4 * q4 + e]; rsl[e] = lrs[2 * CB + 16 * m + 4 * q4 + e]; }
                }
#pragma unroll
                for (int mt = 0; mt < R; ++mt) {
                    const int gd = o.d0 + ld_w, gh = o.h0 + lh_w + mt, gw = o.w0 + n16;
                    if (!(c_ok && gd < a.GD && gh < a.GH && gw < a.GW)) continue;
                    const size_t vox = ((size_t)(o.n * a.OD + gd) * a.OH + gh) * a.OW + gw;
                    Pack<T, 4> pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk.v[e] = (T)acc[m][mt][e];
                    uint2 piece = __builtin_bit_cast(uint2, pk);
                    uint2* dst = (uint2*)((T*)ybase + vox * ypitch_o + (cl - coff));
                    float f[4];
                    if (accum_o) {
                        const uint2 old = *dst;
                        f[0] = __uint_as_float(piece.x << 16) + __uint_as_float(old.x << 16);
                        f[1] = __uint_as_float(piece.x & 0xffff0000u) + __uint_as_float(old.x & 0xffff0000u);
                        f[2] = __uint_as_float(piece.y << 16) + __uint_as_float(old.y << 16);
                        f[3] = __uint_as_float(piece.y & 0xffff0000u) + __uint_as_float(old.y & 0xffff0000u);
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk.v[e] = (T)f[e];
                        piece = __builtin_bit_cast(uint2, pk);
                    }
                    *dst = piece;
                    if (want_stats) {
                        f[0] = __uint_as_float(piece.x << 16); f[1] = __uint_as_float(piece.x & 0xffff0000u);      // values as stored
                        f[2] = __uint_as_float(piece.y << 16); f[3] = __uint_as_float(piece.y & 0xffff0000u);
                        if constexpr (!RED) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { s1[m][e] += f[e]; s2[m][e] = fmaf(f[e], f[e], s2[m][e]); }
                        } else {
                            const uint2 yr = yrp[m][mt];
                            const float yv[4] = {__uint_as_float(yr.x << 16), __uint_as_float(yr.x & 0xffff0000u), __uint_as_float(yr.y << 16),
                                                 __uint_as_float(yr.y & 0xffff0000u)};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float tt = fmaf(rsc[e], yv[e], rsh[e]);
                                const float dz = f[e] * (tt > 0.f ? 1.f : rsl[e]);
                                t1[e] += dz;
                                t2[e] = fmaf(dz, yv[e], t2[e]);
                            }
                        }
                    }
                }
                if constexpr (RED) {
                    if (want_stats) reduce4(t1, t2, m);
                }
            }
        }
        DIAG_STAMP(3);
#ifdef BIU_DIAG
        dsum_[7] += 1;
        if (!have_next && a.diag && tid == 0) {
            for (int q_ = 0; q_ < 8; ++q_) atomicAdd(a.diag + q_, dsum_[q_]);
            atomicAdd(a.diag + 8, __builtin_readcyclecounter() - t0c_);
            atomicAdd(a.diag + 9, __builtin_amdgcn_s_memrealtime() - t0r_);
        }
#endif
        if (!have_next) break;
        __syncthreads();                     // everyone is done reading the tile
        DIAG_STAMP(4);
        commit(nch);
        if (nchunks > 1 && NWB == 2) wcur ^= 1;               // (one chunk per brick: the slab loaded for the first item stays)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // weight DMA landed
        DIAG_STAMP(5);
        __syncthreads();
        DIAG_STAMP(6);
        brick = nbrick; ch = nch; k = nk;
    }
    if constexpr (!RED) {
        if (want_stats) {
#pragma unroll
            for (int m = 0; m < MTL; ++m) reduce4(s1[m], s2[m], m);
        }
    }
    if (want_stats) flush_stats((int)blockIdx.x);
}

// packed weights of the 16-row variant: out[col][kstep32][tap][m][lane]; lane (n = l & 15, q = l >> 4) of tile (col, m) holds
// W[row 16 (col * mtl + m) + n][k = 32 ks + 8 q + e]   (mtl = row tiles per block column: k_conv16_pipe's MTL)
__global__ void k_pack_weights16(const float* __restrict__ w, int cin, int cout, int taps, int kind, int Kc, int Nc, int mtl, uint4* __restrict__ out) {
    using F = Frag<bf16_t>;
    const int nKS = Kc / 32, ncol = (Nc + 16 * mtl - 1) / (16 * mtl);
    const size_t total = (size_t)ncol * nKS * taps * mtl * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx % 64);
        size_t t = idx / 64;
        const int m = (int)(t % mtl); t /= mtl;
        const int tap = (int)(t % taps); t /= taps;
        const int ks = (int)(t % nKS);
        const int col = (int)(t / nKS);
        const int i = (col * mtl + m) * 16 + (lane & 15);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ks * 32 + (lane >> 4) * 8 + e;
            float v = 0.f;
            if (i < Nc && k < Kc) {
                if (kind == 0) v = w[((size_t)i * cin + k) * taps + tap];                 // W[co=i][ci=k][tap]
                else v = w[((size_t)k * cin + i) * taps + (taps - 1 - tap)];              // W[co=k][ci=i][flipped tap]
            }
            f[e] = v;
        }
        out[idx] = F::pack(f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight packing: out[ntile][kstep][tap][lane] (16 B each); lane (r, h) holds W[i = 32*ntile + r][k = KS*kstep + PE*h + e]
// ---------------------------------------------------------------------------------------------------------------
// bf16x3 image of an fp32 weight fragment (k_conv_pipe<f32x3_t>): the k-steps 2s and 2s+1 of a lane hold the hi halves and the lo halves
// of its 8 weights k = 8 (2s + (u >> 2)) + 4 hf + (u & 3), u = 0..7 -- the order the kernel stages the activations in
// (bf16x6: three k-steps 3s, 3s+1, 3s+2 = hi, mid, lo of the same 16 channels)
template <int XP, typename Load>
__device__ __forceinline__ uint4 xs_weight_piece(int ks, int hf, Load&& wload) {
    const int s16 = ks / XP, part = ks % XP;
    float f0[4], f1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { f0[e] = wload(s16 * 16 + hf * 4 + e); f1[e] = wload(s16 * 16 + 8 + hf * 4 + e); }
    uint2 p0[XP], p1[XP];
    split_bf16<XP>(f0, p0);
    split_bf16<XP>(f1, p1);
    uint4 r = make_uint4(p0[0].x, p0[0].y, p1[0].x, p1[0].y);
#pragma unroll
    for (int k = 1; k < XP; ++k) if (part == k) r = make_uint4(p0[k].x, p0[k].y, p1[k].x, p1[k].y);
    return r;
}
template <typename Load>
__device__ __forceinline__ uint4 x3_weight_piece(int mode, int ks, int hf, Load&& wload) {       // mode 1 = bf16x3, 2 = bf16x6
    return mode == 2 ? xs_weight_piece<3>(ks, hf, wload) : xs_weight_piece<2>(ks, hf, wload);
}

template <typename T>
__global__ void k_pack_weights(const float* __restrict__ w, int cin, int cout, int taps, int kind, int Kc, int Nc,
                               int nKS, int ntiles, uint4* __restrict__ out, int x3) {
    using F = Frag<T>;
    constexpr int PE = F::PE;
    const size_t total = (size_t)ntiles * nKS * taps * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx % 64);
        size_t t = idx / 64;
        const int tap = (int)(t % taps); t /= taps;
        const int ks = (int)(t % nKS);
        const int nt = (int)(t / nKS);
        const int i = nt * 32 + (lane & 31);
        if constexpr (sizeof(T) == 4) {
            if (x3) {
                out[idx] = x3_weight_piece(x3, ks, lane >> 5, [&](int k) -> float {
                    if (i >= Nc || k >= Kc) return 0.f;
                    return kind == 0 ? w[((size_t)i * cin + k) * taps + tap] : w[((size_t)k * cin + i) * taps + (taps - 1 - tap)];
                });
                continue;
            }
        }
        float f[PE];
#pragma unroll
        for (int e = 0; e < PE; ++e) {
            const int k = ks * 2 * PE + (lane >> 5) * PE + e;
            float v = 0.f;
            if (i < Nc && k < Kc) {
                if (kind == 0) v = w[((size_t)i * cin + k) * taps + tap];                 // W[co=i][ci=k][tap]
                else v = w[((size_t)k * cin + i) * taps + (taps - 1 - tap)];              // W[co=k][ci=i][flipped tap]
            }
            f[e] = v;
        }
        out[idx] = F::pack(f);
    }
}

static inline int ks_of(int dtype) { return dtype == BIU_BF16 ? 16 : 8; }

// buffer descriptors address one sample with 32-bit byte offsets (0xFFFFFFFF marks a padding piece)
constexpr i64 BIU_MAX_SAMPLE_BYTES = (1LL << 32) - 65536;
static inline i64 sample_bytes(const biu_act* t, size_t es) { return (i64)t->d * t->h * t->w * t->pitch * (i64)es; }

static bool chan_ok(int K, int Nn, int dtype) { return K >= 16 && K % ks_of(dtype) == 0 && Nn >= 16 && Nn % 8 == 0; }

// 16-channel variant (k_conv16_pipe): eligible by channels alone, so the packed buffer carries its fragment image behind the regular one
static bool m16_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("BIU_DISABLE"); v = (e && strstr(e, "m16")) ? 1 : 0; }
    return v == 1;
}
// How the fp32 2-D 3x3 convolutions and ConvTranspose k2 -- forward, data gradient (reduction channels in chunks of 16) and weight gradient --
// multiply (biu_set_fp32_products(mode), or BIU_FP32_PRODUCTS=exact|bf16x3|bf16x6 in the environment):
//   2 = bf16x6 (DEFAULT): fp32-grade split products on the bf16 matrix pipe;  0 = exact: v_mfma_f32_32x32x2_f32;  1 = bf16x3 (opt-in, <= 2^-15 per product).
// The packed weights of such a layer carry the image of the mode (x3_weight_piece).  The mode is fixed by its first use in the process: packed
// images of one mode must never meet launches of another.
static int g_x3_mode = -1;           // -1: not decided yet
static bool g_x3_used = false;
static std::mutex g_x3_mu;
static int fp32_split_mode() {
    std::lock_guard<std::mutex> lock(g_x3_mu);
    if (g_x3_mode < 0) {
        const char* e = getenv("BIU_FP32_PRODUCTS");
        g_x3_mode = (e && strstr(e, "bf16x3")) ? 1 : (e && strstr(e, "exact")) ? 0 : 2;
    }
    g_x3_used = true;
    return g_x3_mode;
}
int biu_mfma_set_fp32_products(int mode) {
    std::lock_guard<std::mutex> lock(g_x3_mu);
    if (mode < 0 || mode > 2) return biu_fail(BIU_ERR_UNSUPPORTED, "set_fp32_products: mode %d (0 = exact fp32 MFMA, 1 = bf16x3, 2 = bf16x6)", mode);
    if (g_x3_used && g_x3_mode != mode)
        return biu_fail(BIU_ERR_UNSUPPORTED, "set_fp32_products: fp32 kernels already ran in another mode (set it before the first forward)");
    g_x3_mode = mode;
    return BIU_OK;
}
// split mode of a launch with K reduction channels (0: the exact fp32 kernels take it)
static int x3_ok(int K, int kd, int dtype) { return (dtype == BIU_F32 && kd == 1 && K >= 16 && K % 16 == 0) ? fp32_split_mode() : 0; }
// k-steps (packed weight fragments per tap and row tile) of a layer with K reduction channels
static int nks_of(int K, int kd, int dtype) { return x3_ok(K, kd, dtype) == 2 ? K / 16 * 3 : K / (dtype == BIU_BF16 ? 16 : 8); }
// 16-row tiles per block column of the 16x16x32 kernel for a layer with K reduction and Nn output channels (0: the layer does not take it):
// 1 for a 16-channel output; 2 for 32-channel tiles when the reduction is ONE 32-channel chunk (its weight slab stays resident in LDS) --
// the data gradient of a layer with 32 output channels (decode5 of cfg4: dy 32 ch -> dx 64 | 32), 32 -> 32 layers both ways.
// BIU_DISABLE=m16 switches all of it off, =m16x2 only the two-tile form.
static int m16_mtl(int K, int Nn, int dtype) {
    if (dtype != BIU_BF16 || K < 32 || K % 32 != 0) return 0;
    if (Nn == 16) return 1;
    static int x2off = -1;
    if (x2off < 0) { const char* e = getenv("BIU_DISABLE"); x2off = (e && strstr(e, "m16x2")) ? 1 : 0; }
    if (!x2off && K == 32 && Nn >= 32 && Nn % 32 == 0) return 2;
    return 0;
}
static bool m16_chan_ok(int K, int Nn, int dtype) { return m16_mtl(K, Nn, dtype) > 0; }
static size_t m16_packed_bytes(int K, int Nn, int taps, int dtype) {
    const int mtl = m16_mtl(K, Nn, dtype);
    return mtl ? (size_t)((Nn + 16 * mtl - 1) / (16 * mtl)) * (K / 32) * taps * mtl * 1024 : 0;
}
static size_t regular_packed_bytes(int K, int Nn, int taps, int dtype) {
    const size_t ntiles = (Nn + 31) / 32, nKS = nks_of(K, taps / 9, dtype);
    return ntiles * nKS * (size_t)taps * 1024;
}

size_t biu_mfma_packed_bytes(int kind, int cin, int cout, int kd, int kh, int kw, int dilation, int dtype) {
    if (dilation != 1 || kh != 3 || kw != 3 || (kd != 1 && kd != 3)) return 0;
    if (dtype != BIU_BF16 && dtype != BIU_F32) return 0;
    const int K = kind == 0 ? cin : cout, Nn = kind == 0 ? cout : cin;
    if (!chan_ok(K, Nn, dtype)) return 0;
    size_t b = regular_packed_bytes(K, Nn, kd * 9, dtype);
    b += m16_packed_bytes(K, Nn, kd * 9, dtype);
    return b;
}

// w: (Cout, cin_stride, taps) with the layer's `cin` input channels starting at w (a channel slice of a wider weight tensor: pass
// w + first_channel * taps and the full tensor's channel count as cin_stride)
static int mfma_pack_strided(int kind, const float* w, int cin_stride, int cin, int cout, int kd, int kh, int kw, int dtype, void* packed, hipStream_t st) {
    const int K = kind == 0 ? cin : cout, Nn = kind == 0 ? cout : cin;
    const int taps = kd * kh * kw, ntiles = (Nn + 31) / 32, nKS = nks_of(K, kd, dtype);
    const size_t total = (size_t)ntiles * nKS * taps * 64;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_pack_weights<T>, dim3(grid_for((i64)total, 256, 4096)), dim3(256), 0, st, w, cin_stride,
                                                 cout, taps, kind, K, Nn, nKS, ntiles, (uint4*)packed, x3_ok(K, kd, dtype)));
    BIU_CHECK_LAUNCH("pack_weights");
    if (m16_chan_ok(K, Nn, dtype)) {
        const size_t total16 = m16_packed_bytes(K, Nn, taps, dtype) / 16;
        hipLaunchKernelGGL(k_pack_weights16, dim3(grid_for((i64)total16, 256, 4096)), dim3(256), 0, st, w, cin_stride, cout, taps, kind, K, Nn,
                           m16_mtl(K, Nn, dtype), (uint4*)((char*)packed + regular_packed_bytes(K, Nn, taps, dtype)));
        BIU_CHECK_LAUNCH("pack_weights16");
    }
    return BIU_OK;
}
int biu_mfma_pack(int kind, const float* w, int cin, int cout, int kd, int kh, int kw, int dtype, void* packed, hipStream_t st) {
    return mfma_pack_strided(kind, w, cin, cin, cout, kd, kh, kw, dtype, packed, st);
}

bool biu_mfma_conv_ok(const biu_act* x, const biu_act* y, int kd, int kh, int kw, int dilation, int dtype) {
    if (dilation != 1 || kh != 3 || kw != 3 || (kd != 1 && kd != 3)) return false;
    if (!chan_ok(x->c, y->c, dtype)) return false;
    const size_t es = dsize(dtype);
    if ((uintptr_t)x->p % 16 || (uintptr_t)y->p % 16 || ((size_t)x->pitch * es) % 16 || ((size_t)y->pitch * es) % 16) return false;
    if (nvox(x) * (i64)x->pitch >= (1LL << 31) || nvox(y) * (i64)y->pitch >= (1LL << 31)) return false;   // 32-bit voxel index math
    if (sample_bytes(x, es) >= BIU_MAX_SAMPLE_BYTES || sample_bytes(y, es) >= BIU_MAX_SAMPLE_BYTES) return false;            // 32-bit buffer offsets per sample
    if (kd == 1 && x->d != 1) return false;
    return true;
}

template <int KD, int KHW, int S, int TD, int TH, int TW>
struct BrickGeo {
    static constexpr int SD = (KD == 1) ? 1 : S;
    static constexpr int HV = ((TD - 1) * SD + KD) * ((TH - 1) * S + KHW) * ((TW - 1) * S + KHW);
};

static int num_cus() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// Blocks per column of a persistent launch with `cols` columns (channel tiles x splits): a multiple of 8 keeps the block id's XCD equal
// to blockIdx.x mod 8 (halo neighbours of the brick walk share an L2), but rounding 85 down to 80 leaves 6 % of the CUs without a
// block -- the exact quotient wins when the rounding would idle more than 1/32 of the chip (the walk then falls back to plain order).
static int grid_per_column(int budget, int cols) {
    const int exact = budget / cols, r8 = exact & ~7;
    int g = ((exact - r8) * cols * 32 > budget) ? exact : r8;
    if (g < 8) g = 8;
    return g;
}

template <typename T, int KD, int KHW, int S, int TD, int TH, int TW, int NT, int CKP, bool RED, int NW>
static int launch_cfg_r(const ConvArgs& a0, int ntiles, int nz, hipStream_t st) {
    ConvArgs a = a0;
    constexpr int HV = BrickGeo<KD, KHW, S, TD, TH, TW>::HV;
    constexpr int PSV = cpad_planes(HV, CKP);
    constexpr int XP = SplitOf<T>::parts;
    constexpr int SPC = XP ? XP : CKP / 2, ACT16 = XP ? 2 * XP * PSV : CKP * PSV;                    // must mirror k_conv_pipe
    constexpr int WN = KD * KHW * KHW * SPC * NT * 64;
    constexpr int NWB = ((size_t)(ACT16 + 2 * WN) * 16 <= conv_lds_budget(NW)) ? 2 : 1;             // must mirror k_conv_pipe::W2
    const size_t lds_bytes = (size_t)(ACT16 + NWB * WN) * 16 + (size_t)3 * a.Cin * sizeof(float) + (size_t)NT * 32 * (2 * NW + 4) * sizeof(float);
    if (lds_bytes > (size_t)(NW == 8 ? 160 : 80) * 1024) return biu_fail(BIU_ERR_UNSUPPORTED, "conv_pipe: %zu bytes of LDS (Cin=%d)", lds_bytes, a.Cin);
    a.nbd = (a.GD + TD - 1) / TD;
    a.nbh = (a.GH + TH - 1) / TH;
    a.nbw = (a.GW + TW - 1) / TW;
    const int nbricks = a.N * a.nbd * a.nbh * a.nbw;
    const int gy = ntiles / NT;
    int g = grid_per_column((NW == 8 ? 1 : 2) * num_cus(), gy * nz);
    if (g > nbricks) g = nbricks;
    dim3 grid((unsigned)g, (unsigned)gy, (unsigned)nz);
    auto kern = k_conv_pipe<T, KD, KHW, S, TD, TH, TW, NT, CKP, RED, NW>;
    static size_t attr_set = 0;
    if (attr_set < lds_bytes) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return biu_fail(BIU_ERR_LAUNCH, "conv_pipe: cannot reserve %zu bytes of LDS", lds_bytes);
        attr_set = lds_bytes;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds_bytes, st, a);
    BIU_CHECK_LAUNCH("conv_pipe");
    return BIU_OK;
}

template <typename T, int KD, int KHW, int S, int TD, int TH, int TW, int NT, int CKP, int NW = 8>
static int launch_cfg(const ConvArgs& a, int ntiles, int nz, hipStream_t st) {
    // the BatchNorm-backward epilogue only exists for stride-1 3x3(x3) data gradients and stride-2 ConvT data gradients
    if constexpr ((S == 1 && KHW == 3) || S == 2 || (KD == 2 && KHW == 2 && S == 1)) {
        if (a.red_mode) return launch_cfg_r<T, KD, KHW, S, TD, TH, TW, NT, CKP, true, NW>(a, ntiles, nz, st);
    } else {
        if (a.red_mode) return biu_fail(BIU_ERR_UNSUPPORTED, "conv_pipe: no fused BatchNorm-backward epilogue for this kernel shape");
    }
    return launch_cfg_r<T, KD, KHW, S, TD, TH, TW, NT, CKP, false, NW>(a, ntiles, nz, st);
}


// (a 4-tile weight slab no longer fits the double buffer next to the activation tile: two tiles is the widest block)
static inline int pick_nt(int ntiles) { return (ntiles % 2 == 0) ? 2 : 1; }

// brick shape per (kernel kind, NT, width class): must stay in sync between launch_* and conv_brick_voxels()
struct BrickDim { int td, th, tw; };
static BrickDim conv3_brick(int kd, int nt, bool wide) {
    if (kd == 3) {
        if (nt == 1) return wide ? BrickDim{4, 8, 32} : BrickDim{4, 16, 16};
        return BrickDim{4, 8, 16};
    }
    if (nt == 1) return wide ? BrickDim{1, 32, 32} : BrickDim{1, 64, 16};
    return wide ? BrickDim{1, 16, 32} : BrickDim{1, 32, 16};
}

// Input-channel split of a 3x3(x3) fp32 launch (ConvArgs::ksplit): when bricks x channel tiles fill under half of the CUs and every
// split still has >= 4 chunks of 8 channels.  1 = no split.  BIU_DISABLE=ksplit switches it off.
int biu_mfma_conv_ksplit(int cin, const biu_act* y, int kd, int dtype) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("BIU_DISABLE"); off = (e && strstr(e, "ksplit")) ? 1 : 0; }
    if (off || dtype != BIU_F32 || (kd != 1 && kd != 3)) return 1;
    const int ntiles = (y->c + 31) / 32, nt = pick_nt(ntiles);
    const BrickDim b = conv3_brick(kd, nt, y->w % 32 == 0);
    const long blocks = (long)y->n * ((y->d + b.td - 1) / b.td) * ((y->h + b.th - 1) / b.th) * ((y->w + b.tw - 1) / b.tw) * (ntiles / nt);
    const int nchunks = x3_ok(cin, kd, dtype) ? cin / 16 : cin / 8;
    int ks = 1;
    while (blocks * ks * 2 <= num_cus() && nchunks / (ks * 2) >= 4 && ks < 32) ks *= 2;
    return ks;
}
// Scratch of a split launch: ks slices laid out like y (then ks like y1), owned by the CALLER (include/biu.h, biu_conv_split_workspace).
size_t biu_mfma_conv_split_bytes(int cin, const biu_act* y, const biu_act* y1, int kd, int dtype) {
    biu_act yall = *y;
    if (y1) yall.c = y->c + y1->c;
    const int ks = biu_mfma_conv_ksplit(cin, &yall, kd, dtype);
    if (ks <= 1) return 0;
    const size_t sl0 = (size_t)nvox(y) * y->pitch * sizeof(float), sl1 = y1 ? (size_t)nvox(y1) * y1->pitch * sizeof(float) : 0;
    return (size_t)ks * (sl0 + sl1);
}

// y[v][c] = (accumulate ? y[v][c] : 0) + sum_z ws[z][v][c]   (ws slices share y's pitch)
__global__ void k_split_reduce(const float* __restrict__ ws, size_t zstride_f, int ks, float* __restrict__ y, long nvoxels, int c, int pitch, int accumulate) {
    const long total = nvoxels * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const size_t o = (size_t)(i / c) * pitch + (size_t)(i % c);
        float v = accumulate ? y[o] : 0.f;
        for (int z = 0; z < ks; ++z) v += ws[z * zstride_f + o];
        y[o] = v;
    }
}

template <typename T>
static int launch_conv(const ConvArgs& a, int kd, hipStream_t st) {
    const int ntiles = (a.Cout + 31) / 32;
    const int nt = pick_nt(ntiles);
    const bool wide = (a.GW % 32 == 0);
    const int nz = a.ksplit > 1 ? a.ksplit : 1;
    if (kd == 3) {
        if (nt == 1) return wide ? launch_cfg<T, 3, 3, 1, 4, 8, 32, 1, 2>(a, ntiles, nz, st) : launch_cfg<T, 3, 3, 1, 4, 16, 16, 1, 2>(a, ntiles, nz, st);
        return launch_cfg<T, 3, 3, 1, 4, 8, 16, 2, 2>(a, ntiles, nz, st);
    }
    if (nt == 1) return wide ? launch_cfg<T, 1, 3, 1, 1, 32, 32, 1, 2>(a, ntiles, nz, st) : launch_cfg<T, 1, 3, 1, 1, 64, 16, 1, 2>(a, ntiles, nz, st);
    return wide ? launch_cfg<T, 1, 3, 1, 1, 16, 32, 2, 2>(a, ntiles, nz, st) : launch_cfg<T, 1, 3, 1, 1, 32, 16, 2, 2>(a, ntiles, nz, st);
}

// fp32 2-D 3x3 as split bf16 products (x3_ok): the bricks of launch_conv<float>, 16-channel chunks
template <typename TS>
static int launch_conv_xs(const ConvArgs& a, hipStream_t st) {
    const int ntiles = (a.Cout + 31) / 32;
    const int nt = pick_nt(ntiles);
    const bool wide = (a.GW % 32 == 0);
    const int nz = a.ksplit > 1 ? a.ksplit : 1;
    // (bf16x6, measured: one output tile per block with a double-buffered weight slab is 5-7 % slower than two tiles with the single slab)
    if (nt == 1) return wide ? launch_cfg<TS, 1, 3, 1, 1, 32, 32, 1, 4>(a, ntiles, nz, st) : launch_cfg<TS, 1, 3, 1, 1, 64, 16, 1, 4>(a, ntiles, nz, st);
    return wide ? launch_cfg<TS, 1, 3, 1, 1, 16, 32, 2, 4>(a, ntiles, nz, st) : launch_cfg<TS, 1, 3, 1, 1, 32, 16, 2, 4>(a, ntiles, nz, st);
}
static int launch_conv_x3(const ConvArgs& a, int mode, hipStream_t st) { return mode == 2 ? launch_conv_xs<f32x6_t>(a, st) : launch_conv_xs<f32x3_t>(a, st); }

// number of bricks of a ConvTranspose data-gradient launch on the coarse tensor dx
int biu_mfma_convt_dgrad_bricks(const biu_act* dx, int kd) {
    const int td = (kd == 2) ? 2 : 1, th = (kd == 2) ? 8 : 16, tw = 16;
    return dx->n * ((dx->d + td - 1) / td) * ((dx->h + th - 1) / th) * ((dx->w + tw - 1) / tw);
}

// partial rows its fused BatchNorm-backward sums occupy: one per workgroup column (must mirror launch_cfg_r's grid computation)
int biu_mfma_convt_dgrad_rows(const biu_act* dx, int kd) {
    const int ntiles = (dx->c + 31) / 32, gy = ntiles / pick_nt(ntiles);
    int g = grid_per_column(num_cus(), gy);
    const int nbricks = biu_mfma_convt_dgrad_bricks(dx, kd);
    return g > nbricks ? nbricks : g;
}

// the 16-row kernel takes a launch when the channels fit (m16_mtl), the input is one tensor and it is not switched off
static bool m16_ok(const biu_act* x, const biu_act* y, int dtype) {
    return x && !m16_disabled() && m16_chan_ok(x->c, y->c, dtype);
}
static BrickDim m16_brick(int kd) { return kd == 3 ? BrickDim{4, 8, 16} : BrickDim{1, 32, 16}; }
static int bricks_of(const biu_act* y, BrickDim b) {
    return y->n * ((y->d + b.td - 1) / b.td) * ((y->h + b.th - 1) / b.th) * ((y->w + b.tw - 1) / b.tw);
}
// blocks per column of a 16-row launch with `cols` block columns (multiple of 8: XCD-grouped brick walk)
static int m16_grid_x(int cols) { return grid_per_column(num_cus(), cols); }      // (3 columns: 85 blocks each, not 80 -- 16 CUs would idle)

// number of bricks of a 3x3(x3) launch writing y (= BatchNorm-backward partial rows of the data-gradient kernels).  x (the tensor the
// launch reads) and dtype select the kernel; without them the count is an upper bound over the kernels that could run (buffer sizing).
int biu_mfma_conv_bricks(const biu_act* y, int kd, const biu_act* x, int dtype) {
    const int ntiles = (y->c + 31) / 32;
    const int reg = bricks_of(y, conv3_brick(kd, pick_nt(ntiles), y->w % 32 == 0));
    if (x) return m16_ok(x, y, dtype) ? bricks_of(y, m16_brick(kd)) : reg;
    const int m16 = (y->c == 16 || y->c % 32 == 0) ? bricks_of(y, m16_brick(kd)) : 0;
    return reg > m16 ? reg : m16;
}
// number of workgroup columns of that launch (= BatchNorm statistics partial rows of the forward kernels: one per block);
// must mirror launch_cfg_r's / launch_conv16's grid computation
int biu_mfma_conv_stat_rows(const biu_act* y, int kd, const biu_act* x, int dtype, bool red) {
    if (x && kd == 3 && biu_conv_roll_ok(x, y, dtype, false, 0, red)) return biu_conv_roll_rows(x, y, dtype);
    if (m16_ok(x, y, dtype)) {
        const int g = m16_grid_x(y->c / (16 * m16_mtl(x->c, y->c, dtype)));
        const int nbricks = bricks_of(y, m16_brick(kd));
        return g > nbricks ? nbricks : g;
    }
    const int ntiles = (y->c + 31) / 32;
    const int nt = pick_nt(ntiles);
    int g = grid_per_column(num_cus(), ntiles / nt);
    const int nbricks = biu_mfma_conv_bricks(y, kd, x ? x : y, x ? dtype : -1);
    return g > nbricks ? nbricks : g;
}

template <int KD, int TD, int TH, int MTL>
static int launch_conv16(ConvArgs a, hipStream_t st) {
    constexpr int HV = (TD + KD - 1) * (TH + 2) * 18;
    constexpr int PSV = (HV + 15) & ~15;
    constexpr int WN = KD * 9 * MTL * 64;
    constexpr int NWB = MTL == 1 ? 2 : 1;
    const size_t lds_bytes = (size_t)(4 * PSV + NWB * WN) * 16 + (size_t)3 * a.Cin * sizeof(float) + (size_t)(8 * 16 * MTL * 2 + 16 * MTL + 48 * MTL) * sizeof(float);
    if (lds_bytes > (size_t)160 * 1024) return biu_fail(BIU_ERR_UNSUPPORTED, "conv16_pipe: %zu bytes of LDS (Cin=%d)", lds_bytes, a.Cin);
    if (MTL > 1 && a.Cin != 32) return biu_fail(BIU_ERR_UNSUPPORTED, "conv16_pipe: the two-tile form takes single-chunk layers (Cin=%d)", a.Cin);
    a.nbd = (a.GD + TD - 1) / TD;
    a.nbh = (a.GH + TH - 1) / TH;
    a.nbw = (a.GW + 15) / 16;
    const int nbricks = a.N * a.nbd * a.nbh * a.nbw;
    const int cols = a.Cout / (16 * MTL);
    int g = m16_grid_x(cols);
    if (g > nbricks) g = nbricks;
    auto launch = [&](auto kern) -> int {
        static size_t attr_set = 0;
        if (attr_set < lds_bytes) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
                return biu_fail(BIU_ERR_LAUNCH, "conv16_pipe: cannot reserve %zu bytes of LDS", lds_bytes);
            attr_set = lds_bytes;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)g, (unsigned)cols), dim3(512), lds_bytes, st, a);
        BIU_CHECK_LAUNCH("conv16_pipe");
        return BIU_OK;
    };
    if (a.red_mode) return launch(k_conv16_pipe<KD, TD, TH, MTL, true>);
    return launch(k_conv16_pipe<KD, TD, TH, MTL, false>);
}

static void clear_cat(ConvArgs& a) {
    a.fold = 0;
    a.ksplit = 1; a.y_zstride = a.y1_zstride = 0;
    a.x1 = nullptr; a.xpitch1 = 0; a.csplit = 0;
    a.xs1 = a.xb1 = a.xl1 = nullptr;
    a.y1 = nullptr; a.ypitch1 = 0; a.osplit = 0; a.accumulate1 = 0;
}

static int fill_xf(ConvArgs& a, const biu_xform* xf) {
    const bool has = xf && (xf->scale || xf->shift || xf->slope);
    if (has) BIU_REQUIRE(xf->scale && xf->shift && xf->slope, BIU_ERR_UNSUPPORTED, "conv_mfma: partial biu_xform (need all three vectors)");
    a.xs = has ? xf->scale : nullptr;
    a.xb = has ? xf->shift : nullptr;
    a.xl = has ? xf->slope : nullptr;
    return BIU_OK;
}

// channel-concatenated input (x0 | x1) -> y through the two-source forms of the three MFMA kernels
bool biu_mfma_conv_cat_ok(const biu_act* x0, const biu_act* x1, const biu_act* y, int kd, int kh, int kw, int dilation, int dtype) {
    if (!biu_mfma_conv_ok(x0, y, kd, kh, kw, dilation, dtype) || !biu_mfma_conv_ok(x1, y, kd, kh, kw, dilation, dtype)) return false;
    if (!biu_mfma_wgrad_ok(x0, y, kd, kh, kw, dilation, dtype) || !biu_mfma_wgrad_ok(x1, y, kd, kh, kw, dilation, dtype)) return false;
    // x0's channels must end on a chunk boundary (forward), a 32-wide tile (weight gradient) and on the data gradient's
    // block tile, which is 64 wide when the total tile count is even
    const int ntiles = (x0->c + x1->c + 31) / 32;
    const int need = pick_nt(ntiles) * 32;
    if (x0->c % need != 0 || x1->c % 32 != 0) return false;
    const i64 plane = (i64)x0->h * x0->w;
    return 10 * plane < (1LL << 24);                    // tile-local voxel offsets go through 24-bit multiplies
}

int biu_mfma_conv(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, int kd, int kh, int kw,
                  const biu_act* y, int accumulate, float* bn_partial, int dtype, hipStream_t st, const BnRedFuse* red, const ConvCat* cat,
                  void* split_ws, size_t split_ws_bytes) {
    ConvArgs a;
    clear_cat(a);
    a.bn_partial = bn_partial;
    a.red_mode = 0; a.red_y = nullptr; a.red_ypitch = 0;
    a.red_scale = a.red_shift = a.red_slope = a.red_mean = a.red_invstd = nullptr;
    if (red) {
        a.red_mode = 1; a.red_y = (const char*)red->y->p; a.red_ypitch = red->y->pitch;
        a.red_scale = red->scale; a.red_shift = red->shift; a.red_slope = red->slope; a.red_mean = red->mean; a.red_invstd = red->invstd;
    }
    a.x = (const char*)x->p;
    a.y = (char*)y->p;
    a.wpk = (const uint4*)packed;
    a.bias = bias;
    int rc = fill_xf(a, xf);
    if (rc) return rc;
    a.xpitch = x->pitch;
    a.ypitch = y->pitch;
    a.N = x->n;
    a.GD = a.ID = a.OD = x->d; a.GH = a.IH = a.OH = x->h; a.GW = a.IW = a.OW = x->w;
    a.osd = a.osh = a.osw = 1;
    a.Cin = x->c; a.Cout = y->c;
    if (cat && cat->x1) {                       // input = concat(x, x1) along the channels
        a.x1 = (const char*)cat->x1->p; a.xpitch1 = cat->x1->pitch; a.csplit = x->c;
        a.Cin = x->c + cat->x1->c;
        const biu_xform* f1 = cat->xf1;
        const bool has1 = f1 && (f1->scale || f1->shift || f1->slope);
        if (has1) BIU_REQUIRE(f1->scale && f1->shift && f1->slope, BIU_ERR_UNSUPPORTED, "conv_mfma: partial biu_xform (second source)");
        a.xs1 = has1 ? f1->scale : nullptr; a.xb1 = has1 ? f1->shift : nullptr; a.xl1 = has1 ? f1->slope : nullptr;
    }
    if (cat && cat->y1) {                       // output = concat(y, y1) along the channels
        a.y1 = (char*)cat->y1->p; a.ypitch1 = cat->y1->pitch; a.osplit = y->c; a.accumulate1 = cat->accumulate1;
        a.Cout = y->c + cat->y1->c;
    }
    a.nKS = nks_of(a.Cin, kd, dtype);
    a.wz_stride = 0;
    a.accumulate = accumulate;
    a.nbd = a.nbh = a.nbw = 0;
#ifdef BIU_DIAG
    a.diag = biu_diag_buffer;
#else
    a.diag = nullptr;
#endif
    a.ksplit = 1;
    a.y_zstride = a.y1_zstride = 0;
    if (!bn_partial && !red) {
        biu_act yall = *y;
        yall.c = a.Cout;
        const int ks = biu_mfma_conv_ksplit(a.Cin, &yall, kd, dtype);
        if (ks > 1) {
            // scratch for the partial results: slices laid out like y (and y1), summed into the outputs afterwards
            const biu_act* y1t = (cat && cat->y1) ? cat->y1 : nullptr;
            const size_t sl0 = (size_t)nvox(y) * y->pitch * sizeof(float), sl1 = y1t ? (size_t)nvox(y1t) * y1t->pitch * sizeof(float) : 0;
            char* ws = (split_ws && split_ws_bytes >= (size_t)ks * (sl0 + sl1)) ? (char*)split_ws : nullptr;     // no scratch given: unsplit launch
            if (ws) {
                ConvArgs b = a;
                b.ksplit = ks;
                b.y = ws; b.y_zstride = sl0; b.accumulate = 0;
                if (y1t) { b.y1 = ws + (size_t)ks * sl0; b.y1_zstride = sl1; b.accumulate1 = 0; }
                rc = x3_ok(b.Cin, kd, dtype) ? launch_conv_x3(b, x3_ok(b.Cin, kd, dtype), st) : launch_conv<float>(b, kd, st);
                if (rc == BIU_OK) {
                    hipLaunchKernelGGL(k_split_reduce, dim3(grid_for((i64)nvox(y) * y->c, 256, 2048)), dim3(256), 0, st, (const float*)ws, sl0 / sizeof(float),
                                       ks, (float*)y->p, (long)nvox(y), y->c, y->pitch, accumulate);
                    if (y1t)
                        hipLaunchKernelGGL(k_split_reduce, dim3(grid_for((i64)nvox(y1t) * y1t->c, 256, 2048)), dim3(256), 0, st,
                                           (const float*)(ws + (size_t)ks * sl0), sl1 / sizeof(float), ks, (float*)y1t->p, (long)nvox(y1t), y1t->c,
                                           y1t->pitch, cat->accumulate1);
                }
                if (rc != BIU_OK) return rc;
                BIU_CHECK_LAUNCH("split_reduce");
                return BIU_OK;
            }
        }
    }
    // the rolling-window kernel with register-resident weights (biu_conv_roll.hip): narrow 3-D bf16 layers on one input and one output tensor
    if (kd == 3 && !(cat && (cat->x1 || cat->y1)) && biu_conv_roll_ok(x, y, dtype, false, accumulate, red != nullptr)) {
        const void* img = biu_conv_roll_mshape(x, y, dtype) == 16 ? (const void*)((const char*)packed + regular_packed_bytes(x->c, y->c, 27, dtype)) : packed;
        return biu_conv_roll(x, xf, img, bias, y, bn_partial, red, st, accumulate);
    }
    {   // the 16-row kernel: one input tensor; one output, or the two outputs of a split data gradient when every 32-channel column lies in one of them
        biu_act yall = *y;
        yall.c = a.Cout;
        const int mtl = (!(cat && cat->x1) && m16_ok(x, &yall, dtype)) ? m16_mtl(x->c, a.Cout, dtype) : 0;
        if (mtl && (!(cat && cat->y1) || (mtl == 2 && a.osplit % 32 == 0))) {
            a.wpk = (const uint4*)((const char*)packed + regular_packed_bytes(x->c, a.Cout, kd * 9, dtype));
            if (mtl == 2) return kd == 3 ? launch_conv16<3, 4, 8, 2>(a, st) : launch_conv16<1, 1, 32, 2>(a, st);
            return kd == 3 ? launch_conv16<3, 4, 8, 1>(a, st) : launch_conv16<1, 1, 32, 1>(a, st);
        }
    }
    if (dtype == BIU_BF16) return launch_conv<bf16_t>(a, kd, st);
    if (x3_ok(a.Cin, kd, dtype)) return launch_conv_x3(a, x3_ok(a.Cin, kd, dtype), st);
    return launch_conv<float>(a, kd, st);
}

// ---------------------------------------------------------------------------------------------------------------
// ConvTranspose k2 s2 on the same kernel
//   forward : 4 (8) one-tap GEMMs, one per output parity a (blockIdx.z), scattered to output voxel 2v + a
//   dgrad   : a stride-2, 2x2(x2)-tap convolution from the fine grid to the coarse grid
// packed weights (biu_mfma_pack_convt): kind 0 -> [a][ntile(co)][kstep(ci)][lane], kind 1 -> [ntile(ci)][kstep(co)][tap a][lane]
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_pack_convt(const float* __restrict__ w, int cin, int cout, int taps, int kind, uint4* __restrict__ out, int x3) {
    using F = Frag<T>;
    constexpr int PE = F::PE;
    const int Kc = kind == 0 ? cin : cout, Nc = kind == 0 ? cout : cin;
    const int nKS = (sizeof(T) == 4 && x3 == 2) ? Kc / 16 * 3 : Kc / (2 * PE), ntiles = (Nc + 31) / 32;
    const size_t total = (size_t)taps * ntiles * nKS * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx % 64);
        size_t t = idx / 64;
        int tap, ks, nt;
        if (kind == 0) { ks = (int)(t % nKS); t /= nKS; nt = (int)(t % ntiles); tap = (int)(t / ntiles); }
        else { tap = (int)(t % taps); t /= taps; ks = (int)(t % nKS); nt = (int)(t / nKS); }
        const int i = nt * 32 + (lane & 31);
        if constexpr (sizeof(T) == 4) {
            if (x3) {
                out[idx] = x3_weight_piece(x3, ks, lane >> 5, [&](int k) -> float {
                    if (i >= Nc || k >= Kc) return 0.f;
                    return kind == 0 ? w[((size_t)k * cout + i) * taps + tap] : w[((size_t)i * cout + k) * taps + tap];
                });
                continue;
            }
        }
        float f[PE];
#pragma unroll
        for (int e = 0; e < PE; ++e) {
            const int k = ks * 2 * PE + (lane >> 5) * PE + e;
            float v = 0.f;
            if (i < Nc && k < Kc) {
                // PyTorch ConvTranspose weight: (Cin, Cout, taps)
                if (kind == 0) v = w[((size_t)k * cout + i) * taps + tap];       // rows i = co, reduce k = ci
                else v = w[((size_t)i * cout + k) * taps + tap];                 // rows i = ci, reduce k = co
            }
            f[e] = v;
        }
        out[idx] = F::pack(f);
    }
}

size_t biu_mfma_convt_packed_bytes(int kind, int cin, int cout, int kd, int dtype) {
    if (dtype != BIU_BF16 && dtype != BIU_F32) return 0;
    if (kd != 1 && kd != 2) return 0;
    const int K = kind == 0 ? cin : cout, Nn = kind == 0 ? cout : cin;
    if (!chan_ok(K, Nn, dtype)) return 0;
    const size_t ntiles = (Nn + 31) / 32, nKS = nks_of(K, kd, dtype);
    return ntiles * nKS * (size_t)(kd * 4) * 1024;
}

int biu_mfma_convt_pack(int kind, const float* w, int cin, int cout, int kd, int dtype, void* packed, hipStream_t st) {
    const size_t total = biu_mfma_convt_packed_bytes(kind, cin, cout, kd, dtype) / 16;
    BIU_REQUIRE(total > 0, BIU_ERR_UNSUPPORTED, "convt_pack: shape is served by the direct kernels");
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_pack_convt<T>, dim3(grid_for((i64)total, 256, 4096)), dim3(256), 0, st, w, cin, cout,
                                                 kd * 4, kind, (uint4*)packed, x3_ok(kind == 0 ? cin : cout, kd, dtype)));
    BIU_CHECK_LAUNCH("pack_convt");
    return BIU_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// batched packing: every layer's weights of a network in ONE launch (blockIdx.y = job); same layouts as k_pack_weights /
// k_pack_convt.  A training step re-packs ~30 small tensors; one launch instead of ~30 removes their launch latencies
// from the step's critical path.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_pack_batch(const biu_pack_job* __restrict__ jobs, int x3_on) {
    using F = Frag<T>;
    constexpr int PE = F::PE;
    const biu_pack_job j = jobs[blockIdx.y];
    const float* __restrict__ w = (const float*)j.w;
    uint4* __restrict__ out = (uint4*)j.packed;
    const int cin = j.cin, cout = j.cout, kind = j.kind;
    const int Kc = kind == 0 ? cin : cout, Nc = kind == 0 ? cout : cin;
    const bool split_img = sizeof(T) == 4 && (x3_on & 1) && j.kd == 1 && Kc % 16 == 0;      // == x3_ok(): the split-product image of a 2-D layer (3x3 conv or ConvTranspose k2)
    const int xmode = (x3_on & 4) ? 2 : 1;
    const int nKS = (split_img && xmode == 2) ? Kc / 16 * 3 : Kc / (2 * PE), ntiles = (Nc + 31) / 32;
    const int taps = j.transposed ? j.kd * 4 : j.kd * j.kh * j.kw;
    const size_t total = (size_t)ntiles * nKS * taps * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx % 64);
        size_t t = idx / 64;
        int tap, ks, nt;
        if (j.transposed && kind == 0) { ks = (int)(t % nKS); t /= nKS; nt = (int)(t % ntiles); tap = (int)(t / ntiles); }
        else { tap = (int)(t % taps); t /= taps; ks = (int)(t % nKS); nt = (int)(t / nKS); }
        const int i = nt * 32 + (lane & 31);
        if constexpr (sizeof(T) == 4) {
            if (split_img) {
                out[idx] = x3_weight_piece(xmode, ks, lane >> 5, [&](int k) -> float {
                    if (i >= Nc || k >= Kc) return 0.f;
                    if (j.transposed) return kind == 0 ? w[((size_t)k * cout + i) * taps + tap] : w[((size_t)i * cout + k) * taps + tap];
                    return kind == 0 ? w[((size_t)i * cin + k) * taps + tap] : w[((size_t)k * cin + i) * taps + (taps - 1 - tap)];
                });
                continue;
            }
        }
        float f[PE];
#pragma unroll
        for (int e = 0; e < PE; ++e) {
            const int k = ks * 2 * PE + (lane >> 5) * PE + e;
            float v = 0.f;
            if (i < Nc && k < Kc) {
                if (j.transposed) v = kind == 0 ? w[((size_t)k * cout + i) * taps + tap] : w[((size_t)i * cout + k) * taps + tap];
                else v = kind == 0 ? w[((size_t)i * cin + k) * taps + tap] : w[((size_t)k * cin + i) * taps + (taps - 1 - tap)];
            }
            f[e] = v;
        }
        out[idx] = F::pack(f);
    }
    // the 16-row kernel's fragment image follows the regular one (biu_mfma_packed_bytes / k_pack_weights16: [col][ks][tap][m][lane])
    if constexpr (sizeof(T) == 2) {
        const int mtl = (j.transposed || Kc < 32 || Kc % 32 != 0) ? 0 : (Nc == 16 ? 1 : ((x3_on & 2) && Kc == 32 && Nc >= 32 && Nc % 32 == 0 ? 2 : 0));   // == m16_mtl()
        if (mtl) {
            uint4* __restrict__ out16 = out + total;
            const int nKS32 = Kc / 32, ncol = (Nc + 16 * mtl - 1) / (16 * mtl);
            const size_t total16 = (size_t)ncol * nKS32 * taps * mtl * 64;
            for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total16; idx += (size_t)gridDim.x * blockDim.x) {
                const int lane = (int)(idx % 64);
                size_t t = idx / 64;
                const int m = (int)(t % mtl); t /= mtl;
                const int tap = (int)(t % taps); t /= taps;
                const int ks = (int)(t % nKS32);
                const int col = (int)(t / nKS32);
                const int i = (col * mtl + m) * 16 + (lane & 15);
                float f[PE];
#pragma unroll
                for (int e = 0; e < PE; ++e) {
                    const int k = ks * 32 + (lane >> 4) * 8 + e;
                    f[e] = (i < Nc && k < Kc) ? (kind == 0 ? w[((size_t)i * cin + k) * taps + tap] : w[((size_t)k * cin + i) * taps + (taps - 1 - tap)]) : 0.f;
                }
                out16[idx] = F::pack(f);
            }
        }
    }
}
int biu_mfma_pack_batch(const biu_pack_job* jobs_device, int n, int dtype, hipStream_t st) {
    if (n <= 0) return BIU_OK;
    // flags: bit 0 = split-product images (fp32), bit 2 = ... of the six-term form; bit 1 = two-tile images of the 16-row kernel (bf16; m16_mtl's switch)
    const int xm = dtype == BIU_F32 ? fp32_split_mode() : 0;
    const int flags = (xm ? 1 : 0) | (xm == 2 ? 4 : 0) | ((dtype == BIU_BF16 && m16_mtl(32, 32, BIU_BF16) == 2) ? 2 : 0);
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_pack_batch<T>, dim3(256, n), dim3(256), 0, st, jobs_device, flags));
    BIU_CHECK_LAUNCH("pack_batch");
    return BIU_OK;
}

static bool ptrs_ok(const biu_act* x, const biu_act* y, int dtype) {
    const size_t es = dsize(dtype);
    if ((uintptr_t)x->p % 16 || (uintptr_t)y->p % 16 || ((size_t)x->pitch * es) % 16 || ((size_t)y->pitch * es) % 16) return false;
    if (nvox(x) * (i64)x->pitch >= (1LL << 31) || nvox(y) * (i64)y->pitch >= (1LL << 31)) return false;
    if (sample_bytes(x, es) >= BIU_MAX_SAMPLE_BYTES || sample_bytes(y, es) >= BIU_MAX_SAMPLE_BYTES) return false;
    return true;
}

bool biu_mfma_convt_ok(int kind, const biu_act* lo, const biu_act* hi, int kd, int dtype) {
    const int K = kind == 0 ? lo->c : hi->c, Nn = kind == 0 ? hi->c : lo->c;
    return (kd == 1 || kd == 2) && chan_ok(K, Nn, dtype) && ptrs_ok(lo, hi, dtype);
}

template <typename T>
static int launch_convt_fwd(const ConvArgs& a, int kd, hipStream_t st) {
    const int ntiles = (a.Cout + 31) / 32, nt = pick_nt(ntiles), nz = kd * 4;
    if (kd == 2) {
        // one-tap GEMMs are all staging and no reuse: take the widest channel chunk the input allows (64, 32, 16 channels),
        // i.e. the fewest (brick, chunk) items and whole 128-byte rows per piece group
        const int e = 16 / (int)sizeof(T);        // channels per 16-byte piece
        {
            if (a.Cin % (8 * e) == 0) return nt == 1 ? launch_cfg<T, 1, 1, 1, 4, 8, 16, 1, 8>(a, ntiles, nz, st) : launch_cfg<T, 1, 1, 1, 4, 8, 16, 2, 8>(a, ntiles, nz, st);
            if (a.Cin % (4 * e) == 0) return nt == 1 ? launch_cfg<T, 1, 1, 1, 4, 8, 16, 1, 4>(a, ntiles, nz, st) : launch_cfg<T, 1, 1, 1, 4, 8, 16, 2, 4>(a, ntiles, nz, st);
        }
        if (nt == 1) return launch_cfg<T, 1, 1, 1, 4, 8, 16, 1, 2>(a, ntiles, nz, st);
        return launch_cfg<T, 1, 1, 1, 4, 8, 16, 2, 2>(a, ntiles, nz, st);
    }
    {
        const int e = 16 / (int)sizeof(T);
        if (a.Cin % (8 * e) == 0) return nt == 1 ? launch_cfg<T, 1, 1, 1, 1, 32, 16, 1, 8>(a, ntiles, nz, st) : launch_cfg<T, 1, 1, 1, 1, 32, 16, 2, 8>(a, ntiles, nz, st);
        if (a.Cin % (4 * e) == 0) return nt == 1 ? launch_cfg<T, 1, 1, 1, 1, 32, 16, 1, 4>(a, ntiles, nz, st) : launch_cfg<T, 1, 1, 1, 1, 32, 16, 2, 4>(a, ntiles, nz, st);
    }
    if (nt == 1) return launch_cfg<T, 1, 1, 1, 1, 32, 16, 1, 2>(a, ntiles, nz, st);
    return launch_cfg<T, 1, 1, 1, 1, 32, 16, 2, 2>(a, ntiles, nz, st);
}

template <typename T>
static int launch_convt_dgrad(const ConvArgs& a, int kd, hipStream_t st) {
    const int ntiles = (a.Cout + 31) / 32, nt = pick_nt(ntiles);
    if (kd == 2) {
        if (nt == 1) return launch_cfg<T, 2, 2, 2, 2, 8, 16, 1, 2>(a, ntiles, 1, st);
        return launch_cfg<T, 2, 2, 2, 2, 8, 16, 2, 2>(a, ntiles, 1, st);
    }
    if (nt == 1) return launch_cfg<T, 1, 2, 2, 1, 16, 16, 1, 2>(a, ntiles, 1, st);
    return launch_cfg<T, 1, 2, 2, 1, 16, 16, 2, 2>(a, ntiles, 1, st);
}

int biu_mfma_convt_fwd(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, int kd, const biu_act* y,
                       int dtype, hipStream_t st) {
    ConvArgs a;
    clear_cat(a);
    a.bn_partial = nullptr;
    a.red_mode = 0; a.red_y = nullptr; a.red_ypitch = 0;
    a.red_scale = a.red_shift = a.red_slope = a.red_mean = a.red_invstd = nullptr;
    a.x = (const char*)x->p;
    a.y = (char*)y->p;
    a.wpk = (const uint4*)packed;
    a.bias = bias;
    int rc = fill_xf(a, xf);
    if (rc) return rc;
    a.xpitch = x->pitch; a.ypitch = y->pitch;
    a.N = x->n;
    a.GD = a.ID = x->d; a.GH = a.IH = x->h; a.GW = a.IW = x->w;
    a.OD = y->d; a.OH = y->h; a.OW = y->w;
    a.osd = kd; a.osh = 2; a.osw = 2;
    a.Cin = x->c; a.Cout = y->c;
    a.nKS = nks_of(x->c, kd, dtype);
    a.wz_stride = ((a.Cout + 31) / 32) * a.nKS * 64;
    a.accumulate = 0;
    a.diag = nullptr;
    a.nbd = a.nbh = a.nbw = 0;
    if (dtype == BIU_BF16) return launch_convt_fwd<bf16_t>(a, kd, st);
    if (const int xm = x3_ok(a.Cin, kd, dtype)) {       // 2-D, fp32 tensors, split bf16 products: 16-channel chunks
        const int ntiles = (a.Cout + 31) / 32;
        if (xm == 2) return pick_nt(ntiles) == 1 ? launch_cfg<f32x6_t, 1, 1, 1, 1, 32, 16, 1, 4>(a, ntiles, 4, st) : launch_cfg<f32x6_t, 1, 1, 1, 1, 32, 16, 2, 4>(a, ntiles, 4, st);
        return pick_nt(ntiles) == 1 ? launch_cfg<f32x3_t, 1, 1, 1, 1, 32, 16, 1, 4>(a, ntiles, 4, st) : launch_cfg<f32x3_t, 1, 1, 1, 1, 32, 16, 2, 4>(a, ntiles, 4, st);
    }
    return launch_convt_fwd<float>(a, kd, st);
}

int biu_mfma_convt_dgrad(const biu_act* dy, const void* packed, int kd, const biu_act* dx, int accumulate, int dtype, hipStream_t st,
                         float* bn_partial, const BnRedFuse* red) {
    ConvArgs a;
    clear_cat(a);
    a.bn_partial = red ? bn_partial : nullptr;
    a.red_mode = 0; a.red_y = nullptr; a.red_ypitch = 0;
    a.red_scale = a.red_shift = a.red_slope = a.red_mean = a.red_invstd = nullptr;
    if (red) {
        a.red_mode = 1; a.red_y = (const char*)red->y->p; a.red_ypitch = red->y->pitch;
        a.red_scale = red->scale; a.red_shift = red->shift; a.red_slope = red->slope; a.red_mean = red->mean; a.red_invstd = red->invstd;
    }
    a.x = (const char*)dy->p;
    a.y = (char*)dx->p;
    a.wpk = (const uint4*)packed;
    a.bias = nullptr;
    a.xs = a.xb = a.xl = nullptr;
    a.xpitch = dy->pitch; a.ypitch = dx->pitch;
    a.N = dx->n;
    a.GD = a.OD = dx->d; a.GH = a.OH = dx->h; a.GW = a.OW = dx->w;
    a.ID = dy->d; a.IH = dy->h; a.IW = dy->w;
    a.osd = a.osh = a.osw = 1;
    a.Cin = dy->c; a.Cout = dx->c;
    a.nKS = nks_of(dy->c, kd, dtype);
    a.wz_stride = 0;
    a.accumulate = accumulate;
    a.diag = nullptr;
    a.nbd = a.nbh = a.nbw = 0;
    if (dtype == BIU_BF16) return launch_convt_dgrad<bf16_t>(a, kd, st);
    if (const int xm = x3_ok(a.Cin, kd, dtype)) {
        const int ntiles = (a.Cout + 31) / 32;
        if (xm == 2) return pick_nt(ntiles) == 1 ? launch_cfg<f32x6_t, 1, 2, 2, 1, 16, 16, 1, 4>(a, ntiles, 1, st) : launch_cfg<f32x6_t, 1, 2, 2, 1, 16, 16, 2, 4>(a, ntiles, 1, st);
        return pick_nt(ntiles) == 1 ? launch_cfg<f32x3_t, 1, 2, 2, 1, 16, 16, 1, 4>(a, ntiles, 1, st) : launch_cfg<f32x3_t, 1, 2, 2, 1, 16, 16, 2, 4>(a, ntiles, 1, st);
    }
    return launch_convt_dgrad<float>(a, kd, st);
}

// ---------------------------------------------------------------------------------------------------------------
// Nearest-neighbour up-sampling (x2 per axis) folded into the 3x3x3 convolution that follows it (round 3; forward):
//   y[2v + p] = bias + sum_{t in {0,1}^3} W'[p][t] . T(x)[v + t - 1 + p]          p = output parity per axis, x = the COARSE tensor
// Along one axis the fine taps k = 0, 1, 2 of output parity 0 read coarse voxels v-1, v, v (t = 0 <- {0}, t = 1 <- {1, 2}), those of parity 1
// read v, v, v+1 (t = 0 <- {0, 1}, t = 1 <- {2}); zero padding of the coarse tensor is exactly the zero padding of the up-sampled one.
// 8 parity classes x 8 taps instead of 27 taps per output voxel (x 0.30 FLOPs), and the up-sampled tensor is not read at all.
// Runs as k_conv_pipe<T, 2, 2, 1, ...> with blockIdx.z = parity: weight slice z, parity-dependent padding, scattered store.
// ---------------------------------------------------------------------------------------------------------------
// folded weight W'[p][co][ci][t] = sum of the fine taps of class (p, t) (fixed order: deterministic)
__device__ __forceinline__ float fold_nearest_weight(const float* __restrict__ w, int cin, int co, int ci, int p, int t) {
    int lo[3], hi[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const int pa = (p >> (2 - ax)) & 1, ta = (t >> (2 - ax)) & 1;
        lo[ax] = pa == 0 ? (ta ? 1 : 0) : (ta ? 2 : 0);
        hi[ax] = pa == 0 ? (ta ? 2 : 0) : (ta ? 2 : 1);
    }
    const float* wp = w + ((size_t)co * cin + ci) * 27;
    float sum = 0.f;
    for (int kd = lo[0]; kd <= hi[0]; ++kd)
        for (int kh = lo[1]; kh <= hi[1]; ++kh)
            for (int kw = lo[2]; kw <= hi[2]; ++kw) sum += wp[(kd * 3 + kh) * 3 + kw];
    return sum;
}
// packed image of the 8 folded kernels: out[parity = blockIdx.y][ntile][kstep][tap t][lane], fragment layout of k_pack_weights
// (wf != nullptr: explicit folded weights [p][co][ci][t] -- the composed ones of a ConvTranspose decoder -- instead of the tap sums of w)
template <typename T>
__global__ void k_pack_upconv(const float* __restrict__ w, int cin, int cout, int nKS, int ntiles, uint4* __restrict__ out, const float* __restrict__ wf = nullptr) {
    using F = Frag<T>;
    constexpr int PE = F::PE;
    const int p = (int)blockIdx.y;
    const size_t slice = (size_t)ntiles * nKS * 8 * 64;
    out += (size_t)p * slice;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < slice; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx % 64);
        size_t r = idx / 64;
        const int t = (int)(r % 8); r /= 8;
        const int ks = (int)(r % nKS);
        const int nt = (int)(r / nKS);
        const int co = nt * 32 + (lane & 31);
        float f[PE];
#pragma unroll
        for (int e = 0; e < PE; ++e) {
            const int ci = ks * 2 * PE + (lane >> 5) * PE + e;
            f[e] = (co < cout && ci < cin) ? (wf ? wf[(((size_t)p * cout + co) * cin + ci) * 8 + t] : fold_nearest_weight(w, cin, co, ci, p, t)) : 0.f;
        }
        out[idx] = F::pack(f);
    }
}

// packed image of the folded DATA GRADIENT: rows = input channels ci, reduction index kv = p * Cout + co over the 8 parity classes,
// tap s = the coarse offset u - p + s it reads: W'[p][co][ci][t = 1 - s per axis].  out[ntile(ci)][kstep(kv)][tap s][lane]
template <typename T>
__global__ void k_pack_upconv_dgrad(const float* __restrict__ w, int cin, int cout, int nKSv, int ntiles, uint4* __restrict__ out, const float* __restrict__ wf = nullptr) {
    using F = Frag<T>;
    constexpr int PE = F::PE;
    const size_t total = (size_t)ntiles * nKSv * 8 * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx % 64);
        size_t r = idx / 64;
        const int sidx = (int)(r % 8); r /= 8;
        const int ks = (int)(r % nKSv);
        const int nt = (int)(r / nKSv);
        const int ci = nt * 32 + (lane & 31);
        float f[PE];
#pragma unroll
        for (int e = 0; e < PE; ++e) {
            const int kv = ks * 2 * PE + (lane >> 5) * PE + e;
            const int p = kv / cout, co = kv - p * cout;
            f[e] = (ci < cin && p < 8) ? (wf ? wf[(((size_t)p * cout + co) * cin + ci) * 8 + (7 - sidx)] : fold_nearest_weight(w, cin, co, ci, p, 7 - sidx)) : 0.f;
        }
        out[idx] = F::pack(f);
    }
}

bool biu_mfma_upconv_ok(const biu_act* x, const biu_act* y, int dtype) {
    if (dtype != BIU_BF16 && dtype != BIU_F32) return false;
    if (y->n != x->n || y->d != 2 * x->d || y->h != 2 * x->h || y->w != 2 * x->w) return false;
    return chan_ok(x->c, y->c, dtype) && ptrs_ok(x, y, dtype);
}
static size_t upconv_slice16(int cin, int cout, int dtype) {                 // packed fragments (16 B) per parity class
    return (size_t)((cout + 31) / 32) * (cin / ks_of(dtype)) * 8 * 64;
}
// kind 0: forward image (8 parity slices);  kind 1: data-gradient image (reduction over 8 x Cout)
size_t biu_mfma_upconv_packed_bytes(int kind, int cin, int cout, int dtype) {
    if ((dtype != BIU_BF16 && dtype != BIU_F32) || (kind != 0 && kind != 1)) return 0;
    if (kind == 0) return chan_ok(cin, cout, dtype) ? 8 * upconv_slice16(cin, cout, dtype) * 16 : 0;
    if (!chan_ok(cout, cin, dtype)) return 0;
    return (size_t)((cin + 31) / 32) * (8 * cout / ks_of(dtype)) * 8 * 64 * 16;
}
int biu_mfma_upconv_pack(int kind, const float* w, int cin, int cout, int dtype, void* packed, hipStream_t st, const float* wf) {
    if (kind == 0) {
        const size_t slice = upconv_slice16(cin, cout, dtype);
        const int ntiles = (cout + 31) / 32, nKS = cin / ks_of(dtype);
        BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_pack_upconv<T>, dim3(grid_for((i64)slice, 256, 512), 8), dim3(256), 0, st, w, cin, cout, nKS, ntiles,
                                                     (uint4*)packed, wf));
    } else {
        const int ntiles = (cin + 31) / 32, nKSv = 8 * cout / ks_of(dtype);
        BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_pack_upconv_dgrad<T>, dim3(grid_for((i64)ntiles * nKSv * 8 * 64, 256, 4096)), dim3(256), 0, st, w, cin,
                                                     cout, nKSv, ntiles, (uint4*)packed, wf));
    }
    BIU_CHECK_LAUNCH("upconv_pack");
    return BIU_OK;
}
template <typename T, int NT, int CKP>
static int launch_upconv_cfg(const ConvArgs& a, int ntiles, hipStream_t st) { return launch_cfg<T, 2, 2, 1, 4, 8, 16, NT, CKP>(a, ntiles, a.fold == 2 ? 1 : 8, st); }
template <typename T>
static int launch_upconv(const ConvArgs& a, hipStream_t st) {
    const int ntiles = (a.Cout + 31) / 32, nt = pick_nt(ntiles);
    const int e = 16 / (int)sizeof(T);                                       // channels per 16-byte piece
    // (a data-gradient chunk must lie in ONE parity class: a.Cin = channels per class there)
    // one output tile (decode5 of UNet3D(32)): 64-channel chunks -- half the items per brick, each with twice the MFMAs behind its two barriers;
    // the 98 KiB tile leaves room for one weight slab (decode5 forward call 1.217 -> 1.183 ms, same box; BIU_DISABLE=foldck8 for the A/B)
    static int ck8 = -1;
    if (ck8 < 0) { const char* e8 = getenv("BIU_DISABLE"); ck8 = (e8 && strstr(e8, "foldck8")) ? 0 : 1; }
    if constexpr (sizeof(T) == 2)
        if (ck8 && nt == 1 && a.Cin % (8 * e) == 0) return launch_upconv_cfg<T, 1, 8>(a, ntiles, st);
    if (a.Cin % (4 * e) == 0) return nt == 1 ? launch_upconv_cfg<T, 1, 4>(a, ntiles, st) : launch_upconv_cfg<T, 2, 4>(a, ntiles, st);
    return nt == 1 ? launch_upconv_cfg<T, 1, 2>(a, ntiles, st) : launch_upconv_cfg<T, 2, 2>(a, ntiles, st);
}
// BatchNorm-statistics rows of a folded launch: one per block and parity class (must mirror launch_cfg_r's grid computation)
int biu_mfma_upconv_stat_rows(const biu_act* x, const biu_act* y) {
    const int ntiles = (y->c + 31) / 32, gy = ntiles / pick_nt(ntiles);
    int g = grid_per_column(num_cus(), gy * 8);
    const int nbricks = x->n * ((x->d + 3) / 4) * ((x->h + 7) / 8) * ((x->w + 15) / 16);
    if (g > nbricks) g = nbricks;
    return 8 * g;
}
int biu_mfma_upconv_fwd(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, const biu_act* y, float* bn_partial,
                        int dtype, hipStream_t st, int accumulate) {
    ConvArgs a;
    clear_cat(a);
    a.bn_partial = bn_partial;
    a.red_mode = 0; a.red_y = nullptr; a.red_ypitch = 0;
    a.red_scale = a.red_shift = a.red_slope = a.red_mean = a.red_invstd = nullptr;
    a.x = (const char*)x->p;
    a.y = (char*)y->p;
    a.wpk = (const uint4*)packed;
    a.bias = bias;
    int rc = fill_xf(a, xf);
    if (rc) return rc;
    a.xpitch = x->pitch; a.ypitch = y->pitch;
    a.N = x->n;
    a.GD = a.ID = x->d; a.GH = a.IH = x->h; a.GW = a.IW = x->w;
    a.OD = y->d; a.OH = y->h; a.OW = y->w;
    a.osd = a.osh = a.osw = 2;
    a.Cin = x->c; a.Cout = y->c;
    a.nKS = x->c / ks_of(dtype);
    a.wz_stride = (int)upconv_slice16(x->c, y->c, dtype);
    a.accumulate = accumulate;
#ifdef BIU_DIAG
    a.diag = biu_diag_buffer;                              // (tools/diag_fold.py: in-kernel stamps of the fold forward)
#else
    a.diag = nullptr;
#endif
    a.nbd = a.nbh = a.nbw = 0;
    a.fold = 1;
    if (dtype == BIU_BF16) return launch_upconv<bf16_t>(a, st);
    return launch_upconv<float>(a, st);
}
// data gradient of the folded up-conv: dx[u] (+)= sum_p sum_s W'[p][1 - s]^T . dy[2 (u - p + s) + p]   (dy fine, dx coarse)
// BatchNorm-backward partial rows of the folded data gradient with the fused reduction (one per block; mirrors launch_cfg_r)
int biu_mfma_upconv_dgrad_rows(const biu_act* dx) {
    const int ntiles = (dx->c + 31) / 32, gy = ntiles / pick_nt(ntiles);
    int g = grid_per_column(num_cus(), gy);
    const int nbricks = dx->n * ((dx->d + 3) / 4) * ((dx->h + 7) / 8) * ((dx->w + 15) / 16);
    return g > nbricks ? nbricks : g;
}
int biu_mfma_upconv_dgrad(const biu_act* dy, const void* packed, const biu_act* dx, int accumulate, int dtype, hipStream_t st, float* bn_partial,
                          const BnRedFuse* red) {
    ConvArgs a;
    clear_cat(a);
    a.bn_partial = red ? bn_partial : nullptr;
    a.red_mode = 0; a.red_y = nullptr; a.red_ypitch = 0;
    a.red_scale = a.red_shift = a.red_slope = a.red_mean = a.red_invstd = nullptr;
    if (red) {
        a.red_mode = 1; a.red_y = (const char*)red->y->p; a.red_ypitch = red->y->pitch;
        a.red_scale = red->scale; a.red_shift = red->shift; a.red_slope = red->slope; a.red_mean = red->mean; a.red_invstd = red->invstd;
    }
    a.x = (const char*)dy->p;
    a.y = (char*)dx->p;
    a.wpk = (const uint4*)packed;
    a.bias = nullptr;
    a.xs = a.xb = a.xl = nullptr;
    a.xpitch = dy->pitch; a.ypitch = dx->pitch;
    a.N = dx->n;
    a.GD = a.OD = a.ID = dx->d; a.GH = a.OH = a.IH = dx->h; a.GW = a.OW = a.IW = dx->w;      // (input extents: those of a parity sub-lattice of dy)
    a.osd = a.osh = a.osw = 1;
    a.Cin = dy->c; a.Cout = dx->c;
    a.nKS = 8 * dy->c / ks_of(dtype);
    a.wz_stride = 0;
    a.accumulate = accumulate;
    a.diag = nullptr;
    a.nbd = a.nbh = a.nbw = 0;
    a.fold = 2;
    if (dtype == BIU_BF16) return launch_upconv<bf16_t>(a, st);
    return launch_upconv<float>(a, st);
}

// ---------------------------------------------------------------------------------------------------------------
// ConvTranspose(k2, s2) + concat + 3x3x3 conv of a decoder level, with the up half folded onto the coarse tensor ("foldt"):
//   y = conv(concat(up, skip)),  up = convT(T(x_low)) + b_T        (unet3d/unet3d.py:52-58,84-90: no non-linearity between the two)
//     = conv_skip(T(skip)) + b_conv + sum_{k inside} Wb[k]  +  fold(T(x_low); W')                       Wb[k] = W_conv[:, up, k] . b_T
//   W'[p][t][ci][co] = sum_{k in class(p, t)} sum_c W_conv[co][c][k] * W_T[ci][c][q(p, k)]              q = (p + k + 1) & 1 per axis: the
// sub-position inside its coarse cell of the fine voxel tap k reads.  A tap that falls outside the fine tensor drops both its x term (coarse
// zero padding does that) and its bias term (the 27-state border table below).  Blob prepared once per weight version (biu_foldt_pack):
//   [fold forward image | fold data-gradient image | skip forward image | skip data-gradient image | W' fp32 | Wb | border fix table | bias sum]
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fold_class(int p, int t, int& lo, int& hi) {          // fine taps of one axis that read coarse tap t under parity p
    lo = p == 0 ? (t ? 1 : 0) : (t ? 2 : 0);
    hi = p == 0 ? (t ? 2 : 0) : (t ? 2 : 1);
}
// Weight-space products of the fold are sums of small fp32 GEMMs: one 32 x 32 output tile per block (256 threads, 2 x 2 outputs each), the
// reduction in chunks of 32 through LDS.  Term `term` multiplies the strided views A = pa + offA[term] (element (m, kk) at m * sa_m + kk * sa_k)
// and B = pb + offB[term] (element (kk, nn) at kk * sb_k + nn * sb_n); M, N, K bound the views (reads outside return 0).  The grids are small
// (a few hundred blocks), so a block's time is its chain of dependent global loads: the (term, chunk) steps run as ONE loop with the next
// step's eight loads per thread in flight under the current step's FMAs, and the lanes of a load run along the view's smaller stride.
struct SgView { const float* p; long s0, s1; };
__device__ __forceinline__ void small_gemm_tile(int nterms, int M, int N, int K, int m0, int n0, SgView A, const long* offA, SgView B, const long* offB,
                                                float (&acc)[2][2]) {
    constexpr int KC = 32;
    __shared__ float As[32][KC + 1];
    __shared__ float Bs[KC][33];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const bool a_m = A.s0 < A.s1, b_k = B.s0 < B.s1;               // lanes along m (else k) of A, along k (else n) of B
    const int nchunks = (K + KC - 1) / KC, nsteps = nterms * nchunks;
    float ra0[4], rb0[4], ra1[4], rb1[4];                          // two steps of loads in flight
    auto fetch = [&](int step, float (&ra)[4], float (&rb)[4]) {
        const int term = step / nchunks, k0 = (step % nchunks) * KC;
        const float* pa = A.p + offA[term];
        const float* pb = B.p + offB[term];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + j * 256;
            const int r = a_m ? (i & 31) : (i >> 5), c = a_m ? (i >> 5) : (i & 31);
            ra[j] = (m0 + r < M && k0 + c < K) ? pa[(long)(m0 + r) * A.s0 + (long)(k0 + c) * A.s1] : 0.f;
            const int rk = b_k ? (i & 31) : (i >> 5), cn = b_k ? (i >> 5) : (i & 31);
            rb[j] = (k0 + rk < K && n0 + cn < N) ? pb[(long)(k0 + rk) * B.s0 + (long)(n0 + cn) * B.s1] : 0.f;
        }
    };
    auto one = [&](int step, float (&ra)[4], float (&rb)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + j * 256;
            As[a_m ? (i & 31) : (i >> 5)][a_m ? (i >> 5) : (i & 31)] = ra[j];
            Bs[b_k ? (i & 31) : (i >> 5)][b_k ? (i >> 5) : (i & 31)] = rb[j];
        }
        __syncthreads();
        if (step + 2 < nsteps) fetch(step + 2, ra, rb);
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const float a0 = As[2 * ty][kk], a1 = As[2 * ty + 1][kk], b0 = Bs[kk][2 * tx], b1 = Bs[kk][2 * tx + 1];
            acc[0][0] = fmaf(a0, b0, acc[0][0]); acc[0][1] = fmaf(a0, b1, acc[0][1]);
            acc[1][0] = fmaf(a1, b0, acc[1][0]); acc[1][1] = fmaf(a1, b1, acc[1][1]);
        }
        __syncthreads();
    };
    if (nsteps > 0) fetch(0, ra0, rb0);
    if (nsteps > 1) fetch(1, ra1, rb1);
    for (int step = 0; step < nsteps; step += 2) {
        one(step, ra0, rb0);
        if (step + 1 < nsteps) one(step + 1, ra1, rb1);
    }
}
__device__ __forceinline__ void foldt_tq(int p, int k, int& t, int& q) {                // coarse tap and sub-position of fine tap k under parity class p
    const int kk[3] = {k / 9, (k / 3) % 3, k % 3};
    t = 0; q = 0;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const int pa = (p >> (2 - ax)) & 1;
        const int ta = pa == 0 ? (kk[ax] >= 1 ? 1 : 0) : (kk[ax] == 2 ? 1 : 0);
        t |= ta << (2 - ax);
        q |= ((pa + kk[ax] + 1) & 1) << (2 - ax);
    }
}
// wfold[p][co][ci][t] = sum_{k in class(p, t)} sum_c W_conv[co][c0 + c][k] W_T[ci][c][q(p, k)]      grid (Cin_low / 32, Cout / 32, 64 = (p, t))
__global__ __launch_bounds__(256) void k_foldt_compose(const float* __restrict__ wc, int ccat, int c0, int cup, int cout, const float* __restrict__ wt,
                                                       int cin_low, float* __restrict__ wfold) {
    __shared__ long offA[8], offB[8];
    __shared__ int nterms_s;
    const int p = (int)blockIdx.z >> 3, t = (int)blockIdx.z & 7;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int k = 0; k < 27; ++k) {
            int tt, q;
            foldt_tq(p, k, tt, q);
            if (tt == t) { offA[n] = (long)c0 * 27 + k; offB[n] = q; ++n; }
        }
        nterms_s = n;
    }
    __syncthreads();
    const int m0 = (int)blockIdx.y * 32, n0 = (int)blockIdx.x * 32;          // rows co, cols ci
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    small_gemm_tile(nterms_s, cout, cin_low, cup, m0, n0, SgView{wc, (long)ccat * 27, 27}, offA, SgView{wt, 8, (long)cup * 8}, offB, acc);
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = m0 + 2 * ty + i, ci = n0 + 2 * tx + j;
            if (co < cout && ci < cin_low) wfold[(((size_t)p * cout + co) * cin_low + ci) * 8 + t] = acc[i][j];
        }
}
// Wb[k][co] = sum_c W_conv[co][c0 + c][k] b_T[c]                                                  (one thread per (k, co))
__global__ void k_foldt_wb(const float* __restrict__ wc, int ccat, int c0, int cup, int cout, const float* __restrict__ bt, float* __restrict__ wb) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= 27 * cout) return;
    const int k = i / cout, co = i % cout;
    float acc = 0.f;
    if (bt)
        for (int c = 0; c < cup; ++c) acc = fmaf(wc[((size_t)co * ccat + c0 + c) * 27 + k], bt[c], acc);
    wb[i] = acc;
}
__device__ __forceinline__ bool tap_outside(int k, int s) {       // tap k falls outside the tensor for a voxel in border state s (per axis 0 first, 1 interior, 2 last)
    const int kd = k / 9, kh = (k / 3) % 3, kw = k % 3, sd = s / 9, sh = (s / 3) % 3, sw = s % 3;
    return (kd == 0 && sd == 0) || (kd == 2 && sd == 2) || (kh == 0 && sh == 0) || (kh == 2 && sh == 2) || (kw == 0 && sw == 0) || (kw == 2 && sw == 2);
}
// fix[state][co] = sum of Wb[k] over the taps OUTSIDE the tensor in that border state;  bias_sum[co] = b_conv[co] + sum_k Wb[k][co]   (thread per (state, co))
__global__ void k_foldt_fix(const float* __restrict__ wb, int cout, const float* __restrict__ bconv, float* __restrict__ fix, float* __restrict__ bias_sum) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= 27 * cout) return;
    const int s_ = i / cout, co = i % cout;
    float out = 0.f, all = 0.f;
    for (int k = 0; k < 27; ++k) {
        const float v = wb[k * cout + co];
        all += v;
        if (tap_outside(k, s_)) out += v;
    }
    fix[i] = out;
    if (s_ == 13) bias_sum[co] = (bconv ? bconv[co] : 0.f) + all;
}
// The border shell of an (n, d, h, w) tensor as three disjoint regular index spaces: A = {d on a face}, B = {d inside, h on a face},
// C = {d, h inside, w on a face}.  shell_voxel maps a flat index to its voxel and border state.
struct ShellDims { int n, d, h, w; long nA, nB, nC; };
__host__ __device__ inline ShellDims shell_dims(int n, int d, int h, int w) {
    ShellDims s{n, d, h, w, 0, 0, 0};
    const long fd = d >= 2 ? 2 : 1, fh = h >= 2 ? 2 : 1, fw = w >= 2 ? 2 : 1, id = d > 2 ? d - 2 : 0, ih = h > 2 ? h - 2 : 0;
    s.nA = (long)n * fd * h * w;
    s.nB = (long)n * id * fh * w;
    s.nC = (long)n * id * ih * fw;
    return s;
}
// (shell voxel counts stay below 2^31 -- the tensors' voxel counts do, biu_mfma_conv_ok --: 32-bit divisions)
__device__ __forceinline__ void shell_voxel(const ShellDims& s, long i64_, long& vox, int& state) {
    int nn, z, y, x;
    unsigned i = (unsigned)i64_;
    const unsigned nA = (unsigned)s.nA, nB = (unsigned)s.nB;
    if (i < nA) {
        x = (int)(i % (unsigned)s.w); i /= (unsigned)s.w; y = (int)(i % (unsigned)s.h); i /= (unsigned)s.h;
        const unsigned fd = s.d >= 2 ? 2u : 1u;
        const int f = (int)(i % fd); nn = (int)(i / fd);
        z = f ? s.d - 1 : 0;
    } else if (i < nA + nB) {
        i -= nA;
        x = (int)(i % (unsigned)s.w); i /= (unsigned)s.w;
        const int f = (int)(i & 1u); i >>= 1;
        z = 1 + (int)(i % (unsigned)(s.d - 2)); nn = (int)(i / (unsigned)(s.d - 2));
        y = f ? s.h - 1 : 0;
    } else {
        i -= nA + nB;
        const int f = (int)(i & 1u); i >>= 1;
        y = 1 + (int)(i % (unsigned)(s.h - 2)); i /= (unsigned)(s.h - 2);
        z = 1 + (int)(i % (unsigned)(s.d - 2)); nn = (int)(i / (unsigned)(s.d - 2));
        x = f ? s.w - 1 : 0;
    }
    const int sd = z == 0 ? 0 : (z == s.d - 1 ? 2 : 1), sh = y == 0 ? 0 : (y == s.h - 1 ? 2 : 1), sw = x == 0 ? 0 : (x == s.w - 1 ? 2 : 1);
    state = (sd * 3 + sh) * 3 + sw;
    vox = (((long)nn * s.d + z) * s.h + y) * s.w + x;
}
// y[o][co] -= fix[state(o)][co] on the border shell (one thread per shell voxel and channel).  Extents >= 2 (biu_mfma_foldt_ok): a 1-voxel axis
// would be first and last at once.
template <typename T>
__global__ void k_foldt_border_fix(char* __restrict__ y, ShellDims sd, int c, int pitch, const float* __restrict__ fix) {
    const int lanes_c = c < 256 ? c : 256, slots = 256 / lanes_c;
    const int cc0 = (int)threadIdx.x % lanes_c, slot = (int)threadIdx.x / lanes_c;
    const long nsv = sd.nA + sd.nB + sd.nC;
    if (slot < slots)
        for (long v = (long)blockIdx.x * slots + slot; v < nsv; v += (long)gridDim.x * slots) {
            long vox; int st;
            shell_voxel(sd, v, vox, st);
            T* row = (T*)y + (size_t)vox * pitch;
            const float* f = fix + st * c;
            for (int cc = cc0; cc < c; cc += lanes_c) row[cc] = (T)((float)row[cc] - f[cc]);
        }
}

struct FoldtBlob { size_t fwd, dg, sfwd, sdg, wfold, wb, fix, bias, lay, total; };
static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static FoldtBlob foldt_blob(int cin_low, int cskip, int cout, int dtype) {
    FoldtBlob b;
    size_t o = 0;
    b.fwd = o;   o += al256(biu_mfma_upconv_packed_bytes(0, cin_low, cout, dtype));
    b.dg = o;    o += al256(biu_mfma_upconv_packed_bytes(1, cin_low, cout, dtype));
    b.sfwd = o;  o += al256(biu_mfma_packed_bytes(0, cskip, cout, 3, 3, 3, 1, dtype));
    b.sdg = o;   o += al256(biu_mfma_packed_bytes(1, cskip, cout, 3, 3, 3, 1, dtype));
    b.wfold = o; o += al256((size_t)8 * cout * cin_low * 8 * sizeof(float));
    b.wb = o;    o += al256((size_t)27 * cout * sizeof(float));
    b.fix = o;   o += al256((size_t)27 * cout * sizeof(float));
    b.bias = o;  o += al256((size_t)cout * sizeof(float));
    // GEMM layouts of W_conv's up channels and of W_T (biu_fold_gemm.hip); the ConvT's output channels are not known here: sized for cup <= cin_low
    b.lay = o;   o += al256(biu_fold_gemm_layout_floats(cin_low, cin_low, cout) * sizeof(float));
    b.total = o;
    return b;
}
// forward (and data gradient, when the channel counts allow its 8-class reduction) served by the folded kernels
bool biu_mfma_foldt_ok(const biu_act* x_low, const biu_act* skip, const biu_act* y, int dtype) {
    if (dtype != BIU_BF16 && dtype != BIU_F32) return false;
    if (!biu_mfma_upconv_ok(x_low, y, dtype) || x_low->d < 1 || y->d < 2 || y->h < 2 || y->w < 2) return false;
    if (skip->n != y->n || skip->d != y->d || skip->h != y->h || skip->w != y->w) return false;
    if (!biu_mfma_conv_ok(skip, y, 3, 3, 3, 1, dtype)) return false;
    if (y->c > 512) return false;                          // (k_foldt_border_sums: 27 x Cout floats of LDS per block)
    return biu_mfma_upconv_packed_bytes(1, x_low->c, y->c, dtype) > 0 && biu_mfma_packed_bytes(1, skip->c, y->c, 3, 3, 3, 1, dtype) > 0;
}
// The composed weights and the chain rule cost 3 x 216 small fp32 GEMMs of Cout x Cup x Cin_low per step, whatever the volume: the fold pays
// where the voxel-space work it saves ((27 - 8) Cup Cout per fine voxel) outweighs them.  Since round 4 those GEMMs run on the fp32 matrix
// pipe (biu_fold_gemm.hip) and the engine keeps them off the critical path (side stream), which moved the break-even down by a factor of
// five: UNet3D(32) at 4 x 128^3 folds all three decoder levels (the 32^3 level: step 11.90 -> 11.64 ms, same box).
bool biu_mfma_foldt_worth(const biu_act* x_low, const biu_act* y) {
    const double vox = (double)y->n * y->d * y->h * y->w;
    return vox * 19.0 >= 12.0 * 648.0 * x_low->c;
}
size_t biu_mfma_foldt_packed_bytes(int cin_low, int cskip, int cout, int dtype) { return foldt_blob(cin_low, cskip, cout, dtype).total; }
// w_conv: (Cout, cup + cskip, 3, 3, 3), concat order (up | skip) [unet3d/unet3d.py:86: torch.cat([up, skip])]; w_t: (Cin_low, cup, 2, 2, 2)
int biu_mfma_foldt_pack(const float* w_conv, const float* b_conv, const float* w_t, const float* b_t, int cin_low, int cup, int cskip, int cout, int dtype,
                        void* packed, hipStream_t st) {
    const FoldtBlob b = foldt_blob(cin_low, cskip, cout, dtype);
    char* base = (char*)packed;
    float* wfold = (float*)(base + b.wfold);
    const int ccat = cup + cskip;
    if (cup <= cin_low && biu_fold_gemm_ok(cin_low, cup, cout)) {
        float* lay = (float*)(base + b.lay);
        int rc = biu_fold_gemm_layouts(w_conv, ccat, cup, cout, w_t, cin_low, lay, st);
        if (rc == BIU_OK) rc = biu_fold_gemm_compose(lay, cin_low, cup, cout, b_t, wfold, (float*)(base + b.wb), st);
        if (rc != BIU_OK) return rc;
    } else {
        hipLaunchKernelGGL(k_foldt_compose, dim3((cin_low + 31) / 32, (cout + 31) / 32, 64), dim3(256), 0, st, w_conv, ccat, 0, cup, cout, w_t, cin_low, wfold);
        hipLaunchKernelGGL(k_foldt_wb, dim3((27 * cout + 127) / 128), dim3(128), 0, st, w_conv, ccat, 0, cup, cout, b_t, (float*)(base + b.wb));
    }
    hipLaunchKernelGGL(k_foldt_fix, dim3((27 * cout + 127) / 128), dim3(128), 0, st, (const float*)(base + b.wb), cout, b_conv, (float*)(base + b.fix),
                       (float*)(base + b.bias));
    BIU_CHECK_LAUNCH("foldt_compose");
    int rc = biu_mfma_upconv_pack(0, nullptr, cin_low, cout, dtype, base + b.fwd, st, wfold);
    if (rc == BIU_OK) rc = biu_mfma_upconv_pack(1, nullptr, cin_low, cout, dtype, base + b.dg, st, wfold);
    if (rc == BIU_OK) rc = mfma_pack_strided(0, w_conv + (size_t)cup * 27, ccat, cskip, cout, 3, 3, 3, dtype, base + b.sfwd, st);
    if (rc == BIU_OK) rc = mfma_pack_strided(1, w_conv + (size_t)cup * 27, ccat, cskip, cout, 3, 3, 3, dtype, base + b.sdg, st);
    return rc;
}
// The rolling-window form (biu_conv_roll.hip): the up half FIRST (k_fold_roll: composed weights of a wave's two parity classes in registers, the
// coarse tile staged once for all eight classes, the ConvT-bias border correction as the accumulators' initial value), then the skip half
// accumulating onto it with the BatchNorm statistics from its epilogue -- y is written twice and read once, rounded to the storage type twice
// (three times and with a border pass in the brick form below).  BIU_DISABLE=froll keeps the brick form.
static bool foldt_roll_ok(const biu_act* x_low, const biu_act* skip, const biu_act* y, int dtype) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("BIU_DISABLE"); off = (e && strstr(e, "froll")) ? 1 : 0; }
    return !off && skip && biu_fold_roll_ok(x_low, y, dtype) && biu_conv_roll_ok(skip, y, dtype, false, 1, false);
}
int biu_mfma_foldt_form(const biu_act* x_low, const biu_act* skip, const biu_act* y, int dtype) { return foldt_roll_ok(x_low, skip, y, dtype) ? 1 : 0; }
int biu_mfma_foldt_stat_rows(const biu_act* x_low, const biu_act* y, const biu_act* skip, int dtype) {
    if (skip && foldt_roll_ok(x_low, skip, y, dtype)) return biu_conv_roll_rows(skip, y, dtype);
    return biu_mfma_upconv_stat_rows(x_low, y);
}
int biu_mfma_foldt_fwd(const biu_act* x_low, const biu_xform* xf_low, const biu_act* skip, const biu_xform* xf_skip, const void* packed, const biu_act* y,
                       float* bn_partial, int dtype, hipStream_t st) {
    const FoldtBlob b = foldt_blob(x_low->c, skip->c, y->c, dtype);
    const char* base = (const char*)packed;
    if (foldt_roll_ok(x_low, skip, y, dtype)) {
        int rc = biu_fold_roll(x_low, xf_low, base + b.fwd, (const float*)(base + b.bias), (const float*)(base + b.fix), y, st);
        if (rc != BIU_OK) return rc;
        return biu_mfma_conv(skip, xf_skip, base + b.sfwd, nullptr, 3, 3, 3, y, 1, bn_partial, dtype, st, nullptr, nullptr, nullptr, 0);
    }
    // 1. skip half + both biases (the full 27-tap ConvT-bias sum; the border shell is corrected next)
    int rc = biu_mfma_conv(skip, xf_skip, base + b.sfwd, (const float*)(base + b.bias), 3, 3, 3, y, 0, nullptr, dtype, st, nullptr, nullptr, nullptr, 0);
    if (rc != BIU_OK) return rc;
    // 2. taps that fall outside the tensor carry no ConvT bias
    const ShellDims sh = shell_dims(y->n, y->d, y->h, y->w);
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_foldt_border_fix<T>, dim3(2048), dim3(256), 0, st, (char*)y->p,
                                                 sh, y->c, y->pitch, (const float*)(base + b.fix)));
    BIU_CHECK_LAUNCH("foldt_border_fix");
    // 3. the up half on the coarse tensor, accumulated; BatchNorm statistics of the finished output from this launch's epilogue
    return biu_mfma_upconv_fwd(x_low, xf_low, base + b.fwd, nullptr, y, bn_partial, dtype, st, 1);
}

// ===============================================================================================================
// weight gradient:  dW[i][j][tap] = sum_v A[v][i] * B[v*S + tap - pad][j]
//   3x3(x3) conv    : A = dy (i = co), B = T(x) (j = ci), S = 1, pad = 1        -> dw (Cout, Cin, taps)
//   ConvTranspose k2: A = T(x) (i = ci), B = dy on the fine grid (j = co), S = 2 -> dw (Cin, Cout, taps)
//
// GEMM view per tap: D[i][j] += A^T[i][k = voxel] * B[k = voxel][j]: the reduction runs over VOXELS, the slow axis
// of a channels-last tensor, so both operands need a transpose on their way into the MFMA.
//   bf16: tiles are staged row-major [voxel][32 ch] in LDS and read with ds_read_b64_tr_b16 (hardware transpose):
//         one read hands each lane 4 consecutive voxels of its channel; two reads = one 32x32x16 operand.
//   fp32: v_mfma_f32_32x32x2_f32 takes ONE element per lane (k = lane >> 5), so a plain ds_read_b32 of
//         [voxel k][channel lane&31] is already the operand.
// A block owns one 32 x 32 tile of dW for ALL taps and a contiguous range of voxel bricks (split-K); its 4 waves
// split the taps, so the A fragment is read once per 16 voxels and re-used for a wave's taps.
// Partial sums are flushed with fp32 atomics into ws[tap][i][j] (two 128-B segments per wave-instruction),
// which a tiny kernel then transposes into the PyTorch layout [i][j][tap].
// ===============================================================================================================
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
    const char* pa;      // plain operand (brick grid extent)
    const char* pb;      // tapped operand
    float* ws;
    const float* as_; const float* ab_; const float* al_;   // transform of A (or null)
    const float* bs_; const float* bb_; const float* bl_;   // transform of B (or null)
    int apitch, bpitch;
    int N, CA, CB;
    int GD, GH, GW;      // extent of A
    int BD, BH, BW;      // extent of B
    int nbd, nbh, nbw, nbricks;
    int njt;             // number of 32-wide j tiles
    int jt_begin, jt_count;   // j tiles this launch covers: blockIdx.y = it * jt_count + (jt - jt_begin)
    int write_back;           // fused BatchNorm backward: this launch overwrites da with dy
    unsigned long long* diag; // BIU_DIAG builds only
    // tapped operand = concat(pb, pb1) along the channels (no concat buffer): channels [0, bsplit) live in pb, the rest in pb1;
    // bsplit is a multiple of 32, so a block's 32-wide j tile lies in one of the two
    const char* pb1; int bpitch1; int bsplit;
    const float* bs1_; const float* bb1_; const float* bl1_;
    int bricks_per_block;
    // optional fused BatchNorm backward on the plain operand: A = dy is computed on the fly from (da = pa, y = py):
    //   dz = da * T'(scale*y + shift),  dy = cA*dz + cB*y + cC ; blocks with jt == 0 also write dy back over da
    const char* py;
    int ypitch;
    const float* bn_scale; const float* bn_shift; const float* bn_slope;
    const float* bn_cA; const float* bn_cB; const float* bn_cC;
    // ConvTranspose launches (S == 2: the tapped tile holds every fine voxel of the brick exactly once): per-channel sums of the tapped
    // operand (= d bias), accumulated while its pieces are committed to LDS, by the blocks of the first plain-operand tile; zeroed by the host
    float* dbias_out;
    // k_wgrad_pipe<T, 2, 2, 1, ...> ("fold": weight gradient of nearest up-sampling + conv, one launch per output parity class):
    // fold_par = -1: off; else p = (pd << 2 | ph << 1 | pw): the plain operand (and y) is the parity-p sub-lattice of a FINE tensor of extents
    // 2 GD x 2 GH x 2 GW (voxel 2v + p, a stride-2 gather), the tapped operand voxel = grid voxel + tap - (1 - p)
    int fold_par;
    size_t fold_slice_f;      // all-parity form (k_wgrad_pipe<..., FALL>): floats between the G slices of two parity classes in ws
};

template <typename T, int PE>
__device__ __forceinline__ uint4 apply_xf16(uint4 v, const float* sc, const float* sh, const float* sl) {
    using F = Frag<T>;
    float f[PE];
    F::unpack(v, f);
    lrelu_affine<PE>(f, sc, sh, sl);
    return F::pack(f);
}


// ---------------------------------------------------------------------------------------------------------------
// Pipelined persistent weight-gradient kernel (the one the launchers use): 512 threads, one block per CU, blocks walk
// bricks; the next brick's two tiles are prefetched into registers while the current one is multiplied, then
// committed to LDS (same issue-early / write-late scheme as k_conv_pipe).  Work items for the 8 waves are
// (tap, half of the brick's voxel groups): 54 items for 27 taps -> 7/7/7/7/7/7/6/6 per wave.  Each block keeps its
// partial dW in registers across ALL its bricks and flushes once.
// ---------------------------------------------------------------------------------------------------------------
// the per-thread piece-coordinate table goes to LDS when tiles + table stay under 150 KB
constexpr bool wgrad_tab_in_lds(size_t tile_bytes, int pieces) { return tile_bytes + (size_t)pieces * 512 * 4 <= 150 * 1024; }

// NI = 32-wide tiles of the plain operand's channels a block owns (its A tile is NI * 32 channels wide): with NI = 2 the tapped
// operand -- the large fine-grid tensor of a ConvTranspose weight gradient -- is read half as often.
// FALL (fold, all parity classes in one block; KD = KHW = 2, S = 1, KSPLIT = 8): the A region holds EIGHT tiles -- the parity sub-lattices of the
// fine plain operand over the brick's coarse voxels -- and wave w owns parity class w with all 8 coarse taps; the tapped (coarse) operand is
// staged ONCE per brick with a one-voxel halo on both sides and read at offset tap + parity.  One launch instead of eight, the coarse operand
// fetched once instead of eight times (the per-class launches were bound by that traffic: 2.7 GB per decode5 call at 4.4 TB/s).
template <typename T, int KD, int KHW, int S, int TD, int TH, int TW, int KSPLIT, int NI, bool RR16 = false, bool FALL = false>
__global__ __launch_bounds__(512, 2) void k_wgrad_pipe(WgradArgs a) {
    using F = Frag<T>;
    constexpr int NTHR = 512, NWAVE = 8;
    constexpr int PE = F::PE;
    constexpr int CT = 32;
    constexpr int PPV = CT / PE;
    // X3 (T = f32x3_t): fp32 tensors, bf16x3 products -- the LDS tiles are TWO bf16 tiles each (hi plane, lo plane: split_bf16x3 while
    // committing), read through the bf16 kernels' transposing fragment reads; three MFMAs per (A, B) fragment pair
    constexpr int XP = SplitOf<T>::parts, XT = SplitOf<T>::terms;      // (bf16x6: three planes, six MFMAs per fragment pair)
    constexpr bool X3 = XP > 0;
    constexpr bool BFM = sizeof(T) == 2 || X3;            // bf16 fragments (32x32x16) out of LDS
    constexpr int LES = BFM ? 2 : 4;                       // bytes per element of an LDS tile
    constexpr int NPL = X3 ? XP : 1;                       // planes per tile
    constexpr int CTA = CT * NI, PPVA = CTA / PE, RSA = CTA * LES;
    constexpr int PD = (KD == 3) ? 1 : 0;
    constexpr int PHW = (KHW == 3) ? 1 : 0;
    constexpr bool FOLD = (KD == 2 && KHW == 2 && S == 1);   // up-sampling folded into the conv: one parity class per launch (WgradArgs::fold_par)
    constexpr int SD = (KD == 1) ? 1 : S;
    static_assert(!FALL || (FOLD && KSPLIT == 8 && NI == 1 && sizeof(T) == 2), "the all-parity form: bf16 fold, one wave per parity class");
    constexpr int HD = (TD - 1) * SD + (FALL ? 3 : KD), HH = (TH - 1) * S + (FALL ? 3 : KHW), HW = (TW - 1) * S + (FALL ? 3 : KHW);
    constexpr int HV = HD * HH * HW;
    constexpr int BVR = TD * TH * TW;                      // voxels of the brick
    constexpr int BV = FALL ? 8 * BVR : BVR;               // rows of the A region (FALL: 8 parity tiles)
    constexpr int TAPS = KD * KHW * KHW;
    constexpr int WPQ = NWAVE / KSPLIT;                    // waves per K split
    constexpr int IPW = (TAPS + WPQ - 1) / WPQ;            // taps per wave
    static_assert(NWAVE % KSPLIT == 0, "K split must divide the wave count");
    constexpr int RS = CT * LES;
    constexpr int NA = (BV * PPVA + NTHR - 1) / NTHR;
    constexpr int NB = (HV * PPV + NTHR - 1) / NTHR;
    constexpr int KUNIT = BFM ? 16 : 2;                    // voxels per MFMA k-step
    constexpr int NKG = BV / KUNIT;
    static_assert(TW % 16 == 0 && NKG % KSPLIT == 0, "brick / split mismatch");

    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    char* at = (char*)lds;                           // [NPL][BV][CT]
    char* bt = at + NPL * BV * RSA;                  // [NPL][HV][CT]
    float* lxf = (float*)(bt + NPL * HV * RS);       // [3][CTA] transform of A, [3][CT] transform of B, [6][CTA] BN-bwd
    constexpr int LA = 0, LB = 3 * CTA, LBN = 3 * CTA + 3 * CT;
    // packed piece coordinates of every thread (brick-invariant): in LDS [NA + NB][NTHR] when they fit next to the tiles
    // (the big-tile kernels have no registers to spare), else in registers
    constexpr bool TAB_LDS = wgrad_tab_in_lds((size_t)NPL * ((size_t)HV * RS + (size_t)BV * RSA), NA + NB);
    unsigned* ltab = (unsigned*)(lxf + LBN + 6 * CTA);
    unsigned xa_[TAB_LDS ? 1 : NA], xb_[TAB_LDS ? 1 : NB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: item tables live in SGPRs
    const int it = blockIdx.y / a.jt_count, jt = a.jt_begin + blockIdx.y % a.jt_count;
    const size_t esz = sizeof(T);
    const int piece = tid % PPV, pieceA = tid % PPVA;
    const int ac0 = it * CTA + pieceA * PE, bc0 = jt * CT + piece * PE;
    const bool apiece_ok = ac0 < a.CA, bpiece_ok = bc0 < a.CB;
    const bool b_src1 = a.pb1 && jt * CT >= a.bsplit;              // this block's B tile comes from the second tensor
    const char* pb_ = b_src1 ? a.pb1 : a.pb;
    const int bpitch_ = b_src1 ? a.bpitch1 : a.bpitch;
    const int bc_local = bc0 - (b_src1 ? a.bsplit : 0);
    const float* bs_ = b_src1 ? a.bs1_ : a.bs_;
    const float* bb_ = b_src1 ? a.bb1_ : a.bb_;
    const float* bl_ = b_src1 ? a.bl1_ : a.bl_;
    const bool a_xf = a.as_ != nullptr, b_xf = bs_ != nullptr;
    const bool bn_fused = !FALL && a.py != nullptr;     // (the all-parity form has no registers left for the staged y pieces: plain dy only)
    if (tid < CTA && bn_fused) {                         // (all NI tiles of the plain operand: [6][CTA])
        const int ca = it * CTA + tid;
        const bool ok = ca < a.CA;
        lxf[LBN + 0 * CTA + tid] = ok ? a.bn_scale[ca] : 1.f;
        lxf[LBN + 1 * CTA + tid] = ok ? a.bn_shift[ca] : 0.f;
        lxf[LBN + 2 * CTA + tid] = (ok && a.bn_slope) ? a.bn_slope[ca] : 1.f;
        lxf[LBN + 3 * CTA + tid] = ok ? a.bn_cA[ca] : 0.f;
        lxf[LBN + 4 * CTA + tid] = ok ? a.bn_cB[ca] : 0.f;
        lxf[LBN + 5 * CTA + tid] = ok ? a.bn_cC[ca] : 0.f;
    }
    if (tid < CTA) {
        const int ca = it * CTA + tid;
        lxf[LA + 0 * CTA + tid] = (a_xf && ca < a.CA) ? a.as_[ca] : 1.f;
        lxf[LA + 1 * CTA + tid] = (a_xf && ca < a.CA) ? a.ab_[ca] : 0.f;
        lxf[LA + 2 * CTA + tid] = (a_xf && ca < a.CA) ? a.al_[ca] : 1.f;
    }
    if (tid < CT) {
        const int cb = jt * CT + tid;
        const int cbl = cb - (b_src1 ? a.bsplit : 0);
        lxf[LB + 0 * CT + tid] = (b_xf && cb < a.CB) ? bs_[cbl] : 1.f;
        lxf[LB + 1 * CT + tid] = (b_xf && cb < a.CB) ? bb_[cbl] : 0.f;
        lxf[LB + 2 * CT + tid] = (b_xf && cb < a.CB) ? bl_[cbl] : 1.f;
    }

    floatx16 acc[IPW][NI];
#pragma unroll
    for (int t = 0; t < IPW; ++t)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][ni][e] = 0.f;

    // Work split: wave w owns K split q = w % KSPLIT (a contiguous run of this brick's k-groups) and the taps
    // w / KSPLIT + WPQ * t, t < IPW -- every wave runs the same branch-free loop over ITS k-groups with IPW MFMAs per A
    // fragment; a tap index past the last one multiplies into an accumulator nobody flushes.
    const int wq = wave % KSPLIT, wtap0 = wave / KSPLIT;
    const bool last_tap_live = (wtap0 + WPQ * (IPW - 1)) < TAPS;      // wave-uniform (wave comes from readfirstlane)
    // Row-reuse variant (RR; 3x3x3, bf16, bricks of 16-voxel rows): wave w < 8 owns the three kh taps of ONE (kd, kw) pair -- pair
    // w / 3, w % 3 -- so the B fragment of (row r, kh tap b) is the fragment of (row r + 1, tap b - 1): walking the rows of a
    // plane it reads ONE new fragment per row instead of three.  The ninth pair (2, 2) has no wave of its own: its three taps are
    // the fourth slot of waves 0, 1, 2 (read per row like before).  2 + 2 (+ 2) transposed reads per row instead of 2 + 6 (+ 2):
    // the weight gradient's MFMAs are fed from LDS with little reuse, and LDS traffic is energy the power-limited kernel pays for.
    constexpr bool RR = BIU_WGRAD_RR && sizeof(T) == 2 && KD == 3 && KHW == 3 && S == 1 && KSPLIT == 1 && NI == 1 && TW == 16 && IPW == 4 && WPQ == 8;
    int tapoff[IPW], tapid[IPW];
#pragma unroll
    for (int t = 0; t < IPW; ++t) {
        int tap = wtap0 + WPQ * t;
        if constexpr (RR) tap = (t < 3) ? ((wave / 3) * 3 + t) * 3 + (wave % 3) : (wave < 3 ? (2 * 3 + wave) * 3 + 2 : TAPS);
        tapid[t] = tap;
        tap %= TAPS;
        const int ta = tap / (KHW * KHW), tb = (tap / KHW) % KHW, tc = tap % KHW;
        tapoff[t] = ((ta * HH + tb) * HW + tc) * RS;
        if constexpr (FALL) tapoff[t] += ((((wave >> 2) & 1) * HH + ((wave >> 1) & 1)) * HW + (wave & 1)) * RS;     // wave = parity class: coarse voxel v + t - 1 + p of a tile whose halo starts at v - 1
    }

    // rr16 (RR kernels, tapped operand with 16 channels -- encode2 of UNet3D(n_filter = 32)): a 32-wide B tile would be half zeros.
    // Instead the upper 16 columns read the same 16 channels at a second (kd, kw) pair: unit u < 3 = pairs (0, u) | (1, u), units
    // 3-5 = pair (2, u - 3) alone (upper columns duplicate the lower ones and are not flushed).  18 MFMA slots per row instead of 27.
    constexpr bool rr16 = RR && RR16;                              // (the launcher picks this instantiation when the tapped operand has 16 channels)
    int u16_unit = 0, u16_r0 = 0, u16_boff = 0, u16_delta = 0;
    bool u16_half = false;
    if (rr16) {
        u16_unit = (wave == 6) ? 0 : (wave == 7) ? 1 : wave;
        u16_half = (wave < 2 || wave > 5);
        u16_r0 = (wave > 5) ? (TD * TH / 2) : 0;
        const int kdA = (u16_unit < 3) ? 0 : 2, kwA = (u16_unit < 3) ? u16_unit : u16_unit - 3;
        u16_boff = ((kdA * HH) * HW + kwA) * RS;
        u16_delta = (u16_unit < 3) ? HH * HW * RS : 0;
    }
    int a_lane, b_lane, b_lane16 = 0;
    if constexpr (BFM) {
        const int g = lane >> 4, li = lane & 15, qrow = li >> 2, p = li & 3, cg = g & 1, h = g >> 1;
        a_lane = (8 * h + qrow) * RSA + (16 * cg + 4 * p) * 2;
        b_lane = (8 * h + qrow) * S * RS + (16 * cg + 4 * p) * 2;
        b_lane16 = (8 * h + qrow) * S * RS + (4 * p) * 2 + (cg ? u16_delta : 0);
    } else {
        a_lane = (lane >> 5) * RSA + (lane & 31) * 4;
        b_lane = (lane >> 5) * S * RS + (lane & 31) * 4;
    }

    uint4 pa[NA], pb[NB], pyv[NA];
    unsigned amask = 0, bmask = 0;
    constexpr bool DBIAS = (S == 2);                                // ConvTranspose weight gradient: d bias rides in the B-tile commit
    const bool dbias_on = DBIAS && a.dbias_out != nullptr && it == 0 && bpiece_ok;
    float bsum[DBIAS ? PE : 1];
#pragma unroll
    for (int e = 0; e < (DBIAS ? PE : 1); ++e) bsum[e] = 0.f;
    // Per-thread piece coordinates, packed in 10-bit fields (d | h << 10 | w << 20); 511 marks a piece this thread does
    // not have.  Per brick: a guard-bit range test, two 24-bit multiply-adds for the offset, one buffer load whose
    // descriptor range check returns zeros for the pieces outside the volume (same scheme as k_conv_pipe).
    constexpr unsigned GBITS = (1u << 9) | (1u << 19) | (1u << 29);
    // (kept in LDS, one dword per piece and thread: the kernel has no registers to spare for them)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int i = tid + NTHR * j;
        static_assert(!FALL || BVR * PPVA == NTHR, "all-parity form: piece j of a thread is its voxel in parity tile j");
        const int q = FALL ? (i / PPVA) % BVR : i / PPVA;
        const int lw = q % TW;
        const int t = q / TW;
        const unsigned xv = (i < BV * PPVA && apiece_ok) ? (unsigned)((t / TH) | ((t % TH) << 10) | (lw << 20)) : 511u;
        if constexpr (TAB_LDS) ltab[j * NTHR + tid] = xv; else xa_[j] = xv;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int i = tid + NTHR * j;
        const int hv = i / PPV;
        const int hw = hv % HW;
        const int t = hv / HW;
        const unsigned xv = (i < HV * PPV && bpiece_ok) ? (unsigned)((t / HH) | ((t % HH) << 10) | (hw << 20)) : 511u;
        if constexpr (TAB_LDS) ltab[(NA + j) * NTHR + tid] = xv; else xb_[j] = xv;
    }
    // (fold: the plain operand is gathered with stride 2 from the fine tensor -- doubled row / plane / voxel strides, parity in the base)
    const int fgs = FOLD ? 2 : 1;
    const int fpd = FOLD ? (a.fold_par >> 2) & 1 : 0, fph = FOLD ? (a.fold_par >> 1) & 1 : 0, fpw = FOLD ? a.fold_par & 1 : 0;
    const unsigned GHa = (unsigned)(a.GH * fgs), GWa = (unsigned)(a.GW * fgs);
    const int rowA1 = a.apitch * (int)esz, rowY1 = a.ypitch * (int)esz;
    const int rowA = rowA1 * fgs, rowY = rowY1 * fgs, rowB = bpitch_ * (int)esz;
    const size_t sampA = (size_t)a.GD * a.GH * a.GW * rowA1 * (FOLD ? 8 : 1), sampY = (size_t)a.GD * a.GH * a.GW * rowY1 * (FOLD ? 8 : 1);
    const size_t sampB = (size_t)a.BD * a.BH * a.BW * rowB;                  // all < 2^31 (checked on the host)
    unsigned ca_hi = 0, cb_lo = 0, cb_hi = 0;
    int brA = 0, brY = 0, brB = 0;
    __amdgpu_buffer_rsrc_t rsA, rsY, rsB;
    auto issue_prep = [&](int brick, bool live) {
        int b = brick;
        const int bw = b % a.nbw; b /= a.nbw;
        const int bh = b % a.nbh; b /= a.nbh;
        const int bd = b % a.nbd;
        const int n = b / a.nbd;
        const int d0 = bd * TD, h0 = bh * TH, w0 = bw * TW;
        ca_hi = GBITS + (unsigned)(min(TD - 1, a.GD - 1 - d0) | (min(TH - 1, a.GH - 1 - h0) << 10) | (min(TW - 1, a.GW - 1 - w0) << 20));
        // (FALL: parity 0's base here; piece j adds its own parity's fine-voxel offset in issue_a)
        const int va = FOLD ? (((2 * d0 + (FALL ? 0 : fpd)) * 2 * a.GH + 2 * h0 + (FALL ? 0 : fph)) * 2 * a.GW + 2 * w0 + (FALL ? 0 : fpw)) : (d0 * a.GH + h0) * a.GW + w0;
        brA = (int)(unsigned)((long long)va * rowA1 + ac0 * (int)esz);         // mod 2^32, see k_conv_pipe
        brY = (int)(unsigned)((long long)va * rowY1 + ac0 * (int)esz);
        const int gd0 = d0 * SD - PD - (FOLD ? (FALL ? 1 : 1 - fpd) : 0), gh0 = h0 * S - PHW - (FOLD ? (FALL ? 1 : 1 - fph) : 0),
                  gw0 = w0 * S - PHW - (FOLD ? (FALL ? 1 : 1 - fpw) : 0);
        cb_lo = GBITS - (unsigned)(max(0, -gd0) | (max(0, -gh0) << 10) | (max(0, -gw0) << 20));
        cb_hi = GBITS + (unsigned)(min(HD - 1, a.BD - 1 - gd0) | (min(HH - 1, a.BH - 1 - gh0) << 10) | (min(HW - 1, a.BW - 1 - gw0) << 20));
        brB = (int)(unsigned)((long long)((gd0 * a.BH + gh0) * a.BW + gw0) * rowB + bc_local * (int)esz);
        // a dead prefetch reads through empty descriptors: zeros, nothing fetched
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.pa + (size_t)n * sampA), 0, live ? (int)(unsigned)sampA : 0, 0x00020000);
        rsY = __builtin_amdgcn_make_buffer_rsrc((void*)(a.py + (size_t)n * sampY), 0, (live && bn_fused) ? (int)(unsigned)sampY : 0, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(pb_ + (size_t)n * sampB), 0, live ? (int)(unsigned)sampB : 0, 0x00020000);
        amask = bmask = 0;
    };
    auto ld128 = [&](const __amdgpu_buffer_rsrc_t& rs, int off) -> uint4 {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        return make_uint4(v[0], v[1], v[2], v[3]);
    };
    auto issue_a = [&](int j) {
        const unsigned x = TAB_LDS ? ltab[j * NTHR + tid] : xa_[TAB_LDS ? 0 : j];
        const bool ok = ((ca_hi - x) & GBITS) == GBITS;
        const int v = (int)__umul24(__umul24(x & 511u, GHa) + ((x >> 10) & 511u), GWa) + (int)(x >> 20);
        // (FALL: piece j lives in parity tile j = (pd, ph, pw): fine voxel + (pd * 2 GH + ph) * 2 GW + pw)
        const unsigned pvo = FALL ? (unsigned)((((j >> 2) & 1) * 2 * a.GH + ((j >> 1) & 1)) * 2 * a.GW + (j & 1)) : 0u;
        pa[j] = ld128(rsA, ok ? (int)(__umul24((unsigned)v, (unsigned)rowA) + (unsigned)brA + pvo * (unsigned)rowA1) : -1);
        if (bn_fused) pyv[j] = ld128(rsY, ok ? (int)(__umul24((unsigned)v, (unsigned)rowY) + (unsigned)brY + pvo * (unsigned)rowY1) : -1);
        amask |= ok ? (1u << j) : 0u;
    };
    auto issue_b = [&](int j) {
        const unsigned x = TAB_LDS ? ltab[(NA + j) * NTHR + tid] : xb_[TAB_LDS ? 0 : j];
        const bool ok = (((x + cb_lo) & (cb_hi - x)) & GBITS) == GBITS;
        const int v = (int)__umul24(__umul24(x & 511u, (unsigned)a.BH) + ((x >> 10) & 511u), (unsigned)a.BW) + (int)(x >> 20);
        pb[j] = ld128(rsB, ok ? (int)(__umul24((unsigned)v, (unsigned)rowB) + (unsigned)brB) : -1);
        bmask |= ok ? (1u << j) : 0u;
    };
    auto issue = [&](int brick, bool live) {
        issue_prep(brick, live);
#pragma unroll
        for (int j = 0; j < NA; ++j) issue_a(j);
#pragma unroll
        for (int j = 0; j < NB; ++j) issue_b(j);
    };
    auto commit = [&]() {
        float sc[PE], sh[PE], sl[PE];
        if (a_xf) {
#pragma unroll
            for (int e = 0; e < PE; ++e) { sc[e] = lxf[LA + pieceA * PE + e]; sh[e] = lxf[LA + CTA + pieceA * PE + e]; sl[e] = lxf[LA + 2 * CTA + pieceA * PE + e]; }
        }
        if (bn_fused) {
            // BatchNorm + LeakyReLU backward of this thread's channel piece, in two sweeps so that only three of the six
            // coefficient vectors are live at a time:  dz = da * T'(scale*y + shift) ;  dy = cA*dz + cB*y + cC
            {
                float ks[PE], kh[PE], kl[PE];
#pragma unroll
                for (int e = 0; e < PE; ++e) { ks[e] = lxf[LBN + pieceA * PE + e]; kh[e] = lxf[LBN + CTA + pieceA * PE + e]; kl[e] = lxf[LBN + 2 * CTA + pieceA * PE + e]; }
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    if ((amask >> j) & 1u) {
                        float g[PE], yy[PE];
                        F::unpack(pa[j], g);
                        F::unpack(pyv[j], yy);
#pragma unroll
                        for (int e = 0; e < PE; ++e) g[e] *= (fmaf(ks[e], yy[e], kh[e]) > 0.f ? 1.f : kl[e]);
                        pa[j] = F::pack(g);          // dz parked in the storage type between the two sweeps (one extra bf16 rounding)
                    }
                }
            }
            {
                float ka[PE], kb[PE], kc[PE];
#pragma unroll
                for (int e = 0; e < PE; ++e) { ka[e] = lxf[LBN + 3 * CTA + pieceA * PE + e]; kb[e] = lxf[LBN + 4 * CTA + pieceA * PE + e]; kc[e] = lxf[LBN + 5 * CTA + pieceA * PE + e]; }
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    if ((amask >> j) & 1u) {
                        float g[PE], yy[PE];
                        F::unpack(pa[j], g);
                        F::unpack(pyv[j], yy);
#pragma unroll
                        for (int e = 0; e < PE; ++e) g[e] = fmaf(ka[e], g[e], fmaf(kb[e], yy[e], kc[e]));
                        pa[j] = F::pack(g);
                        if (a.write_back) {                                      // dy replaces da (only in the launch that owns it)
                            typedef unsigned v4u __attribute__((ext_vector_type(4)));
                            const unsigned x = TAB_LDS ? ltab[j * NTHR + tid] : xa_[TAB_LDS ? 0 : j];
                            const int v = (int)__umul24(__umul24(x & 511u, GHa) + ((x >> 10) & 511u), GWa) + (int)(x >> 20);
                            const unsigned pvo = FALL ? (unsigned)((((j >> 2) & 1) * 2 * a.GH + ((j >> 1) & 1)) * 2 * a.GW + (j & 1)) : 0u;
                            __builtin_amdgcn_raw_buffer_store_b128(v4u{pa[j].x, pa[j].y, pa[j].z, pa[j].w}, rsA,
                                                                   (int)(__umul24((unsigned)v, (unsigned)rowA) + (unsigned)brA + pvo * (unsigned)rowA1), 0, 0);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int i = tid + NTHR * j;
            if (i < BV * PPVA) {
                uint4 v = pa[j];
                if (a_xf && ((amask >> j) & 1u)) v = apply_xf16<T, PE>(v, sc, sh, sl);
                if constexpr (X3) {
                    float f[4];
                    F::unpack(v, f);
                    uint2 parts[NPL];
                    split_bf16<NPL>(f, parts);
#pragma unroll
                    for (int k2 = 0; k2 < NPL; ++k2) ((uint2*)(at + k2 * (BV * RSA)))[i] = parts[k2];
                } else {
                    ((uint4*)at)[i] = v;
                }
            }
        }
        if (b_xf) {
#pragma unroll
            for (int e = 0; e < PE; ++e) { sc[e] = lxf[LB + piece * PE + e]; sh[e] = lxf[LB + CT + piece * PE + e]; sl[e] = lxf[LB + 2 * CT + piece * PE + e]; }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int i = tid + NTHR * j;
            if (i < HV * PPV) {
                uint4 v = pb[j];
                if constexpr (DBIAS) {
                    if (dbias_on && ((bmask >> j) & 1u)) {
                        float f[PE];
                        F::unpack(v, f);
#pragma unroll
                        for (int e = 0; e < PE; ++e) bsum[e] += f[e];
                    }
                }
                if (b_xf && ((bmask >> j) & 1u)) v = apply_xf16<T, PE>(v, sc, sh, sl);
                if constexpr (X3) {
                    float f[4];
                    F::unpack(v, f);
                    uint2 parts[NPL];
                    split_bf16<NPL>(f, parts);
#pragma unroll
                    for (int k2 = 0; k2 < NPL; ++k2) ((uint2*)(bt + k2 * (HV * RS)))[i] = parts[k2];
                } else {
                    ((uint4*)bt)[i] = v;
                }
            }
        }
    };

#ifdef BIU_DIAG
    unsigned long long* wdiag = a.diag;
    unsigned long long tprev_ = __builtin_readcyclecounter();
    unsigned long long dsum_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0c_ = tprev_, t0r_ = __builtin_amdgcn_s_memrealtime();
#define WSTAMP(k_) do { if (wdiag && tid == 0) { unsigned long long now_ = __builtin_readcyclecounter(); dsum_[k_] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define WSTAMP(k_) do { } while (0)
#endif
    // brick walk: round k hands bricks [k*G, (k+1)*G) out so that the blocks of one XCD (block id mod 8) get a contiguous run --
    // neighbouring bricks share their halo through that XCD's L2 (same scheme as k_conv_pipe)
    const int G = gridDim.x;
    auto brick_of = [&](int k) -> int {
        if ((G & 7) == 0) return k * G + (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
        return k * G + (int)blockIdx.x;
    };
    int kround = 0;
    int brick = brick_of(0);
    if (brick < a.nbricks) {
        issue(brick, true);
        __syncthreads();             // lxf visible
        commit();
        __syncthreads();
        while (true) {
            const int nbrick = brick_of(kround + 1);
            const bool have_next = nbrick < a.nbricks;
            // the next brick's loads go out in NG slices between the k-group slices of this brick's MFMA work
            constexpr int KPW = NKG / KSPLIT;                  // k-groups a wave walks per brick
            constexpr int NG = KPW < 8 ? KPW : 8;
            static_assert(NKG % KSPLIT == 0 && KPW % NG == 0, "k-groups must split evenly into slices");
            constexpr int KPG = KPW / NG;
            WSTAMP(0);
            using FragR = typename std::conditional<BFM, bf16x8, float>::type;
            FragR fa[2][NI * NPL], fb[2][IPW * NPL];                  // split products: fa[.. * parts + part], part 0 = hi; their B fragments live in mfma_phase_split (fb unused)
            // 2-D fp32 kernels: 9 taps on 2 x 5 slots leave one empty; a second copy of the phase spills there, a wave-uniform branch around
            // the last slot's reads and 64-cycle MFMA does not (cfg2 weight gradients 96 -> 104-112 TFLOP/s)
            constexpr bool TWO_PHASES_ = IPW > 1 && (IPW * WPQ - TAPS) * BIU_TWO_PHASE_DIV >= IPW * WPQ;
            constexpr bool TAIL_BRANCH = BIU_TAIL_BRANCH && sizeof(T) == 4 && !TWO_PHASES_ && IPW > 1 && IPW * WPQ > TAPS;   // (bf16 2-D: measured neutral)
            auto load_frags = [&](auto ntap_c, int kg, FragR (&af)[NI * NPL], FragR (&bfr)[IPW * NPL]) {
                constexpr int NTAP = decltype(ntap_c)::value;
                const int q0 = kg * KUNIT;
                const int lw0 = q0 % TW;
                const int t = q0 / TW;
                const int lh = t % TH;
                const int ld = FALL ? (t / TH) % TD : t / TH;            // (FALL: row q0 of the A region = voxel q0 % BVR of parity tile q0 / BVR)
                const int hbase = ((ld * SD * HH + lh * S) * HW + lw0 * S) * RS;
                if constexpr (BFM) {
                    typedef bf16x4 __attribute__((address_space(3))) * lp;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) {
                            const char* ap = at + pl * (BV * RSA) + q0 * RSA + a_lane + ni * CT * 2;
                            bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(ap));
                            bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(ap + 4 * RSA));
                            af[ni * NPL + pl] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
#pragma unroll
                    for (int t2 = 0; t2 < NTAP; ++t2) {
                        if (TAIL_BRANCH && t2 == IPW - 1 && !last_tap_live) continue;        // wave-uniform: this wave's last slot holds no tap
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) {
                            const char* bp = bt + pl * (HV * RS) + hbase + tapoff[t2] + b_lane;
                            bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(bp));
                            bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(bp + 4 * S * RS));
                            bfr[t2 * NPL + pl] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                    }
                } else {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) af[ni] = *(const float*)(at + q0 * RSA + a_lane + ni * CT * 4);
#pragma unroll
                    for (int t2 = 0; t2 < NTAP; ++t2) { if (TAIL_BRANCH && t2 == IPW - 1 && !last_tap_live) continue; bfr[t2] = *(const float*)(bt + hbase + tapoff[t2] + b_lane); }
                }
            };
            issue_prep(have_next ? nbrick : brick, have_next);
            // Split products (X3): ONE part of the tapped operand at a time (lo first) against the parts of the plain operand it pairs with
            // (a + b < parts).  A step = (k-group, part of B): IPW fragments of B in flight instead of IPW * parts, so bf16x6 keeps the
            // balanced 2 x 5 tap slots of the exact kernel (K split 4: every SIMD holds a 5-tap and a 4-tap wave) where the all-parts-at-once
            // form needed 15 fragments per k-group and fell back to 4 x 3 slots (5 : 4 taps per SIMD).
            auto load_a_parts = [&](int kg, FragR (&af)[NI * NPL]) {
                if constexpr (BFM) {
                    typedef bf16x4 __attribute__((address_space(3))) * lp;
                    const int q0 = kg * KUNIT;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) {
                            const char* ap = at + pl * (BV * RSA) + q0 * RSA + a_lane + ni * CT * 2;
                            bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(ap));
                            bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(ap + 4 * RSA));
                            af[ni * NPL + pl] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                }
            };
            auto load_b_part = [&](auto ntap_c, int kg, int pl, FragR (&bfr)[IPW]) {
                constexpr int NTAP = decltype(ntap_c)::value;
                if constexpr (BFM) {
                    typedef bf16x4 __attribute__((address_space(3))) * lp;
                    const int q0 = kg * KUNIT;
                    const int lw0 = q0 % TW;
                    const int t = q0 / TW;
                    const int hbase = (((t / TH) * SD * HH + (t % TH) * S) * HW + lw0 * S) * RS;
#pragma unroll
                    for (int t2 = 0; t2 < NTAP; ++t2) {
                        if (TAIL_BRANCH && t2 == IPW - 1 && !last_tap_live) continue;
                        const char* bp = bt + pl * (HV * RS) + hbase + tapoff[t2] + b_lane;
                        bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(bp));
                        bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(bp + 4 * S * RS));
                        bfr[t2] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
            };
            auto mfma_phase_split = [&](auto ntap_c) {
                constexpr int NTAP = decltype(ntap_c)::value;
                FragR fbs[2][IPW];
                load_a_parts(wq * KPW, fa[0]);
                load_b_part(ntap_c, wq * KPW, NPL - 1, fbs[0]);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
#pragma unroll
                    for (int j = pf_lo(g, NA, NG); j < pf_lo(g + 1, NA, NG); ++j) issue_a(j);
#pragma unroll
                    for (int j = pf_lo(g, NB, NG); j < pf_lo(g + 1, NB, NG); ++j) issue_b(j);
#pragma unroll
                    for (int kk = 0; kk < KPG; ++kk) {
                        const int idx = g * KPG + kk;
#pragma unroll
                        for (int kbi = 0; kbi < NPL; ++kbi) {
                            const int kb = NPL - 1 - kbi, step = idx * NPL + kbi;
                            // the next step's fragments, one step ahead, in the other register set
                            if (kbi + 1 < NPL) load_b_part(ntap_c, wq * KPW + idx, kb - 1, fbs[(step + 1) & 1]);
                            else if (idx + 1 < KPW) {
                                load_a_parts(wq * KPW + idx + 1, fa[(idx + 1) & 1]);
                                load_b_part(ntap_c, wq * KPW + idx + 1, NPL - 1, fbs[(step + 1) & 1]);
                            }
#pragma unroll
                            for (int ka = NPL - 1 - kb; ka >= 0; --ka)
#pragma unroll
                                for (int t2 = 0; t2 < NTAP; ++t2)
#pragma unroll
                                    for (int ni = 0; ni < NI; ++ni) {
                                        if (TAIL_BRANCH && t2 == IPW - 1 && !last_tap_live) continue;
                                        if constexpr (BFM) acc[t2][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[idx & 1][ni * NPL + ka], fbs[step & 1][t2], acc[t2][ni], 0, 0, 0);
                                    }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // all-parity form: 8 accumulators per wave leave no room for a second fragment set -- the A fragment of a k-group and the B
            // fragments two taps at a time, the other wave of the SIMD covers the LDS latency
            auto mfma_phase_all = [&]() {
                if constexpr (FALL && BFM) {
                    typedef bf16x4 __attribute__((address_space(3))) * lp;
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
#pragma unroll
                        for (int j = pf_lo(g, NA, NG); j < pf_lo(g + 1, NA, NG); ++j) issue_a(j);
#pragma unroll
                        for (int j = pf_lo(g, NB, NG); j < pf_lo(g + 1, NB, NG); ++j) issue_b(j);
#pragma unroll
                        for (int kk = 0; kk < KPG; ++kk) {
                            const int q0 = (wq * KPW + g * KPG + kk) * KUNIT;
                            const int t = q0 / TW;
                            const int hbase = ((((t / TH) % TD) * SD * HH + (t % TH) * S) * HW + (q0 % TW) * S) * RS;
                            const char* ap = at + q0 * RSA + a_lane;
                            const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(ap));
                            const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(ap + 4 * RSA));
                            const bf16x8 af = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                            for (int t2 = 0; t2 < IPW; t2 += 2) {
                                bf16x8 bfr[2];
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    const char* bp = bt + hbase + tapoff[t2 + u] + b_lane;
                                    const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(bp));
                                    const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(bp + 4 * S * RS));
                                    bfr[u] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                                }
                                acc[t2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[0], acc[t2][0], 0, 0, 0);
                                acc[t2 + 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[1], acc[t2 + 1][0], 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            auto mfma_phase = [&](auto ntap_c) {
            constexpr int NTAP = decltype(ntap_c)::value;
            if constexpr (FALL) { mfma_phase_all(); return; }
#ifndef BIU_WGRAD_SPLIT_STEPPED
#define BIU_WGRAD_SPLIT_STEPPED 1
#endif
            if constexpr (X3 && BIU_WGRAD_SPLIT_STEPPED) { mfma_phase_split(ntap_c); return; }
            load_frags(ntap_c, wq * KPW, fa[0], fb[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int j = pf_lo(g, NA, NG); j < pf_lo(g + 1, NA, NG); ++j) issue_a(j);
#pragma unroll
                for (int j = pf_lo(g, NB, NG); j < pf_lo(g + 1, NB, NG); ++j) issue_b(j);
                // Fragment reads run one k-group ahead of the MFMAs, in a second register set (the loops are fully unrolled, so the
                // set is picked at compile time): left to itself the compiler re-used one register quad for every B fragment and
                // put an lgkmcnt(0) in front of each MFMA, i.e. the full LDS latency per MFMA.  The loop body has no branch: the
                // whole phase exists twice, for waves whose last tap slot holds a tap and for those where it does not (27 taps
                // on 8 x 4 slots leave 5 empty: their reads and MFMAs are energy the power-limited kernel does not have).
#pragma unroll
                for (int kk = 0; kk < KPG; ++kk) {
                    const int idx = g * KPG + kk;
                    if (idx + 1 < KPW) load_frags(ntap_c, wq * KPW + idx + 1, fa[(idx + 1) & 1], fb[(idx + 1) & 1]);
                    if constexpr (X3) {
                        // lo * hi, hi * lo, hi * hi: term-major, so IPW * NI independent accumulators separate the dependent MFMAs
#pragma unroll
                        for (int term = 0; term < XT; ++term)
#pragma unroll
                            for (int t2 = 0; t2 < NTAP; ++t2)
#pragma unroll
                                for (int ni = 0; ni < NI; ++ni) {
                                    if (TAIL_BRANCH && t2 == IPW - 1 && !last_tap_live) continue;
                                    acc[t2][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[idx & 1][ni * NPL + term_a<X3 ? XP : 2>(term)], fb[idx & 1][t2 * NPL + term_b<X3 ? XP : 2>(term)],
                                                                                         acc[t2][ni], 0, 0, 0);
                                }
                    } else {
#pragma unroll
                    for (int t2 = 0; t2 < NTAP; ++t2)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            if (TAIL_BRANCH && t2 == IPW - 1 && !last_tap_live) continue;
                            if constexpr (sizeof(T) == 2) acc[t2][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[idx & 1][ni], fb[idx & 1][t2], acc[t2][ni], 0, 0, 0);
                            else acc[t2][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[idx & 1][ni], fb[idx & 1][t2], acc[t2][ni], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            };
            // row-reuse phase (RR): rows r = 0 .. TD*TH-1 of the brick in order, fragment ring of 4 halo rows, A / extra-tap
            // fragments and the ring's new row read one row ahead; a plane's first three rows are read after the previous row's MFMAs
            auto rr_phase = [&](auto has_x_c, auto nrow_c, const char* bbase, const char* abase) {
                constexpr bool HAS_X = decltype(has_x_c)::value;
                constexpr int NROW = decltype(nrow_c)::value;           // rows this wave walks: the whole brick, or half of it (rr16)
                constexpr int KPG = NROW / NG;
                static_assert(!RR || (TD * TH == KPW && NROW % NG == 0 && NROW % TH == 0), "one k-group per 16-voxel row");
                typedef bf16x4 __attribute__((address_space(3))) * lp;
                auto rd = [&](const char* p_, int step) -> bf16x8 {
                    const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p_));
                    const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p_ + 4 * step));
                    return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                };
                const char* xbase = bt + tapoff[IPW - 1] + b_lane;    // the extra tap (waves 0-2)
                bf16x8 ring[4], fxx[2], faa[2];
                faa[0] = rd(abase, RSA);
                if constexpr (HAS_X) fxx[0] = rd(xbase, RS);
#pragma unroll
                for (int j = 0; j < 3; ++j) ring[j] = rd(bbase + j * HW * RS, RS);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
#pragma unroll
                    for (int j = pf_lo(g, NA, NG); j < pf_lo(g + 1, NA, NG); ++j) issue_a(j);
#pragma unroll
                    for (int j = pf_lo(g, NB, NG); j < pf_lo(g + 1, NB, NG); ++j) issue_b(j);
#pragma unroll
                    for (int kk = 0; kk < KPG; ++kk) {
                        const int r = g * KPG + kk, lh = r % TH;
                        const int r1 = r + 1, ld1 = r1 / TH, lh1 = r1 % TH;
                        if (r1 < NROW) {
                            faa[r1 & 1] = rd(abase + r1 * 16 * RSA, RSA);
                            if constexpr (HAS_X) fxx[r1 & 1] = rd(xbase + ((ld1 * HH + lh1) * HW) * RS, RS);
                            if (lh1 != 0) ring[(lh1 + 2) & 3] = rd(bbase + ((ld1 * HH + lh1 + 2) * HW) * RS, RS);
                        }
#pragma unroll
                        for (int tb = 0; tb < 3; ++tb) acc[tb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(faa[r & 1], ring[(lh + tb) & 3], acc[tb][0], 0, 0, 0);
                        if constexpr (HAS_X) acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(faa[r & 1], fxx[r & 1], acc[3][0], 0, 0, 0);
                        if (r1 < NROW && lh1 == 0) {
#pragma unroll
                            for (int j = 0; j < 3; ++j) ring[j] = rd(bbase + ((ld1 * HH + j) * HW) * RS, RS);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // (a second copy of the phase only where it removes >= 1/8 of the slots: 27 taps on 32; the 2-D kernels' 9 on 10 stay branch-free)
            constexpr bool TWO_PHASES = IPW > 1 && (IPW * WPQ - TAPS) * BIU_TWO_PHASE_DIV >= IPW * WPQ;
            if constexpr (RR) {
                using Full = std::integral_constant<int, TD * TH>;
                using Half = std::integral_constant<int, TD * TH / 2>;
                if (rr16) {
                    // 16-channel tapped operand: columns 0-15 / 16-31 of the B fragment are the SAME 16 channels at two (kd, kw) pairs
                    // (u16_* below); waves 0, 1, 6, 7 walk half of the brick's rows, 2-5 all of them: 144 MFMAs per SIMD and brick
                    const char* bb = bt + u16_boff + b_lane16 + (u16_r0 / TH) * (HH * HW * RS);
                    const char* ab = at + a_lane + u16_r0 * 16 * RSA;
                    if (u16_half) rr_phase(std::false_type{}, Half{}, bb, ab); else rr_phase(std::false_type{}, Full{}, bb, ab);
                } else if (wave < 3) {
                    rr_phase(std::true_type{}, Full{}, bt + tapoff[0] + b_lane, at + a_lane);
                } else {
                    rr_phase(std::false_type{}, Full{}, bt + tapoff[0] + b_lane, at + a_lane);
                }
            } else if constexpr (TWO_PHASES) {
                if (!last_tap_live) mfma_phase(std::integral_constant<int, (IPW > 1 ? IPW - 1 : 1)>{});
                else mfma_phase(std::integral_constant<int, IPW>{});
            } else {
                mfma_phase(std::integral_constant<int, IPW>{});
            }
            WSTAMP(1);
#ifdef BIU_DIAG
            dsum_[7] += 1;
            if (!have_next && wdiag && tid == 0) {
                for (int q_ = 0; q_ < 8; ++q_) atomicAdd(wdiag + q_, dsum_[q_]);
                atomicAdd(wdiag + 8, __builtin_readcyclecounter() - t0c_);
                atomicAdd(wdiag + 9, __builtin_amdgcn_s_memrealtime() - t0r_);
            }
#endif
            if (!have_next) break;
            __syncthreads();
            WSTAMP(2);
            commit();
            WSTAMP(3);
            __syncthreads();
            WSTAMP(4);
            brick = nbrick;
            ++kround;
        }
    }

    if constexpr (DBIAS) {
        if (a.dbias_out != nullptr && it == 0) {                     // block-uniform
            __syncthreads();                                        // every wave is done with the tiles: bt's first floats become the sum
            float* lsum = (float*)bt;
            if (tid < CT) lsum[tid] = 0.f;
            __syncthreads();
            if (dbias_on) {
#pragma unroll
                for (int e = 0; e < PE; ++e) atomicAdd(&lsum[piece * PE + e], bsum[e]);
            }
            __syncthreads();
            if (tid < CT && jt * CT + tid < a.CB) atomicAdd(a.dbias_out + jt * CT + tid, lsum[tid]);
        }
    }
    // ---- flush once per block ------------------------------------------------------------------------------------------
    const int hf = lane >> 5;
    if (rr16) {
        // column j of an accumulator: channel j & 15 of pair A (j < 16) or pair B (j >= 16, double units only)
        const int ch = lane & 15, up = (lane >> 4) & 1;
        const int kd = (u16_unit < 3) ? up : 2, kw = (u16_unit < 3) ? u16_unit : u16_unit - 3;
        if (!(up && u16_unit >= 3)) {
#pragma unroll
            for (int tb = 0; tb < 3; ++tb) {
                const int tap = (kd * 3 + tb) * 3 + kw;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ii = it * CTA + (e & 3) + 8 * (e >> 2) + 4 * hf;
                    if (ii < a.CA) atomicAdd(a.ws + ((size_t)tap * a.CA + ii) * a.CB + ch, acc[tb][0][e]);
                }
            }
        }
        return;
    }
    const int jj = jt * CT + (lane & 31);
    float* wsp = a.ws + (FALL ? (size_t)wave * a.fold_slice_f : (size_t)0);       // (all-parity form: wave = parity class = its own G slice)
    if (jj < a.CB) {
#pragma unroll
        for (int t2 = 0; t2 < IPW; ++t2) {
            const int tap = tapid[t2];
            if (tap < TAPS) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ii = it * CTA + ni * CT + (e & 3) + 8 * (e >> 2) + 4 * hf;
                        if (ii < a.CA) atomicAdd(wsp + ((size_t)tap * a.CA + ii) * a.CB + jj, acc[t2][ni][e]);
                    }
            }
        }
    }
}

// ===============================================================================================================
// Rolling-window weight gradient (3x3x3, stride 1, bf16): the kernel the 3-D conv blocks of a train step use.
//
// k_wgrad_pipe above spends 45 % of a brick outside its MFMA phase: every brick's tiles come through registers (global load ->
// VALU -> ds_write_b128 at ~79 B/clk/CU) between two barriers while the matrix cores wait, and 2 of the 6 halo planes of the
// tapped operand are re-read by the next brick along D (in-kernel stamps: MFMA phase 54 %, commit 20 %, barriers 25 %;
// profiles/r03_wgrad_roll.md).  Here a block walks a COLUMN of the volume -- an 8 x 16 window in (H, W), all of D in steps of
// 2 planes -- and keeps the tapped operand in a RING of 6 halo planes in LDS:
//   * a step needs planes d0-1 .. d0+2; the 2 planes of the NEXT step and its 2 x 8 x 16 plain-operand tile are fetched by
//     LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPRs, no ds_write, out-of-volume pieces arrive as zeros from the descriptor's
//     range check) into the ring slots / the second A buffer nobody reads during the step, issued in front of the step's MFMAs;
//   * halo traffic falls from 2.1x to 1.4x of the tensor (4 of 6 planes are re-used from LDS);
//   * a producer transform (BatchNorm-affine + LeakyReLU of a lazy source) is applied IN PLACE in LDS by the lane that fetched the
//     piece, after its own vmcnt(0) -- no barrier in between; the fused BatchNorm backward of the plain operand (da, y -> dy, written
//     back over da) keeps the register path (2 pieces per thread);
//   * the fetch runs TWO steps ahead (ring of 4 plane pairs, 3 plain-operand buffers): one step (~2.5 us) is shorter than the memory
//     latency of a loaded chip -- with one step of lookahead every step stalled ~1 us on its fetch (ablation: profiles/r03_wgrad_roll.md);
//     the wave waits with a counted vmcnt that leaves the younger fetch in flight;
//   * ONE barrier per step.  MFMA work per step and SIMD: 112 MFMAs = 3584 cycles; everything else ~600.
// Work split inside the step is k_wgrad_pipe's row-reuse scheme: wave w owns the three kh taps of the (kd, kw) pair (w / 3, w % 3),
// the ninth pair's taps are the fourth slot of waves 0-2; a 4-slot fragment ring takes one new transposed B read per 16-voxel row.
// LDS: 4 x 24 KiB ring + 3 x 16 KiB plain-operand tiles + vectors = 145 KiB.
// ===============================================================================================================
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bload_lds16(unsigned voff, v4u_t rsrc, unsigned lds_base_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_base_uniform)
                 : "memory");
}
__device__ __forceinline__ v4u_t raw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long pa = (unsigned long long)base;
    v4u_t r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)pa);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

// Energy ablation builds (tools/build_variant.sh x -DBIU_WROLL_ABL=n; results are WRONG, timing only): bit 0 = no fetches after the first
// step of a column (MFMAs + LDS fragment reads only), bit 1 = fragments read once per step and re-used for every row (no LDS reads).
#ifndef BIU_WROLL_ABL
#define BIU_WROLL_ABL 0
#endif
#ifndef BIU_DIAG_WAVE
#define BIU_DIAG_WAVE 0        // which wave of a block stamps (BIU_DIAG builds)
#endif
constexpr int WR_TD = 2, WR_TH = 8, WR_TW = 16;
constexpr int WR_HH = WR_TH + 2, WR_HW = WR_TW + 2, WR_PL = WR_HH * WR_HW;      // 180 voxels per halo plane
constexpr int WR_RS = 64;                                                       // bytes per voxel row of a 32-channel bf16 tile
constexpr int WR_PLB = WR_PL * WR_RS;                                           // 11520
constexpr int WR_NBI = 24;                                                      // DMA instructions (1 KiB) per pair of planes: 23040 B, rounded up to 3 per wave
constexpr int WR_PAIRB = WR_NBI * 1024;
constexpr int WR_NPAIR = 4;                                                     // ring: 2 pairs read by the step, 2 being fetched (steps s + 1, s + 2)
constexpr int WR_BV = WR_TD * WR_TH * WR_TW;                                    // 256 voxels per step
constexpr int WR_AB = WR_BV * WR_RS;                                            // 16 KiB
constexpr int WR_NAI = WR_AB / 1024;                                            // 16
constexpr int WR_NABUF = 3;
constexpr size_t WR_LDS = (size_t)WR_NPAIR * WR_PAIRB + WR_NABUF * WR_AB + 9 * 32 * sizeof(float);

// A wave-uniform value the optimiser must not see through: keeps per-step offsets (plane * bytes, ring pair * bytes) out of strength
// reduction / loop-invariant hoisting, which turned every (piece, step parity, ring slot) combination into a live VGPR and spilled.
__device__ __forceinline__ unsigned opaque_s(unsigned x) { asm volatile("" : "+s"(x)); return x; }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// K2D: a BATCH of 2-D images through the same kernel -- the images are the planes of the "volume" (the tensors' memory layout [N][H][W][C] is
// that of a volume of depth N), only the centre depth tap exists (kd = 1: the 9 taps of a 3 x 3 kernel), so nothing couples two
// images: a step takes 2 images x 8 rows x 16 voxels, wave w < 6 owns the three kh taps of kw = w % 3 on image w / 3 of the step.  The
// 2-D weight gradients of a U-Net are memory-bound (32-64 channels: 68-140 FLOP/B); what they need from this kernel is its fetch pipeline
// (LDS-DMA two steps ahead, no VGPR staging, no commit phase), not its MFMA schedule.
// A16: the plain operand has 16 channels (dy of a 16-filter layer: decode6 of UNet3D(n_filter = 32)) -- a 32-row tile would multiply 16 rows
// of zeros.  Instead a row of the A tile holds the 16 channels of BOTH planes of the step ([plane 0 | plane 1], 64 bytes), i.e. rows 0-15 of an
// MFMA's result belong to plane 0 and rows 16-31 to plane 1: against halo plane j of the tapped operand that is depth tap kd = j for the upper
// half and kd = j - 1 for the lower one.  Four halo planes x 9 (kh, kw) = 36 accumulators cover all 27 taps of both planes with 288 MFMAs per
// step instead of 432; unit (j, kw): waves 0-3 own (w, 0) and (w, 2), waves 4-7 own (w - 4, 1) -- 72 MFMAs per SIMD and step.
template <bool BNF, bool K2D, bool A16 = false>
__global__ __launch_bounds__(512, 2) void k_wgrad_roll(WgradArgs a) {
    static_assert(!(K2D && A16), "the 16-channel plain operand form exists for volumes only");
    using T = bf16_t;
    using F = Frag<T>;
    constexpr int TD = WR_TD, TH = WR_TH, TW = WR_TW, HW = WR_HW, PL = WR_PL, RS = WR_RS, PLB = WR_PLB, PAIRB = WR_PAIRB;
    constexpr int CT = 32, PE = 8, TAPS = K2D ? 9 : 27, IPW = K2D ? 3 : 4, NROW = TD * TH;
    constexpr int NBK = WR_NBI / 8, NAK = A16 ? 1 : WR_NAI / 8;               // DMA instructions per wave and fetch: tapped pair 3, plain tile 2 (A16: 1)
    constexpr int NACC = A16 ? 6 : IPW;

    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    char* bring = (char*)lds;                          // [4][PAIRB]: pair p holds the halo planes gp with ((gp + 1) / 2) % 4 == p
    char* abuf = bring + WR_NPAIR * PAIRB;             // [3][WR_AB]: plain-operand tile of step s in buffer s % 3, rows [voxel][32 ch]
    float* lxf = (float*)(abuf + WR_NABUF * WR_AB);    // [3][CT] transform of B, [6][CT] BatchNorm backward of A
    constexpr int LB = 0, LBN = 3 * CT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int it = blockIdx.y / a.jt_count, jt = a.jt_begin + blockIdx.y % a.jt_count;
    const int piece = lane & 3;
    const int apc = A16 ? (piece & 1) : piece;           // channel piece of the plain operand this lane stages (A16: pieces 2, 3 are plane 1)
    const int ac0 = it * CT + apc * PE, bc0 = jt * CT + piece * PE;
    const bool apiece_ok = ac0 < a.CA, bpiece_ok = bc0 < a.CB;
    const bool b_src1 = a.pb1 && jt * CT >= a.bsplit;
    const char* pb_ = b_src1 ? a.pb1 : a.pb;
    const int bpitch_ = b_src1 ? a.bpitch1 : a.bpitch;
    const int bc_local = bc0 - (b_src1 ? a.bsplit : 0);
    const float* bs_ = b_src1 ? a.bs1_ : a.bs_;
    const float* bb_ = b_src1 ? a.bb1_ : a.bb_;
    const float* bl_ = b_src1 ? a.bl1_ : a.bl_;
    const bool b_xf = bs_ != nullptr;
    constexpr bool bn_fused = BNF;                       // (a.py != nullptr: the launcher picks the instantiation)
    const bool wb = bn_fused && a.write_back;
    if (tid < CT && bn_fused) {
        const int ca = it * CT + tid;
        const bool ok = ca < a.CA;
        lxf[LBN + 0 * CT + tid] = ok ? a.bn_scale[ca] : 1.f;
        lxf[LBN + 1 * CT + tid] = ok ? a.bn_shift[ca] : 0.f;
        lxf[LBN + 2 * CT + tid] = (ok && a.bn_slope) ? a.bn_slope[ca] : 1.f;
        lxf[LBN + 3 * CT + tid] = ok ? a.bn_cA[ca] : 0.f;
        lxf[LBN + 4 * CT + tid] = ok ? a.bn_cB[ca] : 0.f;
        lxf[LBN + 5 * CT + tid] = ok ? a.bn_cC[ca] : 0.f;
    }
    if (tid < CT) {
        const int cb = jt * CT + tid;
        const int cbl = cb - (b_src1 ? a.bsplit : 0);
        lxf[LB + 0 * CT + tid] = (b_xf && cb < a.CB) ? bs_[cbl] : 1.f;
        lxf[LB + 1 * CT + tid] = (b_xf && cb < a.CB) ? bb_[cbl] : 0.f;
        lxf[LB + 2 * CT + tid] = (b_xf && cb < a.CB) ? bl_[cbl] : 1.f;
    }

    floatx16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    // wave w: (kd, kw) pair (w / 3, w % 3); waves 0-2 also take tap (2, kh = w, 2) of the ninth pair.  K2D: wave w < 6 = (image w / 3 of
    // the step, kw = w % 3), taps kh = 0..2 -> 2-D tap kh * 3 + kw; waves 6, 7 only fetch.
    const int wkd = wave / 3, wkw = wave % 3;
    const bool has_x = A16 ? (wave < 4) : (!K2D && wave < 3);        // A16: waves 0-3 own a second unit
    int tapid[IPW];
#pragma unroll
    for (int t = 0; t < IPW; ++t) {
        if constexpr (K2D) tapid[t] = (wave < 6) ? t * 3 + wkw : TAPS;
        else tapid[t] = (t < 3) ? (wkd * 3 + t) * 3 + wkw : (has_x ? (2 * 3 + wave) * 3 + 2 : TAPS);
    }
    int ab_lane;
    {
        const int g = lane >> 4, li = lane & 15, qrow = li >> 2, p = li & 3, cg = g & 1, h = g >> 1;
        ab_lane = (8 * h + qrow) * RS + (16 * cg + 4 * p) * 2;                  // same lane map for both operands (row stride 64 B, stride 1)
    }

    // ---- this thread's DMA pieces (column-invariant): tapped pair instruction wave + 8k (k < 3), plain tile instruction wave + 8k (k < 2)
    unsigned bco[NBK], aco[NAK];                         // packed (plane | hh << 10 | hw << 20) / (ld | lh << 10 | lw << 20); 0xffffffff = no piece
#pragma unroll
    for (int k = 0; k < NBK; ++k) {
        const int v = ((wave + 8 * k) * 64 + lane) >> 2;
        const int pl = v / PL, r = v % PL;
        bco[k] = (v < 2 * PL && bpiece_ok) ? (unsigned)(pl | ((r / HW) << 10) | ((r % HW) << 20)) : 0xffffffffu;
    }
#pragma unroll
    for (int k = 0; k < NAK; ++k) {
        const int v = ((wave + 8 * k) * 64 + lane) >> 2;
        if constexpr (A16) aco[k] = apiece_ok ? (unsigned)((piece >> 1) | ((v >> 4) << 10) | ((v & 15) << 20)) : 0xffffffffu;     // row v = (lh, lw), plane = piece >> 1
        else aco[k] = apiece_ok ? (unsigned)((v >> 7) | (((v >> 4) & 7) << 10) | ((v & 15) << 20)) : 0xffffffffu;
    }
    const unsigned rowA = (unsigned)a.apitch * 2u, rowY = (unsigned)a.ypitch * 2u, rowB = (unsigned)bpitch_ * 2u;
    const size_t sampA = (size_t)a.GD * a.GH * a.GW * rowA, sampY = (size_t)a.GD * a.GH * a.GW * rowY, sampB = (size_t)a.BD * a.BH * a.BW * rowB;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)bring);

    // column state
    int cn = 0, ch0 = 0, cw0 = 0;
    v4u_t rsB, rsA;
    __amdgpu_buffer_rsrc_t rsAr, rsYr;
    // two fetches are in flight at any time (steps s + 1 and s + 2): the fetch of step k keeps its masks / staging registers in slot k & 1
    struct Slot { unsigned bmask, amask; uint4 pa[BNF ? NAK : 1], pyv[BNF ? NAK : 1]; };
    Slot sl0, sl1;
    sl0.bmask = sl0.amask = sl1.bmask = sl1.amask = 0;
    // per column and piece: byte offset of the piece at plane 0 of its kind (tapped: halo plane gp = 0; plain: d = 0) inside the sample, or
    // 0xffffffff when the piece lies outside the volume in (H, W) / has no channels -- a fetch then only adds plane * (bytes per plane)
    unsigned bbase[NBK], abase_[NAK], ybase_[BNF ? NAK : 1];
    unsigned planeB = 0, planeA = 0, planeY = 0;
    auto set_column = [&](int col) __attribute__((always_inline)) {
        int c = col;
        const int wb_ = c % a.nbw; c /= a.nbw;
        const int hb = c % a.nbh;
        cn = c / a.nbh;
        ch0 = hb * TH; cw0 = wb_ * TW;
        rsB = raw_rsrc(pb_ + (size_t)cn * sampB, (unsigned)sampB);
        rsA = raw_rsrc(a.pa + (size_t)cn * sampA, (unsigned)sampA);
        rsAr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.pa + (size_t)cn * sampA), 0, (int)(unsigned)sampA, 0x00020000);
        rsYr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.py + (size_t)cn * sampY), 0, bn_fused ? (int)(unsigned)sampY : 0, 0x00020000);
        planeB = (unsigned)(a.BH * a.BW) * rowB; planeA = (unsigned)(a.GH * a.GW) * rowA; planeY = (unsigned)(a.GH * a.GW) * rowY;
#pragma unroll
        for (int k = 0; k < NBK; ++k) {
            const unsigned x = bco[k];
            const int gh = ch0 - 1 + (int)((x >> 10) & 1023u), gw = cw0 - 1 + (int)(x >> 20);
            const bool ok = x != 0xffffffffu && (unsigned)gh < (unsigned)a.BH && (unsigned)gw < (unsigned)a.BW;
            bbase[k] = ok ? (unsigned)(gh * a.BW + gw) * rowB + (unsigned)bc_local * 2u + (x & 1023u) * planeB : 0xffffffffu;
        }
#pragma unroll
        for (int k = 0; k < NAK; ++k) {
            const unsigned x = aco[k];
            const int gh = ch0 + (int)((x >> 10) & 1023u), gw = cw0 + (int)(x >> 20);
            const bool ok = x != 0xffffffffu && gh < a.GH && gw < a.GW;
            const unsigned vox = (unsigned)(gh * a.GW + gw);
            abase_[k] = ok ? vox * rowA + (unsigned)ac0 * 2u + (x & 1023u) * planeA : 0xffffffffu;
            if constexpr (BNF) ybase_[k] = ok ? vox * rowY + (unsigned)ac0 * 2u + (x & 1023u) * planeY : 0xffffffffu;
        }
    };
    // A fetch = NBK pieces of the tapped operand (halo planes gp0, gp0 + 1 -> ring pair `pair`) + NAK pieces of the plain operand (planes
    // d0, d0 + 1 -> A buffer `buf` by LDS-DMA, or -- fused BatchNorm backward -- (da, y) into the slot's registers).  Piece by piece, so that
    // a step can spread them over its MFMA rows: issued back to back at the head of the step the 5 DMA instructions of each of the 8 waves
    // took ~1000 cycles during which no wave multiplied (in-kernel stamps, profiles/r03_wgrad_roll.md).
    struct FetchCtx { int pair, buf; bool bd0, bd1, ad0, ad1; unsigned dofB, dofA, dofY; };
    auto fetch_ctx = [&](int pair, int gp0, int buf, int d0) __attribute__((always_inline)) -> FetchCtx {
        FetchCtx f;
        f.pair = pair; f.buf = buf;
        f.bd0 = (unsigned)gp0 < (unsigned)a.BD; f.bd1 = (unsigned)(gp0 + 1) < (unsigned)a.BD;
        f.ad0 = d0 < a.GD; f.ad1 = d0 + 1 < a.GD;
        f.dofB = opaque_s((unsigned)gp0 * planeB);            // (mod 2^32: gp0 = -1 pairs with plane index 1 or an invalid plane)
        f.dofA = opaque_s((unsigned)d0 * planeA); f.dofY = opaque_s((unsigned)d0 * planeY);
        f.pair = (int)opaque_s((unsigned)pair); f.buf = (int)opaque_s((unsigned)buf);
        return f;
    };
    auto fetch_piece = [&](Slot& sl, const FetchCtx& f, int i) __attribute__((always_inline)) {      // i < NBK: tapped piece i; else plain piece i - NBK
        if (i < NBK) {
            const int k = i;
            const bool ok = bbase[k] != 0xffffffffu && ((bco[k] & 1u) ? f.bd1 : f.bd0);
            bload_lds16(ok ? bbase[k] + f.dofB : 0xffffffffu, rsB, lds0 + (unsigned)(f.pair * PAIRB + (wave + 8 * k) * 1024));
            sl.bmask = (k == 0 ? 0u : sl.bmask) | (ok ? (1u << k) : 0u);
        } else {
            const int k = i - NBK;
            const bool ok = abase_[k] != 0xffffffffu && ((aco[k] & 1u) ? f.ad1 : f.ad0);
            if constexpr (BNF) {
                const auto v0 = __builtin_amdgcn_raw_buffer_load_b128(rsAr, ok ? (int)(abase_[k] + f.dofA) : -1, 0, 0);
                const auto v1 = __builtin_amdgcn_raw_buffer_load_b128(rsYr, ok ? (int)(ybase_[k] + f.dofY) : -1, 0, 0);
                sl.pa[k] = make_uint4(v0[0], v0[1], v0[2], v0[3]);
                sl.pyv[k] = make_uint4(v1[0], v1[1], v1[2], v1[3]);
            } else {
                bload_lds16(ok ? abase_[k] + f.dofA : 0xffffffffu, rsA, lds0 + (unsigned)(WR_NPAIR * PAIRB + f.buf * WR_AB + (wave + 8 * k) * 1024));
            }
            sl.amask = (k == 0 ? 0u : sl.amask) | (ok ? (1u << k) : 0u);
        }
    };
    auto fetch_all = [&](Slot& sl, const FetchCtx& f, bool with_a) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NBK + NAK; ++i)
            if (i < NBK || with_a) fetch_piece(sl, f, i);
    };
    // once this thread's own fetch of a slot has landed: producer transform of its tapped pieces in place; BatchNorm backward of its
    // plain pieces into the A buffer (+ dy written back over da: always NAK store instructions per wave, masked lanes out of range)
    auto finish = [&](Slot& sl, int pair_, int buf_, int d0, bool with_a) __attribute__((always_inline)) {
        const int pair = (int)opaque_s((unsigned)pair_), buf = (int)opaque_s((unsigned)buf_);
        const unsigned dofA = opaque_s((unsigned)d0 * planeA);
        if (b_xf) {
            float sc[PE], sh[PE], sl_[PE];
#pragma unroll
            for (int e = 0; e < PE; ++e) { sc[e] = lxf[LB + piece * PE + e]; sh[e] = lxf[LB + CT + piece * PE + e]; sl_[e] = lxf[LB + 2 * CT + piece * PE + e]; }
#pragma unroll
            for (int k = 0; k < NBK; ++k) {
                if ((sl.bmask >> k) & 1u) {
                    uint4* p_ = (uint4*)(bring + pair * PAIRB + (wave + 8 * k) * 1024 + lane * 16);
                    *p_ = apply_xf16<T, PE>(*p_, sc, sh, sl_);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BNF) if (with_a) {
            // two sweeps, three coefficient vectors live at a time (dz parked in the storage type in between, as k_wgrad_pipe does:
            // same rounding sequence, so both kernels write the same dy)
            {
                float ks[PE], kh[PE], kl[PE];
#pragma unroll
                for (int e = 0; e < PE; ++e) { ks[e] = lxf[LBN + apc * PE + e]; kh[e] = lxf[LBN + CT + apc * PE + e]; kl[e] = lxf[LBN + 2 * CT + apc * PE + e]; }
#pragma unroll
                for (int k = 0; k < NAK; ++k) {
                    if ((sl.amask >> k) & 1u) {
                        float g[PE], yy[PE];
                        F::unpack(sl.pa[k], g);
                        F::unpack(sl.pyv[k], yy);
#pragma unroll
                        for (int e = 0; e < PE; ++e) g[e] *= (fmaf(ks[e], yy[e], kh[e]) > 0.f ? 1.f : kl[e]);
                        sl.pa[k] = F::pack(g);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                float ka[PE], kb[PE], kc[PE];
#pragma unroll
                for (int e = 0; e < PE; ++e) { ka[e] = lxf[LBN + 3 * CT + apc * PE + e]; kb[e] = lxf[LBN + 4 * CT + apc * PE + e]; kc[e] = lxf[LBN + 5 * CT + apc * PE + e]; }
#pragma unroll
                for (int k = 0; k < NAK; ++k) {
                    uint4 v = make_uint4(0, 0, 0, 0);
                    const bool ok = (sl.amask >> k) & 1u;
                    if (ok) {
                        float g[PE], yy[PE];
                        F::unpack(sl.pa[k], g);
                        F::unpack(sl.pyv[k], yy);
#pragma unroll
                        for (int e = 0; e < PE; ++e) g[e] = fmaf(ka[e], g[e], fmaf(kb[e], yy[e], kc[e]));
                        v = F::pack(g);
                    }
                    if (wb) __builtin_amdgcn_raw_buffer_store_b128(v4u_t{v.x, v.y, v.z, v.w}, rsAr, ok ? (int)(abase_[k] + dofA) : -1, 0, 0);
                    *(uint4*)(abuf + buf * WR_AB + (wave + 8 * k) * 1024 + lane * 16) = v;
                }
            }
        }
    };
    // wait until this wave's fetch of the OLDER slot has landed while the younger one (issued this step) stays in flight.  Vector-memory
    // operations retire in order; younger than the fetch waited for are: the write-back stores of the previous finish (NAK, when dy is
    // written back) and this step's fetch (NBK DMA + NAK DMA, or NBK DMA + 2 NAK register loads with the fused BatchNorm backward).
    auto wait_older = [&](bool younger_in_flight) __attribute__((always_inline)) {
        if (!younger_in_flight) { wait_vmcnt<0>(); return; }
        if (wb) wait_vmcnt<NAK + NBK + 2 * NAK>();
        else if (bn_fused) wait_vmcnt<NBK + 2 * NAK>();
        else wait_vmcnt<NBK + NAK>();
    };

    typedef bf16x4 __attribute__((address_space(3))) * lp;
    auto rd = [&](const char* p_) -> bf16x8 {
        const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p_));
        const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(p_ + 4 * RS));
        return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // one step: rows r = 0 .. 15 (plane ld = r / 8, row lh = r % 8) of the tile in A buffer `buf`; halo plane j (0..3) of the step at pj(j)
    auto mfma_step = [&](auto has_x_c, int s4, int buf, auto&& issue) __attribute__((always_inline)) {
        constexpr bool HAS_X = decltype(has_x_c)::value;
        auto pj = [&](int j) -> const char* { return bring + opaque_s((unsigned)(((s4 + (j >> 1)) & 3) * PAIRB + (j & 1) * PLB)); };
        if constexpr (A16) {
            const int uj = wave & 3, ukw = wave >> 2;
            const char* b0 = pj(uj) + ukw * RS + ab_lane;            // primary unit (halo plane uj, kw = 0 | 1)
            const char* c0 = pj(uj) + 2 * RS + ab_lane;              // second unit of waves 0-3 (halo plane uj, kw = 2)
            const char* ab = abuf + opaque_s((unsigned)(buf * WR_AB)) + ab_lane;
            bf16x8 ring[4], ring2[HAS_X ? 4 : 1], faa[2];
            faa[0] = rd(ab);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                ring[j] = rd(b0 + j * HW * RS);
                if constexpr (HAS_X) ring2[j] = rd(c0 + j * HW * RS);
            }
#pragma unroll
            for (int r = 0; r < TH; ++r) {
                if (r + 1 < TH) {
                    faa[(r + 1) & 1] = rd(ab + (r + 1) * 16 * RS);
                    ring[(r + 3) & 3] = rd(b0 + (r + 3) * HW * RS);
                    if constexpr (HAS_X) ring2[(r + 3) & 3] = rd(c0 + (r + 3) * HW * RS);
                }
#pragma unroll
                for (int tb = 0; tb < 3; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(faa[r & 1], ring[(r + tb) & 3], acc[tb], 0, 0, 0);
                if constexpr (HAS_X) {
#pragma unroll
                    for (int tb = 0; tb < 3; ++tb) acc[3 + tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(faa[r & 1], ring2[(r + tb) & 3], acc[3 + tb], 0, 0, 0);
                }
                if ((r & 1) == 0 && r / 2 < NBK + NAK) issue(r / 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        if constexpr (K2D) {
            // image ld = wave / 3 of the step (halo plane j = ld + 1: the centre depth tap), its 8 rows in order, ring over the kh taps
            if (wave < 6) {                                              // wave-uniform; waves 6, 7 issue their share of the fetch below
                const char* b0 = pj(wkd + 1) + wkw * RS + ab_lane;
                const char* ab = abuf + opaque_s((unsigned)(buf * WR_AB + wkd * TH * 16 * RS)) + ab_lane;
                bf16x8 ring[4], faa[2];
                faa[0] = rd(ab);
#pragma unroll
                for (int j = 0; j < 3; ++j) ring[j] = rd(b0 + j * HW * RS);
#pragma unroll
                for (int r = 0; r < TH; ++r) {
                    if (r + 1 < TH) {
                        faa[(r + 1) & 1] = rd(ab + (r + 1) * 16 * RS);
                        ring[(r + 3) & 3] = rd(b0 + (r + 3) * HW * RS);
                    }
#pragma unroll
                    for (int tb = 0; tb < 3; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(faa[r & 1], ring[(r + tb) & 3], acc[tb], 0, 0, 0);
                    if (r < NBK + NAK) issue(r);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NBK + NAK; ++i) issue(i);
            }
            return;
        }
        const char* b0 = pj(wkd) + wkw * RS + ab_lane;           // plane of row-plane 0 for this wave's kd
        const char* b1 = pj(wkd + 1) + wkw * RS + ab_lane;       // ... of row-plane 1
        const char* x0 = pj(2) + (wave * HW + 2) * RS + ab_lane; // extra tap (2, kh = wave, 2)
        const char* x1 = pj(3) + (wave * HW + 2) * RS + ab_lane;
        const char* ab = abuf + opaque_s((unsigned)(buf * WR_AB)) + ab_lane;
        bf16x8 ring[4], fxx[2], faa[2];
        faa[0] = rd(ab);
        if constexpr (HAS_X) fxx[0] = rd(x0);
#pragma unroll
        for (int j = 0; j < 3; ++j) ring[j] = rd(b0 + j * HW * RS);
        if (BIU_WROLL_ABL & 2) {                         // ablation: every fragment register holds real data, none is read again
            faa[1] = rd(ab + 16 * RS);
            ring[3] = rd(b0 + 3 * HW * RS);
            if constexpr (HAS_X) fxx[1] = rd(x0 + HW * RS);
        }
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            const int lh = r % TH;
            const int r1 = r + 1, ld1 = r1 / TH, lh1 = r1 % TH;
            if (r1 < NROW && !(BIU_WROLL_ABL & 2)) {
                faa[r1 & 1] = rd(ab + r1 * 16 * RS);
                if constexpr (HAS_X) fxx[r1 & 1] = rd((ld1 ? x1 : x0) + lh1 * HW * RS);
                if (lh1 != 0) ring[(lh1 + 2) & 3] = rd((ld1 ? b1 : b0) + (lh1 + 2) * HW * RS);
            }
#pragma unroll
            for (int tb = 0; tb < 3; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(faa[r & 1], ring[(lh + tb) & 3], acc[tb], 0, 0, 0);
            if constexpr (HAS_X) acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(faa[r & 1], fxx[r & 1], acc[3], 0, 0, 0);
            if (r1 < NROW && lh1 == 0 && !(BIU_WROLL_ABL & 2)) {
#pragma unroll
                for (int j = 0; j < 3; ++j) ring[j] = rd((ld1 ? b1 : b0) + j * HW * RS);
            }
            if (r % 3 == 1 && r / 3 < NBK + NAK) issue(r / 3);          // one piece of the fetch behind rows 1, 4, 7, 10, 13
            if ((r & 1) == 1 || r % 3 == 1) __builtin_amdgcn_sched_barrier(0);   // keeps the scheduler from hoisting a whole step's fragment reads (spills)
        }
    };

    const int G = gridDim.x;
    const int ncols = a.N * a.nbh * a.nbw, nsteps = (a.GD + TD - 1) / TD;
    auto column_of = [&](int k) -> int {
        if ((G & 7) == 0) return k * G + (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
        return k * G + (int)blockIdx.x;
    };
    // step s: the fetch of step s + 2 goes out (slot s & 1), the MFMAs of step s run, the fetch of step s + 1 (slot (s + 1) & 1) is finished
#ifdef BIU_DIAG
    unsigned long long* wdiag = a.diag;
    unsigned long long tprev_ = __builtin_readcyclecounter();
    unsigned long long dsum_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0c_ = tprev_, t0r_ = __builtin_amdgcn_s_memrealtime();
#define RSTAMP(k_) do { if (wdiag && (tid & 63) == 0 && (tid >> 6) == (BIU_DIAG_WAVE)) { unsigned long long now_ = __builtin_readcyclecounter(); dsum_[k_] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define RSTAMP(k_) do { } while (0)
#endif
    auto step = [&](Slot& mine, Slot& other, int s) __attribute__((always_inline)) {
        const bool f2 = (s + 2 < nsteps) && !((BIU_WROLL_ABL & 1) && s >= 1);
        const bool f1 = (s + 1 < nsteps) && !((BIU_WROLL_ABL & 1) && s >= 2);
        RSTAMP(0);
        const FetchCtx fc = fetch_ctx((s + 3) & 3, 2 * s + 5, (s + 2) % 3, 2 * s + 4);
        RSTAMP(1);
        auto issue = [&](int i) __attribute__((always_inline)) { if (f2) fetch_piece(mine, fc, i); };           // block-uniform f2
        if (has_x) mfma_step(std::true_type{}, s & 3, s % 3, issue); else mfma_step(std::false_type{}, s & 3, s % 3, issue);
        __builtin_amdgcn_sched_barrier(0);
        RSTAMP(2);
        if (f1) {
            wait_older(f2);
            RSTAMP(3);
            finish(other, (s + 2) & 3, (s + 1) % 3, 2 * s + 2, true);
        }
        RSTAMP(4);
        __syncthreads();
        RSTAMP(5);
#ifdef BIU_DIAG
        dsum_[7] += 1;
#endif
    };
    __syncthreads();                                     // lxf visible
    for (int kc = 0;; ++kc) {
        const int col = column_of(kc);
        if (col >= ncols) break;                         // block-uniform
        set_column(col);
        // prologue: planes -1, 0 -> pair 0 (finished at once); step 0: planes 1, 2 -> pair 1 + A buffer 0 (slot 0); step 1: planes 3, 4 ->
        // pair 2 + A buffer 1 (slot 1)
        fetch_all(sl0, fetch_ctx(0, -1, 0, 0), false);
        wait_vmcnt<0>();
        finish(sl0, 0, 0, 0, false);
        fetch_all(sl0, fetch_ctx(1, 1, 0, 0), true);
        if (nsteps > 1) fetch_all(sl1, fetch_ctx(2, 3, 1, 2), true);
        wait_vmcnt<0>();
        finish(sl0, 1, 0, 0, true);                      // (step 1's fetch, slot 1, is finished by step 0 like every later one)
        __syncthreads();
        for (int s = 0; s < nsteps; s += 2) {
            step(sl0, sl1, s);
            if (s + 1 < nsteps) step(sl1, sl0, s + 1);
        }
    }

#ifdef BIU_DIAG
    if (wdiag && (tid & 63) == 0 && (tid >> 6) == (BIU_DIAG_WAVE)) {
        for (int q_ = 0; q_ < 8; ++q_) atomicAdd(wdiag + q_, dsum_[q_]);
        atomicAdd(wdiag + 8, __builtin_readcyclecounter() - t0c_);
        atomicAdd(wdiag + 9, __builtin_amdgcn_s_memrealtime() - t0r_);
    }
#endif
    // ---- flush once per block (k_wgrad_pipe's layout: ws[tap][i][j], two 128-B segments per wave-instruction) ----------------------
    const int hf = lane >> 5;
    const int jj = jt * CT + (lane & 31);
    if constexpr (A16) {
        // accumulator (unit u, kh = tb): rows 0-15 = tap (kd = j, kh, kw) of plane 0, rows 16-31 = tap (kd = j - 1, kh, kw) of plane 1
        const int uj = wave & 3;
        if (jj < a.CB) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !has_x) break;
                const int kw = u ? 2 : (wave >> 2);
#pragma unroll
                for (int tb = 0; tb < 3; ++tb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int kd = uj - (e >> 3), ii = (e & 3) + 8 * ((e >> 2) & 1) + 4 * hf;
                        if (kd >= 0 && kd <= 2 && ii < a.CA) atomicAdd(a.ws + ((size_t)((kd * 3 + tb) * 3 + kw) * a.CA + ii) * a.CB + jj, acc[u * 3 + tb][e]);
                    }
            }
        }
        return;
    }
    if (jj < a.CB) {
#pragma unroll
        for (int t2 = 0; t2 < IPW; ++t2) {
            const int tap = tapid[t2];
            if (tap < TAPS) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ii = it * CT + (e & 3) + 8 * (e >> 2) + 4 * hf;
                    if (ii < a.CA) atomicAdd(a.ws + ((size_t)tap * a.CA + ii) * a.CB + jj, acc[t2][e]);
                }
            }
        }
    }
}

static bool wroll_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("BIU_DISABLE"); v = (e && strstr(e, "wroll")) ? 1 : 0; }
    return v == 1;
}

// Worth it when a column is long enough to amortise its prologue (two exposed fetch latencies per column) and there are enough columns
// to give every block of a (plain tile, tapped tile) pair at least one.
static bool wroll2d_disabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("BIU_DISABLE"); v = (e && strstr(e, "wroll2d")) ? 1 : 0; }
    return v == 1;
}
static bool wroll_fits(const WgradArgs& a) {
    const int ncols = a.N * ((a.GH + WR_TH - 1) / WR_TH) * ((a.GW + WR_TW - 1) / WR_TW);
    return a.GD >= 8 && ncols >= 8;
}
// 2-D (kd = 1): the batch is the depth axis (images are contiguous, [N][1][H][W][C] == [D = N][H][W][C]); worth it from 8 images on, when
// the whole batch stays inside one buffer descriptor (32-bit offsets)
static bool wroll2d_fits(const WgradArgs& a, i64 bytesA, i64 bytesB, i64 bytesY) {
    const int ncols = ((a.GH + WR_TH - 1) / WR_TH) * ((a.GW + WR_TW - 1) / WR_TW);
    // (measured, cfg3 shapes: 64->64 @256^2 -7 %, 256->256 @64^2 -11 %, 32->32 @512^2 +2 % -- a single-tile layer already sits on the HBM roofline)
    return a.N >= 8 && ncols >= 8 && a.CA * a.CB > 1024 && bytesA < BIU_MAX_SAMPLE_BYTES && bytesB < BIU_MAX_SAMPLE_BYTES && bytesY < BIU_MAX_SAMPLE_BYTES;
}

static int launch_wgrad_roll(WgradArgs a, hipStream_t st, bool k2d = false) {
    if (k2d) { a.GD = a.BD = a.N; a.N = 1; }
    static int a16off = -1;
    if (a16off < 0) { const char* e = getenv("BIU_DISABLE"); a16off = (e && strstr(e, "wroll16")) ? 1 : 0; }
    const bool a16 = !k2d && a.CA == 16 && !a16off;      // 16-channel plain operand: both planes of a step share the tile's 32 rows
    a.nbd = 1;
    a.nbh = (a.GH + WR_TH - 1) / WR_TH;
    a.nbw = (a.GW + WR_TW - 1) / WR_TW;
    const int ncols = a.N * a.nbh * a.nbw;
    a.nbricks = ncols;
    const int nit = (a.CA + 31) / 32;
    a.njt = (a.CB + 31) / 32;
    a.bricks_per_block = 0;
#ifdef BIU_DIAG
    a.diag = biu_diag_buffer;
#else
    a.diag = nullptr;
#endif
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_wgrad_roll<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WR_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_wgrad_roll<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WR_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_wgrad_roll<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WR_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_wgrad_roll<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WR_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_wgrad_roll<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WR_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_wgrad_roll<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WR_LDS) != hipSuccess)
            return biu_fail(BIU_ERR_LAUNCH, "wgrad_roll: cannot reserve %zu bytes of LDS", WR_LDS);
        attr_set = true;
    }
    auto launch = [&](int jt_begin, int jt_count, int write_back, bool with_bn) {
        WgradArgs b = a;
        if (!with_bn) b.py = nullptr;
        b.jt_begin = jt_begin; b.jt_count = jt_count; b.write_back = write_back;
        const int pairs = nit * jt_count;
        int g = num_cus() / pairs;
        if (g >= 16) g = grid_per_column(num_cus(), pairs);
        if (g < 1) g = 1;
        if (g > ncols) g = ncols;
        if (k2d) {
            if (with_bn) hipLaunchKernelGGL((k_wgrad_roll<true, true>), dim3(g, pairs), dim3(512), WR_LDS, st, b);
            else hipLaunchKernelGGL((k_wgrad_roll<false, true>), dim3(g, pairs), dim3(512), WR_LDS, st, b);
        } else if (a16) {
            if (with_bn) hipLaunchKernelGGL((k_wgrad_roll<true, false, true>), dim3(g, pairs), dim3(512), WR_LDS, st, b);
            else hipLaunchKernelGGL((k_wgrad_roll<false, false, true>), dim3(g, pairs), dim3(512), WR_LDS, st, b);
        } else {
            if (with_bn) hipLaunchKernelGGL((k_wgrad_roll<true, false>), dim3(g, pairs), dim3(512), WR_LDS, st, b);
            else hipLaunchKernelGGL((k_wgrad_roll<false, false>), dim3(g, pairs), dim3(512), WR_LDS, st, b);
        }
    };
    if (a.py && a.njt > 1) {             // as launch_wgrad: the first input-channel tile turns da into dy in place, the others read the finished dy
        launch(0, 1, 1, true);
        BIU_CHECK_LAUNCH("wgrad_roll");
        launch(1, a.njt - 1, 0, false);
    } else {
        launch(0, a.njt, a.py ? 1 : 0, a.py != nullptr);
    }
    BIU_CHECK_LAUNCH("wgrad_roll");
    return BIU_OK;
}

// ws[tap][i][j] -> dw[i][j][tap]
// (ld_cols > 0: dw is a channel slice [c_off, c_off + cols) of a (rows, ld_cols, taps) tensor)
__global__ void k_wgrad_finalize(const float* __restrict__ ws, int rows, int cols, int taps, float* __restrict__ dw, int ld_cols = 0, int c_off = 0) {
    const size_t total = (size_t)rows * cols * taps;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const size_t r = i / taps;
        const int c = (int)(r % cols);
        const int rr = (int)(r / cols);
        const size_t o = ld_cols > 0 ? ((size_t)rr * ld_cols + c_off + c) * taps + tap : i;
        dw[o] = ws[((size_t)tap * rows + rr) * cols + c];
    }
}

// zero-fill by a kernel of our own: a hipMemsetAsync captured in a hipGraph (bio_image_unet_amd/graph.py) was seen to lose its order against
// the kernels around it when eager work ran between two replays -- weight gradients accumulated onto stale workspace contents
__global__ void k_zero_f32(float* __restrict__ p, size_t n, float* __restrict__ q, int nq) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
    if (q && blockIdx.x == 0)
        for (int i = threadIdx.x; i < nq; i += blockDim.x) q[i] = 0.f;
}
static int zero_ws(void* ws, size_t bytes, float* extra, int nextra, hipStream_t st) {
    const size_t n = bytes / sizeof(float);
    hipLaunchKernelGGL(k_zero_f32, dim3(grid_for((i64)n, 256, 1024)), dim3(256), 0, st, (float*)ws, n, extra, nextra);
    if (hipGetLastError() != hipSuccess) return biu_fail(BIU_ERR_LAUNCH, "wgrad: zero-fill launch failed");
    return BIU_OK;
}

static bool wgrad_chan_ok(int cin, int cout) { return cin >= 16 && cin % 8 == 0 && cout >= 16 && cout % 8 == 0; }

static size_t wgrad_acc_bytes(int cin, int cout, int taps) { return (((size_t)cin * cout * taps * sizeof(float)) + 255) & ~(size_t)255; }

size_t biu_mfma_wgrad_workspace(int cin, int cout, int kd, int kh, int kw, int dtype) {
    if (!wgrad_chan_ok(cin, cout)) return 0;
    const size_t red = biu_chan_sum_workspace(cout);           // partials of the dbias reduction
    if (kh == 3 && kw == 3 && (kd == 1 || kd == 3)) return wgrad_acc_bytes(cin, cout, kd * 9) + red;
    if (kh == 2 && kw == 2 && (kd == 1 || kd == 2)) return wgrad_acc_bytes(cin, cout, kd * 4) + red;
    return 0;
}

static bool wgrad_ptrs_ok(const biu_act* x, const biu_act* dy, int dtype) {
    const size_t es = dsize(dtype);
    if ((uintptr_t)x->p % 16 || (uintptr_t)dy->p % 16 || ((size_t)x->pitch * es) % 16 || ((size_t)dy->pitch * es) % 16) return false;
    // 32-bit buffer offsets inside one sample; 24-bit multiplies on tile-local voxel offsets (a tile spans <= 10 planes)
    if (sample_bytes(x, es) >= BIU_MAX_SAMPLE_BYTES || sample_bytes(dy, es) >= BIU_MAX_SAMPLE_BYTES) return false;
    const i64 plane_x = (i64)x->h * x->w, plane_y = (i64)dy->h * dy->w;
    return 10 * (plane_x > plane_y ? plane_x : plane_y) < (1LL << 24);
}

bool biu_mfma_wgrad_ok(const biu_act* x, const biu_act* dy, int kd, int kh, int kw, int dilation, int dtype) {
    if (dilation != 1 || kh != 3 || kw != 3 || (kd != 1 && kd != 3)) return false;
    if (!wgrad_chan_ok(x->c, dy->c) || !wgrad_ptrs_ok(x, dy, dtype)) return false;
    if (kd == 1 && x->d != 1) return false;
    return true;
}

bool biu_mfma_convt_wgrad_ok(const biu_act* x, const biu_act* dy, int kd, int dtype) {
    return (kd == 1 || kd == 2) && wgrad_chan_ok(x->c, dy->c) && wgrad_ptrs_ok(x, dy, dtype);
}

template <typename T, int KD, int KHW, int S, int TD, int TH, int TW, int KSPLIT, int NI = 1, bool RR16 = false, bool FALL = false>
static int launch_wgrad(WgradArgs a, hipStream_t st) {
    constexpr int SD = (KD == 1) ? 1 : S;
    constexpr int HV = ((TD - 1) * SD + (FALL ? 3 : KD)) * ((TH - 1) * S + (FALL ? 3 : KHW)) * ((TW - 1) * S + (FALL ? 3 : KHW));     // must mirror k_wgrad_pipe
    constexpr int BV = (FALL ? 8 : 1) * TD * TH * TW;
    constexpr int PPV_ = 32 / (16 / (int)sizeof(T));
    constexpr int NA_ = (BV * PPV_ * NI + 511) / 512, NB_ = (HV * PPV_ + 511) / 512;
    const size_t tile_bytes = (size_t)(HV + BV * NI) * 32 * (SplitOf<T>::parts ? 2 * SplitOf<T>::parts : sizeof(T));    // split products: one bf16 plane per part
    const size_t lds_bytes = tile_bytes + (3 + 9 * NI) * 32 * sizeof(float) + (wgrad_tab_in_lds(tile_bytes, NA_ + NB_) ? (size_t)(NA_ + NB_) * 512 * sizeof(unsigned) : 0);
    a.nbd = (a.GD + TD - 1) / TD;
    a.nbh = (a.GH + TH - 1) / TH;
    a.nbw = (a.GW + TW - 1) / TW;
    a.nbricks = a.N * a.nbd * a.nbh * a.nbw;
    const int nit = (a.CA + 32 * NI - 1) / (32 * NI);
    a.njt = (a.CB + 31) / 32;
    a.bricks_per_block = 0;
    auto kern = k_wgrad_pipe<T, KD, KHW, S, TD, TH, TW, KSPLIT, NI, RR16, FALL>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return biu_fail(BIU_ERR_LAUNCH, "wgrad_pipe: cannot reserve %zu bytes of LDS", lds_bytes);
        attr_set = true;
    }
    auto launch = [&](int jt_begin, int jt_count, int write_back, bool with_bn) {
        WgradArgs b = a;
#ifdef BIU_DIAG
        b.diag = biu_diag_buffer;
#else
        b.diag = nullptr;
#endif
        if (!with_bn) b.py = nullptr;
        b.jt_begin = jt_begin; b.jt_count = jt_count; b.write_back = write_back;
        const int pairs = nit * jt_count;
        int g = num_cus() / pairs;                        // persistent: about one block per CU in total
        if (g >= 16) g = grid_per_column(num_cus(), pairs);   // multiple of 8 (linear block id mod 8 == blockIdx.x mod 8) unless that idles > 3 % of the CUs
        if (g < 1) g = 1;
        if (g > b.nbricks) g = b.nbricks;
        hipLaunchKernelGGL(kern, dim3(g, pairs), dim3(512), lds_bytes, st, b);
    };
    if (a.py && a.njt > 1) {
        // Fused BatchNorm backward rewrites da as dy IN PLACE.  The first input-channel tile does that (BatchNorm math in
        // its loader, dy written back); the remaining tiles are launched after it and read the finished dy as a plain
        // operand -- no y loads, no BatchNorm math, and stream order is the only ordering between workgroups relied on.
        launch(0, 1, 1, true);
        BIU_CHECK_LAUNCH("wgrad_pipe");
        launch(1, a.njt - 1, 0, false);
    } else {
        launch(0, a.njt, a.py ? 1 : 0, a.py != nullptr);
    }
    BIU_CHECK_LAUNCH("wgrad_pipe");
    return BIU_OK;
}

static int wgrad_xf(const biu_xform* xf, const float** s, const float** b, const float** l) {
    const bool has = xf && (xf->scale || xf->shift || xf->slope);
    if (has) BIU_REQUIRE(xf->scale && xf->shift && xf->slope, BIU_ERR_UNSUPPORTED, "wgrad_mfma: partial biu_xform");
    *s = has ? xf->scale : nullptr;
    *b = has ? xf->shift : nullptr;
    *l = has ? xf->slope : nullptr;
    return BIU_OK;
}

int biu_mfma_wgrad(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, int kh, int kw, float* dw, float* dbias,
                   void* ws, size_t ws_bytes, int dtype, hipStream_t st, const BnBwdFuse* bn, const biu_act* x1, const biu_xform* xf1,
                   int dw_ld_cols, int dw_c_off) {
    WgradArgs a;
    int rc_ = BIU_OK;
    // Small volumes with several input-channel tiles: the fused form runs as TWO launches (the first tile's blocks turn da into dy, the others
    // read the finished dy), each a fraction of the chip at these sizes (4 x 16^3: 32 + 96 blocks).  A separate BatchNorm-backward pass over a
    // tensor that lives in L2 costs less than the second launch's latency: 128 -> 256 @ 4 x 16^3 102 -> ~70 us.  (BIU_DISABLE=wsplitbn: fused)
    static int split_off = -1;
    if (split_off < 0) { const char* e = getenv("BIU_DISABLE"); split_off = (e && strstr(e, "wsplitbn")) ? 1 : 0; }
    if (bn && !split_off && dtype == BIU_BF16 && kd == 3 && x->c + (x1 ? x1->c : 0) > 32 && nvox(dy) <= 4 * 32 * 32 * 32) {
        rc_ = biu_bn_bwd_apply(dy, bn->y, bn->scale, bn->shift, bn->slope, bn->cA, bn->cB, bn->cC, dy, dtype, (biu_stream)st);
        if (rc_ != BIU_OK) return rc_;
        bn = nullptr;
    }
    if (bn) {
        a.py = (const char*)bn->y->p; a.ypitch = bn->y->pitch;
        a.bn_scale = bn->scale; a.bn_shift = bn->shift; a.bn_slope = bn->slope;
        a.bn_cA = bn->cA; a.bn_cB = bn->cB; a.bn_cC = bn->cC;
    } else {
        a.py = nullptr; a.ypitch = 0;
        a.bn_scale = a.bn_shift = a.bn_slope = a.bn_cA = a.bn_cB = a.bn_cC = nullptr;
    }
    a.pa = (const char*)dy->p;  a.apitch = dy->pitch;  a.CA = dy->c;      // plain operand: dy  (rows i = co)
    a.pb = (const char*)x->p;   a.bpitch = x->pitch;   a.CB = x->c;       // tapped operand: x (cols j = ci)
    a.dbias_out = nullptr;
    a.fold_par = -1;
    a.pb1 = nullptr; a.bpitch1 = 0; a.bsplit = 0; a.bs1_ = a.bb1_ = a.bl1_ = nullptr;
    if (x1) {                                                              // x = concat(x, x1)
        a.pb1 = (const char*)x1->p; a.bpitch1 = x1->pitch; a.bsplit = x->c; a.CB = x->c + x1->c;
        rc_ = wgrad_xf(xf1, &a.bs1_, &a.bb1_, &a.bl1_);
        if (rc_) return rc_;
    }
    a.ws = (float*)ws;
    a.as_ = a.ab_ = a.al_ = nullptr;
    int rc = wgrad_xf(xf, &a.bs_, &a.bb_, &a.bl_);
    if (rc) return rc;
    a.N = x->n;
    a.GD = a.BD = x->d; a.GH = a.BH = x->h; a.GW = a.BW = x->w;
    const int taps = kd * 9;
    const size_t need = wgrad_acc_bytes(a.CA, a.CB, taps);
    BIU_REQUIRE(ws_bytes >= need + (dbias ? biu_chan_sum_workspace(dy->c) : 0), BIU_ERR_WORKSPACE, "wgrad_mfma: workspace %zu too small", ws_bytes);
    if (int zr = zero_ws(ws, need, nullptr, 0, st)) return zr;
    static int rr16_off = -1;
    if (rr16_off < 0) { const char* e = getenv("BIU_DISABLE"); rr16_off = (e && strstr(e, "rr16")) ? 1 : 0; }
    if (dtype == BIU_BF16 && kd == 3 && a.CB == 16 && !x1 && !rr16_off) rc = launch_wgrad<bf16_t, 3, 3, 1, 4, 8, 16, 1, 1, true>(a, st);   // paired taps
    else if (dtype == BIU_BF16 && kd == 3 && !wroll_disabled() && wroll_fits(a)) rc = launch_wgrad_roll(a, st);      // rolling window + LDS-DMA
    else if (dtype == BIU_BF16 && kd == 1 && !wroll_disabled() && !wroll2d_disabled() &&
             wroll2d_fits(a, (i64)nvox(dy) * dy->pitch * 2, (i64)nvox(x) * x->pitch * 2 > (x1 ? (i64)nvox(x1) * x1->pitch * 2 : 0) ? (i64)nvox(x) * x->pitch * 2 : (i64)nvox(x1) * x1->pitch * 2,
                          bn ? (i64)nvox(bn->y) * bn->y->pitch * 2 : 0))
        rc = launch_wgrad_roll(a, st, true);                                                                                 // batch of images as the depth axis
    else if (dtype == BIU_BF16) rc = (kd == 3) ? launch_wgrad<bf16_t, 3, 3, 1, 4, 8, 16, 1>(a, st) : launch_wgrad<bf16_t, 1, 3, 1, 1, 16, 32, 4>(a, st);
    else if (kd == 1 && fp32_split_mode() == 2) rc = launch_wgrad<f32x6_t, 1, 3, 1, 1, 16, 16, BIU_WGRAD_SPLIT_STEPPED ? 4 : 2>(a, st);  // fp32 tensors, bf16x6 products
    else if (kd == 1 && fp32_split_mode() == 1) rc = launch_wgrad<f32x3_t, 1, 3, 1, 1, 16, 16, 4>(a, st);  // fp32 tensors, bf16x3 products
    else rc = (kd == 3) ? launch_wgrad<float, 3, 3, 1, 4, 4, 16, 1>(a, st) : launch_wgrad<float, 1, 3, 1, 1, 16, 16, 4>(a, st);
    if (rc != BIU_OK) return rc;
    hipLaunchKernelGGL(k_wgrad_finalize, dim3(grid_for((i64)a.CA * a.CB * taps, 256, 2048)), dim3(256), 0, st, (const float*)ws,
                       a.CA, a.CB, taps, dw, dw_ld_cols, dw_c_off);
    BIU_CHECK_LAUNCH("wgrad_finalize");
    if (dbias) return biu_chan_sum_vec(dy, dbias, (char*)ws + need, dtype, st);
    return BIU_OK;
}

// ---- weight gradient of nearest up-sampling + 3x3x3 conv on the coarse tensor ("fold") ------------------------------------------
//   G[p][t][co][ci] = sum_v dy[2v + p][co] * T(x)[v + t - 1 + p][ci]          one launch per output parity class p (k_wgrad_pipe<T, 2, 2, 1>)
//   dW[co][ci][k]   = sum_p G[p][t_p(k)]                                      t_p(k): the coarse tap the fine tap k reads under parity p
__global__ void k_upconv_wgrad_unfold(const float* __restrict__ ws, size_t slice_f, int cout, int cin, float* __restrict__ dw) {
    const size_t total = (size_t)cout * cin * 27;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % 27);
        const size_t r = idx / 27;
        const int ci = (int)(r % cin), co = (int)(r / cin);
        const int kk[3] = {k / 9, (k / 3) % 3, k % 3};
        float sum = 0.f;
        for (int p = 0; p < 8; ++p) {
            int t = 0;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const int pa = (p >> (2 - ax)) & 1;
                const int ta = pa == 0 ? (kk[ax] >= 1 ? 1 : 0) : (kk[ax] == 2 ? 1 : 0);
                t |= ta << (2 - ax);
            }
            sum += ws[(size_t)p * slice_f + ((size_t)t * cout + co) * cin + ci];
        }
        dw[idx] = sum;
    }
}
size_t biu_mfma_upconv_wgrad_workspace(int cin, int cout, int dtype) {
    if ((dtype != BIU_BF16 && dtype != BIU_F32) || !wgrad_chan_ok(cin, cout)) return 0;
    return 8 * wgrad_acc_bytes(cout, cin, 8);
}
int biu_mfma_upconv_wgrad(const biu_act* x, const biu_xform* xf, const biu_act* dy, float* dw, void* ws, size_t ws_bytes, int dtype,
                          hipStream_t st, const BnBwdFuse* bn) {
    WgradArgs a;
    if (bn) {
        a.py = (const char*)bn->y->p; a.ypitch = bn->y->pitch;
        a.bn_scale = bn->scale; a.bn_shift = bn->shift; a.bn_slope = bn->slope;
        a.bn_cA = bn->cA; a.bn_cB = bn->cB; a.bn_cC = bn->cC;
    } else {
        a.py = nullptr; a.ypitch = 0;
        a.bn_scale = a.bn_shift = a.bn_slope = a.bn_cA = a.bn_cB = a.bn_cC = nullptr;
    }
    a.pa = (const char*)dy->p;  a.apitch = dy->pitch;  a.CA = dy->c;      // plain operand: the parity sub-lattice of the FINE dy (rows i = co)
    a.pb = (const char*)x->p;   a.bpitch = x->pitch;   a.CB = x->c;       // tapped operand: the coarse x (cols j = ci)
    a.dbias_out = nullptr;
    a.pb1 = nullptr; a.bpitch1 = 0; a.bsplit = 0; a.bs1_ = a.bb1_ = a.bl1_ = nullptr;
    a.as_ = a.ab_ = a.al_ = nullptr;
    int rc = wgrad_xf(xf, &a.bs_, &a.bb_, &a.bl_);
    if (rc) return rc;
    a.N = x->n;
    a.GD = a.BD = x->d; a.GH = a.BH = x->h; a.GW = a.BW = x->w;          // brick grid = the coarse grid
    const size_t slice = wgrad_acc_bytes(a.CA, a.CB, 8);
    BIU_REQUIRE(ws_bytes >= 8 * slice, BIU_ERR_WORKSPACE, "upconv_wgrad: workspace %zu too small (need %zu)", ws_bytes, 8 * slice);
    if (int zr = zero_ws(ws, 8 * slice, nullptr, 0, st)) return zr;
    a.fold_slice_f = slice / sizeof(float);
    static int fall_off = -1;
    if (fall_off < 0) { const char* e = getenv("BIU_DISABLE"); fall_off = (e && strstr(e, "foldall")) ? 1 : 0; }
    static int fall_ca = -1;                               // (BIU_FALL_MAXCA moves the rule below for A/B runs)
    if (fall_ca < 0) { const char* e = getenv("BIU_FALL_MAXCA"); fall_ca = e ? atoi(e) : 128; }
    if (dtype == BIU_BF16 && a.CA <= fall_ca && bn == nullptr && !fall_off) {
        // all eight parity classes in one launch: wave = class, the coarse operand staged once per 2 x 4 x 16 brick (k_wgrad_pipe<..., FALL>);
        // plain dy only (no registers left for the y pieces of a fused BatchNorm backward).  Same-box: decode5 (dy 32 ch) 608 -> 410 us,
        // cfg4 step 12.69 -> 12.46 ms; with decode3 (dy 64 ch: two tiles, operands staged per tile) 12.30 ms; round 4, with the 32^3 level
        // folded: dy 128 ch (four tiles) 449 -> 352 us.  Wider dy stays on the per-class launches below, which stage two dy tiles per block.
        a.ws = (float*)ws;
        a.fold_par = 8;
        rc = launch_wgrad<bf16_t, 2, 2, 1, 2, 4, 16, 8, 1, false, true>(a, st);
        if (rc != BIU_OK) return rc;
    } else
    for (int p = 0; p < 8; ++p) {
        a.ws = (float*)((char*)ws + (size_t)p * slice);
        a.fold_par = p;
        // two 32-wide tiles of dy's channels per block when it has them: twice the MFMA work per staged tile of the coarse operand
        if (a.CA > 32) rc = dtype == BIU_BF16 ? launch_wgrad<bf16_t, 2, 2, 1, 4, 8, 16, 1, 2>(a, st) : launch_wgrad<float, 2, 2, 1, 4, 4, 16, 1, 2>(a, st);
        else rc = dtype == BIU_BF16 ? launch_wgrad<bf16_t, 2, 2, 1, 4, 8, 16, 1>(a, st) : launch_wgrad<float, 2, 2, 1, 4, 4, 16, 1>(a, st);
        if (rc != BIU_OK) return rc;
    }
    if (!dw) return BIU_OK;                              // (foldt: the caller turns G[p][t][co][ci] in ws into the gradients of both weight tensors)
    hipLaunchKernelGGL(k_upconv_wgrad_unfold, dim3(grid_for((i64)a.CA * a.CB * 27, 256, 2048)), dim3(256), 0, st, (const float*)ws, slice / sizeof(float),
                       a.CA, a.CB, dw);
    BIU_CHECK_LAUNCH("upconv_wgrad_unfold");
    return BIU_OK;
}

// ---- foldt backward ------------------------------------------------------------------------------------------------------------
// data gradients: d skip through the channel-sliced data-gradient image of the conv, d x_low through the composed fold (optionally with the
// BatchNorm-backward sums of x_low's producer from its epilogue)
int biu_mfma_foldt_dgrad(const biu_act* dy, const void* packed, const biu_act* dx_low, int acc_low, const biu_act* dskip, int acc_skip, int dtype,
                         hipStream_t st, float* bn_partial_low, const BnRedFuse* red_low, void* ws, size_t ws_bytes) {
    const FoldtBlob b = foldt_blob(dx_low->c, dskip->c, dy->c, dtype);
    const char* base = (const char*)packed;
    int rc = biu_mfma_conv(dy, nullptr, base + b.sdg, nullptr, 3, 3, 3, dskip, acc_skip, nullptr, dtype, st, nullptr, nullptr, ws, ws_bytes);
    if (rc != BIU_OK) return rc;
    return biu_mfma_upconv_dgrad(dy, base + b.dg, dx_low, acc_low, dtype, st, bn_partial_low, red_low);
}
// border sums of dy: R[state][co] = sum of dy over the voxels of border state (sd, sh, sw) != interior.  Threads = (voxel slot, channel); every
// slot keeps its OWN [27][c] table in LDS, so each table entry has one owner thread: plain read-modify-write, no atomics (most shell voxels share
// one of six face states -- one shared table had every wave of the block queueing on the same 32 addresses: 77 us at decode5).  Every block writes
// the sum of its slots' tables -- no global atomics, the sum over blocks (k_foldt_reduce_tables) runs in a fixed order
constexpr int FOLDT_SUM_BLOCKS = 1024;
inline size_t foldt_sum_lds(int c) { return (size_t)27 * (c > 256 ? c : 256) * sizeof(float); }
template <typename T>
__global__ __launch_bounds__(256) void k_foldt_border_sums(const char* __restrict__ dy, ShellDims sd, int c, int pitch, float* __restrict__ partial) {
    extern __shared__ float tab[];                        // [slots][27][c]
    const int lanes_c = c < 256 ? c : 256, slots = 256 / lanes_c;
    for (int i = threadIdx.x; i < slots * 27 * c; i += 256) tab[i] = 0.f;
    __syncthreads();
    const int cc0 = (int)threadIdx.x % lanes_c, slot = (int)threadIdx.x / lanes_c;
    const long nsv = sd.nA + sd.nB + sd.nC;
    // (a thread's voxels are a chain of dependent gathers: four of them in flight at a time)
    const long stride = (long)gridDim.x * slots;
    if (slot < slots) {
        float* mine = tab + (size_t)slot * 27 * c;
        for (long v = (long)blockIdx.x * slots + slot; v < nsv; v += 4 * stride) {
            const T* p[4]; float* r[4]; bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long vv = v + u * stride;
                ok[u] = vv < nsv;
                long vox = 0; int st = 0;
                if (ok[u]) shell_voxel(sd, vv, vox, st);
                p[u] = (const T*)dy + (size_t)vox * pitch;
                r[u] = mine + st * c;
            }
            for (int cc = cc0; cc < c; cc += lanes_c) {
                float val[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) val[u] = ok[u] ? (float)p[u][cc] : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) if (ok[u]) r[u][cc] += val[u];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * c; i += 256) {
        float t = 0.f;
        for (int sl = 0; sl < slots; ++sl) t += tab[(size_t)sl * 27 * c + i];
        partial[(size_t)blockIdx.x * 27 * c + i] = t;
    }
}
// chain rule from G[p][t][co][ci] (ws, parity slices `slice_f` floats apart) to the gradients of both weight tensors and the ConvT bias:
//   dW_conv[co][c][k] = sum_p sum_ci W_T[ci][c][q(p,k)] G[p][t_p(k)][co][ci]                          (c < cup: the up half of the concat)
//   dW_T[ci][c][q]    = sum_{(p,k): q(p,k) = q} sum_co W_conv[co][c][k] G[p][t_p(k)][co][ci]
//   db_T[c]           = sum_k sum_co W_conv[co][c][k] S_k[co],  S_k = sum of dy over the voxels where tap k stays inside = -(sum over the
//                       border states where it does not): sum_v dy = 0 exactly behind a train-mode BatchNorm
// R[split][state][co] = sum over a contiguous share of the blocks' tables (fixed order): grid (27 states, FOLDT_RED_SPLIT), threads over (part of the
// share, channel) with four independent loads in flight, LDS tree over the parts; k_foldt_inside_sums adds the shares
constexpr int FOLDT_RED_SPLIT = 8;
__global__ __launch_bounds__(256) void k_foldt_reduce_tables(const float* __restrict__ partial, int nblocks, int cout, float* __restrict__ R) {
    __shared__ float red[256];
    const int s_ = (int)blockIdx.x, per = (nblocks + (int)gridDim.y - 1) / (int)gridDim.y;
    const int b0 = (int)blockIdx.y * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    for (int c0 = 0; c0 < cout; c0 += 32) {                               // 32 channels x 8 parts per pass
        const int co = c0 + (int)(threadIdx.x & 31), part = (int)(threadIdx.x >> 5);
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        if (co < cout) {
            int b = b0 + part;
            for (; b + 24 < b1; b += 32)
#pragma unroll
                for (int u = 0; u < 4; ++u) s4[u] += partial[((size_t)(b + 8 * u) * 27 + s_) * cout + co];
            for (; b < b1; b += 8) s4[0] += partial[((size_t)b * 27 + s_) * cout + co];
        }
        red[threadIdx.x] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        __syncthreads();
        if (threadIdx.x < 32 && co < cout) {
            float t = 0.f;
            for (int pp = 0; pp < 8; ++pp) t += red[pp * 32 + threadIdx.x];
            R[((size_t)blockIdx.y * 27 + s_) * cout + co] = t;
        }
        __syncthreads();
    }
}
// S_k[co] = sum of dy over the voxels whose tap k stays inside = total - (sum of R over the border states where it does not), summed over the shares of
// k_foldt_reduce_tables.  grid (27 taps, Cout / 32), threads over (share, channel): 27 unconditional loads each, LDS tree over the shares
// (total = per-channel sum of dy over ALL voxels, or NULL when it is identically zero: dy behind a train-mode BatchNorm)
__global__ __launch_bounds__(256) void k_foldt_inside_sums(const float* __restrict__ R, int nsplit, int cout, float* __restrict__ Sk, const float* __restrict__ total) {
    __shared__ float red[256];
    const int k = (int)blockIdx.x, co = (int)blockIdx.y * 32 + (int)(threadIdx.x & 31), part = (int)(threadIdx.x >> 5);
    float sum = (total && part == 0 && co < cout) ? total[co] : 0.f;
    if (co < cout)
        for (int sp = part; sp < nsplit; sp += 8) {
#pragma unroll 9
            for (int s_ = 0; s_ < 27; ++s_) {
                const float v = R[((size_t)sp * 27 + s_) * cout + co];
                sum -= (s_ != 13 && tap_outside(k, s_)) ? v : 0.f;
            }
        }
    red[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x < 32 && co < cout) {
        float t = 0.f;
        for (int pp = 0; pp < 8; ++pp) t += red[pp * 32 + threadIdx.x];
        Sk[k * cout + co] = t;
    }
}
// (the ConvT bias is part of `up`: dW_conv[co][c][k] also gets b_T[c] S_k[co])
// dW_conv[co][c][k] = sum_p sum_ci G[p][t_p(k)][co][ci] W_T[ci][c][q(p,k)]  (+ b_T[c] S_k[co])          grid (cup / 32, Cout / 32, 27 = k)
__global__ __launch_bounds__(256) void k_foldt_chain_wconv(const float* __restrict__ G, size_t slice_f, const float* __restrict__ wt, int cin_low, int cup, int cout,
                                                           int ccat, float* __restrict__ dwc, const float* __restrict__ bt, const float* __restrict__ Sk) {
    __shared__ long offA[8], offB[8];
    const int k = (int)blockIdx.z;
    if (threadIdx.x < 8) {
        int t, q;
        foldt_tq((int)threadIdx.x, k, t, q);
        offA[threadIdx.x] = (long)threadIdx.x * (long)slice_f + (long)t * cout * cin_low;
        offB[threadIdx.x] = q;
    }
    __syncthreads();
    const int m0 = (int)blockIdx.y * 32, n0 = (int)blockIdx.x * 32;             // rows co, cols c; reduction ci
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    small_gemm_tile(8, cout, cup, cin_low, m0, n0, SgView{G, cin_low, 1}, offA, SgView{wt, (long)cup * 8, 8}, offB, acc);
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = m0 + 2 * ty + i, c = n0 + 2 * tx + j;
            if (co < cout && c < cup) dwc[((size_t)co * ccat + c) * 27 + k] = bt ? fmaf(bt[c], Sk[k * cout + co], acc[i][j]) : acc[i][j];
        }
}
// dW_T[ci][c][q] = sum_{(p,k): q(p,k) = q} sum_co G[p][t_p(k)][co][ci] W_conv[co][c][k]                    grid (cup / 32, Cin_low / 32, 8 = q)
// (for every fine tap k exactly one parity class has sub-position q: p = (q + k + 1) & 1 per axis)
__global__ __launch_bounds__(256) void k_foldt_chain_wt(const float* __restrict__ G, size_t slice_f, const float* __restrict__ wc, int cin_low, int cup, int cout,
                                                        int ccat, float* __restrict__ dwt) {
    __shared__ long offA[27], offB[27];
    const int q = (int)blockIdx.z;
    if (threadIdx.x < 27) {
        const int k = (int)threadIdx.x, kk[3] = {k / 9, (k / 3) % 3, k % 3};
        int p = 0, t = 0;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const int qa = (q >> (2 - ax)) & 1, pa = (qa + kk[ax] + 1) & 1;
            const int ta = pa == 0 ? (kk[ax] >= 1 ? 1 : 0) : (kk[ax] == 2 ? 1 : 0);
            p |= pa << (2 - ax); t |= ta << (2 - ax);
        }
        offA[k] = (long)p * (long)slice_f + (long)t * cout * cin_low;
        offB[k] = k;
    }
    __syncthreads();
    const int m0 = (int)blockIdx.y * 32, n0 = (int)blockIdx.x * 32;             // rows ci, cols c; reduction co
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    small_gemm_tile(27, cin_low, cup, cout, m0, n0, SgView{G, 1, cin_low}, offA, SgView{wc, (long)ccat * 27, 27}, offB, acc);
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = m0 + 2 * ty + i, c = n0 + 2 * tx + j;
            if (ci < cin_low && c < cup) dwt[((size_t)ci * cup + c) * 8 + q] = acc[i][j];
        }
}
__global__ __launch_bounds__(256) void k_foldt_chain_bt(const float* __restrict__ Sk, const float* __restrict__ wc, int cup, int cout, int ccat, float* __restrict__ dbt) {
    __shared__ float red[256];
    const int c = (int)blockIdx.x;                       // one block per ConvT output channel, threads over (co, k)
    float sum = 0.f;
    for (int i = threadIdx.x; i < cout * 27; i += 256) {
        const int co = i / 27, k = i % 27;
        sum = fmaf(wc[((size_t)co * ccat + c) * 27 + k], Sk[k * cout + co], sum);
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) dbt[c] = red[0];
}
size_t biu_mfma_foldt_wgrad_workspace(int cin_low, int cskip, int cout, int dtype) {
    if (!wgrad_chan_ok(cin_low, cout) || !wgrad_chan_ok(cskip, cout)) return 0;
    const size_t g = 8 * wgrad_acc_bytes(cout, cin_low, 8), sk = biu_mfma_wgrad_workspace(cskip, cout, 3, 3, 3, dtype);
    return (g > sk ? g : sk) + al256((size_t)FOLDT_SUM_BLOCKS * 27 * cout * sizeof(float)) + al256((size_t)(1 + FOLDT_RED_SPLIT) * 27 * cout * sizeof(float)) +
           al256(biu_fold_gemm_layout_floats(cin_low, cin_low, cout) * sizeof(float));          // (GEMM layouts of the chain rule, cup <= cin_low)
}
// da -> dy in place (BatchNorm + LeakyReLU backward in the loader of the skip half's weight gradient, which runs first); then G on the finished dy
int biu_mfma_foldt_wgrad(const biu_act* x_low, const biu_xform* xf_low, const biu_act* skip, const biu_xform* xf_skip, const biu_act* da, const BnBwdFuse* bn,
                         const float* dy_sum, const float* w_conv, const float* w_t, const float* b_t, int cup, float* dw_conv, float* dw_t, float* db_t, void* ws, size_t ws_bytes,
                         int dtype, hipStream_t st, int phases) {
    // phases: 1 = the skip half (da -> dy in place, its slice of dw_conv); 4 = G on the finished dy into ws; 2 = border sums of dy + the chain
    // rule on the tables.  Parts 4 and 2 read dy, x_low, ws and the parameters: they may run on another stream once part 1 is complete
    const int cin_low = x_low->c, cskip = skip->c, cout = da->c, ccat = cup + cskip;
    const size_t need = biu_mfma_foldt_wgrad_workspace(cin_low, cskip, cout, dtype);
    BIU_REQUIRE(need > 0 && ws_bytes >= need, BIU_ERR_WORKSPACE, "foldt_wgrad: workspace %zu too small (need %zu)", ws_bytes, need);
    const size_t pbytes = al256((size_t)FOLDT_SUM_BLOCKS * 27 * cout * sizeof(float)), sbytes = al256((size_t)(1 + FOLDT_RED_SPLIT) * 27 * cout * sizeof(float));
    const size_t lbytes = al256(biu_fold_gemm_layout_floats(cin_low, cin_low, cout) * sizeof(float));
    const size_t main_bytes = need - pbytes - sbytes - lbytes;
    float* R = (float*)((char*)ws + main_bytes);                    // per-block border tables
    float* Sk = (float*)((char*)ws + main_bytes + pbytes);
    float* lay = (float*)((char*)ws + main_bytes + pbytes + sbytes);
    const bool has_bias = b_t != nullptr || db_t != nullptr;
    if (phases & 1) {
        // 1. skip half of dW_conv (its slice of the channel axis), BatchNorm backward in the loader: da becomes dy
        int rc = biu_mfma_wgrad(skip, xf_skip, da, 3, 3, 3, dw_conv, nullptr, ws, main_bytes, dtype, st, bn, nullptr, nullptr, ccat, cup);
        if (rc != BIU_OK) return rc;
    }
    if (phases & 4) {
        // 2. G[p][t] on the finished dy (ws: the skip half's accumulators were flushed into dw_conv by its finalize on the same stream --
        //    a caller that runs this part on another stream orders it behind part 1 with an event)
        int rc = biu_mfma_upconv_wgrad(x_low, xf_low, da, nullptr, ws, main_bytes, dtype, st, nullptr);
        if (rc != BIU_OK) return rc;
    }
    if (!(phases & 2)) return BIU_OK;
    // 3. the ConvT bias: border sums of the finished dy (its shell only: 25 MB at 4 x 128^3), one table per block -> S_k (taps inside), needed
    //    by dW_conv (b_T is part of `up`) and by db_T.  Part of phase 2: nothing on the caller's critical path waits for it.
    if (has_bias) {
        const ShellDims sh = shell_dims(da->n, da->d, da->h, da->w);
        BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_foldt_border_sums<T>, dim3(FOLDT_SUM_BLOCKS), dim3(256), foldt_sum_lds(cout), st,
                                                     (const char*)da->p, sh, da->c, da->pitch, R));
        BIU_CHECK_LAUNCH("foldt_border_sums");
        float* Rsum = Sk + 27 * cout;                                   // (rest of the Sk region: FOLDT_RED_SPLIT tables)
        hipLaunchKernelGGL(k_foldt_reduce_tables, dim3(27, FOLDT_RED_SPLIT), dim3(256), 0, st, (const float*)R, FOLDT_SUM_BLOCKS, cout, Rsum);
        hipLaunchKernelGGL(k_foldt_inside_sums, dim3(27, (cout + 31) / 32), dim3(256), 0, st, (const float*)Rsum, FOLDT_RED_SPLIT, cout, Sk, dy_sum);
        BIU_CHECK_LAUNCH("foldt_inside_sums");
    }
    // 4. chain rule to the up half of dW_conv, to dW_T and to db_T
    const size_t slice_f = wgrad_acc_bytes(cout, cin_low, 8) / sizeof(float);
    if (cup <= cin_low && biu_fold_gemm_ok(cin_low, cup, cout)) {
        int rc = biu_fold_gemm_layouts(w_conv, ccat, cup, cout, w_t, cin_low, lay, st);
        if (rc == BIU_OK) rc = biu_fold_gemm_chain((const float*)ws, slice_f, lay, cin_low, cup, cout, ccat, dw_conv, dw_t, b_t, (const float*)Sk, st);
        if (rc != BIU_OK) return rc;
        if (db_t) hipLaunchKernelGGL(k_foldt_chain_bt, dim3(cup), dim3(256), 0, st, (const float*)Sk, w_conv, cup, cout, ccat, db_t);
        BIU_CHECK_LAUNCH("foldt_chain_bt");
        return BIU_OK;
    }
    hipLaunchKernelGGL(k_foldt_chain_wconv, dim3((cup + 31) / 32, (cout + 31) / 32, 27), dim3(256), 0, st, (const float*)ws, slice_f, w_t, cin_low, cup, cout, ccat,
                       dw_conv, b_t, (const float*)Sk);
    hipLaunchKernelGGL(k_foldt_chain_wt, dim3((cup + 31) / 32, (cin_low + 31) / 32, 8), dim3(256), 0, st, (const float*)ws, slice_f, w_conv, cin_low, cup, cout, ccat,
                       dw_t);
    if (db_t) hipLaunchKernelGGL(k_foldt_chain_bt, dim3(cup), dim3(256), 0, st, (const float*)Sk, w_conv, cup, cout, ccat, db_t);
    BIU_CHECK_LAUNCH("foldt_chain");
    return BIU_OK;
}

int biu_mfma_convt_wgrad(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, float* dw, float* dbias, void* ws,
                         size_t ws_bytes, int dtype, hipStream_t st) {
    WgradArgs a;
    a.py = nullptr; a.ypitch = 0;
    a.bn_scale = a.bn_shift = a.bn_slope = a.bn_cA = a.bn_cB = a.bn_cC = nullptr;
    a.pa = (const char*)x->p;   a.apitch = x->pitch;   a.CA = x->c;       // plain operand: x on the coarse grid (rows i = ci)
    a.pb = (const char*)dy->p;  a.bpitch = dy->pitch;  a.CB = dy->c;      // tapped operand: dy on the fine grid (cols j = co)
    a.fold_par = -1;
    a.pb1 = nullptr; a.bpitch1 = 0; a.bsplit = 0; a.bs1_ = a.bb1_ = a.bl1_ = nullptr;
    a.ws = (float*)ws;
    int rc = wgrad_xf(xf, &a.as_, &a.ab_, &a.al_);
    if (rc) return rc;
    a.bs_ = a.bb_ = a.bl_ = nullptr;
    a.N = x->n;
    a.GD = x->d; a.GH = x->h; a.GW = x->w;
    a.BD = dy->d; a.BH = dy->h; a.BW = dy->w;
    const int taps = kd * 4;
    const size_t need = wgrad_acc_bytes(a.CA, a.CB, taps);
    BIU_REQUIRE(ws_bytes >= need + (dbias ? biu_chan_sum_workspace(dy->c) : 0), BIU_ERR_WORKSPACE, "convt_wgrad_mfma: workspace %zu too small", ws_bytes);
    a.dbias_out = dbias;                                   // d bias = channel sums of dy, taken while the kernel stages dy (no second pass)
    if (int zr = zero_ws(ws, need, dbias, dbias ? dy->c : 0, st)) return zr;
    // two 32-wide tiles of x's channels per block when x has them: the fine-grid dy is then read half as often
    const bool wide = a.CA > 32;
    if (dtype == BIU_BF16) {
        if (wide) rc = (kd == 2) ? launch_wgrad<bf16_t, 2, 2, 2, 2, 4, 16, 2, 2>(a, st) : launch_wgrad<bf16_t, 1, 2, 2, 1, 8, 16, 4, 2>(a, st);
        else rc = (kd == 2) ? launch_wgrad<bf16_t, 2, 2, 2, 2, 4, 16, 2>(a, st) : launch_wgrad<bf16_t, 1, 2, 2, 1, 8, 16, 4>(a, st);
    } else {
        if (kd == 1 && fp32_split_mode() == 2) rc = wide ? launch_wgrad<f32x6_t, 1, 2, 2, 1, 8, 16, 4, 2>(a, st) : launch_wgrad<f32x6_t, 1, 2, 2, 1, 8, 16, 4>(a, st);
        else if (kd == 1 && fp32_split_mode() == 1) rc = wide ? launch_wgrad<f32x3_t, 1, 2, 2, 1, 8, 16, 4, 2>(a, st) : launch_wgrad<f32x3_t, 1, 2, 2, 1, 8, 16, 4>(a, st);
        else if (wide && kd == 1) rc = launch_wgrad<float, 1, 2, 2, 1, 8, 16, 4, 2>(a, st);      // (the fp32 3-D tiles leave no LDS for a second A tile)
        else rc = (kd == 2) ? launch_wgrad<float, 2, 2, 2, 2, 4, 16, 2>(a, st) : launch_wgrad<float, 1, 2, 2, 1, 8, 16, 4>(a, st);
    }
    if (rc != BIU_OK) return rc;
    hipLaunchKernelGGL(k_wgrad_finalize, dim3(grid_for((i64)a.CA * a.CB * taps, 256, 2048)), dim3(256), 0, st, (const float*)ws,
                       a.CA, a.CB, taps, dw);
    BIU_CHECK_LAUNCH("convt_wgrad_finalize");
    return BIU_OK;
}
