// Rolling-window 3x3x3 convolution with the weights RESIDENT IN REGISTERS (bf16, gfx950): forward and data gradient of the narrow
// full-resolution layers of a 3-D U-Net (16 <-> 32 and 32 -> 32 channels at 128^3: encode2 / decode6 of UNet3D(n_filter = 32), reference
// unet3d/unet3d.py:24-25, 48-49, and the skip halves of the folded decoder levels).
//
// Why another kernel: in k_conv_pipe / k_conv16_pipe these layers are one (brick, chunk) item per brick, i.e. every 512 voxels pay a
// register-staged halo tile (global load -> VALU transform -> ds_write_b128), two barriers and an epilogue for 1.7 k cycles of MFMA per
// wave, and every MFMA re-reads its weight fragment from LDS: in-kernel stamps put 60 % of an item outside the MFMA phase with the LDS at
// 190 of its 256 B/clk (profiles/r03_mfma_shape_ab.md section 4).  Here
//   * a block is 4 waves, ONE per SIMD, 512 registers each: the 27 x (Cin x Cout) weights of the layer sit in registers / AGPRs for the
//     life of the block (108 registers for 16 <-> 32, 216 for 32 -> 32) -- no weight fragment is ever read from LDS;
//   * a block walks a COLUMN of the volume: an 8 x 32 window in (H, W), all planes of a depth segment, PS output planes per step; the
//     input lives in a RING of halo planes in LDS, filled by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPR staging, no ds_write, no
//     commit phase; pieces outside the volume arrive as zeros from the descriptor's range check = the zero padding), LA steps ahead,
//     each wave waiting with a counted vmcnt for its own pieces only; every input plane is fetched once per column (halo 1.33x in
//     (H, W), none along D) instead of 1.5x per brick along D on top;
//   * the producer's BatchNorm-affine + LeakyReLU is applied IN PLACE in LDS by the lane that fetched the piece, after its own vmcnt,
//     spread between the MFMA groups of the step in front of the one that reads the plane; zero padding pieces are skipped (T(0) != 0);
//   * a wave owns a 2-row x PS-plane patch of the window: a row fragment read once serves every (kd, kh) tap that touches it --
//     0.33-0.67 KiB of LDS reads per MFMA-issue slot instead of 1.5 KiB;
//   * ONE barrier per step; BatchNorm statistics (forward) or the upstream block's BatchNorm-backward sums (data gradient) are kept in
//     registers across the whole column and reduced once per block: one partial row per block, as the brick kernels deliver.
// LDS image of a halo plane (10 x 34 voxels, flat index hv = hh * 34 + hw): blocks of 16 voxels, [block][8-channel piece][voxel] x 16 B,
// so that (a) one DMA instruction (64 lanes x 16 B) is a whole block (32 channels) or two (16 channels): its global addresses are 16 whole
// voxel rows; (b) the 16 lanes of a `ds_read_b128` group read 16 consecutive voxels of one piece = 256 contiguous bytes: conflict-free at
// every tap offset.  Fragment addresses of a lane are column-invariant registers (row x kw), the plane slot is added per step.
//
// MFMA shapes: 32 output channels per block column on v_mfma_f32_32x32x16_bf16 (voxels on the lane axis: a lane ends with 16 channels
// of ONE voxel, 16-byte stores after a permlane32 swap) or 16 output channels on v_mfma_f32_16x16x32_bf16 (8-byte stores), the
// packed weight images are the ones biu_mfma_pack already writes for k_conv_pipe / k_conv16_pipe.
#include "biu_internal.h"
#include <cstdlib>
#include <cstring>
#include <utility>
#include <type_traits>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4m __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

struct RollArgs {
    const char* x;
    char* y;
    const uint4* wpk;              // fragment image of this launch's MFMA shape
    const float* bias;
    const float* xs; const float* xb; const float* xl;      // input transform (all three or none)
    int xpitch, ypitch;            // elements
    int N, D, H, W;
    int Cin, Cout;
    int nbh, nbw;                  // windows per plane
    int nseg, seg;                 // depth segments per column, planes per segment (a multiple of the step)
    float* bn_partial;             // [gridDim.x][Cout][2] or null
    const char* red_y; int red_ypitch;
    const float* red_scale; const float* red_shift; const float* red_slope; const float* red_mean; const float* red_invstd;
    unsigned long long* diag;
};

#ifdef BIU_DIAG
extern "C" unsigned long long* biu_diag_buffer;
#endif

namespace {

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) -- the slot machinery of the step needs
// every index as a constant BEFORE the optimiser runs (an unrolled run-time loop over it is too large for the unroller's budget, stays a
// loop, and the register arrays it indexes land in scratch)
template <typename F, int... I>
__device__ __forceinline__ void rl_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void rl_static_for(F&& f) { rl_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int RL_TH = 8, RL_TW = 32, RL_HH = RL_TH + 2, RL_HW = RL_TW + 2, RL_PLV = RL_HH * RL_HW;       // 340 halo voxels per plane
constexpr int RL_NBLK = (RL_PLV + 15) / 16;                                                            // 22 blocks of 16 voxels

__device__ __forceinline__ void rl_bload_lds16(unsigned voff, v4u_t rsrc, unsigned lds_base_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_base_uniform)
                 : "memory");
}
__device__ __forceinline__ v4u_t rl_rsrc(const void* base, unsigned bytes) {
    const unsigned long long pa = (unsigned long long)base;
    v4u_t r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)pa);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ unsigned rl_opaque(unsigned x) { asm volatile("" : "+s"(x)); return x; }
template <int N> __device__ __forceinline__ void rl_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void rl_unpack8(const uint4& v, float* f) {
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(u[i] << 16);
        f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 rl_pack8(const float* f) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)f[i];
    return __builtin_bit_cast(uint4, o);
}
// T(v) = max(t, slope * t), t = scale * v + shift: the arithmetic of k_conv_pipe's lrelu_affine (same rounding, so both kernels stage the
// same bf16 operands)
__device__ __forceinline__ void rl_lrelu_affine8(float (&f)[8], const float* sc, const float* sh, const float* sl) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const floatx2 v = {f[e], f[e + 1]}, s = {sc[e], sc[e + 1]}, b = {sh[e], sh[e + 1]}, l = {sl[e], sl[e + 1]};
        const floatx2 t = __builtin_elementwise_fma(s, v, b);
        const floatx2 u = l * t;
        f[e] = fmaxf(t[0], u[0]);
        f[e + 1] = fmaxf(t[1], u[1]);
    }
}

// MFMA with the weight operand pinned to the accumulation-register file (BIU_ROLL_ASM): past 256 registers hipcc parks long-lived values in
// AGPRs and copies them back (4 x v_accvgpr_read + a hazard s_nop) in front of EVERY use -- two VALU instructions per MFMA for weights that
// could be read where they lie.  The asm form reads them in place; the hazards the compiler no longer sees are covered explicitly (results
// are read by VALU only after the s_nop block in front of pack_pieces; a dependent MFMA on the same accumulator needs no wait states).
#ifndef BIU_ROLL_ASM
#define BIU_ROLL_ASM 0
#endif
#ifndef BIU_ROLL_SGB
#define BIU_ROLL_SGB 0
#endif
#ifndef BIU_ROLL_ABL
#define BIU_ROLL_ABL 0            // timing ablations (results are WRONG): bit 0 = no fetch inside the steps, bit 1 = no output stores / sums
#endif
__device__ __forceinline__ void rl_mfma32(floatx16& acc, const uint4& w, const uint4& b) {
#if BIU_ROLL_ASM
    const v4u_t wv = {w.x, w.y, w.z, w.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wv), "v"(bv));
#else
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
#endif
}
__device__ __forceinline__ void rl_mfma16(floatx4m& acc, const uint4& w, const uint4& b) {
#if BIU_ROLL_ASM
    const v4u_t wv = {w.x, w.y, w.z, w.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wv), "v"(bv));
#else
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
#endif
}

#ifdef BIU_DIAG
#define RL_STAMP(k_) do { if (a.diag && tid == 0) { unsigned long long now_ = __builtin_readcyclecounter(); dsum_[k_] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define RL_STAMP(k_) do { } while (0)
#endif

// CIN: input channels (16 | 32).  M: output channels per block column = MFMA rows (32: 32x32x16, 16: 16x16x32, CIN = 32 only).
// XF: the input carries a lazy producer transform.  PS: output planes per step.  LA: fetch lookahead in steps (>= 2).  ST: epilogue sums -- 0 none, 1 BatchNorm statistics (sum y, sum y^2) of the
// stored output, 2 ("RED") the output is d loss / d a of an upstream conv block whose raw output is red_y: (sum dz, sum dz * yhat)
// (k_conv_pipe's red_mode 1).
template <int CIN, int M, int PS, int LA, int ST, bool XF, bool ACC = false>
__global__ __launch_bounds__(256, 1) void k_conv_roll(RollArgs a) {
    constexpr bool RED = ST == 2;
    static_assert(!(RED && ACC) && (!ACC || M == 32), "the accumulate form exists for 32-row tiles without the BatchNorm-backward sums");
    static_assert(CIN == 16 || CIN == 32, "input channels");
    static_assert(M == 32 || (M == 16 && CIN == 32), "MFMA shape");
    static_assert(LA >= 2 && LA <= 4 && (PS == 1 || PS == 2), "pipeline depth");
    constexpr int TH = RL_TH, TW = RL_TW, HW = RL_HW, PLV = RL_PLV;
    constexpr int PCS = CIN / 8;                       // 16-byte pieces per voxel
    constexpr int BLKB = 256 * PCS;                    // bytes of a 16-voxel block
    constexpr int PB = RL_NBLK * BLKB;                 // bytes of a plane image
    constexpr int NPI = PB / 1024;                     // DMA instructions per plane
    constexpr int KPP = (NPI + 3) / 4;                 // ... per wave and plane: instruction wave + 4 kk (a padding one where that is >= NPI)
    constexpr int NK = PS * KPP;                       // ... per wave and fetch unit (PS planes): piece k = pl * KPP + kk
    constexpr int NPL = PS * (LA + 1) + 2;             // ring slots (planes)
    constexpr int NKS = CIN / 16;                      // k-steps of the 32x32x16 shape
    constexpr bool G32 = M == 32;
    constexpr int NRL = (RED || ACC) ? 4 * PS : 0;     // vector-memory instructions a wave issues per step besides the fetch: y loads ...
    constexpr int NST = 4 * PS;                        // ... and output stores

    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    char* ring = (char*)lds;                                    // [NPL][PB]
    char* dump = ring + NPL * PB;                               // 1 KiB: where the padding DMA instructions land
    float* lxf = (float*)(dump + 1024);                         // [3][CIN] input transform
    float* lred = lxf + 3 * CIN;                                // [4 waves][M][2]
    float* lrs_ = lred + 4 * M * 2;                             // [3][M] RED: scale / shift / slope of the upstream block
    float* lbias = lrs_ + 3 * M;                                // [M]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool has_xf = XF;                                 // (a.xs != nullptr: the launcher picks the instantiation)
    const int co0 = (int)blockIdx.y * M;
    constexpr bool want_stats = ST != 0;

    for (int i = tid; i < 4 * M * 2; i += 256) lred[i] = 0.f;
    if (tid < CIN) {
        lxf[tid] = has_xf ? a.xs[tid] : 1.f;
        lxf[CIN + tid] = has_xf ? a.xb[tid] : 0.f;
        lxf[2 * CIN + tid] = has_xf ? a.xl[tid] : 1.f;
    }
    if constexpr (RED) {
        if (tid < M) {
            const int co = co0 + tid;
            const bool okc = co < a.Cout && a.red_scale != nullptr;
            lrs_[tid] = okc ? a.red_scale[co] : 0.f;
            lrs_[M + tid] = okc ? a.red_shift[co] : 0.f;
            lrs_[2 * M + tid] = (okc && a.red_slope) ? a.red_slope[co] : 1.f;
        }
    }

    // ---- weights: registers for the life of the block ---------------------------------------------------------------------------
    // 32x32x16: image [n-tile][k-step][tap][lane] (k_pack_weights); 16x16x32: [column][k-step 32][tap][m = 0][lane] (k_pack_weights16, mtl = 1)
    uint4 wr[27][G32 ? NKS : 1];
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int ks = 0; ks < (G32 ? NKS : 1); ++ks)
            wr[t][ks] = a.wpk[(((size_t)blockIdx.y * (G32 ? NKS : 1) + ks) * 27 + t) * 64 + lane];

    // ---- this lane's share of a fetch unit (column-invariant): instruction j = wave + 4 k of the unit covers plane j / NPI, LDS bytes
    // [1024 (j % NPI), + 1024) of it; lane l of the instruction is (block, piece, voxel) in image order
    const int l_piece = (lane >> 4) % PCS;                      // the 8-channel piece this lane stages, in EVERY instruction
    const int l_vox = (PCS == 4) ? (lane & 15) : ((lane >> 5) * 16 + (lane & 15));      // voxel inside the instruction's 16 (32) voxels
    const unsigned rowB = (unsigned)a.xpitch * 2u;
    const unsigned planeB = (unsigned)(a.H * a.W) * rowB;       // (a sample is < 4 GB: checked on the host)
    const size_t sampX = (size_t)a.D * planeB;
    const unsigned rowY = (unsigned)a.ypitch * 2u;
    const size_t sampY = (size_t)a.D * a.H * a.W * rowY;
    const unsigned rowR = RED ? (unsigned)a.red_ypitch * 2u : 0u;
    const size_t sampR = (size_t)a.D * a.H * a.W * rowR;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)ring);
    constexpr unsigned BAD = 0x40000000u, BAD2 = 0x80000000u;   // padding marks of the 32-bit offsets (see set_item / set_out_bases)

    // ---- fragment addresses (column-invariant): halo rows hh = 2 wave + rr of a plane, kw = 0..2
    //  32x32x16: lane (r = lane & 31 -> voxel, hf = lane >> 5): piece 2 ks + hf of voxel hv = hh * 34 + kw + r
    //  16x16x32: lane (n = lane & 15 -> voxel, q = lane >> 4 = piece): voxel hv = hh * 34 + 16 c + kw + n of half c
    constexpr int NFA = G32 ? 12 : 24;
    unsigned fa[NFA];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int c = 0; c < (G32 ? 1 : 2); ++c)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int hv = (2 * wave + rr) * HW + kw + (G32 ? (lane & 31) : (16 * c + (lane & 15)));
                const int pc = G32 ? (lane >> 5) : (lane >> 4);
                fa[(rr * (G32 ? 1 : 2) + c) * 3 + kw] = (unsigned)((hv >> 4) * BLKB + (hv & 15) * 16 + pc * 256);
            }

    // ---- epilogue state ------------------------------------------------------------------------------------------------------------
    // 32x32x16: lane holds, per row, channels 8 g + 4 hf + i of its voxel; after the permlane32 swap pieces p = 0, 1 of channels
    // 16 p + 8 hf .. + 7.  16x16x32: lane holds channels 4 q .. + 3 of voxel n.
    constexpr int NP = G32 ? 2 : 1, CPP = G32 ? 8 : 4;
    float s1[NP][CPP], s2[NP][CPP];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int e = 0; e < CPP; ++e) s1[p][e] = s2[p][e] = 0.f;
    if (tid < M) lbias[tid] = (a.bias && co0 + tid < a.Cout) ? a.bias[co0 + tid] : 0.f;

    const int G = gridDim.x;
    const int ncols = a.N * a.nbh * a.nbw, nitems = ncols * a.nseg;
    auto item_of = [&](int k) -> int {
        if ((G & 7) == 0) return k * G + (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
        return k * G + (int)blockIdx.x;
    };

    // column state
    int cn = 0, h0 = 0, w0 = 0, d0 = 0, dend = 0;
    v4u_t rsX;
    __amdgpu_buffer_rsrc_t rsY, rsR;
    // byte offset of the lane's piece k at plane 0 of the sample; BAD = padding / no piece.  A fetch adds the plane's offset (or BAD2 for a
    // plane outside the volume): every sum with a BAD term lies in [2^30, 2^32) without wrapping -- beyond the descriptor's range (a sample
    // is < 2^30 bytes, checked on the host), so the hardware's range check IS the padding test and `voff < sample bytes` the lane's own
    unsigned bbase[NK];
    auto set_item = [&](int item) __attribute__((always_inline)) {
        int c = item;
        const int sg = c % a.nseg; c /= a.nseg;
        const int wb = c % a.nbw; c /= a.nbw;
        const int hb = c % a.nbh;
        cn = c / a.nbh;
        h0 = hb * TH; w0 = wb * TW;
        d0 = sg * a.seg;
        dend = min(a.D, d0 + a.seg);
        rsX = rl_rsrc(a.x + (size_t)cn * sampX, (unsigned)sampX);
        rsY = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (size_t)cn * sampY), 0, (int)(unsigned)sampY, 0x00020000);
        rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(RED ? a.red_y + (size_t)cn * sampR : a.y), 0, RED ? (int)(unsigned)sampR : 0, 0x00020000);
#pragma unroll
        for (int kk = 0; kk < KPP; ++kk) {
            const int ji = wave + 4 * kk;
            const int hv = ji * (64 / PCS) + l_vox;                     // (an instruction holds 64 / PCS voxels)
            const int hh = hv / HW, hw = hv - hh * HW;
            const int gh = h0 - 1 + hh, gw = w0 - 1 + hw;
            const bool ok = ji < NPI && hv < PLV && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
            const unsigned b = ok ? (unsigned)(gh * a.W + gw) * rowB + (unsigned)l_piece * 16u : BAD;
#pragma unroll
            for (int pl = 0; pl < PS; ++pl) bbase[pl * KPP + kk] = b;
        }
    };
    // (set_out_bases follows set_item in the item loop: it needs w0)
    auto slot_of = [&](int q) -> int { return (q + 1) % NPL; };            // ring slot of relative plane q >= -1
    // offset of the lane's piece k of the unit whose first relative plane is q0 (>= sample bytes: padding), and its LDS address
    auto piece_voff = [&](int q0, int k) __attribute__((always_inline)) -> unsigned {
        const int dg = d0 + q0 + k / KPP;
        const unsigned dof = rl_opaque((unsigned)dg < (unsigned)a.D ? (unsigned)dg * planeB : BAD2);
        return bbase[k] + dof;
    };
    auto piece_lds = [&](int q0, int k) __attribute__((always_inline)) -> unsigned {
        const int kk = k % KPP, ji = wave + 4 * kk;
        const unsigned in_ring = (unsigned)(slot_of(q0 + k / KPP) * PB) + (unsigned)ji * 1024u;
        if (4 * kk + 3 < NPI) return rl_opaque(in_ring);
        return rl_opaque(ji < NPI ? in_ring : (unsigned)(NPL * PB));           // a padding instruction lands in the dump slot
    };
    auto fetch_piece = [&](int q0, int k) __attribute__((always_inline)) {
        rl_bload_lds16(piece_voff(q0, k), rsX, lds0 + piece_lds(q0, k));
    };
    auto load_xf = [&](float* sc, float* sh, float* sl) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = lxf[l_piece * 8 + e]; sh[e] = lxf[CIN + l_piece * 8 + e]; sl[e] = lxf[2 * CIN + l_piece * 8 + e]; }
    };
    // once the lane's own piece has landed: producer transform in place (padding pieces stay zero).  Branch-free and in two halves -- the LDS
    // read in one MFMA group, arithmetic + write-back in the next -- so that neither the read's latency nor a basic-block boundary keeps the
    // MFMAs of a group from issuing (a lane rewrites its own padding pieces unchanged)
    auto fin_read = [&](int q0, int k) __attribute__((always_inline)) -> uint4 {
        return *(const uint4*)(ring + piece_lds(q0, k) + lane * 16);
    };
    auto fin_write = [&](int q0, int k, const uint4& v) __attribute__((always_inline)) {
        float f[8], sc[8], sh[8], sl[8];
        load_xf(sc, sh, sl);                            // (from LDS each time: 24 registers less across the MFMA phase)
        rl_unpack8(v, f);
        rl_lrelu_affine8(f, sc, sh, sl);
        const uint4 t = rl_pack8(f);
        const bool ok = piece_voff(q0, k) < (unsigned)sampX;
        *(uint4*)(ring + piece_lds(q0, k) + lane * 16) = make_uint4(ok ? t.x : v.x, ok ? t.y : v.y, ok ? t.z : v.z, ok ? t.w : v.w);
    };
    auto finish_piece = [&](int q0, int k) __attribute__((always_inline)) { fin_write(q0, k, fin_read(q0, k)); };

#ifdef BIU_DIAG
    unsigned long long tprev_ = __builtin_readcyclecounter();
    unsigned long long dsum_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0c_ = tprev_, t0r_ = __builtin_amdgcn_s_memrealtime();
#endif

    // Output (and upstream-output) offsets: a per-lane base (column + channel bytes, BAD when the lane's column lies outside the volume) plus a
    // wave-uniform row offset (BAD2 when the row / plane lies outside the volume or the segment): as for the fetch, every sum with a BAD term
    // is >= 2^30 > the sample's bytes, so the descriptor's range check drops the store (returns zeros for the load) and `off < sample bytes`
    // is the lane's own validity -- no selects, no branches, two registers.  (channels: the plan guarantees Cout = columns x M)
    unsigned vY[G32 ? 1 : 2], vR[G32 ? 1 : 2];
    auto set_out_bases = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < (G32 ? 1 : 2); ++c) {
            const int gw = w0 + (G32 ? (lane & 31) : (16 * c + (lane & 15)));
            const unsigned cb = (unsigned)(co0 + (G32 ? 8 * (lane >> 5) : 4 * (lane >> 4))) * 2u;
            vY[c] = gw < a.W ? (unsigned)gw * rowY + cb : BAD;
            vR[c] = gw < a.W ? (unsigned)gw * rowR + cb : BAD;
        }
    };
    auto row_off = [&](int d, int r2, unsigned rowbytes) __attribute__((always_inline)) -> unsigned {
        const int gh = h0 + 2 * wave + r2;
        return rl_opaque((d < dend && gh < a.H) ? (unsigned)((d * a.H + gh) * a.W) * rowbytes : BAD2);
    };

    // fragments of one MFMA group: the 4 halo rows of the wave at (relative plane q, kw) -- per k-step (32x32x16) or per half (16x16x32)
    constexpr int NF = G32 ? NKS * 4 : 8;
    constexpr int NGRP = (PS + 2) * 3;
    constexpr int F0 = NGRP - NK - 1;                                 // first group that reads a piece of the next step's unit for its transform
    static_assert(F0 >= 0 && NST <= NGRP, "the groups of a step carry its fetch, the transform of the next unit and the previous step's epilogue");
    uint4 frb[2][NF];
    auto load_frags = [&](uint4 (&fr)[NF], int q, int kw) __attribute__((always_inline)) {
        const char* pbase = ring + rl_opaque((unsigned)(slot_of(q) * PB));
        if constexpr (G32) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) fr[ks * 4 + rr] = *(const uint4*)(pbase + fa[rr * 3 + kw] + ks * 512);
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) fr[c * 4 + rr] = *(const uint4*)(pbase + fa[(rr * 2 + c) * 3 + kw]);
        }
    };
    auto load_frag1 = [&](uint4 (&fr)[NF], int q, int kw, int j) __attribute__((always_inline)) {       // fragment j of that set
        const char* pbase = ring + rl_opaque((unsigned)(slot_of(q) * PB));
        if constexpr (G32) fr[j] = *(const uint4*)(pbase + fa[(j % 4) * 3 + kw] + (j / 4) * 512);
        else fr[j] = *(const uint4*)(pbase + fa[((j % 4) * 2 + j / 4) * 3 + kw]);
    };
    // RED: upstream raw output at this lane's output voxels of the step whose first plane is d (always NRL load instructions)
    struct YQ { uint4 v4[((RED || ACC) && G32) ? 4 * PS : 1]; v2u_t v2[(RED && !G32) ? 4 * PS : 1]; };
    struct Acc { floatx16 a32[G32 ? PS : 1][2]; floatx4m a16[G32 ? 1 : PS][2][2]; };
    struct Park { uint4 p4[G32 ? 4 * PS : 1]; uint2 p2[G32 ? 1 : 4 * PS]; };     // a step's output pieces in the storage type, as they will be stored
    auto load_y = [&](int d, uint4* o4, v2u_t* o2) __attribute__((always_inline)) {
        if constexpr (ACC) {                            // the output's present content at the step's voxels (the sum starts from it)
#pragma unroll
            for (int pl = 0; pl < PS; ++pl)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    const unsigned ro = row_off(d + pl, r2, rowY);
#pragma unroll
                    for (int x2 = 0; x2 < 2; ++x2) {
                        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)(vY[0] + ro + 32u * x2), 0, 0);
                        o4[(pl * 2 + r2) * 2 + x2] = make_uint4(v[0], v[1], v[2], v[3]);
                    }
                }
        }
        if constexpr (RED) {
#pragma unroll
            for (int pl = 0; pl < PS; ++pl)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    const unsigned ro = row_off(d + pl, r2, rowR);
#pragma unroll
                    for (int x2 = 0; x2 < 2; ++x2) {
                        if constexpr (G32) {
                            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)(vR[0] + ro + 32u * x2), 0, 0);
                            o4[(pl * 2 + r2) * 2 + x2] = make_uint4(v[0], v[1], v[2], v[3]);
                        } else {
                            o2[(pl * 2 + r2) * 2 + x2] = __builtin_amdgcn_raw_buffer_load_b64(rsR, (int)(vR[x2] + ro), 0, 0);
                        }
                    }
                }
        }
    };

    __syncthreads();                                    // lxf / lrs / lred visible
    for (int kc = 0;; ++kc) {
        const int item = item_of(kc);
        if (item >= nitems) break;                      // block-uniform
        set_item(item);
        set_out_bases();
        const int nsteps = (dend - d0 + PS - 1) / PS;
        // ---- prologue: planes -1 .. LA * PS in flight, the first PS + 2 of them landed and transformed ---------------------------
        constexpr int NU0 = (LA * PS + 2) / PS;         // units of the prologue: q0 = -1, -1 + PS, ...
        constexpr int NW0 = (PS + 2 + PS - 1) / PS;     // ... of which step 0 needs the first NW0 (PS = 1: 3, PS = 2: 2)
        // (RED: the y loads of step 0 go out with step 0 itself: its epilogue runs during step 1.  ACC: the output's content at step 0's voxels is
        //  requested here, in front of every fetch, so that the prologue's wait lands it)
        YQ yq_cur, yq_prev;
        if constexpr (ACC) load_y(d0, yq_cur.v4, yq_cur.v2);
#pragma unroll
        for (int u = 0; u < NU0; ++u)
#pragma unroll
            for (int k = 0; k < NK; ++k) fetch_piece(-1 + u * PS, k);
        rl_wait_vmcnt<(NU0 - NW0) * NK>();
        if (has_xf) {
#pragma unroll
            for (int u = 0; u < NW0; ++u)
#pragma unroll
                for (int k = 0; k < NK; ++k) finish_piece(-1 + u * PS, k);
        }
        __syncthreads();
        // units in flight behind the landed ones: NU0 - NW0 (they cover the planes steps 1 .. need first); step s waits for the unit
        // that completes step s + 1's planes, q0 = (s + 1) * PS + 1 - ... see fetch bookkeeping below
        // Unit bookkeeping: step s reads planes s PS - 1 .. s PS + PS.  Planes <= NW0 * PS - 2 have landed.  Step s needs, beyond step
        // s - 1, planes s PS + 1 .. s PS + PS = unit with q0 = s PS + 1 ("unit of step s").  The prologue issued the units of steps
        // 0 .. LA - 1 (and, PS = 1, the two planes in front of them); step s issues the unit of step s + LA and, at its start, waits for
        // the unit of step s + 1 and transforms it while it multiplies.
        load_frags(frb[0], -1, 0);                      // group 0 of step 0
        // Piece i of a step's epilogue (NST per step and wave): convert, (swap), store, sums -- branch-free: lanes outside the volume store
        // through an out-of-range offset and add zeros.  It runs one step LATE, piece by piece between the MFMA groups of the next step
        // (one wave per SIMD: nothing else would hide its ~50 VALU instructions per piece), on that step's parked accumulators.
        // pack: accumulators -> pieces in the storage type (bf16; the two half-waves exchange their channel groups so that a lane holds 16 bytes)
        auto pack_pieces = [&](const Acc& pa, Park& pk) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int pl = i / 4, r2 = (i / 2) % 2, x2 = i % 2;
                if constexpr (G32) {
                    bf16x4 g0, g1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { g0[e] = (__bf16)pa.a32[pl][r2][8 * x2 + e]; g1[e] = (__bf16)pa.a32[pl][r2][8 * x2 + 4 + e]; }
                    const uint2 ua = __builtin_bit_cast(uint2, g0), ub = __builtin_bit_cast(uint2, g1);
                    const auto sx = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
                    pk.p4[i] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                } else {
                    bf16x4 g0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) g0[e] = (__bf16)pa.a16[pl][r2][x2][e];
                    pk.p2[i] = __builtin_bit_cast(uint2, g0);
                }
            }
        };
        // ---- the work that rides between the MFMAs of a step, cut into STAGES of at most ~8 instructions ------------------------------------
        // One wave per SIMD: while a 32-cycle MFMA executes the wave can issue ~6 other instructions for free, but only if they are NEXT to
        // it -- hipcc's scheduler clusters them (a run of MFMAs, then a run of VALU during which the matrix pipe idles: measured, the transform
        // of a step's six pieces cost 1 600 exposed cycles, the epilogue 1 300).  So the step is written as slots  { a few stages ; one MFMA ;
        // sched_barrier }  and the streams below are dealt over the slots at compile time:
        //   O(i), i < NST: epilogue piece i of the PREVIOUS step  -- O0 store, then the sums (ST = 1: 3 stages, ST = 2: 8 stages)
        //   D(k), k < NK : fetch piece k of the unit of step s + LA
        //   X(k), k < NK : transform piece k of the unit of step s + 1 in place -- X0 read .. X6 select + write-back
        // in the order O(0) D(0) X(0) O(1) D(1) X(1) ...: behind the unit's last fetch piece D(NK - 1) a step issues max(0, NST - NK) stores.
        constexpr int SO = ST == 0 ? 1 : (ST == 1 ? 4 : 9), SX = XF ? 7 : 0;
        constexpr int NWI = NST > NK ? NST : NK;                         // work items i: O(i) (i < NST), D(i) (i < NK), X(i) (i < NK)
        auto w_len = [](int i) constexpr -> int { return (i < NST ? SO : 0) + (i < NK ? 1 + SX : 0); };
        constexpr int WT = [] { int t = 0; for (int i = 0; i < NWI; ++i) t += (i < NST ? SO : 0) + (i < NK ? 1 + SX : 0); return t; }();
        constexpr int NMT = G32 ? 54 * NKS * PS : 108 * PS;              // MFMAs = slots of a step
        // stage state (one O piece and one X piece are in flight at a time)
        uint4 o_piece = make_uint4(0, 0, 0, 0);
        float o_f[CPP], o_y[CPP], o_m[CPP], o_rsc[CPP], o_rsh[CPP], o_rsl[CPP];
        uint4 x_v = make_uint4(0, 0, 0, 0);
        float x_f[8], x_t[8], x_u[8], x_sc[8], x_sh[8], x_sl[8];
        if constexpr (XF && ST != 2) load_xf(x_sc, x_sh, x_sl);          // (registers to spare: the vectors stay resident; ST = 2 reads them per piece)
        Park park;
        if constexpr (RED) {                            // (step 0 runs a fully masked epilogue on these: 0 x garbage could be a NaN)
#pragma unroll
            for (int i = 0; i < ((RED && G32) ? 4 * PS : 1); ++i) yq_cur.v4[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < ((RED && !G32) ? 4 * PS : 1); ++i) yq_cur.v2[i] = v2u_t{0u, 0u};
        }
        auto o_stage = [&](auto I_, auto ST_, const Park& pk, const YQ& yq, int dprev) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, st = decltype(ST_)::value;
            constexpr int pl = i / 4, r2 = (i / 2) % 2, x2 = i % 2;
            constexpr int sx = G32 ? x2 : 0;                                // which of the lane's statistic sets the piece belongs to
            if constexpr (st == 0) {
                if constexpr (G32) {
                    const unsigned off = vY[0] + row_off(dprev + pl, r2, rowY) + 32u * x2;
                    const bool live = off < (unsigned)sampY;
                    const uint4 pq = pk.p4[i];
                    o_piece = make_uint4(live ? pq.x : 0u, live ? pq.y : 0u, live ? pq.z : 0u, live ? pq.w : 0u);
                    __builtin_amdgcn_raw_buffer_store_b128(v4u_t{o_piece.x, o_piece.y, o_piece.z, o_piece.w}, rsY, (int)off, 0, 0);
                } else {
                    const unsigned off = vY[x2] + row_off(dprev + pl, r2, rowY);
                    const bool live = off < (unsigned)sampY;
                    const uint2 pq = pk.p2[i];
                    o_piece = make_uint4(live ? pq.x : 0u, live ? pq.y : 0u, 0u, 0u);
                    __builtin_amdgcn_raw_buffer_store_b64(v2u_t{o_piece.x, o_piece.y}, rsY, (int)off, 0, 0);
                }
                if constexpr (RED) {                                     // the upstream block's vectors for this piece's channels (used from stage 3 on)
                    const float* lrs = lrs_ + rl_opaque(0u);
                    const int ci0 = G32 ? 16 * x2 + 8 * (lane >> 5) : 4 * (lane >> 4);
#pragma unroll
                    for (int h4 = 0; h4 < CPP / 4; ++h4) {
                        const float4 t0 = *(const float4*)(lrs + ci0 + 4 * h4), t1 = *(const float4*)(lrs + M + ci0 + 4 * h4),
                                     t2 = *(const float4*)(lrs + 2 * M + ci0 + 4 * h4);
                        o_rsc[4 * h4] = t0.x; o_rsc[4 * h4 + 1] = t0.y; o_rsc[4 * h4 + 2] = t0.z; o_rsc[4 * h4 + 3] = t0.w;
                        o_rsh[4 * h4] = t1.x; o_rsh[4 * h4 + 1] = t1.y; o_rsh[4 * h4 + 2] = t1.z; o_rsh[4 * h4 + 3] = t1.w;
                        o_rsl[4 * h4] = t2.x; o_rsl[4 * h4 + 1] = t2.y; o_rsl[4 * h4 + 2] = t2.z; o_rsl[4 * h4 + 3] = t2.w;
                    }
                }
            } else if constexpr (st == 1) {                               // values as stored (zeros outside the volume)
                const unsigned u[4] = {o_piece.x, o_piece.y, o_piece.z, o_piece.w};
#pragma unroll
                for (int e = 0; e < CPP / 2; ++e) { o_f[2 * e] = __uint_as_float(u[e] << 16); o_f[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
            } else if constexpr (ST == 1) {
                if constexpr (st == 2) {
#pragma unroll
                    for (int e = 0; e < CPP; ++e) s1[sx][e] += o_f[e];
                } else {
#pragma unroll
                    for (int e = 0; e < CPP; ++e) s2[sx][e] = fmaf(o_f[e], o_f[e], s2[sx][e]);
                }
            } else if constexpr (ST == 2) {
                if constexpr (st == 2) {
                    unsigned u[4];
                    if constexpr (G32) { const uint4 q = yq.v4[(pl * 2 + r2) * 2 + x2]; u[0] = q.x; u[1] = q.y; u[2] = q.z; u[3] = q.w; }
                    else { const v2u_t q = yq.v2[(pl * 2 + r2) * 2 + x2]; u[0] = q[0]; u[1] = q[1]; u[2] = u[3] = 0u; }
#pragma unroll
                    for (int e = 0; e < CPP / 2; ++e) { o_y[2 * e] = __uint_as_float(u[e] << 16); o_y[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
                } else if constexpr (st == 3) {
#pragma unroll
                    for (int e = 0; e < CPP; ++e) o_m[e] = fmaf(o_rsc[e], o_y[e], o_rsh[e]);                  // tt
                } else if constexpr (st == 4) {
#pragma unroll
                    for (int e = 0; e < CPP / 2; ++e) o_m[e] = o_m[e] > 0.f ? 1.f : o_rsl[e];
                } else if constexpr (st == 5) {
#pragma unroll
                    for (int e = CPP / 2; e < CPP; ++e) o_m[e] = o_m[e] > 0.f ? 1.f : o_rsl[e];
                } else if constexpr (st == 6) {
#pragma unroll
                    for (int e = 0; e < CPP; ++e) o_f[e] *= o_m[e];                                            // dz
                } else if constexpr (st == 7) {
#pragma unroll
                    for (int e = 0; e < CPP; ++e) s1[sx][e] += o_f[e];
                } else {
#pragma unroll
                    for (int e = 0; e < CPP; ++e) s2[sx][e] = fmaf(o_f[e], o_y[e], s2[sx][e]);                 // raw; centred when the partial row is written
                }
            }
        };
        auto x_stage = [&](auto K_, auto ST_, int q_fin) __attribute__((always_inline)) {
            constexpr int k = decltype(K_)::value, st = decltype(ST_)::value;
            if constexpr (st == 0) {
                x_v = fin_read(q_fin, k);
                if constexpr (ST == 2) load_xf(x_sc, x_sh, x_sl);
            } else if constexpr (st == 1) {
                rl_unpack8(x_v, x_f);
            } else if constexpr (st == 2) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const floatx2 v = {x_f[e], x_f[e + 1]}, sc2 = {x_sc[e], x_sc[e + 1]}, b = {x_sh[e], x_sh[e + 1]};
                    const floatx2 t = __builtin_elementwise_fma(sc2, v, b);
                    x_t[e] = t[0]; x_t[e + 1] = t[1];
                }
            } else if constexpr (st == 3) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const floatx2 t = {x_t[e], x_t[e + 1]}, l = {x_sl[e], x_sl[e + 1]};
                    const floatx2 u = l * t;
                    x_u[e] = u[0]; x_u[e + 1] = u[1];
                }
            } else if constexpr (st == 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x_f[e] = fmaxf(x_t[e], x_u[e]);
            } else if constexpr (st == 5) {
                const uint4 t = rl_pack8(x_f);
                x_f[0] = __uint_as_float(t.x); x_f[1] = __uint_as_float(t.y); x_f[2] = __uint_as_float(t.z); x_f[3] = __uint_as_float(t.w);
            } else {
                const bool ok = piece_voff(q_fin, k) < (unsigned)sampX;
                *(uint4*)(ring + piece_lds(q_fin, k) + lane * 16) =
                    make_uint4(ok ? __float_as_uint(x_f[0]) : x_v.x, ok ? __float_as_uint(x_f[1]) : x_v.y, ok ? __float_as_uint(x_f[2]) : x_v.z,
                               ok ? __float_as_uint(x_f[3]) : x_v.w);
            }
        };
        // stage w (0 .. WT - 1) of the step's work list
        auto run_stage = [&](auto W_, const Park& pk, const YQ& yq, int dprev, int q_issue, int q_fin) __attribute__((always_inline)) {
            constexpr int w = decltype(W_)::value;
            rl_static_for<NWI>([&](auto I_) {
                constexpr int i = decltype(I_)::value;
                constexpr int base = [] { int b = 0; for (int i2 = 0; i2 < i; ++i2) b += (i2 < NST ? SO : 0) + (i2 < NK ? 1 + SX : 0); return b; }();
                constexpr int so = i < NST ? SO : 0, sd = i < NK ? 1 : 0, sxx = i < NK ? SX : 0;
                if constexpr (w >= base && w < base + so) {
                    if constexpr (!(BIU_ROLL_ABL & 2)) o_stage(I_, std::integral_constant<int, w - base>{}, pk, yq, dprev);
                } else if constexpr (w >= base + so && w < base + so + sd) {
                    if constexpr (!(BIU_ROLL_ABL & 1)) fetch_piece(q_issue, i);
                } else if constexpr (w >= base + so + sd && w < base + so + sd + sxx) {
                    x_stage(I_, std::integral_constant<int, w - base - so - sd>{}, q_fin);
                }
            });
        };
        Acc cur;
        // Vector-memory instructions of a wave per step, in issue order: NRL y loads, then O(0) store, D(0), O(1) store, D(1), ...
        auto step = [&](int s) __attribute__((always_inline)) {
            RL_STAMP(0);
            // -- wait for this wave's pieces of the unit of step s + 1: issued by step s + 1 - LA (behind its last piece: that step's stores
            //    O(NK) .. O(NST - 1), then LA - 2 whole steps) or, in the first LA - 1 steps, by the prologue (behind it: the prologue's later
            //    units and the steps so far)
            constexpr int PER = NRL + NK + NST, TAILST = NST > NK ? NST - NK : 0;
            if (s >= LA - 1) rl_wait_vmcnt<TAILST + (LA - 2) * PER>();
            else if (s == 0) rl_wait_vmcnt<(LA - 2) * NK>();
            else if (s == 1) rl_wait_vmcnt<(LA >= 3 ? (LA - 3) * NK + PER : 0)>();
            else rl_wait_vmcnt<(LA >= 4 ? (LA - 4) * NK + 2 * PER : 0)>();
            RL_STAMP(1);
            const int dcur = d0 + s * PS;
            const int dprev = s > 0 ? dcur - PS : 0x3fffffff;              // (no previous step: every store of the deferred epilogue is masked)
            // -- RED: the upstream block's raw output at this step's output voxels, needed by ITS epilogue, i.e. during the next step: a wait for a
            //    load retires everything older in the vector-memory queue, so these loads must be OLDER than the fetches that are to stay in flight
            if constexpr (RED) {
                yq_prev = yq_cur;
                load_y(dcur, yq_cur.v4, yq_cur.v2);
            }
            if constexpr (ACC) {                        // (yq_cur: this step's initial values, requested a step ago; yq_prev is reused for the next step's)
                yq_prev = yq_cur;
                load_y(dcur + PS, yq_cur.v4, yq_cur.v2);
            }
            // -- accumulators start at the bias
            if constexpr (ACC) {
                // the accumulators start at the stored output (+ bias): a stored piece is channels 16 x2 + 8 hf .. + 7 of the lane's voxel; the
                // epilogue's permlane32 swap is its own inverse and hands back the lane's MFMA rows 8 g + 4 hf + i
#pragma unroll
                for (int pl = 0; pl < PS; ++pl)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int x2 = 0; x2 < 2; ++x2) {
                            const uint4 P = yq_prev.v4[(pl * 2 + r2) * 2 + x2];
                            const auto sx = __builtin_amdgcn_permlane32_swap(P.x, P.z, false, false);
                            const auto sy = __builtin_amdgcn_permlane32_swap(P.y, P.w, false, false);
                            const unsigned u[4] = {sx[0], sy[0], sx[1], sy[1]};
                            const float4 b0 = *(const float4*)(lbias + 16 * x2 + 4 * (lane >> 5)), b1 = *(const float4*)(lbias + 16 * x2 + 8 + 4 * (lane >> 5));
                            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                cur.a32[pl][r2][8 * x2 + 2 * e] = __uint_as_float(u[e] << 16) + bb[2 * e];
                                cur.a32[pl][r2][8 * x2 + 2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u) + bb[2 * e + 1];
                            }
                        }
            } else if constexpr (G32) {
                // (lane holds channels 8 g + 4 hf + i of its voxel: four float4 reads of the bias row)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 b4 = *(const float4*)(lbias + 8 * g4 + 4 * (lane >> 5));
#pragma unroll
                    for (int pl = 0; pl < PS; ++pl)
#pragma unroll
                        for (int r2 = 0; r2 < 2; ++r2) {
                            cur.a32[pl][r2][4 * g4] = b4.x; cur.a32[pl][r2][4 * g4 + 1] = b4.y; cur.a32[pl][r2][4 * g4 + 2] = b4.z; cur.a32[pl][r2][4 * g4 + 3] = b4.w;
                        }
                }
            } else {
                const float4 b4 = *(const float4*)(lbias + 4 * (lane >> 4));
#pragma unroll
                for (int pl = 0; pl < PS; ++pl)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int c = 0; c < 2; ++c) { cur.a16[pl][r2][c][0] = b4.x; cur.a16[pl][r2][c][1] = b4.y; cur.a16[pl][r2][c][2] = b4.z; cur.a16[pl][r2][c][3] = b4.w; }
            }
            // -- MFMA phase: groups (input plane p of the step, kw), one slot per MFMA.  Slot t of the step runs the stages
            //    [t WT / NMT, (t + 1) WT / NMT) of the work list and -- in the first NF slots of a group -- one fragment read for the NEXT group
            const int q_issue = (s + LA) * PS + 1, q_fin = (s + 1) * PS + 1;
            // (slot index t = gbase(g) + j in closed form; every index is a template constant)
            constexpr int MAXW = (WT + NMT - 1) / NMT + 1;
            auto slot = [&](auto G_, auto J_) __attribute__((always_inline)) {
                constexpr int g = decltype(G_)::value, j = decltype(J_)::value;
                constexpr int t = [] {
                    int tt = j;
                    for (int g2 = 0; g2 < g; ++g2) {
                        int nv = 0;
                        for (int pl = 0; pl < PS; ++pl) nv += (g2 / 3 - pl >= 0 && g2 / 3 - pl <= 2) ? 1 : 0;
                        tt += nv * 6 * (G32 ? NKS : 2);
                    }
                    return tt;
                }();
                if constexpr (g + 1 < NGRP && j < NF) load_frag1(frb[(g + 1) & 1], s * PS - 1 + (g + 1) / 3, (g + 1) % 3, j);
                constexpr int lo = (t * WT) / NMT, hi = ((t + 1) * WT) / NMT;
                rl_static_for<MAXW>([&](auto DW_) {
                    constexpr int w = lo + decltype(DW_)::value;
                    if constexpr (w < hi) run_stage(std::integral_constant<int, w>{}, park, yq_prev, dprev, q_issue, q_fin);
                });
            };
            rl_static_for<NGRP>([&](auto G_) {
                constexpr int g = decltype(G_)::value;
                constexpr int p = g / 3, kw = g % 3;                               // input plane: relative q = s PS - 1 + p
                const uint4 (&fr)[NF] = frb[g & 1];
                // (MFMA order inside a group: k-step / half, valid output plane, kh, row; plx = index of pl among the valid ones)
                constexpr int nv = [] { int n = 0; for (int pl = 0; pl < PS; ++pl) n += (p - pl >= 0 && p - pl <= 2) ? 1 : 0; return n; }();
                rl_static_for<(G32 ? NKS : 2)>([&](auto C_) {
                    constexpr int c = decltype(C_)::value;                        // k-step (32x32x16) or half (16x16x32)
                    rl_static_for<PS>([&](auto PL_) {
                        constexpr int pl = decltype(PL_)::value, kd = p - pl;     // depth tap of input plane p for output plane pl
                        if constexpr (kd >= 0 && kd <= 2) {
                            constexpr int plx = [] { int n = 0; for (int p2 = 0; p2 < pl; ++p2) n += (p - p2 >= 0 && p - p2 <= 2) ? 1 : 0; return n; }();
                            rl_static_for<6>([&](auto E_) {
                                constexpr int kh = decltype(E_)::value / 2, r2 = decltype(E_)::value % 2;
                                slot(G_, std::integral_constant<int, ((c * nv + plx) * 3 + kh) * 2 + r2>{});
                                if constexpr (G32) rl_mfma32(cur.a32[pl][r2], wr[(kd * 3 + kh) * 3 + kw][c], fr[c * 4 + r2 + kh]);
                                else rl_mfma16(cur.a16[pl][r2][c], wr[(kd * 3 + kh) * 3 + kw][0], fr[c * 4 + r2 + kh]);
                                __builtin_amdgcn_sched_barrier(0);
                            });
                        }
                    });
                });
            });
#if BIU_ROLL_ASM
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");            // MFMA results -> VALU reads: wait states the compiler cannot count for an asm MFMA
#endif
            pack_pieces(cur, park);                                     // the accumulators are free for the next step; its slots store `park`
            RL_STAMP(2);
            load_frags(frb[0], (s + 1) * PS - 1, 0);    // group 0 of the next step reads a plane that has been visible for a step
            RL_STAMP(3);
            __syncthreads();                            // the transformed unit of step s + 1 is visible; the slots of step s are free
            RL_STAMP(4);
#ifdef BIU_DIAG
            dsum_[7] += 1;
#endif
        };
        // the last step's epilogue in one piece
        auto out_all = [&](const Park& pk, const YQ& yq, int dprev) __attribute__((always_inline)) {
            rl_static_for<NST>([&](auto I_) { rl_static_for<SO>([&](auto S_) { o_stage(I_, S_, pk, yq, dprev); }); });
        };
        for (int s = 0; s < nsteps; ++s) step(s);
        // the last step's epilogue
        {
            const int dlast = d0 + (nsteps - 1) * PS;
            out_all(park, yq_cur, dlast);
        }
        rl_wait_vmcnt<0>();                             // (the padding fetches behind the segment's end, the last stores)
        __syncthreads();
    }

    // ---- one partial row per block: DPP sums over the lanes that share a channel set, the waves' rows through LDS ---------------------
    if constexpr (want_stats) {
        if constexpr (G32) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float* slot = lred + ((size_t)(wave * M + p * 16 + (lane >> 5) * 8)) * 2;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float u = s1[p][e], v = s2[p][e];
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
                    if ((lane & 31) == 16) { slot[2 * e] = u; slot[2 * e + 1] = v; }      // lanes 16 / 48: the half-wave's sum (each slot has one writer)
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = s1[0][e], v = s2[0][e];
                u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x128, 0xf, 0xf, false));
                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
                u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x124, 0xf, 0xf, false));
                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
                u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x122, 0xf, 0xf, false));
                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));
                u += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(u), 0x121, 0xf, 0xf, false));
                v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
                if ((lane & 15) == 0) { float* sl_ = lred + ((size_t)(wave * M + 4 * (lane >> 4) + e)) * 2; sl_[0] = u; sl_[1] = v; }
            }
        }
        __syncthreads();
        if (tid < M) {
            const int co = co0 + tid;
            float l0 = 0.f, l1 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) { l0 += lred[(w2 * M + tid) * 2]; l1 += lred[(w2 * M + tid) * 2 + 1]; }
            if (co < a.Cout) {
                float* dstp = a.bn_partial + ((size_t)blockIdx.x * a.Cout + co) * 2;
                dstp[0] = l0;
                if constexpr (RED) dstp[1] = a.red_invstd[co] * (l1 - a.red_mean[co] * l0);      // sum dz * yhat
                else dstp[1] = l1;
            }
        }
    }
#ifdef BIU_DIAG
    if (a.diag && tid == 0) {
        for (int q_ = 0; q_ < 8; ++q_) atomicAdd(a.diag + q_, dsum_[q_]);
        atomicAdd(a.diag + 8, __builtin_readcyclecounter() - t0c_);
        atomicAdd(a.diag + 9, __builtin_amdgcn_s_memrealtime() - t0r_);
    }
#endif
}

// =====================================================================================================================================
// The UP HALF of a folded decoder level (ConvTranspose(k2, s2) + concat + 3x3x3 conv as one op) on the same machinery:
//   y[2v + p] = bias_eff[border state] + sum_{t in {0,1}^3} W'[p][t] . T(x_low)[v + t - 1 + p]        8 parity classes p x 8 coarse taps t
// as the FIRST writer of y (the skip half then accumulates onto it in k_conv_roll's ACC form and rounds the sum once).
// k_conv_pipe<.., 2, 2, 1, ..> ran this as one block per parity class with the coarse tile re-staged eight times, a weight fragment from LDS
// for every two MFMAs and a read-modify-write epilogue: an item spent 8.8 k cycles for 1 k cycles of MFMA (profiles/r03_mfma_shape_ab.md
// section 5).  Here a block is 4 waves = the 4 (pd, ph) class pairs; a wave keeps the composed weights of ITS two classes (pw = 0, 1) in
// registers (2 x 8 taps x 4 k-steps = 256 registers for 64 -> 32 channels) and the block walks a coarse column: a 4 x 32 window in (H, W),
// one coarse plane per step, the 64-channel coarse planes in an LDS ring filled by LDS-DMA and transformed in place.  Every wave reads the
// same ring: the coarse tile is staged ONCE for all eight classes.  A step is two half-steps (pw = 0, then 1) of 128 MFMAs each; the
// half-step's 8 output pieces are parked in the storage type and stored between the MFMAs of the next half-step.
// The ConvTranspose bias reaches the output through the conv's taps: bias_eff[state][co] = b_conv + sum_k Wb[k] - (the Wb[k] of the taps
// that fall outside the fine tensor in that border state) -- the table biu_foldt_pack already builds (fix, bias) -- is the accumulators'
// initial value, looked up per output row (d, h states) and lane (w state): no border pass over y.
// =====================================================================================================================================
struct FoldArgs {
    const char* x; char* y;            // coarse input (ID, IH, IW), fine output (2 ID, 2 IH, 2 IW)
    const uint4* wpk;                  // k_pack_upconv image: [class][n-tile][k-step][tap][lane]
    const float* bias; const float* fix;          // [Cout], [27][Cout]
    const float* xs; const float* xb; const float* xl;
    int xpitch, ypitch;
    int N, D, H, W;                    // coarse extents
    int Cout;
    int nbh, nbw, nseg, seg;
    unsigned long long* diag;
};
constexpr int FR_TH = 4, FR_TW = 32, FR_HH = FR_TH + 2, FR_HW = FR_TW + 2, FR_PLV = FR_HH * FR_HW;      // 204 halo voxels per coarse plane
constexpr int FR_NBLK = (FR_PLV + 15) / 16;                                                           // 13 blocks of 16 voxels

template <int LA, bool XF>
__global__ __launch_bounds__(256, 1) void k_fold_roll(FoldArgs a) {
    constexpr int CIN = 64, M = 32, NKS = CIN / 16, PCS = CIN / 8;
    constexpr int TH = FR_TH, TW = FR_TW, HW = FR_HW, PLV = FR_PLV;
    constexpr int BLKB = 256 * PCS, PB = FR_NBLK * BLKB, NPI = PB / 1024, KPP = (NPI + 3) / 4, NK = KPP;      // 26 instructions per plane, 7 per wave
    constexpr int NPL = LA + 3;                                  // ring slots: 3 planes being read, LA - 1 in flight, 1 being transformed
    constexpr int NST = 8;                                       // output pieces (stores) of a HALF-step and wave: 4 rows x 2 pieces
    static_assert(LA == 2, "the 64-channel ring holds five planes");

    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    char* ring = (char*)lds;
    char* dump = ring + NPL * PB;
    float* lxf = (float*)(dump + 1024);                         // [3][CIN]
    float* lbe = lxf + 3 * CIN;                                 // [27][M] bias_eff

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pd = wave >> 1, ph = wave & 1;
    const int co0 = (int)blockIdx.y * M;
    if (tid < CIN) {
        lxf[tid] = XF ? a.xs[tid] : 1.f;
        lxf[CIN + tid] = XF ? a.xb[tid] : 0.f;
        lxf[2 * CIN + tid] = XF ? a.xl[tid] : 1.f;
    }
    for (int i = tid; i < 27 * M; i += 256) {
        const int st = i / M, co = co0 + i % M;
        lbe[i] = co < a.Cout ? (a.bias ? a.bias[co] : 0.f) - (a.fix ? a.fix[st * a.Cout + co] : 0.f) : 0.f;
    }
    // weights of this wave's two classes: [pw][tap][k-step]
    uint4 wr[2][8][NKS];
    {
        const size_t slice = (size_t)((a.Cout + 31) / 32) * NKS * 8 * 64;
#pragma unroll
        for (int pw = 0; pw < 2; ++pw)
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    wr[pw][t][ks] = a.wpk[(size_t)((pd << 2) | (ph << 1) | pw) * slice + (((size_t)blockIdx.y * NKS + ks) * 8 + t) * 64 + lane];
    }
    const int l_piece = (wave & 1) * 4 + (lane >> 4);           // instruction wave + 4 kk covers half-block (wave + 4 kk) & 1 = wave & 1
    const unsigned rowB = (unsigned)a.xpitch * 2u;
    const unsigned planeB = (unsigned)(a.H * a.W) * rowB;
    const size_t sampX = (size_t)a.D * planeB;
    const unsigned rowY = (unsigned)a.ypitch * 2u;
    const int OD = 2 * a.D, OH = 2 * a.H, OW = 2 * a.W;
    const size_t sampY = (size_t)OD * OH * OW * rowY;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)ring);
    constexpr unsigned BAD = 0x40000000u, BAD2 = 0x80000000u;

    // fragment addresses: halo rows ph + j (j = 0..4), column offset kwv = pw + tw (0..2); piece 2 ks + hf -> + ks * 512
    unsigned fa[5 * 3];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int kwv = 0; kwv < 3; ++kwv) {
            const int hv = (ph + j) * HW + kwv + (lane & 31);
            fa[j * 3 + kwv] = (unsigned)((hv >> 4) * BLKB + (hv & 15) * 16 + (lane >> 5) * 256);
        }

    const int G = gridDim.x;
    const int ncols = a.N * a.nbh * a.nbw, nitems = ncols * a.nseg;
    auto item_of = [&](int k) -> int {
        if ((G & 7) == 0) return k * G + (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
        return k * G + (int)blockIdx.x;
    };
    int cn = 0, h0 = 0, w0 = 0, d0 = 0, dend = 0;
    v4u_t rsX;
    __amdgpu_buffer_rsrc_t rsY;
    unsigned bbase[NK];
    unsigned vY[2];                                             // per-lane output base of the two classes (pw = 0, 1), BAD outside the volume
    int lsw[2];                                                 // the lane's w border state for pw = 0, 1 (0 first, 1 interior, 2 last)
    auto set_item = [&](int item) __attribute__((always_inline)) {
        int c = item;
        const int sg = c % a.nseg; c /= a.nseg;
        const int wb = c % a.nbw; c /= a.nbw;
        const int hb = c % a.nbh;
        cn = c / a.nbh;
        h0 = hb * TH; w0 = wb * TW;
        d0 = sg * a.seg;
        dend = min(a.D, d0 + a.seg);
        rsX = rl_rsrc(a.x + (size_t)cn * sampX, (unsigned)sampX);
        rsY = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (size_t)cn * sampY), 0, (int)(unsigned)sampY, 0x00020000);
#pragma unroll
        for (int kk = 0; kk < KPP; ++kk) {
            const int ji = wave + 4 * kk;
            const int hv = (ji >> 1) * 16 + (lane & 15);
            const int hh = hv / HW, hw = hv - hh * HW;
            const int gh = h0 - 1 + hh, gw = w0 - 1 + hw;
            const bool ok = ji < NPI && hv < PLV && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
            bbase[kk] = ok ? (unsigned)(gh * a.W + gw) * rowB + (unsigned)l_piece * 16u : BAD;
        }
        const int gw = w0 + (lane & 31);
#pragma unroll
        for (int pw = 0; pw < 2; ++pw) {
            const int ow = 2 * gw + pw;
            vY[pw] = gw < a.W ? (unsigned)ow * rowY + (unsigned)(co0 + 8 * (lane >> 5)) * 2u : BAD;
            lsw[pw] = ow == 0 ? 0 : (ow == OW - 1 ? 2 : 1);
        }
    };
    auto slot_of = [&](int q) -> int { return (q + 1) % NPL; };
    auto piece_voff = [&](int q, int k) __attribute__((always_inline)) -> unsigned {
        const int dg = d0 + q;
        const unsigned dof = rl_opaque((unsigned)dg < (unsigned)a.D ? (unsigned)dg * planeB : BAD2);
        return bbase[k] + dof;
    };
    auto piece_lds = [&](int q, int k) __attribute__((always_inline)) -> unsigned {
        const int ji = wave + 4 * k;
        const unsigned in_ring = (unsigned)(slot_of(q) * PB) + (unsigned)ji * 1024u;
        if (4 * k + 3 < NPI) return rl_opaque(in_ring);
        return rl_opaque(ji < NPI ? in_ring : (unsigned)(NPL * PB));
    };
    auto fetch_piece = [&](int q, int k) __attribute__((always_inline)) { rl_bload_lds16(piece_voff(q, k), rsX, lds0 + piece_lds(q, k)); };
    auto load_xf = [&](float* sc, float* sh, float* sl) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = lxf[l_piece * 8 + e]; sh[e] = lxf[CIN + l_piece * 8 + e]; sl[e] = lxf[2 * CIN + l_piece * 8 + e]; }
    };
    auto finish_piece = [&](int q, int k) __attribute__((always_inline)) {       // (prologue only: the steps run the staged form)
        float f[8], sc[8], sh[8], sl[8];
        load_xf(sc, sh, sl);
        uint4* p_ = (uint4*)(ring + piece_lds(q, k) + lane * 16);
        const uint4 v = *p_;
        rl_unpack8(v, f);
        rl_lrelu_affine8(f, sc, sh, sl);
        const uint4 t = rl_pack8(f);
        const bool ok = piece_voff(q, k) < (unsigned)sampX;
        *p_ = make_uint4(ok ? t.x : v.x, ok ? t.y : v.y, ok ? t.z : v.z, ok ? t.w : v.w);
    };
#ifdef BIU_DIAG
    unsigned long long tprev_ = __builtin_readcyclecounter();
    unsigned long long dsum_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0c_ = tprev_, t0r_ = __builtin_amdgcn_s_memrealtime();
#endif
    // a half-step's MFMA groups: (td, tw, ks) -> 16 groups of 8 MFMAs; fragments of a group: the 5 halo rows at (plane td, kwv = pw + tw, ks)
    constexpr int NGRP = 16, NF = 5, NMT = 128;
    uint4 frb[2][NF];
    auto load_frag1 = [&](uint4 (&fr)[NF], int s, int pw, int g, int j) __attribute__((always_inline)) {
        const int td = g >> 3, tw = (g >> 2) & 1, ks = g & 3;
        const char* pbase = ring + rl_opaque((unsigned)(slot_of(s + pd - 1 + td) * PB));
        fr[j] = *(const uint4*)(pbase + fa[j * 3 + pw + tw] + ks * 512);
    };
    __syncthreads();
    for (int kc = 0;; ++kc) {
        const int item = item_of(kc);
        if (item >= nitems) break;
        set_item(item);
        const int nsteps = dend - d0;
        // prologue: planes -1, 0, 1 landed and transformed, plane 2 in flight (LA = 2)
#pragma unroll
        for (int u = 0; u < LA + 2; ++u)
#pragma unroll
            for (int k = 0; k < NK; ++k) fetch_piece(-1 + u, k);
        rl_wait_vmcnt<(LA - 1) * NK>();
        if constexpr (XF) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int k = 0; k < NK; ++k) finish_piece(-1 + u, k);
        }
        __syncthreads();
        // ---- stage streams of a half-step (see k_conv_roll): O(i) i < 8: store piece i of the PREVIOUS half-step (2 stages); D(k): fetch piece k
        // of plane s + LA + 1; X(k): transform piece k of plane s + 2.  Half 0 carries k = 0..3, half 1 k = 4..6.
        constexpr int SO = 2, SX = XF ? 7 : 0;
        uint4 park[NST];
        uint4 o_piece = make_uint4(0, 0, 0, 0);
        unsigned o_off = 0;
        uint4 x_v = make_uint4(0, 0, 0, 0);
        float x_f[8], x_t[8], x_u[8], x_sc[8], x_sh[8], x_sl[8];
        floatx16 acc[TH];
        auto row_off = [&](int dcv, int r, int hpw) __attribute__((always_inline)) -> unsigned {      // fine row (2 dcv + pd, 2 (h0 + r) + ph)
            const int gh = h0 + r;
            return rl_opaque((dcv < dend && dcv >= d0 && gh < a.H) ? (unsigned)(((2 * dcv + pd) * OH + 2 * gh + ph) * OW) * rowY : BAD2);
        };
        auto o_stage = [&](auto I_, auto ST_, int dcv, int hpw) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, st = decltype(ST_)::value;
            constexpr int r = i / 2, x2 = i % 2;
            if constexpr (st == 0) {
                o_off = vY[hpw] + row_off(dcv, r, hpw) + 32u * x2;
                o_piece = park[i];
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(v4u_t{o_piece.x, o_piece.y, o_piece.z, o_piece.w}, rsY, (int)o_off, 0, 0);
            }
        };
        auto x_stage = [&](int k, auto ST_, int q_fin) __attribute__((always_inline)) {
            constexpr int st = decltype(ST_)::value;
            if constexpr (st == 0) {
                x_v = *(const uint4*)(ring + piece_lds(q_fin, k) + lane * 16);
                load_xf(x_sc, x_sh, x_sl);
            } else if constexpr (st == 1) {
                rl_unpack8(x_v, x_f);
            } else if constexpr (st == 2) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const floatx2 v = {x_f[e], x_f[e + 1]}, sc2 = {x_sc[e], x_sc[e + 1]}, b = {x_sh[e], x_sh[e + 1]};
                    const floatx2 t = __builtin_elementwise_fma(sc2, v, b);
                    x_t[e] = t[0]; x_t[e + 1] = t[1];
                }
            } else if constexpr (st == 3) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const floatx2 t = {x_t[e], x_t[e + 1]}, l = {x_sl[e], x_sl[e + 1]};
                    const floatx2 u = l * t;
                    x_u[e] = u[0]; x_u[e + 1] = u[1];
                }
            } else if constexpr (st == 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x_f[e] = fmaxf(x_t[e], x_u[e]);
            } else if constexpr (st == 5) {
                const uint4 t = rl_pack8(x_f);
                x_f[0] = __uint_as_float(t.x); x_f[1] = __uint_as_float(t.y); x_f[2] = __uint_as_float(t.z); x_f[3] = __uint_as_float(t.w);
            } else {
                const bool ok = piece_voff(q_fin, k) < (unsigned)sampX;
                *(uint4*)(ring + piece_lds(q_fin, k) + lane * 16) =
                    make_uint4(ok ? __float_as_uint(x_f[0]) : x_v.x, ok ? __float_as_uint(x_f[1]) : x_v.y, ok ? __float_as_uint(x_f[2]) : x_v.z,
                               ok ? __float_as_uint(x_f[3]) : x_v.w);
            }
        };
        // work list of half `HP` (compile time): items i = 0..7: O(i) [2 stages], then for i < NKH: D(KB + i) [1], X(KB + i) [SX]
        auto half_step = [&](auto HP_, int s, int dprev, int hprev) __attribute__((always_inline)) {
            constexpr int hp = decltype(HP_)::value;                      // = pw of this half
            constexpr int KB = hp == 0 ? 0 : 4, NKH = hp == 0 ? 4 : NK - 4;
            constexpr int WT = NST * SO + NKH * (1 + SX);
            constexpr int MAXW = (WT + NMT - 1) / NMT + 1;
            const int q_issue = s + LA + 1, q_fin = s + 2;
            // accumulators start at bias_eff[state]: d / h states are wave-uniform per row, the w state is the lane's
            {
                const int od = 2 * (d0 + s) + pd;
                const int sd = od == 0 ? 0 : (od == OD - 1 ? 2 : 1);
#pragma unroll
                for (int r = 0; r < TH; ++r) {
                    const int oh = 2 * (h0 + r) + ph;
                    const int sh = oh == 0 ? 0 : (oh >= OH - 1 ? 2 : 1);
                    const float* row = lbe + ((sd * 3 + sh) * 3 + lsw[hp]) * M + 4 * (lane >> 5);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float4 b4 = *(const float4*)(row + 8 * g4);
                        acc[r][4 * g4] = b4.x; acc[r][4 * g4 + 1] = b4.y; acc[r][4 * g4 + 2] = b4.z; acc[r][4 * g4 + 3] = b4.w;
                    }
                }
            }
            auto run_stage = [&](auto W_) __attribute__((always_inline)) {
                constexpr int w = decltype(W_)::value;
                rl_static_for<NST>([&](auto I_) {
                    constexpr int i = decltype(I_)::value;
                    constexpr int base = [] { int b = 0; for (int i2 = 0; i2 < i; ++i2) b += SO + (i2 < NKH ? 1 + SX : 0); return b; }();
                    constexpr int sd_ = i < NKH ? 1 : 0, sx_ = i < NKH ? SX : 0;
                    if constexpr (w >= base && w < base + SO) {
                        o_stage(I_, std::integral_constant<int, w - base>{}, dprev, hprev);
                    } else if constexpr (w >= base + SO && w < base + SO + sd_) {
                        fetch_piece(q_issue, KB + i);
                    } else if constexpr (w >= base + SO + sd_ && w < base + SO + sd_ + sx_) {
                        x_stage(KB + i, std::integral_constant<int, w - base - SO - sd_>{}, q_fin);
                    }
                });
            };
            rl_static_for<NGRP>([&](auto G_) {
                constexpr int g = decltype(G_)::value;
                constexpr int td = g >> 3, tw = (g >> 2) & 1, ks = g & 3;
                const uint4 (&fr)[NF] = frb[g & 1];
                rl_static_for<8>([&](auto E_) {
                    constexpr int th = decltype(E_)::value / 4, r = decltype(E_)::value % 4, j = decltype(E_)::value;
                    constexpr int t = g * 8 + j;                                       // slot of the half-step
                    if constexpr (g + 1 < NGRP && j < NF) load_frag1(frb[(g + 1) & 1], s, hp, g + 1, j);
                    constexpr int lo = (t * WT) / NMT, hi = ((t + 1) * WT) / NMT;
                    rl_static_for<MAXW>([&](auto DW_) {
                        constexpr int w = lo + decltype(DW_)::value;
                        if constexpr (w < hi) run_stage(std::integral_constant<int, w>{});
                    });
                    rl_mfma32(acc[r], wr[hp][td * 4 + th * 2 + tw][ks], fr[r + th]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            // park the half-step's outputs in the storage type (bf16; the half-waves exchange channel groups so that a lane holds 16 bytes)
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int r = i / 2, x2 = i % 2;
                bf16x4 g0, g1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { g0[e] = (__bf16)acc[r][8 * x2 + e]; g1[e] = (__bf16)acc[r][8 * x2 + 4 + e]; }
                const uint2 ua = __builtin_bit_cast(uint2, g0), ub = __builtin_bit_cast(uint2, g1);
                const auto sx = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
                park[i] = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
        };
        // Vector-memory instructions of a wave per step, in issue order: half 0: O0 D0 O1 D1 O2 D2 O3 D3 O4 .. O7; half 1: O0 D4 O1 D5 O2 D6 O3 .. O7
        for (int s = 0; s < nsteps; ++s) {
            RL_STAMP(0);
            // wait for this wave's pieces of plane s + 2 (issued by step s - 1, or by the prologue): behind its last piece D6 that step's O3 .. O7
            if (s >= 1) rl_wait_vmcnt<5>(); else rl_wait_vmcnt<0>();
            RL_STAMP(1);
#pragma unroll
            for (int j = 0; j < NF; ++j) load_frag1(frb[0], s, 0, 0, j);
            half_step(std::integral_constant<int, 0>{}, s, s > 0 ? d0 + s - 1 : -0x40000000, 1);
#pragma unroll
            for (int j = 0; j < NF; ++j) load_frag1(frb[0], s, 1, 0, j);
            half_step(std::integral_constant<int, 1>{}, s, d0 + s, 0);
            RL_STAMP(2);
            RL_STAMP(3);
            __syncthreads();
            RL_STAMP(4);
#ifdef BIU_DIAG
            dsum_[7] += 1;
#endif
        }
        // the last half-step's stores
        rl_static_for<NST>([&](auto I_) { rl_static_for<SO>([&](auto S_) { o_stage(I_, S_, d0 + nsteps - 1, 1); }); });
        rl_wait_vmcnt<0>();
        __syncthreads();
    }
#ifdef BIU_DIAG
    if (a.diag && tid == 0) {
        for (int q_ = 0; q_ < 8; ++q_) atomicAdd(a.diag + q_, dsum_[q_]);
        atomicAdd(a.diag + 8, __builtin_readcyclecounter() - t0c_);
        atomicAdd(a.diag + 9, __builtin_amdgcn_s_memrealtime() - t0r_);
    }
#endif
}


int roll_num_cus() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// 0: off, 1: the size rule, 2: wherever the kernel serves the shapes (BIU_ROLL=always: the tests)
int roll_mode() {
    static int v = -1;
    if (v < 0) {
        const char* d = getenv("BIU_DISABLE");
        const char* e = getenv("BIU_ROLL");
        v = (d && strstr(d, "croll")) ? 0 : ((e && strstr(e, "always")) ? 2 : 1);
    }
    return v;
}

struct RollPlan { int ok, m, ps, la, nbh, nbw, nseg, seg, grid, cols; size_t lds; };

RollPlan roll_plan(const biu_act* x, const biu_act* y, int dtype, bool has_cat, int accumulate, bool red = false) {
    RollPlan p{};
    if (roll_mode() == 0 || dtype != BIU_BF16 || has_cat) return p;
    if (accumulate && (red || !(x->c == 32 && y->c == 32))) return p;      // the accumulate form: 32 -> 32 only (the skip half of a folded decoder level)
    const int cin = x->c, cout = y->c;
    if (!(cin == 16 || cin == 32)) return p;
    if (cout == 16 && cin == 32) p.m = 16;
    else if (cout == 32) p.m = 32;
    else return p;
    if (x->d < 4 || x->h < 2 || x->w < 2) return p;
    // 32-bit offsets per sample with the padding marks of the kernel (BAD / BAD2): every tensor's sample under 2^30 bytes
    const long long lim = 1LL << 30;
    if ((long long)x->d * x->h * x->w * x->pitch * 2 >= lim || (long long)y->d * y->h * y->w * y->pitch * 2 >= lim) return p;
    if ((uintptr_t)x->p % 16 || (uintptr_t)y->p % 16 || (x->pitch * 2) % 16 || (y->pitch * 2) % 16) return p;
    // pipeline shape per (Cin, MFMA shape): the ring must fit beside the partial-sum rows
    // (a unit is awaited LA - 1 steps after its issue and transformed during the step after that)
    if (cin == 32 && p.m == 32) {                                  // step = 108 MFMAs of 32 cycles; ring 6 x 22 KiB
        if (red) return RollPlan{};                                // (216 weight registers leave no room for the BatchNorm-backward sums: brick kernels)
        p.ps = 1; p.la = 3;
    }
    else if (cin == 32) { p.ps = 1; p.la = 4; }                    // 16x16x32: step = 108 MFMAs of 16 cycles; ring 7 x 22 KiB
    else if (red) { p.ps = 1; p.la = 4; }                          // 16 input channels with the BatchNorm-backward sums: one plane per step (registers)
    else { p.ps = 2; p.la = 3; }                                   // 16 input channels: step = 108 MFMAs of 32 cycles; ring 10 x 11 KiB
    const int pcs = cin / 8, pb = RL_NBLK * 256 * pcs, npl = p.ps * (p.la + 1) + 2;
    p.lds = (size_t)npl * pb + 1024 + (size_t)(3 * cin + 4 * p.m * 2 + 4 * p.m) * sizeof(float);
    if (p.lds > (size_t)160 * 1024) return p;
    p.nbh = (x->h + RL_TH - 1) / RL_TH;
    p.nbw = (x->w + RL_TW - 1) / RL_TW;
    p.cols = cout / p.m;
    const int ncols = x->n * p.nbh * p.nbw;
    const int budget = roll_num_cus() / p.cols > 0 ? roll_num_cus() / p.cols : 1;
    // depth segments: enough items to give every block of a column at least one, segments of at least 8 planes (a segment's prologue
    // re-fetches two halo planes and exposes one memory latency)
    int nseg = 1;
    while (ncols * nseg < budget && (x->d / (nseg * 2)) >= 8) nseg *= 2;
    int seg = (x->d + nseg - 1) / nseg;
    seg = (seg + 1) / 2 * 2;                                        // (a multiple of every variant's step: the grid must not depend on the epilogue form)
    nseg = (x->d + seg - 1) / seg;
    p.nseg = nseg; p.seg = seg;
    const int items = ncols * nseg;
    p.grid = items < budget ? items : budget;
    if ((p.grid & 7) != 0 && p.grid > 8) p.grid &= ~7;             // XCD-grouped walk
    // worth it when the window covers the plane reasonably and the chip is filled (BIU_ROLL=always: wherever it is correct)
    if (roll_mode() != 2) {
        const double cover = (double)x->h * x->w / ((double)p.nbh * RL_TH * p.nbw * RL_TW);
        if (cover < 0.75 || items * p.cols * 2 < roll_num_cus() || x->d < 16) return RollPlan{};
    }
    p.ok = 1;
    return p;
}

// REDF: 0 = only the forms without the BatchNorm-backward sums, 1 = only that form, 2 = all; ACCF: the accumulate forms are built too
template <int CIN, int M, int PS, int LA, int REDF, bool ACCF = false>
int roll_launch(const RollArgs& a, const RollPlan& p, bool red, hipStream_t st, bool acc = false) {
    auto go = [&](auto kern) -> int {
        static size_t attr_set = 0;
        if (attr_set < p.lds) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) != hipSuccess)
                return biu_fail(BIU_ERR_LAUNCH, "conv_roll: cannot reserve %zu bytes of LDS", p.lds);
            attr_set = p.lds;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)p.grid, (unsigned)p.cols), dim3(256), p.lds, st, a);
        BIU_CHECK_LAUNCH("conv_roll");
        return BIU_OK;
    };
    const bool xf = a.xs != nullptr;
    if constexpr (ACCF) {
        if (acc && !red) {
            if (a.bn_partial) return xf ? go(k_conv_roll<CIN, M, PS, LA, 1, true, true>) : go(k_conv_roll<CIN, M, PS, LA, 1, false, true>);
            return xf ? go(k_conv_roll<CIN, M, PS, LA, 0, true, true>) : go(k_conv_roll<CIN, M, PS, LA, 0, false, true>);
        }
    }
    if (acc) return biu_fail(BIU_ERR_UNSUPPORTED, "conv_roll: no accumulate form for %d input channels on the %d-row shape", CIN, M);
    if constexpr (REDF >= 1) {
        if (red) return xf ? go(k_conv_roll<CIN, M, PS, LA, 2, true>) : go(k_conv_roll<CIN, M, PS, LA, 2, false>);
    }
    if constexpr (REDF != 1) {
        if (!red && a.bn_partial) return xf ? go(k_conv_roll<CIN, M, PS, LA, 1, true>) : go(k_conv_roll<CIN, M, PS, LA, 1, false>);
        if (!red) return xf ? go(k_conv_roll<CIN, M, PS, LA, 0, true>) : go(k_conv_roll<CIN, M, PS, LA, 0, false>);
    }
    return biu_fail(BIU_ERR_UNSUPPORTED, "conv_roll: this epilogue form is not built for %d input channels on the %d-row shape", CIN, M);
}

struct FoldPlan { int ok, nbh, nbw, nseg, seg, grid; size_t lds; };
FoldPlan fold_plan(const biu_act* x_low, const biu_act* y, int dtype) {
    FoldPlan p{};
    if (roll_mode() == 0 || dtype != BIU_BF16 || x_low->c != 64 || y->c != 32) return p;
    if (y->n != x_low->n || y->d != 2 * x_low->d || y->h != 2 * x_low->h || y->w != 2 * x_low->w || x_low->d < 2) return p;
    const long long lim = 1LL << 30;
    if ((long long)x_low->d * x_low->h * x_low->w * x_low->pitch * 2 >= lim || (long long)y->d * y->h * y->w * y->pitch * 2 >= lim) return p;
    if ((uintptr_t)x_low->p % 16 || (uintptr_t)y->p % 16 || (x_low->pitch * 2) % 16 || (y->pitch * 2) % 16) return p;
    p.lds = (size_t)5 * FR_NBLK * 2048 + 1024 + (size_t)(3 * 64 + 27 * 32) * sizeof(float);
    p.nbh = (x_low->h + FR_TH - 1) / FR_TH;
    p.nbw = (x_low->w + FR_TW - 1) / FR_TW;
    const int ncols = x_low->n * p.nbh * p.nbw, budget = roll_num_cus();
    int nseg = 1;
    while (ncols * nseg < budget && (x_low->d / (nseg * 2)) >= 8) nseg *= 2;
    int seg = (x_low->d + nseg - 1) / nseg;
    nseg = (x_low->d + seg - 1) / seg;
    p.nseg = nseg; p.seg = seg;
    const int items = ncols * nseg;
    p.grid = items < budget ? items : budget;
    if ((p.grid & 7) != 0 && p.grid > 8) p.grid &= ~7;
    if (roll_mode() != 2) {
        const double cover = (double)x_low->h * x_low->w / ((double)p.nbh * FR_TH * p.nbw * FR_TW);
        if (cover < 0.75 || items * 2 < roll_num_cus() || x_low->d < 16) return FoldPlan{};
    }
    p.ok = 1;
    return p;
}

}  // namespace

// The up half of a folded decoder level as the first writer of y (k_fold_roll): 64 -> 32 channels, bf16.
bool biu_fold_roll_ok(const biu_act* x_low, const biu_act* y, int dtype) { return fold_plan(x_low, y, dtype).ok != 0; }
int biu_fold_roll(const biu_act* x_low, const biu_xform* xf, const void* packed_fwd, const float* bias_sum, const float* fix, const biu_act* y, hipStream_t st) {
    const FoldPlan p = fold_plan(x_low, y, BIU_BF16);
    BIU_REQUIRE(p.ok, BIU_ERR_UNSUPPORTED, "fold_roll: shape not served");
    FoldArgs a;
    a.x = (const char*)x_low->p; a.y = (char*)y->p; a.wpk = (const uint4*)packed_fwd; a.bias = bias_sum; a.fix = fix;
    const bool has = xf && (xf->scale || xf->shift || xf->slope);
    if (has) BIU_REQUIRE(xf->scale && xf->shift && xf->slope, BIU_ERR_UNSUPPORTED, "fold_roll: partial biu_xform (need all three vectors)");
    a.xs = has ? xf->scale : nullptr; a.xb = has ? xf->shift : nullptr; a.xl = has ? xf->slope : nullptr;
    a.xpitch = x_low->pitch; a.ypitch = y->pitch;
    a.N = x_low->n; a.D = x_low->d; a.H = x_low->h; a.W = x_low->w;
    a.Cout = y->c;
    a.nbh = p.nbh; a.nbw = p.nbw; a.nseg = p.nseg; a.seg = p.seg;
#ifdef BIU_DIAG
    a.diag = biu_diag_buffer;
#else
    a.diag = nullptr;
#endif
    auto go = [&](auto kern) -> int {
        static size_t attr_set = 0;
        if (attr_set < p.lds) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) != hipSuccess)
                return biu_fail(BIU_ERR_LAUNCH, "fold_roll: cannot reserve %zu bytes of LDS", p.lds);
            attr_set = p.lds;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)p.grid, 1), dim3(256), p.lds, st, a);
        BIU_CHECK_LAUNCH("fold_roll");
        return BIU_OK;
    };
    return has ? go(k_fold_roll<2, true>) : go(k_fold_roll<2, false>);
}

bool biu_conv_roll_ok(const biu_act* x, const biu_act* y, int dtype, bool has_cat, int accumulate, bool red) { return roll_plan(x, y, dtype, has_cat, accumulate, red).ok != 0; }
int biu_conv_roll_mshape(const biu_act* x, const biu_act* y, int dtype) { return roll_plan(x, y, dtype, false, 0).m; }
// partial rows of its epilogue sums: one per block of a column (the same with and without the BatchNorm-backward sums)
int biu_conv_roll_rows(const biu_act* x, const biu_act* y, int dtype) { return roll_plan(x, y, dtype, false, 0).grid; }

// `packed`: the fragment image of the MFMA shape the plan picked (biu_conv_roll_mshape: 32 = k_pack_weights' image, 16 = k_pack_weights16's)
int biu_conv_roll(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, const biu_act* y, float* bn_partial, const BnRedFuse* red,
                  hipStream_t st, int accumulate) {
    const RollPlan p = roll_plan(x, y, BIU_BF16, false, accumulate, red != nullptr);
    BIU_REQUIRE(p.ok, BIU_ERR_UNSUPPORTED, "conv_roll: shape not served");
    RollArgs a;
    a.x = (const char*)x->p; a.y = (char*)y->p; a.wpk = (const uint4*)packed; a.bias = bias;
    const bool has = xf && (xf->scale || xf->shift || xf->slope);
    if (has) BIU_REQUIRE(xf->scale && xf->shift && xf->slope, BIU_ERR_UNSUPPORTED, "conv_roll: partial biu_xform (need all three vectors)");
    a.xs = has ? xf->scale : nullptr; a.xb = has ? xf->shift : nullptr; a.xl = has ? xf->slope : nullptr;
    a.xpitch = x->pitch; a.ypitch = y->pitch;
    a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w;
    a.Cin = x->c; a.Cout = y->c;
    a.nbh = p.nbh; a.nbw = p.nbw; a.nseg = p.nseg; a.seg = p.seg;
    a.bn_partial = bn_partial;
    a.red_y = nullptr; a.red_ypitch = 0;
    a.red_scale = a.red_shift = a.red_slope = a.red_mean = a.red_invstd = nullptr;
    if (red) {
        a.red_y = (const char*)red->y->p; a.red_ypitch = red->y->pitch;
        a.red_scale = red->scale; a.red_shift = red->shift; a.red_slope = red->slope; a.red_mean = red->mean; a.red_invstd = red->invstd;
    }
#ifdef BIU_DIAG
    a.diag = biu_diag_buffer;
#else
    a.diag = nullptr;
#endif
    const bool r = red != nullptr;
    if (red) {
        const long long rb = (long long)red->y->d * red->y->h * red->y->w * red->y->pitch * 2;
        BIU_REQUIRE(rb < (1LL << 30) && (uintptr_t)red->y->p % 16 == 0 && (red->y->pitch * 2) % 16 == 0, BIU_ERR_UNSUPPORTED,
                    "conv_roll: the upstream output must be 16-byte aligned with samples under 2^30 bytes");
    }
    if (x->c == 32 && p.m == 32) return roll_launch<32, 32, 1, 3, 0, true>(a, p, r, st, accumulate != 0);
    if (x->c == 32 && p.m == 16) return roll_launch<32, 16, 1, 4, 2>(a, p, r, st);
    if (x->c == 16 && p.m == 32) return r ? roll_launch<16, 32, 1, 4, 1>(a, p, r, st) : roll_launch<16, 32, 2, 3, 0>(a, p, r, st);
    return biu_fail(BIU_ERR_UNSUPPORTED, "conv_roll: no instantiation for %d -> %d channels", x->c, y->c);
}
