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

struct ConvArgs {
    const char* x;
    char* y;
    const uint4* wpk;
    const float* bias;
    const float* xs;     // input transform (all three non-null or all null)
    const float* xb;
    const float* xl;
    int xpitch, ypitch;  // elements
    int N, D, H, W, Cin, Cout;
    int nbd, nbh, nbw;
    int nKS;             // total k-steps = Cin / (2 * PE)
    int accumulate;
};

constexpr int cpad_planes(int hv, int ckp) {     // plane stride in 16-B units: == 8/ckp (mod 8) -> conflict-free staging writes
    int want = 8 / ckp;
    int v = hv;
    while (v % 8 != want % 8) ++v;
    return v;
}

template <typename T, int KD, int TD, int TH, int TW, int NT, int CKP>
__global__ __launch_bounds__(256, 2) void k_conv_mfma(ConvArgs a) {
    using F = Frag<T>;
    constexpr int PE = F::PE;
    constexpr int PD = (KD == 3) ? 1 : 0;
    constexpr int HD = TD + 2 * PD, HH = TH + 2, HW = TW + 2;
    constexpr int HV = HD * HH * HW;
    constexpr int PSV = cpad_planes(HV, CKP);
    constexpr int TILES = TD * TH * TW / 32;
    static_assert(TILES % 4 == 0, "brick must give a multiple of 4 voxel tiles");
    constexpr int MT = TILES / 4;
    constexpr int TAPS = KD * 9;
    constexpr int SPC = CKP / 2;                     // k-steps per chunk
    constexpr int NITEMS = HV * CKP;
    constexpr int NPASS = (NITEMS + 255) / 256;
    constexpr int CK = CKP * PE;                     // channels per chunk

    extern __shared__ __attribute__((aligned(16))) uint4 lds[];   // [CKP][PSV]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;

    int b = blockIdx.x;
    const int bw = b % a.nbw; b /= a.nbw;
    const int bh = b % a.nbh; b /= a.nbh;
    const int bd = b % a.nbd;
    const int n = b / a.nbd;
    const int d0 = bd * TD, h0 = bh * TH, w0 = bw * TW;

    // ---- staging plan: item i = tid + 256*j  ->  (halo voxel hv = i / CKP, piece p = i % CKP) --------------------
    const int p_mine = tid % CKP;
    int voxidx[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
        const int i = tid + 256 * j;
        const int hv = i / CKP;
        const int hw = hv % HW;
        const int t = hv / HW;
        const int hh = t % HH;
        const int hd = t / HH;
        const int gd = d0 - PD + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
        const bool inb = (hv < HV) && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        voxidx[j] = inb ? ((n * a.D + gd) * a.H + gh) * a.W + gw : (hv < HV ? -1 : -2);
    }

    // ---- per-lane LDS base of each of this wave's voxel tiles (tap (0,0,0) corner), in 16-B units ----------------
    int hvb[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (wave * MT + mt) * 32 + r;
        const int lw = q % TW;
        const int t = q / TW;
        const int lh = t % TH;
        const int ld = t / TH;
        hvb[mt] = hf * PSV + (ld * HH + lh) * HW + lw;
    }

    floatx16 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nt][mt][e] = 0.f;

    const bool has_xf = a.xs != nullptr;
    const int nchunks = a.Cin / CK;
    const size_t esz = sizeof(T);
    const uint4* wbase = a.wpk + ((size_t)blockIdx.y * NT * a.nKS * TAPS) * 64 + lane;

    for (int ch = 0; ch < nchunks; ++ch) {
        // -------- stage the halo tile of channels [ch*CK, ch*CK + CK) --------------------------------------------
        const int c0 = ch * CK + p_mine * PE;
        float sc[PE], sh[PE], sl[PE];
        if (has_xf) {
#pragma unroll
            for (int e = 0; e < PE; ++e) { sc[e] = a.xs[c0 + e]; sh[e] = a.xb[c0 + e]; sl[e] = a.xl[c0 + e]; }
        }
        const char* xsrc = a.x + (size_t)c0 * esz;
#pragma unroll
        for (int j0 = 0; j0 < NPASS; j0 += 4) {
            uint4 v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j0 + jj;
                if (j < NPASS) {
                    v[jj] = make_uint4(0, 0, 0, 0);
                    if (voxidx[j] >= 0) v[jj] = *(const uint4*)(xsrc + (size_t)voxidx[j] * a.xpitch * esz);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j0 + jj;
                if (j < NPASS) {
                    if (has_xf && voxidx[j] >= 0) {
                        float f[PE];
                        F::unpack(v[jj], f);
#pragma unroll
                        for (int e = 0; e < PE; ++e) {
                            const float t = fmaf(sc[e], f[e], sh[e]);
                            f[e] = fmaxf(t, sl[e] * t);          // LeakyReLU for 0 <= slope <= 1
                        }
                        v[jj] = F::pack(f);
                    }
                    if (voxidx[j] != -2) lds[p_mine * PSV + (tid + 256 * j) / CKP] = v[jj];
                }
            }
        }
        __syncthreads();

        // -------- 27 (9) taps x SPC k-steps of MFMA ----------------------------------------------------------------
        const uint4* wch = wbase + (size_t)(ch * SPC) * TAPS * 64;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            constexpr int dummy = 0; (void)dummy;
            const int ta = tap / 9, tb = (tap / 3) % 3, tc = tap % 3;
            const int tapoff = (ta * HH + tb) * HW + tc;
#pragma unroll
            for (int s = 0; s < SPC; ++s) {
                uint4 wf[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    wf[nt] = wch[((size_t)nt * a.nKS * TAPS + (size_t)s * TAPS + tap) * 64];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const uint4 bf = lds[hvb[mt] + 2 * s * PSV + tapoff];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) F::mma(wf[nt], bf, acc[nt][mt]);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds, per tile, channels {4*hf + 8*q + i} of voxel r ------------------------------------
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (wave * MT + mt) * 32 + r;
        const int lw = q % TW;
        const int t = q / TW;
        const int lh = t % TH;
        const int ld = t / TH;
        const int gd = d0 + ld, gh = h0 + lh, gw = w0 + lw;
        if (gd >= a.D || gh >= a.H || gw >= a.W) continue;
        const size_t vox = ((size_t)(n * a.D + gd) * a.H + gh) * a.W + gw;
        T* yrow = (T*)a.y + vox * a.ypitch;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int co = (blockIdx.y * NT + nt) * 32 + 8 * qq + 4 * hf;
                if (co >= a.Cout) continue;
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = acc[nt][mt][4 * qq + i] + (a.bias ? a.bias[co + i] : 0.f);
                Pack<T, 4>* dst = (Pack<T, 4>*)(yrow + co);
                if (a.accumulate) {
                    Pack<T, 4> old = *dst;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] += (float)old.v[i];
                }
                Pack<T, 4> pk;
#pragma unroll
                for (int i = 0; i < 4; ++i) pk.v[i] = (T)o[i];
                *dst = pk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight packing: out[ntile][kstep][tap][lane] (16 B each); lane (r, h) holds W[i = 32*ntile + r][k = KS*kstep + PE*h + e]
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_pack_weights(const float* __restrict__ w, int cin, int cout, int taps, int kind, int Kc, int Nc,
                               int nKS, int ntiles, uint4* __restrict__ out) {
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

static bool chan_ok(int K, int Nn, int dtype) { return K >= 16 && K % ks_of(dtype) == 0 && Nn >= 16 && Nn % 8 == 0; }

size_t biu_mfma_packed_bytes(int kind, int cin, int cout, int kd, int kh, int kw, int dilation, int dtype) {
    if (dilation != 1 || kh != 3 || kw != 3 || (kd != 1 && kd != 3)) return 0;
    if (dtype != BIU_BF16 && dtype != BIU_F32) return 0;
    const int K = kind == 0 ? cin : cout, Nn = kind == 0 ? cout : cin;
    if (!chan_ok(K, Nn, dtype)) return 0;
    const size_t ntiles = (Nn + 31) / 32, nKS = K / ks_of(dtype);
    return ntiles * nKS * (size_t)(kd * 9) * 1024;
}

int biu_mfma_pack(int kind, const float* w, int cin, int cout, int kd, int kh, int kw, int dtype, void* packed, hipStream_t st) {
    const int K = kind == 0 ? cin : cout, Nn = kind == 0 ? cout : cin;
    const int taps = kd * kh * kw, ntiles = (Nn + 31) / 32, nKS = K / ks_of(dtype);
    const size_t total = (size_t)ntiles * nKS * taps * 64;
    BIU_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_pack_weights<T>, dim3(grid_for((i64)total, 256, 4096)), dim3(256), 0, st, w, cin,
                                                 cout, taps, kind, K, Nn, nKS, ntiles, (uint4*)packed));
    BIU_CHECK_LAUNCH("pack_weights");
    return BIU_OK;
}

bool biu_mfma_conv_ok(const biu_act* x, const biu_act* y, int kd, int kh, int kw, int dilation, int dtype) {
    if (dilation != 1 || kh != 3 || kw != 3 || (kd != 1 && kd != 3)) return false;
    if (!chan_ok(x->c, y->c, dtype)) return false;
    const size_t es = dsize(dtype);
    if ((uintptr_t)x->p % 16 || (uintptr_t)y->p % 16 || ((size_t)x->pitch * es) % 16 || ((size_t)y->pitch * es) % 16) return false;
    if (nvox(x) * (i64)x->pitch >= (1LL << 31) || nvox(y) * (i64)y->pitch >= (1LL << 31)) return false;   // 32-bit voxel index math
    if (kd == 1 && x->d != 1) return false;
    return true;
}

template <typename T, int KD, int TD, int TH, int TW, int NT, int CKP>
static int launch_cfg(const ConvArgs& a0, int ntiles, hipStream_t st) {
    ConvArgs a = a0;
    constexpr int PD = (KD == 3) ? 1 : 0;
    constexpr int HV = (TD + 2 * PD) * (TH + 2) * (TW + 2);
    constexpr int PSV = cpad_planes(HV, CKP);
    const size_t lds_bytes = (size_t)CKP * PSV * 16;
    a.nbd = (a.D + TD - 1) / TD;
    a.nbh = (a.H + TH - 1) / TH;
    a.nbw = (a.W + TW - 1) / TW;
    dim3 grid((unsigned)((size_t)a.N * a.nbd * a.nbh * a.nbw), (unsigned)(ntiles / NT));
    auto kern = k_conv_mfma<T, KD, TD, TH, TW, NT, CKP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, a);
    BIU_CHECK_LAUNCH("conv_mfma");
    return BIU_OK;
}

template <typename T>
static int launch_conv(const ConvArgs& a, int kd, hipStream_t st) {
    const int ntiles = (a.Cout + 31) / 32;
    const int nt = (ntiles % 4 == 0) ? 4 : (ntiles % 2 == 0 ? 2 : 1);
    const bool wide = (a.W % 32 == 0);
    if (kd == 3) {
        if (nt == 1) return wide ? launch_cfg<T, 3, 4, 8, 32, 1, 2>(a, ntiles, st) : launch_cfg<T, 3, 4, 16, 16, 1, 2>(a, ntiles, st);
        if (nt == 2) return launch_cfg<T, 3, 4, 8, 16, 2, 2>(a, ntiles, st);
        return launch_cfg<T, 3, 4, 4, 16, 4, 2>(a, ntiles, st);
    }
    if (nt == 1) return wide ? launch_cfg<T, 1, 1, 32, 32, 1, 2>(a, ntiles, st) : launch_cfg<T, 1, 1, 64, 16, 1, 2>(a, ntiles, st);
    if (nt == 2) return wide ? launch_cfg<T, 1, 1, 16, 32, 2, 2>(a, ntiles, st) : launch_cfg<T, 1, 1, 32, 16, 2, 2>(a, ntiles, st);
    return launch_cfg<T, 1, 1, 16, 16, 4, 2>(a, ntiles, st);
}

int biu_mfma_conv(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, int kd, int kh, int kw,
                  const biu_act* y, int accumulate, int dtype, hipStream_t st) {
    ConvArgs a;
    a.x = (const char*)x->p;
    a.y = (char*)y->p;
    a.wpk = (const uint4*)packed;
    a.bias = bias;
    const bool has = xf && (xf->scale || xf->shift || xf->slope);
    if (has) BIU_REQUIRE(xf->scale && xf->shift && xf->slope, BIU_ERR_UNSUPPORTED, "conv_mfma: partial biu_xform (need all three vectors)");
    a.xs = has ? xf->scale : nullptr;
    a.xb = has ? xf->shift : nullptr;
    a.xl = has ? xf->slope : nullptr;
    a.xpitch = x->pitch;
    a.ypitch = y->pitch;
    a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w;
    a.Cin = x->c; a.Cout = y->c;
    a.nKS = x->c / ks_of(dtype);
    a.accumulate = accumulate;
    a.nbd = a.nbh = a.nbw = 0;
    if (dtype == BIU_BF16) return launch_conv<bf16_t>(a, kd, st);
    return launch_conv<float>(a, kd, st);
}

// weight gradient: not covered yet -> direct kernels
size_t biu_mfma_wgrad_workspace(int, int, int, int, int, int) { return 0; }
bool biu_mfma_wgrad_ok(const biu_act*, const biu_act*, int, int, int, int, int) { return false; }
int biu_mfma_wgrad(const biu_act*, const biu_xform*, const biu_act*, int, int, int, float*, float*, void*, size_t, int, hipStream_t) {
    return biu_fail(BIU_ERR_UNSUPPORTED, "mfma wgrad: not built");
}
