// Weight-space products of a folded decoder level (ConvTranspose(k2, s2) + concat + 3x3x3 conv as one op) on the fp32 matrix pipe.
//
//   compose      W'[p][t][co][ci] = sum_{k in class(p, t)} sum_c W_conv[co][c][k] W_T[ci][c][q(p, k)]          (forward: biu_foldt_pack)
//   chain, conv  dW_conv[co][c][k] = sum_p sum_ci G[p][t_p(k)][co][ci] W_T[ci][c][q(p, k)] + b_T[c] S_k[co]     (backward)
//   chain, ConvT dW_T[ci][c][q]    = sum_{(p, k): q(p, k) = q} sum_co G[p][t_p(k)][co][ci] W_conv[co][c][k]
//
// 3 x 216 GEMMs of Cout x Cup x Cin_low per step, whatever the volume.  Round 3 ran them as 32 x 32 tiles of scalar FMAs straight off the
// PyTorch layouts (27- and 8-float strides: one 64-byte line per 4-byte load), 8 TFLOP/s -- 0.44 + 0.43 ms at the 256 -> 256 | 128 -> 128
// level of UNet3D(32), which is why that level stayed unfolded.  Here: (1) one gather pass puts the two weight tensors into GEMM layouts
// (W_conv -> [k][co][c], W_T -> [q][ci][c]: every operand of the three products then has a unit-stride axis), (2) 64 x 64 tiles per block on
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation: same arithmetic as the scalar kernels up to the order of the sums),
// operands staged through LDS in K-chunks of 32 with the next chunk's 16-byte loads in flight.
#include <cstdlib>
#include <cstring>

#include "biu_common.h"
#include "biu_internal.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void fg_tq(int p, int k, int& t, int& q) {     // coarse tap and sub-position of fine tap k under parity class p
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

// wct[k][co][c] = W_conv[co][c][k] (c < cup: the up channels), wtq[q][ci][c] = W_T[ci][c][q]
__global__ void k_fold_layouts(const float* __restrict__ wc, int ccat, int cup, int cout, const float* __restrict__ wt, int cin_low,
                               float* __restrict__ wct, float* __restrict__ wtq) {
    const long n1 = (long)27 * cout * cup, n2 = (long)8 * cin_low * cup;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += (long)gridDim.x * blockDim.x) {
        if (i < n1) {
            const int c = (int)(i % cup);
            const long r = i / cup;
            const int co = (int)(r % cout), k = (int)(r / cout);
            wct[i] = wc[((long)co * ccat + c) * 27 + k];
        } else {
            const long j = i - n1;
            const int c = (int)(j % cup);
            const long r = j / cup;
            const int ci = (int)(r % cin_low), q = (int)(r / cin_low);
            wtq[j] = wt[((long)ci * cup + c) * 8 + q];
        }
    }
}

// Wb[k][co] = sum_c W_conv[co][c][k] b_T[c] off the [k][co][c] layout: one wave per (k, co), unit-stride reads
__global__ __launch_bounds__(256) void k_fold_wb(const float* __restrict__ wct, int cup, int cout, const float* __restrict__ bt, float* __restrict__ wb) {
    const int o = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= 27 * cout) return;                          // (wave-uniform)
    float s = 0.f;
    if (bt)
        for (int c = lane; c < cup; c += 64) s = fmaf(wct[(size_t)o * cup + c], bt[c], s);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) wb[o] = s;
}

constexpr int FG_T = 64, FG_KC = 32, FG_LD = FG_KC + 1;

// One 64 x 64 tile of  C = sum_term A_term B_term  per block of 256 threads (wave (wm, wn) owns a 32 x 32 quarter: 16 accumulator registers).
// A_KC: A_term[m][k] at pa + offA[term] + m * lda + k (unit stride along k), else at ... + k * lda + m (unit stride along m);
// B_KC: B_term[k][n] at pb + offB[term] + n * ldb + k, else ... + k * ldb + n.  M, N, K are multiples of 4 (channel counts are multiples of 8)
// and every offset / leading dimension is a multiple of 4 floats: all global reads are 16-byte loads along the unit-stride axis.
// Both operands land in LDS as [row or column][k], 33 floats apart: the lanes of an MFMA operand read (32 rows x 2 k) hit 64 distinct banks.
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void fg_tile(int nterms, int M, int N, int K, int m0, int n0, const float* __restrict__ pa, long lda, const long* offA,
                                        const float* __restrict__ pb, long ldb, const long* offB, floatx16& acc) {
    __shared__ float As[FG_T * FG_LD];
    __shared__ float Bs[FG_T * FG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int nchunks = (K + FG_KC - 1) / FG_KC, nsteps = nterms * nchunks;
    float4 ra[2][2], rb[2][2];                           // two steps of loads in flight, two 16-byte pieces per operand and thread
    auto fetch = [&](int step, float4 (&a)[2], float4 (&b)[2]) {
        const int term = step / nchunks, k0 = (step % nchunks) * FG_KC;
        const float* qa = pa + offA[term];
        const float* qb = pb + offB[term];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 256;                 // 512 pieces of 4 floats = 64 x 32
            {
                const int r = A_KC ? (i >> 3) : ((i & 15) * 4), c = A_KC ? ((i & 7) * 4) : (i >> 4);      // r: tile row m, c: k inside the chunk
                const bool ok = m0 + r < M && k0 + c < K;
                a[j] = ok ? *(const float4*)(qa + (A_KC ? (long)(m0 + r) * lda + (k0 + c) : (long)(k0 + c) * lda + (m0 + r))) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            {
                const int r = B_KC ? (i >> 3) : ((i & 15) * 4), c = B_KC ? ((i & 7) * 4) : (i >> 4);      // r: tile column n
                const bool ok = n0 + r < N && k0 + c < K;
                b[j] = ok ? *(const float4*)(qb + (B_KC ? (long)(n0 + r) * ldb + (k0 + c) : (long)(k0 + c) * ldb + (n0 + r))) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto stage = [&](const float4 (&a)[2], const float4 (&b)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 256;
            const float av[4] = {a[j].x, a[j].y, a[j].z, a[j].w}, bv[4] = {b[j].x, b[j].y, b[j].z, b[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (A_KC) As[(i >> 3) * FG_LD + (i & 7) * 4 + e] = av[e];
                else As[((i & 15) * 4 + e) * FG_LD + (i >> 4)] = av[e];
                if (B_KC) Bs[(i >> 3) * FG_LD + (i & 7) * 4 + e] = bv[e];
                else Bs[((i & 15) * 4 + e) * FG_LD + (i >> 4)] = bv[e];
            }
        }
    };
    const float* ar = As + (wm * 32 + (lane & 31)) * FG_LD + (lane >> 5);
    const float* br = Bs + (wn * 32 + (lane & 31)) * FG_LD + (lane >> 5);
    auto one = [&](int step, float4 (&a)[2], float4 (&b)[2]) {
        stage(a, b);
        __syncthreads();
        if (step + 2 < nsteps) fetch(step + 2, a, b);
#pragma unroll
        for (int kk = 0; kk < FG_KC; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[kk], br[kk], acc, 0, 0, 0);
        __syncthreads();
    };
    if (nsteps > 0) fetch(0, ra[0], rb[0]);
    if (nsteps > 1) fetch(1, ra[1], rb[1]);
    for (int step = 0; step < nsteps; step += 2) {
        one(step, ra[0], rb[0]);
        if (step + 1 < nsteps) one(step + 1, ra[1], rb[1]);
    }
}
// element e of a wave's accumulator: row (e & 3) + 8 (e >> 2) + 4 (lane >> 5), column lane & 31 of its 32 x 32 quarter
__device__ __forceinline__ int fg_row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

// grid (Cin_low / 64, Cout / 64, 64 = (p, t)): rows co, columns ci, reduction c
__global__ __launch_bounds__(256) void k_fold_compose(const float* __restrict__ wct, const float* __restrict__ wtq, int cin_low, int cup, int cout,
                                                      float* __restrict__ wfold) {
    __shared__ long offA[8], offB[8];
    __shared__ int nterms_s;
    const int p = (int)blockIdx.z >> 3, t = (int)blockIdx.z & 7;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int k = 0; k < 27; ++k) {
            int tt, q;
            fg_tq(p, k, tt, q);
            if (tt == t) { offA[n] = (long)k * cout * cup; offB[n] = (long)q * cin_low * cup; ++n; }
        }
        nterms_s = n;
    }
    __syncthreads();
    const int m0 = (int)blockIdx.y * FG_T, n0 = (int)blockIdx.x * FG_T;
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    fg_tile<true, true>(nterms_s, cout, cin_low, cup, m0, n0, wct, cup, offA, wtq, cup, offB, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ci = n0 + (wave & 1) * 32 + (lane & 31);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int co = m0 + (wave >> 1) * 32 + fg_row(e, lane);
        if (co < cout && ci < cin_low) wfold[(((size_t)p * cout + co) * cin_low + ci) * 8 + t] = acc[e];
    }
}
// grid (Cup / 64, Cout / 64, 27 = k): rows co, columns c, reduction ci over the eight parity classes
__global__ __launch_bounds__(256) void k_fold_chain_wconv(const float* __restrict__ G, long slice_f, const float* __restrict__ wtq, int cin_low, int cup, int cout,
                                                          int ccat, float* __restrict__ dwc, const float* __restrict__ bt, const float* __restrict__ Sk) {
    __shared__ long offA[8], offB[8];
    const int k = (int)blockIdx.z;
    if (threadIdx.x < 8) {
        int t, q;
        fg_tq((int)threadIdx.x, k, t, q);
        offA[threadIdx.x] = (long)threadIdx.x * slice_f + (long)t * cout * cin_low;
        offB[threadIdx.x] = (long)q * cin_low * cup;
    }
    __syncthreads();
    const int m0 = (int)blockIdx.y * FG_T, n0 = (int)blockIdx.x * FG_T;
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    fg_tile<true, false>(8, cout, cup, cin_low, m0, n0, G, cin_low, offA, wtq, cup, offB, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = n0 + (wave & 1) * 32 + (lane & 31);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int co = m0 + (wave >> 1) * 32 + fg_row(e, lane);
        if (co < cout && c < cup) dwc[((size_t)co * ccat + c) * 27 + k] = bt ? fmaf(bt[c], Sk[k * cout + co], acc[e]) : acc[e];
    }
}
// grid (Cup / 64, Cin_low / 64, 8 = q): rows ci, columns c, reduction co over the 27 fine taps (for every tap exactly one class has sub-position q)
__global__ __launch_bounds__(256) void k_fold_chain_wt(const float* __restrict__ G, long slice_f, const float* __restrict__ wct, int cin_low, int cup, int cout,
                                                       float* __restrict__ dwt) {
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
        offA[k] = (long)p * slice_f + (long)t * cout * cin_low;
        offB[k] = (long)k * cout * cup;
    }
    __syncthreads();
    const int m0 = (int)blockIdx.y * FG_T, n0 = (int)blockIdx.x * FG_T;
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    fg_tile<false, false>(27, cin_low, cup, cout, m0, n0, G, cin_low, offA, wct, cup, offB, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = n0 + (wave & 1) * 32 + (lane & 31);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int ci = m0 + (wave >> 1) * 32 + fg_row(e, lane);
        if (ci < cin_low && c < cup) dwt[((size_t)ci * cup + c) * 8 + q] = acc[e];
    }
}

}  // namespace

// floats of scratch the two GEMM layouts take (W_conv's up channels as [k][co][c], W_T as [q][ci][c])
size_t biu_fold_gemm_layout_floats(int cin_low, int cup, int cout) { return (size_t)27 * cout * cup + (size_t)8 * cin_low * cup; }
bool biu_fold_gemm_ok(int cin_low, int cup, int cout) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("BIU_DISABLE"); off = (e && strstr(e, "foldgemm")) ? 1 : 0; }
    return !off && cin_low % 4 == 0 && cup % 4 == 0 && cout % 4 == 0;
}
int biu_fold_gemm_layouts(const float* w_conv, int ccat, int cup, int cout, const float* w_t, int cin_low, float* layouts, hipStream_t st) {
    const long n = (long)biu_fold_gemm_layout_floats(cin_low, cup, cout);
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fold_layouts, dim3((unsigned)blocks), dim3(256), 0, st, w_conv, ccat, cup, cout, w_t, cin_low, layouts, layouts + (size_t)27 * cout * cup);
    BIU_CHECK_LAUNCH("fold_layouts");
    return BIU_OK;
}
int biu_fold_gemm_compose(const float* layouts, int cin_low, int cup, int cout, const float* b_t, float* wfold, float* wb, hipStream_t st) {
    const float* wct = layouts;
    const float* wtq = layouts + (size_t)27 * cout * cup;
    hipLaunchKernelGGL(k_fold_wb, dim3((27 * cout + 3) / 4), dim3(256), 0, st, wct, cup, cout, b_t, wb);
    hipLaunchKernelGGL(k_fold_compose, dim3((cin_low + FG_T - 1) / FG_T, (cout + FG_T - 1) / FG_T, 64), dim3(256), 0, st, wct, wtq, cin_low, cup, cout, wfold);
    BIU_CHECK_LAUNCH("fold_compose");
    return BIU_OK;
}
int biu_fold_gemm_chain(const float* G, size_t slice_f, const float* layouts, int cin_low, int cup, int cout, int ccat, float* dw_conv, float* dw_t,
                        const float* b_t, const float* Sk, hipStream_t st) {
    const float* wct = layouts;
    const float* wtq = layouts + (size_t)27 * cout * cup;
    hipLaunchKernelGGL(k_fold_chain_wconv, dim3((cup + FG_T - 1) / FG_T, (cout + FG_T - 1) / FG_T, 27), dim3(256), 0, st, G, (long)slice_f, wtq, cin_low, cup, cout,
                       ccat, dw_conv, b_t, Sk);
    hipLaunchKernelGGL(k_fold_chain_wt, dim3((cup + FG_T - 1) / FG_T, (cin_low + FG_T - 1) / FG_T, 8), dim3(256), 0, st, G, (long)slice_f, wct, cin_low, cup, cout,
                       dw_t);
    BIU_CHECK_LAUNCH("fold_chain");
    return BIU_OK;
}
