// Probe (gfx950): sustained bf16 MFMA rate on RANDOM operands, v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 at the same
// 64 x 64 output tile per wave (64 accumulator registers), operands in registers (rotated over 8 random fragments so that the datapath
// toggles), every CU busy, 1 or 2 waves per SIMD.  The chip is power-limited on such loops: the question is which shape delivers more
// FLOP/s at the cap (MI355X_MICROARCH.md 'DVFS give-back' item 7).   hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k_mfma(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    bf16x8 fa[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + i) & 0xfffff]);
        fb[i] = __builtin_bit_cast(bf16x8, src[(tid * 16 + 8 + i) & 0xfffff]);
    }
    float res = 0.f;
    if constexpr (SHAPE == 32) {
        floatx16 acc[2][2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {            // 4 k-steps of 16: 16 MFMAs x 32 cycles
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2 * u + i], fb[2 * u + j], acc[i][j], 0, 0, 0);
            }
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) res += acc[i][j][e];
    } else {
        floatx4 acc[4][4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {            // 2 k-steps of 32: 32 MFMAs x 16 cycles
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[4 * u + i], fb[4 * u + j], acc[i][j], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) res += acc[i][j][e];
    }
    out[tid] = res;
}

static double run(int shape, int threads, int iters, const uint4* src, float* out, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        if (shape == 32) hipLaunchKernelGGL(k_mfma<32>, dim3(256), dim3(threads), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(k_mfma<16>, dim3(256), dim3(threads), 0, 0, src, out, iters);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)reps * 256.0 * (threads / 64) * iters * 16.0 * 32768.0;      // per iteration and wave: 16 x 32768 FLOP in both shapes
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t n = 1 << 20;
    std::vector<uint32_t> h(n * 4);
    uint4* d; float* o;
    hipMalloc(&d, n * 16);
    hipMalloc(&o, 256 * 512 * 4);
    for (int data = 0; data < 2; ++data) {
        srand(1);
        for (auto& v : h) {
            if (data == 1) { v = 0; continue; }
            // two random bf16 in [-2, 2): sign, exponent 127 or 126.., random mantissa
            uint32_t a = ((rand() & 1) << 15) | ((126 + (rand() & 1)) << 7) | (rand() & 127);
            uint32_t b = ((rand() & 1) << 15) | ((126 + (rand() & 1)) << 7) | (rand() & 127);
            v = a | (b << 16);
        }
        hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
        for (int threads : {256, 512}) {
            for (int round = 0; round < 3; ++round) {
                const double t32 = run(32, threads, 20000, d, o, 20);
                const double t16 = run(16, threads, 20000, d, o, 20);
                printf("%s data, %d waves/SIMD, round %d: 32x32x16 %.0f TFLOP/s   16x16x32 %.0f TFLOP/s   ratio %.3f\n", data ? "zero" : "random", threads / 256, round,
                       t32, t16, t16 / t32);
            }
        }
    }
    return 0;
}
