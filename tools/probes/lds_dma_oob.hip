// Probe (gfx950): what does `buffer_load_dwordx4 ... lds` write for (a) lanes whose offset is beyond the descriptor's range,
// (b) lanes switched off in EXEC?   Build: hipcc --offload-arch=gfx950 -O2 tools/probes/lds_dma_oob.hip -o tools/probes/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bload_lds16(unsigned voff, v4u rsrc, unsigned lds_base_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 :
                 : "v"(voff), "s"(rsrc), "s"(lds_base_uniform)
                 : "memory");
}

__global__ void k_probe(const uint32_t* src, int nbytes, uint32_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[64 * 4 * 2];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 4 * 2; i += 64) buf[i] = 0xDEADBEEFu;
    __syncthreads();
    const uint64_t pa = (uint64_t)src;          // raw V# : base[47:0], stride 0, num_records = bytes, flags as make_buffer_rsrc(.., 0x00020000)
    v4u r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)pa);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane((unsigned)nbytes);
    r[3] = 0x00020000u;
    // lane l reads 16 bytes at 16*l, except: mode 1 -> odd lanes read out of range (offset -1 = 0xFFFFFFFF); mode 2 -> odd lanes are
    // masked by EXEC; mode 3 -> lanes >= 32 read beyond nbytes (nbytes = 512)
    unsigned off = 16u * lane;
    if (mode == 1 && (lane & 1)) off = 0xFFFFFFFFu;
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)buf));
    if (mode == 2) {
        if (!(lane & 1)) bload_lds16(off, r, lbase);
    } else {
        bload_lds16(off, r, lbase);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 64 * 4 * 2; i += 64) out[i] = buf[i];
}

int main() {
    std::vector<uint32_t> h(64 * 4 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000u + (uint32_t)i;
    uint32_t *d, *o;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&o, 64 * 4 * 2 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        const int nbytes = (mode == 3) ? 512 : 1024;
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, nbytes, o, mode);
        std::vector<uint32_t> r(64 * 4 * 2);
        hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
        int ok_data = 0, zeros = 0, untouched = 0, other = 0;
        for (int l = 0; l < 64; ++l) {
            const uint32_t v = r[l * 4];
            if (v == 0x1000u + l * 4) ++ok_data; else if (v == 0) ++zeros; else if (v == 0xDEADBEEFu) ++untouched; else ++other;
        }
        int tail_untouched = 0;
        for (int i = 256; i < 512; ++i) tail_untouched += (r[i] == 0xDEADBEEFu);
        printf("mode %d: lanes with their data %d, zeros %d, untouched %d, other %d; second KiB untouched dwords %d/256; lane1 dwords: %08x %08x %08x %08x; lane33: %08x\n",
               mode, ok_data, zeros, untouched, other, tail_untouched, r[4], r[5], r[6], r[7], r[33 * 4]);
    }
    return 0;
}
