// How many operand bytes per cycle can ONE CU pull from L2, as a function of how many waves issue and of the path?
// (round 6: the GEMM K loops run at ~25 B/clk/CU of staged operands -- a per-wave issue cost, or a CU limit?)
// hipcc --offload-arch=gfx950 -O2 tools/probes/lds_dma_rate.hip -o tools/probes/lds_dma_rate && tools/probes/lds_dma_rate
// 256 workgroups (one per CU, 64 KB of LDS each so that no second one fits), NW waves per workgroup; every wave streams
// its own 1-KiB pieces of a small (L2 / MALL resident) buffer REP times:
//   mode 0: global_load_lds_dwordx4 (saddr form), 8 pieces in flight per wave, into LDS
//   mode 1: global_load_dwordx4 into registers, 8 in flight per wave (data discarded)
//   mode 2: mode 1 + ds_write_b128 of the registers (register-staged)
// Prints bytes per shader cycle per CU (s_memtime around the loop of workgroup 0) and the aggregate rate from the wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP 256

template <int MODE>
__global__ __launch_bounds__(1024) void probe(const unsigned char* src, size_t span, long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + wave * 8192;
    // each workgroup walks its own window of the buffer (stride so that neighbouring CUs touch different lines)
    // every CU reads the same `span` bytes (like a GEMM's operand panels: L2 hits once warm), each wave walking its own 256 KB window
    const size_t off0 = ((size_t)blockIdx.x * 24576 + (size_t)wave * 262144) % span;
    const uint32_t voff = lane * 16;
    uint4 acc = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
        const unsigned char* bv = src + (off0 + (size_t)(r & 31) * 8192) % span;
        const uint64_t bu = (uint64_t)(uintptr_t)bv;
        const unsigned char* b = (const unsigned char*)(uintptr_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(bu >> 32)) << 32) |
                                                                   __builtin_amdgcn_readfirstlane((uint32_t)bu));
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(voff), "s"(b + q * 1024), "s"(lds_base + q * 1024) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            uint4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const uint4*>(b + q * 1024 + voff);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (MODE == 2) *reinterpret_cast<uint4*>(lds + wave * 8192 + q * 1024 + voff) = v[q];
                else { acc.x ^= v[q].x; acc.y ^= v[q].y; acc.z ^= v[q].z; acc.w ^= v[q].w; }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (MODE != 0 && acc.x == 0x12345678u) sink[threadIdx.x] = (float)acc.y;
    if (MODE == 2 && lds[threadIdx.x] == 77) sink[threadIdx.x] = 1.f;
    (void)nw;
}

template <int MODE>
static void run(const char* name, const unsigned char* src, size_t span, long long* cyc, float* sink) {
    for (int nw : {1, 2, 4, 8, 16}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(nw * 64), 160 * 1024 - 1024, 0, src, span, cyc, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(nw * 64), 160 * 1024 - 1024, 0, src, span, cyc, sink);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(256);
        hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += v;
        avg /= 256;
        const double bytes_cu = (double)nw * REP * 8192;
        printf("%-44s %2d waves: %6.1f B/clk/CU in the loop (%.0f cycles), %6.2f TB/s aggregate incl. launch (%.1f us)\n", name, nw,
               bytes_cu / avg, avg, bytes_cu * 256 / (ms * 1e-3) / 1e12, ms * 1e3);
    }
}

int main() {
    const size_t span = 2u << 20;       // 2 MB shared by all CUs: resident in every XCD's 4 MB L2 after the warm-up launch
    unsigned char* src; long long* cyc; float* sink;
    hipMalloc(&src, span + (1 << 20)); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 4096);
    hipMemset(src, 1, span + (1 << 20));
    hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    run<0>("LDS-DMA (global_load_lds_dwordx4, L2 hits)", src, span, cyc, sink);
    run<1>("global_load_dwordx4 -> registers", src, span, cyc, sink);
    run<2>("global_load_dwordx4 -> registers -> ds_write", src, span, cyc, sink);
    return 0;
}
