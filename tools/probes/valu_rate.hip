// What does one wave64 VALU instruction cost on gfx950, alone on its SIMD and with a second wave beside it?  (round 6: the
// attention kernel's softmax is VALU-bound -- which of its instructions are the expensive ones?)
// hipcc --offload-arch=gfx950 -O2 tools/probes/valu_rate.hip -o tools/probes/valu_rate && tools/probes/valu_rate
// Each kernel runs REP x 32 instructions of one kind on 8 independent register chains between two s_memtime reads;
// 256 threads = one wave per SIMD, 512 = two, 1024 = four.  Prints shader cycles per instruction PER WAVE (so a pipe shared by
// two waves at full rate shows the single-wave figure doubled).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP 64
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(8))) __bf16 b8;

template <int OP>
__global__ void probe(float* out, long long* cyc, float seed) {
    float a[8];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = seed + threadIdx.x * 1e-3f + i;
        p[i] = f2{a[i], a[i] + 0.5f};
    }
    const float c1 = 1.0001f, c2 = 1e-4f;
    const f2 pc1 = {c1, c1}, pc2 = {c2, c2};
    f16v acc0 = {}, acc1 = {};
    b8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc1), "v"(pc2));
                if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
                if (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc2));
                if (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
                if (OP == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
                if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc1));
                if (OP == 8) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c1));
                if (OP == 9) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
                if (OP == 10) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 11) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(1));
                if (OP == 12) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
                if (OP == 13) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
                if (OP == 14) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(c1));
                if (OP == 15) {      // 1 MFMA 32x32x16 + 7 v_fma per 8 slots
                    if (i == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
                }
                if (OP == 16) {      // 1 MFMA + 7 v_exp per 8 slots
                    if (i == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                    else asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                }
                if (OP == 17) {      // 1 MFMA + 3 v_exp + 4 v_fma per 8 slots
                    if (i == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                    else if (i < 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
                }
                if (OP == 18) {      // MFMAs alone, two accumulators
                    if (i & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc0, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1, 0, 0, 0);
                }
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    s += acc0[0] + acc1[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
static void run(const char* name, float* out, long long* cyc) {
    for (int threads : {256, 512, 1024}) {
        const int nblk = 256;      // one workgroup per CU at 1024 threads; more than one may share a CU at 256 (reported as is)
        hipLaunchKernelGGL(probe<OP>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.5f);       // warm-up, one workgroup: clean figure
        hipDeviceSynchronize();
        hipLaunchKernelGGL(probe<OP>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.5f);
        hipDeviceSynchronize();
        std::vector<long long> h(threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double mx = 0;
        for (auto v : h) mx = v > mx ? v : mx;
        printf("%-34s %4d threads (%d wave%s / SIMD): %7.2f cycles per instruction per wave\n", name, threads, threads / 256,
               threads > 256 ? "s" : " ", mx / (REP * 32.0));
        (void)nblk;
    }
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 4 * 256); hipMalloc(&cyc, 8 * 16 * 256);
    // s_memtime ticks: is it the shader clock?  (constant 100 MHz on some parts: then everything below is in those units)
    run<8>("v_mov_b32", out, cyc);
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_pk_fma_f32", out, cyc);
    run<3>("v_add_f32", out, cyc);
    run<4>("v_pk_add_f32", out, cyc);
    run<7>("v_pk_mul_f32", out, cyc);
    run<5>("v_max3_f32", out, cyc);
    run<6>("v_cvt_pk_bf16_f32", out, cyc);
    run<2>("v_exp_f32", out, cyc);
    run<9>("v_exp_f16", out, cyc);
    run<10>("v_rcp_f32", out, cyc);
    run<11>("v_ldexp_f32", out, cyc);
    run<12>("v_fract_f32", out, cyc);
    run<13>("v_pk_fma_f16", out, cyc);
    run<14>("v_lshl_add_u32", out, cyc);
    run<18>("mfma 32x32x16 bf16 alone", out, cyc);
    run<15>("1 mfma + 7 v_fma per 8", out, cyc);
    run<16>("1 mfma + 7 v_exp per 8", out, cyc);
    run<17>("1 mfma + 3 v_exp + 4 v_fma per 8", out, cyc);
    return 0;
}
