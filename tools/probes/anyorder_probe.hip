// Probe (tools only): can dependent launches of ONE stream overlap on gfx950 when the consumer is launched with
// hipExtAnyOrderLaunch (AQL barrier bit clear) and waits for its producers INSIDE the kernel?
//
// A chain of L launches ping-pongs two buffers; launch i reads the tile that workgroup (b + 37) % G of launch i-1 wrote
// (another XCD: workgroups are dealt round-robin over the 8 XCDs) and writes tile b, value + 1.  After L launches every
// element must be initial + L, so a missing write-back / invalidate across the XCDs' L2s shows up as a wrong sum.
//   mode 0  ordered launches, no counters                    (what a hipGraph chain does today)
//   mode 1  ordered launches + release fence / counter add / acquire (price of the fences alone)
//   mode 2  any-order launches, consumer waits for ALL tiles of the producer (one counter per launch)
//   mode 3  any-order launches, consumer waits for the ONE producer tile it reads (a flag word per tile)
//   mode 4  = mode 2 with WRITE-THROUGH stores (sc0 sc1) + s_waitcnt vmcnt(0) instead of the release fence's L2 write-back
//   mode 5  = mode 1 with the same write-through stores (ordered launches: the price of that hand-off alone)
// Each workgroup stamps the 100 MHz wall clock at entry, after its wait and at exit: the overlap of launch i+1's entries with
// launch i's exits is read off the stamps.  Spins are bounded (a grid that is not dispatched in order costs ms, never hangs).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/anyorder_probe.hip -o tools/probes/anyorder_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <chrono>

typedef unsigned long long u64;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

struct Args {
    const float* in;
    float* out;
    int tile_floats;          // per workgroup
    int spin;                 // extra arithmetic per element (emulates the K loop)
    unsigned* wait_ctr;       // mode 1/2: counter of the producing launch (null: none)
    unsigned wait_for;
    unsigned* done_ctr;       // this launch's counter
    unsigned* wait_flags;     // mode 3: one word per producer tile
    unsigned* done_flags;
    unsigned epoch;           // value the flags / counters carry for this launch
    u64* stamps;              // [G][3]
    unsigned* timeouts;
    int shift;
    int wt;                   // 1: write-through stores + vmcnt(0) instead of fence(release, agent)
};

__global__ __launch_bounds__(256) void chain_kernel(Args a) {
    extern __shared__ float lds[];
    const int b = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    u64 t0 = wall_clock64();
    const int src = (b + a.shift) % G;
    if (tid == 0) {
        long n = 0;
        if (a.wait_ctr) {
            while (__hip_atomic_load(a.wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.wait_for) {
                __builtin_amdgcn_s_sleep(2);
                if (++n > 2000000) { atomicAdd(a.timeouts, 1u); break; }
            }
        } else if (a.wait_flags) {
            while (__hip_atomic_load(a.wait_flags + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch - 1) {
                __builtin_amdgcn_s_sleep(2);
                if (++n > 2000000) { atomicAdd(a.timeouts, 1u); break; }
            }
        }
        if (a.wait_ctr || a.wait_flags) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    u64 t1 = wall_clock64();
    const float4* in = reinterpret_cast<const float4*>(a.in + (size_t)src * a.tile_floats);
    float4* out = reinterpret_cast<float4*>(a.out + (size_t)b * a.tile_floats);
    const int n4 = a.tile_floats / 4;
    for (int i = tid; i < n4; i += 256) {
        float4 v = in[i];
        float z = 0.f;
        for (int k = 0; k < a.spin; ++k) z = __builtin_fmaf(z, 0.5f, v.x * 1e-30f);
        v.x += 1.f + z; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        if (a.wt) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v vv = {v.x, v.y, v.z, v.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(out + i), "v"(vv) : "memory");
        } else {
            out[i] = v;
        }
    }
    if (tid == 0) lds[0] = 1.f;
    __syncthreads();                                   // every thread's stores are issued
    if (a.done_ctr || a.done_flags) {
        if (a.wt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // write-through stores: acknowledged = at the memory side
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // each wave waits for its own stores + writes L2 back
        __syncthreads();
        if (tid == 0) {
            if (a.done_ctr) __hip_atomic_fetch_add(a.done_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.done_flags) __hip_atomic_store(a.done_flags + b, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    u64 t2 = wall_clock64();
    if (tid == 0) { a.stamps[b * 3 + 0] = t0; a.stamps[b * 3 + 1] = t1; a.stamps[b * 3 + 2] = t2; }
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 256;
    const int tile_floats = argc > 2 ? atoi(argv[2]) : 16384;    // 64 KB per workgroup
    const int spin = argc > 3 ? atoi(argv[3]) : 64;
    const int L = argc > 4 ? atoi(argv[4]) : 64;
    const int lds = argc > 5 ? atoi(argv[5]) : 100 * 1024;       // one workgroup per CU, like the region GEMMs
    const int reps = 5;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d  G %d tile %d KB spin %d L %d lds %d\n", prop.gcnArchName, prop.multiProcessorCount, G, tile_floats / 256, spin, L, lds);
    CK(hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    size_t n = (size_t)G * tile_floats;
    float *A, *B; unsigned *ctr, *flags, *timeouts; u64* stamps;
    CK(hipMalloc(&A, n * 4)); CK(hipMalloc(&B, n * 4));
    CK(hipMalloc(&ctr, (L + 1) * 4)); CK(hipMalloc(&flags, 2 * G * 4)); CK(hipMalloc(&timeouts, 4));
    CK(hipMalloc(&stamps, (size_t)L * G * 3 * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> host(n);
    std::vector<u64> hst((size_t)L * G * 3);
    for (int mode = 0; mode < 6; ++mode) {
        double best = 1e30, host_us = 0;
        long bad = 0; unsigned to = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemsetAsync(A, 0, n * 4, s)); CK(hipMemsetAsync(B, 0, n * 4, s));
            CK(hipMemsetAsync(ctr, 0, (L + 1) * 4, s)); CK(hipMemsetAsync(flags, 0, 2 * G * 4, s)); CK(hipMemsetAsync(timeouts, 0, 4, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            auto h0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < L; ++i) {
                Args a{};
                a.in = (i & 1) ? B : A; a.out = (i & 1) ? A : B;
                a.tile_floats = tile_floats; a.spin = spin; a.shift = 37;
                a.stamps = stamps + (size_t)i * G * 3; a.timeouts = timeouts; a.epoch = i + 1;
                if (mode == 1 || mode == 2 || mode >= 4) { a.wait_ctr = i ? ctr + i - 1 : nullptr; a.wait_for = G; a.done_ctr = ctr + i; }
                a.wt = mode >= 4;
                if (mode == 3) { a.wait_flags = i ? flags + ((i - 1) & 1) * G : nullptr; a.done_flags = flags + (i & 1) * G; }
                int fl = ((mode == 2 || mode == 3 || mode == 4) && i > 0) ? hipExtAnyOrderLaunch : 0;
                hipExtLaunchKernelGGL(chain_kernel, dim3(G), dim3(256), lds, s, nullptr, nullptr, fl, a);
            }
            auto h1 = std::chrono::high_resolution_clock::now();
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, (double)ms * 1000.0 / L);
            host_us = std::chrono::duration<double, std::micro>(h1 - h0).count() / L;
            CK(hipMemcpy(host.data(), (L & 1) ? B : A, n * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n; ++i) bad += host[i] != (float)L;
            unsigned t; CK(hipMemcpy(&t, timeouts, 4, hipMemcpyDeviceToHost)); to += t;
        }
        CK(hipMemcpy(hst.data(), stamps, hst.size() * 8, hipMemcpyDeviceToHost));
        // overlap statistics of the last repetition: for launch i >= 1, (first entry of i) - (last exit of i-1), and how many
        // workgroups of i entered before the last exit of i-1; mean wait (t1 - t0) and body (t2 - t1) in us
        double gap = 0, early = 0, waitus = 0, body = 0, order_viol = 0;
        for (int i = 1; i < L; ++i) {
            u64 last_exit = 0, first_entry = ~0ull, last_entry_prev = 0;
            for (int b = 0; b < G; ++b) { last_exit = std::max(last_exit, hst[((size_t)(i - 1) * G + b) * 3 + 2]); last_entry_prev = std::max(last_entry_prev, hst[((size_t)(i - 1) * G + b) * 3]); }
            int ne = 0;
            for (int b = 0; b < G; ++b) {
                u64 t0 = hst[((size_t)i * G + b) * 3], t1 = hst[((size_t)i * G + b) * 3 + 1], t2 = hst[((size_t)i * G + b) * 3 + 2];
                first_entry = std::min(first_entry, t0);
                ne += t0 < last_exit;
                waitus += (double)(t1 - t0) / 100.0; body += (double)(t2 - t1) / 100.0;
            }
            gap += ((double)first_entry - (double)last_exit) / 100.0;
            early += ne;
            order_viol += first_entry < last_entry_prev;
        }
        printf("mode %d: %.2f us/launch (best of %d)  host %.2f us/launch  wrong elements %ld  timeouts %u | first entry - prev last exit %.2f us, early workgroups %.1f / %d, "
               "launches entered before prev fully dispatched %.0f, mean wait %.2f us, mean body %.2f us\n",
               mode, best, reps, host_us, bad, to, gap / (L - 1), early / (L - 1), G, order_viol, waitus / ((L - 1) * G), body / ((L - 1) * G));
    }
    return 0;
}
