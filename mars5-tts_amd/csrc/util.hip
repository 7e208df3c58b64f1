// ABI version, hipGraph capture helpers and HIP-event timing.
#include "common.h"

extern "C" int m5_version(void) { return 1; }
extern "C" const char* m5_build_info(void) { return "mars5_hip gfx950 wave64 (f32/f16/bf16 MFMA) built " __DATE__; }

extern "C" int m5_graph_begin(void* stream) {
    if (!stream) return M5_ERR_ARG;   // the legacy default stream cannot be captured
    return hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}

extern "C" int m5_graph_end(void* stream, void** graph_exec) {
    if (!stream || !graph_exec) return M5_ERR_ARG;
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture((hipStream_t)stream, &g) != hipSuccess || !g) return M5_ERR_LAUNCH;
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return M5_ERR_LAUNCH;
    *graph_exec = (void*)ex;
    return M5_OK;
}

extern "C" int m5_graph_launch(void* graph_exec, void* stream) {
    if (!graph_exec) return M5_ERR_ARG;
    return hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}

extern "C" int m5_graph_destroy(void* graph_exec) {
    if (!graph_exec) return M5_ERR_ARG;
    return hipGraphExecDestroy((hipGraphExec_t)graph_exec) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}

extern "C" int m5_event_create(void** ev) {
    if (!ev) return M5_ERR_ARG;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return M5_ERR_LAUNCH;
    *ev = (void*)e;
    return M5_OK;
}
extern "C" int m5_event_record(void* ev, void* stream) {
    if (!ev) return M5_ERR_ARG;
    return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}
extern "C" int m5_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
    if (!ev_start || !ev_stop || !ms) return M5_ERR_ARG;
    if (hipEventSynchronize((hipEvent_t)ev_stop) != hipSuccess) return M5_ERR_LAUNCH;
    return hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}
extern "C" int m5_event_destroy(void* ev) {
    if (!ev) return M5_ERR_ARG;
    return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? M5_OK : M5_ERR_LAUNCH;
}
// In-graph timing: a one-lane kernel that stores the 100 MHz wall clock (s_memrealtime).  Captured between the launches of a
// hipGraph it runs after its predecessor has drained and before its successor starts, so slot[i + 1] - slot[i] is the time the
// launch between them occupies in the replay (its launch boundary included) -- what HIP events cannot give inside a capture.
namespace {
__global__ void clock_stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
}  // namespace
extern "C" int m5_clock_stamp(uint64_t* slot, void* stream) {
    if (!slot) return M5_ERR_ARG;
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slot);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

#ifdef M5_TOOLS   // probes behind tools/*.py (dispatch census, launch floor, operand-feed probe, grid barrier): tools library only

// ---- placement census (diagnostics; tools/census.py): where does the dispatcher put the
// workgroups of a grid with this shape?  Each workgroup records {XCC_ID, HW_ID, start, end clock}
// and spins for `spin` clock ticks so that the whole grid is co-resident when it fits.
namespace {
__global__ void census_kernel(uint32_t* out, int lds_bytes, int spin) {
    extern __shared__ unsigned char dyn[];
    uint32_t xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const uint64_t t0 = __builtin_readcyclecounter();
    if (lds_bytes > 0) dyn[threadIdx.x % lds_bytes] = (unsigned char)threadIdx.x;
    while ((int64_t)(__builtin_readcyclecounter() - t0) < spin) __builtin_amdgcn_s_sleep(8);
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        uint32_t* o = out + (size_t)blockIdx.x * 6;
        o[0] = xcc; o[1] = hwid; o[2] = (uint32_t)t0; o[3] = (uint32_t)(t0 >> 32); o[4] = (uint32_t)t1; o[5] = (uint32_t)(t1 >> 32);
    }
}
}  // namespace

extern "C" int m5_debug_census(uint32_t* out, int nblocks, int threads, int lds_bytes, int spin, void* stream) {
    if (!out || nblocks <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024) return M5_ERR_ARG;
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(census_kernel, dim3(nblocks), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream, out, lds_bytes, spin);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

// ---- launch-floor probe (diagnostics; tools/launch_floor.py): n dependent trivial launches on a stream.
namespace {
__global__ void probe_kernel(int* p, int touch) {
    if (touch && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
}  // namespace
extern "C" int m5_debug_launch_chain(int* buf, int n, int blocks, int threads, int touch, void* stream) {
    if (!buf || n <= 0 || blocks <= 0 || threads <= 0) return M5_ERR_ARG;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, buf, touch);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

// ---- operand-feed probe (diagnostics; tools/feed_probe.py): how many bytes per second can one CU
// pull from an L2-resident panel into LDS, by LDS-DMA vs by register staging?  Every workgroup
// (NW waves) repeatedly stages the same `panel_bytes` slice (its XCD-mates share it through L2).
namespace {
__device__ inline void probe_glds16(const unsigned char* gsrc, uint32_t lds_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int MODE>   // 0: LDS-DMA, 1: global_load_dwordx4 -> ds_write_b128, 2: global loads only (no LDS)
__global__ void feed_probe_kernel(const unsigned char* src, int64_t panel_bytes, int iters, int row_bytes, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    // panel = rows of `row_bytes` (128-byte K-slab of each row, like a GEMM stage); XCD x reads panel x
    const unsigned char* base = src + (int64_t)(blockIdx.x & 7) * panel_bytes;
    const int64_t nchunk = panel_bytes / 1024;                  // 1 KiB wave-instructions in the panel
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)dyn;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (int64_t c = wave; c < nchunk; c += nw * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t cc = min(c + (int64_t)u * nw, nchunk - 1);
                // 8 rows x 128 B per instruction: lane -> row lane>>3, 16-B chunk lane&7, rows `row_bytes` apart
                const unsigned char* g = base + (cc * 8 + (lane >> 3)) * (int64_t)row_bytes % panel_bytes + (lane & 7) * 16;
                const uint32_t slot = (uint32_t)(((wave + u * nw) & 31) * 1024);
                if (MODE == 0) probe_glds16(g, lds_base + slot);
                else v[u] = *reinterpret_cast<const uint4*>(g);
            }
            if (MODE == 1) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    *reinterpret_cast<uint4*>(dyn + ((wave + u * nw) & 31) * 1024 + lane * 16) = v[u];
            } else if (MODE == 2) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].w;
            }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc ^= *reinterpret_cast<const uint32_t*>(dyn + (tid & 1023) * 16);
    }
    if (acc == 0x12345678u) sink[0] = 1.f;
}
}  // namespace
extern "C" int m5_debug_feed_probe(const void* src, int64_t panel_bytes, int iters, int row_bytes, int mode, int blocks, int threads,
                                   float* sink, void* stream) {
    if (!src || !sink || panel_bytes < 8192 || (panel_bytes % 1024) || iters <= 0 || blocks <= 0 || threads % 64 || row_bytes % 128) return M5_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = 32 * 1024;
    switch (mode) {
        case 0: hipLaunchKernelGGL(feed_probe_kernel<0>, dim3(blocks), dim3(threads), lds, s, (const unsigned char*)src, panel_bytes, iters, row_bytes, sink); break;
        case 1: hipLaunchKernelGGL(feed_probe_kernel<1>, dim3(blocks), dim3(threads), lds, s, (const unsigned char*)src, panel_bytes, iters, row_bytes, sink); break;
        case 2: hipLaunchKernelGGL(feed_probe_kernel<2>, dim3(blocks), dim3(threads), lds, s, (const unsigned char*)src, panel_bytes, iters, row_bytes, sink); break;
        default: return M5_ERR_ARG;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}

// ---- grid-barrier probe (diagnostics; tools/grid_barrier.py): what does a device-wide barrier between the
// co-resident workgroups of ONE launch cost on this part (agent-scope atomic arrive + acquire spin), against the
// 1.6 us dependent-launch floor?  That number decides whether a persistent decode-step kernel can beat one launch
// per stage.  mode 1 adds a produce/consume of one word per workgroup across the barrier (visibility check).
// Spins are bounded: a workgroup that waits more than 2^20 polls counts an error and moves on (never hangs).
namespace {
__global__ void grid_barrier_kernel(unsigned* ctr, unsigned* data, unsigned* err, int iters, int mode) {
    const unsigned nb = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        if (mode && threadIdx.x == 0)
            __hip_atomic_store(&data[blockIdx.x], (unsigned)it * nb + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * nb;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1 << 20)) { atomicAdd(err, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (mode && threadIdx.x == 0) {
            const unsigned nbr = (blockIdx.x + 1) % nb;
            const unsigned v = __hip_atomic_load(&data[nbr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the neighbour may already have crossed into the next iteration and published its next word
            if (v != (unsigned)it * nb + nbr && v != (unsigned)(it + 1) * nb + nbr) atomicAdd(err + 1, 1u);
        }
    }
}
}  // namespace
extern "C" int m5_debug_grid_barrier(uint32_t* scratch, int blocks, int threads, int iters, int mode, void* stream) {
    // scratch: >= blocks + 4 words; [0] arrive counter, [1..2] error counts (timeouts, stale reads), [4..] data
    if (!scratch || blocks <= 0 || blocks > 1024 || threads <= 0 || threads > 1024 || iters <= 0) return M5_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(scratch, 0, sizeof(uint32_t) * (blocks + 4), s) != hipSuccess) return M5_ERR_ARG;
    hipLaunchKernelGGL(grid_barrier_kernel, dim3(blocks), dim3(threads), 0, s, scratch, scratch + 4, scratch + 1, iters, mode);
    M5_CHECK_LAUNCH();
    return M5_OK;
}


// ---- L2-retention probe (diagnostics; tools/l2_retention.py): does data a launch pulled into an XCD's L2 still hit there
// in the NEXT launch of the stream?  Workgroup b streams chunk (b + shift) % gridDim.x of the buffer (plain or nt 16-byte
// loads).  A timed launch (shift 0) right after a warming launch with shift 0 finds its chunk where the same XCD left it;
// after shift 1 the chunk was left in the neighbouring XCD's L2; after a launch over another buffer it is cold.  That
// decides whether producer -> consumer XCD affinity / same-stream prefetch can pay on this part.
namespace {
template <bool NT>
__global__ void l2_touch_kernel(const unsigned char* buf, int64_t chunk_bytes, int shift, float* sink) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int64_t c = (blockIdx.x + shift) % gridDim.x;
    const u32x4* p = reinterpret_cast<const u32x4*>(buf + c * chunk_bytes);
    const int n = (int)(chunk_bytes / 16);
    unsigned acc = 0;
#pragma unroll 8
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const u32x4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = 1.f;
}
}  // namespace
extern "C" int m5_debug_l2_touch(const void* buf, int64_t chunk_bytes, int blocks, int threads, int shift, int nt, float* sink, void* stream) {
    if (!buf || !sink || chunk_bytes <= 0 || (chunk_bytes % 16) || blocks <= 0 || threads <= 0 || threads > 1024) return M5_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (nt) hipLaunchKernelGGL(l2_touch_kernel<true>, dim3(blocks), dim3(threads), 0, s, (const unsigned char*)buf, chunk_bytes, shift, sink);
    else hipLaunchKernelGGL(l2_touch_kernel<false>, dim3(blocks), dim3(threads), 0, s, (const unsigned char*)buf, chunk_bytes, shift, sink);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

// ---- all-gather edge probe (diagnostics; tools/edge_probe.py): what does one dependency edge of a persistent decode
// step cost on this part?  Every workgroup produces `per` fp32 values of an n-vector as 8-byte {value, tag} granules (one
// agent-scope relaxed 8-byte store each: value and tag can never be seen torn), then every workgroup sweeps the whole
// vector until all n tags carry the edge's tag, and consumes it (a sum, checked on the host).  `stream_kb` > 0 adds a
// non-temporal read of that many KiB per workgroup and edge from `wbuf` (the weight stream a decode phase runs under).
// Two granule buffers alternate, so a fast workgroup never overwrites values a slow one still waits for.
// Spins are bounded: err[0] counts workgroups that gave up (the kernel never hangs).
namespace {
__global__ void edge_probe_kernel(unsigned long long* gran, int n, int per, int iters, unsigned base_tag,
                                  const unsigned char* wbuf, int stream_kb, unsigned* err, float* sums) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, nt = blockDim.x, b = blockIdx.x;
    __shared__ float red[16];
    __shared__ int bad;
    float total = 0.f;
    unsigned sink = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned tag = base_tag + (unsigned)it + 1u;
        unsigned long long* g = gran + (size_t)(it & 1) * n;
        // the phase's weight stream (issued first, consumed after the edge)
        u32x4 w[8];
        const int nld = stream_kb * 1024 / (nt * 16);            // 16-byte loads per thread
        const unsigned char* wp = wbuf + ((size_t)b * iters + it) % 4096 * ((size_t)stream_kb * 1024);
        for (int i = 0; i < 8; ++i) w[i] = (i < nld) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + ((size_t)i * nt + tid) * 16)) : u32x4{0, 0, 0, 0};
        // produce
        if (tid < per && b * per + tid < n) {
            const float v = (float)((it + b * per + tid) & 1023);
            const unsigned long long word = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
            __hip_atomic_store(&g[b * per + tid], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // consume: thread t owns granules t, t + nt, ...
        float s = 0.f;
        if (tid == 0) bad = 0;
        __syncthreads();
        for (int i = tid; i < n; i += nt) {
            unsigned long long word;
            int spins = 0;
            while (true) {
                word = __hip_atomic_load(&g[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(word >> 32) == tag) break;
                if (++spins > (1 << 18)) { bad = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            s += __uint_as_float((unsigned)word);
        }
        for (int i = 0; i < 8; ++i) sink ^= w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
        // block sum (the consumer's use of the vector)
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) { float t = 0.f; for (int k = 0; k < nt / 64; ++k) t += red[k]; total += t; if (bad) atomicAdd(err, 1u); }
        __syncthreads();
        if (bad) break;
    }
    if (tid == 0) sums[b] = total + (sink == 0x12345u ? 1.f : 0.f);
}
}  // namespace
extern "C" int m5_debug_edge_probe(uint64_t* gran, int n, int per, int blocks, int threads, int iters, uint32_t base_tag,
                                   const void* wbuf, int stream_kb, uint32_t* err, float* sums, void* stream) {
    // gran: 2 n words (zeroed by the caller once; tags only grow); wbuf: >= 4096 * stream_kb KiB when stream_kb > 0
    if (!gran || !err || !sums || n <= 0 || per <= 0 || blocks <= 0 || blocks > 1024 || threads < 64 || threads > 1024 || (threads % 64) || iters <= 0) return M5_ERR_ARG;
    if (per > threads || (int64_t)blocks * per < n || stream_kb < 0 || stream_kb * 1024 > threads * 16 * 8 || (stream_kb && !wbuf)) return M5_ERR_ARG;
    hipLaunchKernelGGL(edge_probe_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (unsigned long long*)gran, n, per, iters, base_tag,
                       (const unsigned char*)wbuf, stream_kb, err, sums);
    M5_CHECK_LAUNCH();
    return M5_OK;
}
#endif  // M5_TOOLS
