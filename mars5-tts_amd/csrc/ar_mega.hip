// AR decode step, batch 1: the 26 Mistral layers of one token as ONE persistent launch (reference mars5/nn_future.py:
// 235-274, 326-333 -- TransformerBlock.forward / Attention.forward decode branch / FeedForward.forward).
//
// The 5-launches-per-layer form (ar_decode.hip) spends ~3.8 us per launch outside its weight stream (boundary, first
// byte, drain) and streams a layer's 52 MB in ~8 us: 27.4 us per layer.  Here 256 co-resident workgroups (one per CU, 8
// waves) walk the layers themselves: 17.8 us per layer, 520 vs 740 us per token (profiles/r3_ar_persistent_step_log.txt).
// The five launches become phases of the same arithmetic, row for row and lane for lane (a workgroup owns exactly the rows
// workgroup blockIdx.x of the corresponding launch owned, with the same lane / k mapping and the same reduction trees, so
// every dot product, RMSNorm sum and softmax merge is bit-identical to ar_decode.hip; tests/test_gpu_parity16.py):
//   P1 RMSNorm -> Wqkv rows -> RoPE -> KV-cache slot + {q, k, v} of this token          (18 rows: 9 pairs over waves 0-6)
//   P2 workgroups 0..191: (head, key split) cache scan;  192..215: merge the 8 splits of one head
//   P3 Wo rows -> x += .                                                                  (waves 0-5: 1 row each)
//   P4 RMSNorm -> interleaved (W1, W3) rows -> silu(a) * b                                (waves 0-6: 4 rows each)
//   P5 W2 rows -> x += .                                                                  (waves 0-5: 1 row each)
//
// Edges.  What a phase needs from OTHER workgroups crosses as 8-byte granules {32-bit payload, tag} written by one
// agent-scope store each (payload and tag cannot be seen torn; no fence, no flag): tag = (pos + 1) * 256 + layer * 8 + edge
// is unique within an utterance, and the host zeroes the granule buffer at prefill.  fp32 values (residual stream, split-KV
// partials) travel one per granule, 16-bit values (q | k | v, merged attention output, SwiGLU output) two per granule.  The
// consumers sweep the vector (every granule of a wave's share in flight, repeated until all tags match) into LDS.
//   * ONE store instruction publishes a workgroup's outputs of a phase (collected through LDS).  An edge costs ~0.3 us per
//     store instruction of a different wave / CU that lands in the same 128-byte line: the SwiGLU edge took 4.7 us while
//     each of 7 waves x 256 workgroups stored its own granule, 1.7 us with one instruction per workgroup (-17 % per token).
//   * A CU's polls queue behind its own DMA returns (~25 GB/s per CU), and a wave's polls behind everything that wave has
//     in flight (the VM counter retires in order).  So no wave that polls ever has a weight load in flight.
// A granule buffer is reused by every layer: safe, because a workgroup publishes edge e of layer l + 1 only after it has
// gathered everything up to edge e - 1 of that layer, which every workgroup published after consuming edge e of layer l.
//
// Weights.  Wave 7 is the loader: it alone requests weight rows, by LDS-DMA (no data registers: hipcc spilled every
// register-held prefetch of this kernel), as flat copies of the workgroup's contiguous rows into two LDS regions
// (A: Wqkv rows, then W1|W3 rows; B: Wo rows, then W2 rows), one to three phases ahead and -- as far as the regions
// allow -- at points where the workgroup is about to compute rather than about to wait for an edge.  It makes sure a set
// has landed (counted vmcnt: the sets retire in issue order) before it joins the barrier in front of the products that
// read it.  Waves 0-6 only ever touch LDS and granules.
//
// Workgroup barriers order LDS only (lgkmcnt(0) + s_barrier, never the VM counter).  Spins are bounded: a gatherer that
// waits ~2^16 sweeps raises err[0] and its workgroup leaves; the others then run into their own bound.  err is sticky
// (later launches return at once), so a broken co-residency assumption costs milliseconds, never a hang; the host raises.
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int MD = 1536, MF = 3584, MH = 24, MNS = 8;
constexpr int E_X2 = 1, E_QKV = 2, E_PART = 3, E_O = 4, E_X1 = 5, E_H = 6;
constexpr int PART_H = 576;                                     // 8 x 66 partial words of a head, padded to 9 x 64 (whole gather sweeps)
// q | k | v, the merged attention output and the SwiGLU output are 16-bit values: two per granule
constexpr int G_X2 = 0, G_QKV = G_X2 + MD, G_PART = G_QKV + 3 * MD / 2, G_O = G_PART + MH * PART_H, G_X1 = G_O + MD / 2,
              G_H = G_X1 + MD, G_WORDS = G_H + MF / 2;
static_assert(G_WORDS == M5_AR_MEGA_GRANULES, "granule buffer size in the header");
constexpr int SPIN_LIMIT = 1 << 16;

// Workgroup barrier that orders LDS only (every LDS operation of the wave retired, then s_barrier).  __syncthreads() would
// also wait for the VM counter, i.e. for the weight rows a compute wave has in flight for LATER phases and for its granule
// stores -- exactly the overlap this kernel exists for.  Cross-workgroup data only travels in granules (single atomic
// words), so no global-memory fence is needed here.
__device__ inline void bar() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// diagnostics (tools/ar_mega_clock.py, tools build): workgroups 0, 192, 255 and 100 stamp the 100 MHz wall clock at their phase boundaries
#ifdef M5_TOOLS
__device__ inline void mstamp(unsigned long long* d, int l, int k) {
    const int sel = blockIdx.x == 0 ? 0 : (blockIdx.x == 192 ? 1 : (blockIdx.x == 255 ? 2 : (blockIdx.x == 100 ? 3 : -1)));
    if (d && sel >= 0 && (threadIdx.x & 63) == 0) d[((sel * 32 + l) * 16 + k) * 8 + (threadIdx.x >> 6)] = wall_clock64();
}
#else
__device__ inline void mstamp(unsigned long long*, int, int) {}
#endif

__device__ inline void publish(u64* g, int i, float v, unsigned tag) {
    __hip_atomic_store(g + i, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void publish_raw(u64* g, int i, unsigned pay, unsigned tag) {
    __hip_atomic_store(g + i, ((u64)tag << 32) | (u64)pay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned pack16(unsigned short a, unsigned short b) { return (unsigned)a | ((unsigned)b << 16); }
// two 16-bit values (storage bits of the operand type) in one granule: element 2 i in the low half
template <typename T>
__device__ inline void publish2(u64* g, int i, typename T::storage lo, typename T::storage hi, unsigned tag) {
    unsigned short a, b;
    __builtin_memcpy(&a, &lo, 2);
    __builtin_memcpy(&b, &hi, 2);
    __hip_atomic_store(g + i, ((u64)tag << 32) | (u64)((unsigned)a | ((unsigned)b << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ inline float half_to_f32(unsigned bits) {
    const unsigned short u = (unsigned short)bits;
    typename T::storage v;
    __builtin_memcpy(&v, &u, 2);
    return T::to_f32(v);
}

// One wave gathers granules (j, lane) = g[j * sj + lane], j < PER, lane < nl, into dst: every sweep has all PER loads of
// a lane in flight; a granule is accepted once its tag matches.  PACK: a granule carries two 16-bit values, written as
// fp32 to dst[2 (j nl + lane)], [.. + 1]; else one fp32 value to dst[j nl + lane].  false: gave up.
// Addresses: wave-uniform base (SGPRs) + one shared 32-bit lane offset; the buffer is readable up to (PER - 1) sj + 64 words.
// PACK: 0 = granules carry one fp32 value each; 1 = two operand-type values, unpacked to fp32 (the head's q | k | v for the cache
// scan); 2 = two operand-type values stored as they travelled (the products' operand vector, common.h dot8)
template <typename T, int PER, int PACK>
__device__ inline bool gather(const u64* g, int sj, int nl, unsigned tag, void* dst_, int lane, int n_total = PER * 64) {
    float* dst = reinterpret_cast<float*>(dst_);                // PACK: granule i = elements 2i, 2i + 1 of an operand-type vector
    const unsigned lane8 = (unsigned)lane * 8u;
    unsigned pend = 0;                                          // granule (j, lane) is wanted iff lane < nl and j nl + lane < n_total
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (lane < nl && j * nl + lane < n_total) pend |= 1u << j;
    for (int spins = 0; spins < SPIN_LIMIT; ++spins) {
        u64 w[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j)
            w[j] = __hip_atomic_load(reinterpret_cast<const u64*>(reinterpret_cast<const unsigned char*>(g + (size_t)j * sj) + lane8),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (((pend >> j) & 1u) && (unsigned)(w[j] >> 32) == tag) {
                const unsigned pay = (unsigned)w[j];
                if (PACK == 2) {
                    reinterpret_cast<unsigned*>(dst_)[j * nl + lane] = pay;      // the pair as it travelled: the products' operand (dot8)
                } else if (PACK == 1) {
                    *reinterpret_cast<float2*>(dst + 2 * (j * nl + lane)) = make_float2(half_to_f32<T>(pay & 0xffffu), half_to_f32<T>(pay >> 16));
                } else {
                    dst[j * nl + lane] = __uint_as_float(pay);
                }
                pend &= ~(1u << j);
            }
        if (__ballot(pend != 0) == 0) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// The wave's R weight rows, whole K, on their way into ITS slice of LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
// instruction, lane l -> bytes [16 l, 16 l + 16) -- the lane -> k mapping of gemv_stream_kernel's register loads, so the dot
// products below see the same operands in the same order).  No data VGPRs: hipcc spilled every register-held prefetch of
// this kernel (a load, a wait, a scratch store per 1 KiB).  Non-temporal: each row is read once, by one CU.
// Issued from inline asm (M0 = LDS base, saved / restored): the compiler does not count these on the VM counter; the
// consuming wave waits vmcnt(0) itself (wait_dma) before it reads its slice -- only its own DMAs land there.
__device__ inline void glds16_nt(const unsigned char* gsrc, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
// The loader wave: `pieces` KiB of a workgroup's contiguous weight rows -> a contiguous LDS region (a workgroup's rows of a
// phase are consecutive rows of the matrix, and the waves' slices are laid out in the same order)
// (Four pieces per M0 write through the instruction's immediate offset -- it advances the global AND the LDS address --
// issues them faster and measured SLOWER, 576 vs 564 us / token: the burst reaches the edges' polls.)
__device__ inline void dma_flat(uint32_t dst, const unsigned char* src, int pieces, int lane) {
    const unsigned char* p = src + lane * 16;
    for (int k = 0; k < pieces; ++k) glds16_nt(p + (int64_t)k * 1024, dst + k * 1024);
}

// Wait until at most N of this wave's VM operations are outstanding.  The counter retires in order, so N = the number of
// DMAs the wave has issued SINCE the rows it is about to read (never more than that: an over-count would let the wait
// return early; operations the count does not know of -- granule stores, polls -- only make it wait longer).
template <int N>
__device__ inline void wait_dma() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// acc[r] = row r . xs with gemv_stream_kernel's summation order (per lane: k chunks it = 0..NIT-1, 8 elements each, one
// fmaf chain per row; then the xor-shuffle tree); the rows come from the wave's LDS slice
template <typename T, int R, int NIT>
__device__ inline void dot_rows(float (&acc)[R], const unsigned char* slice, const typename T::storage* xs, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k0 = (it * 64 + lane) * 8;
        const uint4 xv = *reinterpret_cast<const uint4*>(xs + k0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint4 wv = *reinterpret_cast<const uint4*>(slice + (r * NIT + it) * 1024 + lane * 16);
            acc[r] = dot8<T>(wv, xv, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
}

// RMSNorm of the gathered vector (graw, MD values) into xs with the thread mapping of a gemv_stream launch of NWP waves
// (chunk c = tid + j * NWP * 64 of MD / 4 float4 chunks; block sum = wave trees, then the waves in order).  All 512
// threads call it (one workgroup barrier inside).
template <typename T, int NWP>
__device__ inline void rms_to_xs(const float* graw, const float* nws, float eps, typename T::storage* xs, float* red, int tid, int lane, int wave) {
    constexpr int NT = NWP * 64, NCH = MD / 4, JN = (NCH + NT - 1) / NT;
    float4 xin[JN], nwv[JN];
    float ss = 0.f;
    if (tid < NT) {
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int c = min(tid + j * NT, NCH - 1);
            xin[j] = *reinterpret_cast<const float4*>(graw + c * 4);
            nwv[j] = *reinterpret_cast<const float4*>(nws + c * 4);
        }
#pragma unroll
        for (int j = 0; j < JN; ++j)
            if (tid + j * NT < NCH) ss += xin[j].x * xin[j].x + xin[j].y * xin[j].y + xin[j].z * xin[j].z + xin[j].w * xin[j].w;
        ss = wave_sum(ss);
    }
    if (wave < NWP && lane == 0) red[wave] = ss;             // (the caller alternates two `red` arrays: no barrier needed before)
    bar();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NWP; ++i) tot += red[i];
    const float rstd = rsqrtf(tot / (float)MD + eps);
    if (tid < NT) {
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int c = tid + j * NT;
            if (c < NCH) {
                *reinterpret_cast<uint2*>(xs + c * 4) = m5_pack4<T>((xin[j].x * rstd) * nwv[j].x, (xin[j].y * rstd) * nwv[j].y,
                                                                    (xin[j].z * rstd) * nwv[j].z, (xin[j].w * rstd) * nwv[j].w);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(512) void ar_mega_kernel(M5ArMegaArgs a) {
    using st = typename T::storage;
    // LDS (156 KB of the CU's 160): graw | xs | weight region A (Wqkv rows, then W1|W3 rows: 84 KB) | region B (Wo rows,
    // then W2 rows: 42 KB) | small arrays.  Exactly one weight set is in flight / resident per region at a time.
    constexpr int OFF_A = 2 * MF * 4, OFF_B = OFF_A + 7 * 12 * 1024, OFF_END = OFF_B + 6 * 7 * 1024;
    __shared__ __attribute__((aligned(16))) unsigned char lds[OFF_END];
    float* graw = reinterpret_cast<float*>(lds);                // the gatherers' target (raw granule values)
    st* xs = reinterpret_cast<st*>(lds + MF * 4);               // the phase's activation vector as the dot products read it (operand type)
    float* nws = graw + 2048;                                   // RMSNorm weights of the phase (P1 / P4 gather 1536 values only)
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    __shared__ float red[2][8];
    __shared__ float sm[4][8][10];                              // attention: per-wave partial (o[8], m, l) of 8 d-groups
    __shared__ float xloc[8];                                   // this workgroup's 6 rows of the residual stream
    __shared__ unsigned pub[16];                                // a phase's outputs of this workgroup, published by ONE store instruction
    __shared__ int fail;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    if (a.state[M5_ST_DONE] || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const int pos = a.state[M5_ST_POS];
    const unsigned tag0 = (unsigned)(pos + 1) << 8;
    u64* g = reinterpret_cast<u64*>(a.gran);
    if (tid == 0) fail = 0;
    if (tid < 6) xloc[tid] = a.xres[b * 6 + tid];

    // RoPE factors of this lane's (q | k) row pair: the same in every layer
    // P1: the workgroup's 18 Wqkv rows = 9 (even, odd) pairs.  A row's product and reduction are the same whichever wave
    // runs them, so the pairs are spread over all 7 compute waves (2, 2, 1, 1, 1, 1, 1) instead of the 3 x 3 of the launch.
    const int qpair0 = wave < 2 ? 2 * wave : wave + 2, qnp = wave < 2 ? 2 : 1;      // first pair, pairs of this wave
    float rcs = 1.f, rsn = 0.f;
    if (wave < 7 && lane < qnp) {
        const int n = b * 18 + 2 * (qpair0 + lane);
        if (n < 2 * MD) {
            const int d = (n % MD) & 63;
            rcs = a.rope[((int64_t)pos * 32 + (d >> 1)) * 2];
            rsn = a.rope[((int64_t)pos * 32 + (d >> 1)) * 2 + 1];
        }
    }
    // attention geometry (attn_decode_kernel, 16-bit: 8 lanes per position, 8 positions per wave instruction)
    const int n_valid = min(pos + 1, a.window), slot_cur = pos % a.window;
    int chunk = (n_valid + MNS - 1) / MNS;
    chunk = (chunk + 31) / 32 * 32;
    const int ah = b >> 3, asplit = b & 7;
    const int astart = asplit * chunk, aend = min(n_valid, astart + chunk);
    const int sub = lane & 7, grp = lane >> 3;

    const int64_t cache_l = (int64_t)MH * a.w_alloc * 64;      // elements per layer of the KV cache
    const unsigned char* Wq = (const unsigned char*)a.wqkv;
    // Roles: waves 0-6 own rows; waves 3-6 also gather (they never have a DMA in flight, so their polls return as fast as
    // the fabric allows); wave 7 is the loader: it alone issues the weight DMAs, one phase or more ahead, and makes sure a
    // set has landed (counted vmcnt) before it joins the barrier in front of the products that read it.
    const bool loader = wave == 7;
    const unsigned char* Wo = (const unsigned char*)a.wo;
    if (loader) {
        dma_flat(lds_base + OFF_A, Wq + ((int64_t)a.layer0 * 3 * MD + (int64_t)b * 18) * MD * 2, 54, lane);
    }

    for (int l = a.layer0; l < a.layer1; ++l) {
        const unsigned tl = tag0 + (unsigned)l * 8u;
        mstamp(a.dbg, l, 0);
        // ------------------------------------------------------------------ P1: RMSNorm -> QKV rows -> RoPE
        if (wave < 6) {                                       // every wave but the loader gathers (none has anything in flight)
            const int q0 = wave * (MD / 6);
#pragma unroll
            for (int j = 0; j < MD / 384; ++j) nws[q0 + j * 64 + lane] = a.attn_norm[(int64_t)l * MD + q0 + j * 64 + lane];
            if (l == a.layer0) {
#pragma unroll
                for (int j = 0; j < MD / 384; ++j) graw[q0 + j * 64 + lane] = a.xres[q0 + j * 64 + lane];
            } else if (!gather<T, MD / 384, 0>(g + G_X2 + q0, 64, 64, tl - 8u + E_X2, graw + q0, lane)) {
                fail = 1;
            }
        }
        bar();
        if (fail) { if (tid == 0) atomicAdd(a.err, 1u); return; }
        mstamp(a.dbg, l, 1);
        // loader: this layer's Wo rows -> region B (the previous layer's W2 rows were consumed before its last barrier).  Requested HERE,
        // behind the x gather, not at the end of the previous layer: 18 pieces fewer in flight while the CU polls the x edge
        // (round 6, same box, alternated builds: 503.5 -> 495 us per token, profiles/r6f_ar_dma_placement_ab.txt)
        if (loader) dma_flat(lds_base + OFF_B, Wo + ((int64_t)l * MD + (int64_t)b * 6) * MD * 2, 18, lane);
        rms_to_xs<T, 3>(graw, nws, a.eps, xs, red[0], tid, lane, wave);
        mstamp(a.dbg, l, 11);
        if (loader) wait_dma<18>();                           // the Wqkv rows have landed (the 18 Wo pieces are younger)
        bar();
        mstamp(a.dbg, l, 12);
        if (wave < 7) {
            float acc[4];
            const unsigned char* rows = lds + OFF_A + qpair0 * 2 * 3 * 1024;
            if (wave < 2) {
                dot_rows<T, 4, 3>(acc, rows, xs, lane);
            } else {
                float a2[2];
                dot_rows<T, 2, 3>(a2, rows, xs, lane);
                acc[0] = a2[0]; acc[1] = a2[1]; acc[2] = 0.f; acc[3] = 0.f;
            }
            if (lane < qnp) {
                const float va = lane == 0 ? acc[0] : acc[2], vb = lane == 0 ? acc[1] : acc[3];
                const int n = b * 18 + 2 * (qpair0 + lane);
                const int sec = n / MD, c = n - sec * MD, h = c >> 6, d = c & 63;
                const float x0 = round_dt<T>(va), x1 = round_dt<T>(vb);
                float o0 = x0, o1 = x1;
                if (sec < 2) {
                    o0 = x0 * rcs - x1 * rsn;
                    o1 = x0 * rsn + x1 * rcs;
                }
                const st s0 = T::from_f32(o0), s1 = T::from_f32(o1);
                if (sec > 0) {
                    st* base = reinterpret_cast<st*>(sec == 1 ? a.kcache : a.vcache) + (int64_t)l * cache_l;
                    st* dst = base + ((int64_t)h * a.w_alloc + slot_cur) * 64 + d;
                    dst[0] = s0;
                    dst[1] = s1;
                }
                unsigned short u0, u1;
                __builtin_memcpy(&u0, &s0, 2);
                __builtin_memcpy(&u1, &s1, 2);
                pub[qpair0 + lane] = pack16(u0, u1);          // granule (n >> 1) = b * 9 + pair
            }
        }
        mstamp(a.dbg, l, 13);
        bar();                                                // every wave is done with the Wqkv rows: region A is free
        // One store instruction publishes the workgroup's outputs of a phase.  Measured: an edge costs ~0.3 us per store
        // instruction (of different waves / CUs) that hits the same 128-byte line -- 1.3 us for the merged attention output
        // (one instruction per line) against 4.7 us for h when each of 7 waves x 256 workgroups stored its own granule.
        if (wave == 0 && lane < 9) publish_raw(g + G_QKV, b * 9 + lane, pub[lane], tl + E_QKV);
        // loader: the W1 | W3 rows.  A CU's polls queue behind its own DMA returns (~25 GB/s per CU), so a set is requested
        // where the workgroup computes rather than where it waits for an edge: in a cache-scan workgroup half behind the
        // q | k | v gather (drains under the scan) and half behind the scan; elsewhere at once.  (Measured: all 84 pieces
        // here in every workgroup -> the q | k | v edge slows from 2.9 to 4.2 us, 737 vs 705 us / token.)
        const unsigned char* W13 = (const unsigned char*)a.w13 + ((int64_t)l * 2 * MF + (int64_t)b * 28) * MD * 2;
        const bool scans = b < MH * MNS;
        if (loader && !scans) dma_flat(lds_base + OFF_A, W13, 84, lane);      // (no barrier of theirs is due before the O edge)

        mstamp(a.dbg, l, 2);
        // ------------------------------------------------------------------ P2: cache scan per (head, split); merge per head
        if (b < MH * MNS) {
            const st* Kh = reinterpret_cast<const st*>(a.kcache) + (int64_t)l * cache_l + (int64_t)ah * a.w_alloc * 64;
            const st* Vh = reinterpret_cast<const st*>(a.vcache) + (int64_t)l * cache_l + (int64_t)ah * a.w_alloc * 64;
            // the first 128 positions of this split's range are requested before q is known
            Vec16<T> kv[4], vv[4];
            if (wave < 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = astart + wave * 8 + u * 32 + grp;
                    if (p < aend) {
                        kv[u].load(Kh + (int64_t)p * 64 + sub * 8);
                        vv[u].load(Vh + (int64_t)p * 64 + sub * 8);
                    } else {
                        kv[u].zero();
                        vv[u].zero();
                    }
                }
            }
            if (wave == 4 && !gather<T, 3, 1>(g + G_QKV + ah * 32, MD / 2, 32, tl + E_QKV, graw, lane)) fail = 1;
            bar();
            if (fail) { if (tid == 0) atomicAdd(a.err, 1u); return; }
            // all 84 W1 | W3 pieces in ONE request: drains under the scan's arithmetic and the O edge (round 6: two requests of 42,
            // the second behind the scan, were 1.5 us per token slower -- profiles/r6f_ar_dma_placement_ab.txt)
            if (loader) dma_flat(lds_base + OFF_A, W13, 84, lane);
            mstamp(a.dbg, l, 3);
            float m = -INFINITY, lsum = 0.f, o[8];
            if (wave < 4) {
                float qv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) qv[e] = graw[sub * 8 + e] * a.scale;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = 0.f;
                for (int base0 = astart + wave * 8; base0 < aend; base0 += 128) {
                    if (base0 != astart + wave * 8) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int p = base0 + u * 32 + grp;
                            if (p < aend) {
                                kv[u].load(Kh + (int64_t)p * 64 + sub * 8);
                                vv[u].load(Vh + (int64_t)p * 64 + sub * 8);
                            } else {
                                kv[u].zero();
                                vv[u].zero();
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int p = base0 + u * 32 + grp;
                        float kf[8], vf[8];
                        kv[u].to_float(kf);
                        vv[u].to_float(vf);
                        if (p == slot_cur) {                  // this token's own k / v: written in this launch, taken from the edge
#pragma unroll
                            for (int e = 0; e < 8; ++e) { kf[e] = graw[64 + sub * 8 + e]; vf[e] = graw[128 + sub * 8 + e]; }
                        }
                        float sc = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) sc = fmaf(qv[e], kf[e], sc);
                        sc += lane_xor1(sc);
                        sc += lane_xor2(sc);
                        sc += lane_xor4(sc);
                        if (p < aend) {
                            const float mn = fmaxf(m, sc);
                            const float al = expf(m - mn), pe = expf(sc - mn);
                            lsum = lsum * al + pe;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = o[e] * al + pe * vf[e];
                            m = mn;
                        }
                    }
                }
                // merge of the wave's 8 position groups: partners lane ^ 8, ^ 16, ^ 32 (attn_decode_kernel's shuffle loop)
                auto merge_with = [&](auto partner) {
                    const float m2 = partner(m), l2 = partner(lsum);
                    const float mn = fmaxf(m, m2);
                    const float w1 = (m == -INFINITY) ? 0.f : expf(m - mn);
                    const float w2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
                    lsum = lsum * w1 + l2 * w2;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float o2 = partner(o[e]);
                        o[e] = o[e] * w1 + o2 * w2;
                    }
                    m = mn;
                };
                merge_with([&](float v) { return lane_xor8(v); });
                merge_with([&](float v) { return lane_xor16(v); });
                merge_with([&](float v) { return lane_xor32(v); });
                if (grp == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) sm[wave][sub][e] = o[e];
                    sm[wave][sub][8] = m;
                    sm[wave][sub][9] = lsum;
                }
            }
            bar();
            if (wave == 0) {
                // merge of the 4 waves (attn_decode_kernel's 8-thread tail, one (d-group, element) per lane: the same sums in
                // the same order), then two store instructions publish the 66 words
                const int ms = lane >> 3, me = lane & 7;
                float mm = -INFINITY;
                for (int w = 0; w < 4; ++w) mm = fmaxf(mm, sm[w][ms][8]);
                float ll = 0.f, oo = 0.f;
                for (int w = 0; w < 4; ++w) {
                    const float mw = sm[w][ms][8];
                    const float wt = (mw == -INFINITY) ? 0.f : expf(mw - mm);
                    ll += wt * sm[w][ms][9];
                    oo += wt * sm[w][ms][me];
                }
                u64* dst = g + G_PART + ah * PART_H + asplit * M5_ATTN_PART;
                publish(dst, lane, oo, tl + E_PART);
                if (lane < 2) publish(dst, 64 + lane, lane == 0 ? mm : ll, tl + E_PART);
            }
        } else if (b < MH * MNS + MH) {
            const int hh = b - MH * MNS;
            // one sweep over the head's 8 x 66 words (the region is padded to 9 x 64 granules)
            if (wave == 4 && !gather<T, 9, 0>(g + G_PART + hh * PART_H, 64, 64, tl + E_PART, graw, lane, MNS * M5_ATTN_PART)) fail = 1;
            bar();
            if (fail) { if (tid == 0) atomicAdd(a.err, 1u); return; }
            if (wave == 0) {                                  // lane = d; the merge of gemv_stream_kernel's PRO_ATTN prologue
                float pm[8], w[8];
                float mx = -INFINITY;
#pragma unroll
                for (int s = 0; s < 8; ++s) { pm[s] = graw[s * M5_ATTN_PART + 64]; mx = fmaxf(mx, pm[s]); }
                float lt = 0.f, ov = 0.f;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    w[s] = expf(pm[s] - mx);
                    lt += w[s] * graw[s * M5_ATTN_PART + 65];
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) ov += w[s] * graw[s * M5_ATTN_PART + lane];
                const float oq = ov / lt, on = __shfl_down(oq, 1);
                if (!(lane & 1)) publish2<T>(g + G_O, hh * 32 + (lane >> 1), T::from_f32(oq), T::from_f32(on), tl + E_O);
            }
            bar();                                            // graw is the gatherer's again
        }

        mstamp(a.dbg, l, 4);
        // ------------------------------------------------------------------ P3: Wo rows -> x += .
        // The gathered values ARE the products' operand (already rounded): they go straight to xs (its last readers, P1's
        // products, are behind the post-P1 barrier), and the loader's wait sits in front of the one barrier of this phase.
        if (wave < 6 && !gather<T, 2, 2>(g + G_O + wave * 128, 64, 64, tl + E_O, xs + wave * 256, lane)) fail = 1;
        if (loader) wait_dma<63>();                           // 18 Wo + 84 W1|W3 pieces issued since: <= 63 outstanding => the Wo rows landed
        bar();
        if (fail) { if (tid == 0) atomicAdd(a.err, 1u); return; }
        mstamp(a.dbg, l, 5);
        // loader: the W2 rows that do not overlap the Wo rows (region B bytes [18 K, 42 K)), while the others run their products
        const unsigned char* W2 = (const unsigned char*)a.w2 + ((int64_t)l * MD + (int64_t)b * 6) * MF * 2;
        if (loader) dma_flat(lds_base + OFF_B + 18 * 1024, W2 + 18 * 1024, 24, lane);
        if (wave < 6) {                                       // one row per wave (the launch ran 3 waves x 2 rows: same sums)
            float acc[1];
            dot_rows<T, 1, 3>(acc, lds + OFF_B + wave * 3 * 1024, xs, lane);
            if (lane == 0) xloc[wave] = xloc[wave] + acc[0];
        }
        bar();
        if (wave == 0 && lane < 6) publish(g + G_X1, b * 6 + lane, xloc[lane], tl + E_X1);

        mstamp(a.dbg, l, 6);
        // ------------------------------------------------------------------ P4: RMSNorm -> (W1, W3) rows -> SwiGLU
        if (wave < 6) {
            const int q0 = wave * (MD / 6);
#pragma unroll
            for (int j = 0; j < MD / 384; ++j) nws[q0 + j * 64 + lane] = a.ffn_norm[(int64_t)l * MD + q0 + j * 64 + lane];
            if (!gather<T, MD / 384, 0>(g + G_X1 + q0, 64, 64, tl + E_X1, graw + q0, lane)) fail = 1;
        }
        bar();
        if (fail) { if (tid == 0) atomicAdd(a.err, 1u); return; }
        mstamp(a.dbg, l, 7);
        rms_to_xs<T, 7>(graw, nws, a.eps, xs, red[1], tid, lane, wave);
        mstamp(a.dbg, l, 14);
        if (loader) wait_dma<24>();                           // the W1 | W3 rows have landed (the 24 W2 pieces of P3 are younger)
        bar();
        mstamp(a.dbg, l, 15);
        // loader: the remaining W2 rows -> region B bytes [0, 18 K) (the Wo rows were consumed in P3)
        if (loader) dma_flat(lds_base + OFF_B, W2, 18, lane);
        if (wave < 7) {
            float acc[4];
            dot_rows<T, 4, 3>(acc, lds + OFF_A + wave * 12 * 1024, xs, lane);
            if (lane < 2) {
                float va = 0.f, vb = 0.f;
#pragma unroll
                for (int r = 0; r + 1 < 4; r += 2) if (lane == r / 2) { va = acc[r]; vb = acc[r + 1]; }
                const float x1 = round_dt<T>(va), x3 = round_dt<T>(vb);
                const float sl = round_dt<T>(silu_f(x1));
                const float hv = sl * x3, hn = __shfl_down(hv, 1);
                if (lane == 0) {
                    const typename T::storage h0 = T::from_f32(hv), h1 = T::from_f32(hn);
                    unsigned short u0, u1;
                    __builtin_memcpy(&u0, &h0, 2);
                    __builtin_memcpy(&u1, &h1, 2);
                    pub[wave] = pack16(u0, u1);
                }
            }
        }
        bar();
        if (wave == 0 && lane < 7) publish_raw(g + G_H, b * 7 + lane, pub[lane], tl + E_H);

        mstamp(a.dbg, l, 8);
        // ------------------------------------------------------------------ P5: W2 rows -> x += .
        if (wave < 7 && !gather<T, 4, 2>(g + G_H + wave * 256, 64, 64, tl + E_H, xs + wave * 512, lane)) fail = 1;   // (straight to xs, as in P3)
        if (loader) wait_dma<0>();                            // the W2 rows have landed
        bar();
        if (fail) { if (tid == 0) atomicAdd(a.err, 1u); return; }
        mstamp(a.dbg, l, 9);
        // loader: next layer's Wqkv rows -> region A (the W1 | W3 rows were consumed in P4), while the others run their products
        const bool more = l + 1 < a.layer1;
        // (requested one barrier earlier, in front of the h gather: neutral, 503.3 vs 503.7 us per token, same file)
        if (loader && more) dma_flat(lds_base + OFF_A, Wq + ((int64_t)(l + 1) * 3 * MD + (int64_t)b * 18) * MD * 2, 54, lane);
        if (wave < 6) {
            float acc[1];
            dot_rows<T, 1, 7>(acc, lds + OFF_B + wave * 7 * 1024, xs, lane);
            if (lane == 0) {
                const float x2 = xloc[wave] + acc[0];
                xloc[wave] = x2;
            }
        }
        bar();
        if (wave == 0 && lane < 6) {
            if (more) publish(g + G_X2, b * 6 + lane, xloc[lane], tl + E_X2);
            else a.xres[b * 6 + lane] = xloc[lane];
        }
        mstamp(a.dbg, l, 10);
    }
}

}  // namespace

extern "C" int m5_ar_layers_persistent(int dtype, const M5ArMegaArgs* a, void* stream) {
    if (!a || !a->wqkv || !a->wo || !a->w13 || !a->w2 || !a->attn_norm || !a->ffn_norm || !a->xres || !a->rope || !a->state ||
        !a->kcache || !a->vcache || !a->gran || !a->err)
        return M5_ERR_ARG;
    if (dtype != M5_F16 && dtype != M5_BF16) return M5_ERR_UNSUPPORTED;
    if (a->dim != MD || a->hidden != MF || a->n_heads != MH || a->layer0 < 0 || a->layer1 <= a->layer0 || a->layer1 > 31 ||
        a->window <= 0 || a->w_alloc <= 0)
        return M5_ERR_UNSUPPORTED;
    if ((((uintptr_t)a->wqkv | (uintptr_t)a->wo | (uintptr_t)a->w13 | (uintptr_t)a->w2) & 15) || ((uintptr_t)a->gran & 7)) return M5_ERR_ARG;
    // every workgroup must be resident at once: one per CU of the device the CURRENT context runs on (queried per call: the
    // attribute read is a table lookup, and a process may drive several GPUs from several threads)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return M5_ERR_LAUNCH;
    if (cus < 256) return M5_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == M5_F16) hipLaunchKernelGGL(ar_mega_kernel<F16T>, dim3(256), dim3(512), 0, s, *a);
    else hipLaunchKernelGGL(ar_mega_kernel<BF16T>, dim3(256), dim3(512), 0, s, *a);
    M5_CHECK_LAUNCH();
    return M5_OK;
}
