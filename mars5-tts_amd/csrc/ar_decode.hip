// AR decode step, batch 1: HBM-bound weight streaming.
//
// One token = 5 launches per Mistral layer + head + sampler, all reading the position /
// token state from DEVICE memory so a single captured hipGraph serves every step:
//   K1 gemv<RMS, QKV_ROPE>   RMSNorm -> Wqkv stream -> RoPE(q,k) -> q buffer + KV-cache slot
//   K2 attn_decode           q . K[0:n] softmax . V, split over keys, partial (o, m, l)
//   K3 gemv<ATTN, RESIDUAL>  combine partials -> Wo stream -> x += .
//   K4 gemv<RMS, SWIGLU>     RMSNorm -> interleaved (W1,W3) stream -> silu(a)*b
//   K5 gemv<DT, RESIDUAL>    W2 stream -> x += .
// GEMV: the activation vector lives in LDS as fp32 (<= 18 KB); every wave streams R weight
// rows with 16-byte loads per lane (1 KiB per wave instruction, fully coalesced, no LDS
// round trip for the streamed operand), fp32 accumulate, xor-shuffle reduction.
#include "common.h"
#include <mutex>

namespace {

// diagnostics (tools/ar_phase_clock.py): when set, workgroup 0 of every decode launch stamps the 100 MHz
// wall clock at its phases into dbg[slot * 8 + k]; slot advances on the host per launch
#ifdef M5_TOOLS
__device__ inline void ar_stamp(unsigned long long* d, int k) {
    if (d && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) d[k] = wall_clock64();
}
#else
__device__ inline void ar_stamp(unsigned long long*, int) {}     // product build: M5GemvArgs.dbg is ignored
#endif

// Prefetch workgroup `pid` of `pf.wgs` (appended behind a launch's compute workgroups; `lin` = its linear block id, whose
// residue mod 8 is -- observed, speed only -- its XCD): touches the chunks j = first + r + 8 k of the region whose residue
// matches, sharing them with the other prefetchers of that residue.  One dword per 64 bytes pulls the line towards L2.
__device__ inline void prefetch_wg(const M5Prefetch& pf, int pid, int npf, int lin, int nthreads) {
    const int r = lin & 7;
    const int per_res = max(npf >> 3, 1), idx = (pid >> 3) % per_res;       // npf: prefetch workgroups actually launched
    const int nk = (pf.n_chunks - r + 7) >> 3;                       // chunks of this residue
    const int kper = (nk + per_res - 1) / per_res;
    const int units = (int)(pf.chunk_bytes >> 6);
    const unsigned char* base = (const unsigned char*)pf.ptr;
    int acc = 0;
    for (int k = idx * kper; k < min((idx + 1) * kper, nk); ++k) {
        const unsigned char* cb = base + (int64_t)(pf.first_chunk + r + 8 * k) * pf.chunk_bytes;
#pragma unroll 8
        for (int u = threadIdx.x; u < units; u += nthreads) acc += *reinterpret_cast<const int*>(cb + (int64_t)u * 64);
    }
    asm volatile("" :: "v"(acc));
}

// ----------------------------------------------------------------------------- GEMV
template <typename T, int PRO, int EPI, int R>
__global__ __launch_bounds__(256) void gemv_kernel(M5GemvArgs a) {
    using st = typename T::storage;
    constexpr int EPL = T::EPL;
    extern __shared__ __attribute__((aligned(16))) float xs[];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.state && a.state[M5_ST_DONE]) return;
    const int K = a.K;

    // ---- prologue: activation vector -> LDS (fp32 values already rounded to dtype)
    if constexpr (PRO == M5_PRO_RMS) {
        float ss = 0.f;
        for (int i = tid; i < K; i += 256) { const float v = a.x_f32[i]; ss += v * v; }
        const float tot = block_sum<4>(ss, red);
        const float rstd = rsqrtf(tot / (float)K + a.eps);
        for (int i = tid; i < K; i += 256) {
            const float n = a.x_f32[i] * rstd;
            xs[i] = round_dt<T>(n * a.norm_w[i]);
        }
    } else if constexpr (PRO == M5_PRO_DT) {
        const st* x = reinterpret_cast<const st*>(a.x_dt);
        for (int i = tid; i < K; i += 256) xs[i] = T::to_f32(x[i]);
    } else {   // M5_PRO_ATTN: merge the split-KV partials of every head
        for (int i = tid; i < K; i += 256) {
            const int h = i >> 6, d = i & 63;
            const float* pp = a.part + (int64_t)h * a.nsplit * M5_ATTN_PART;
            float mx = -INFINITY;
            for (int s = 0; s < a.nsplit; ++s) mx = fmaxf(mx, pp[s * M5_ATTN_PART + 64]);
            float o = 0.f, l = 0.f;
            for (int s = 0; s < a.nsplit; ++s) {
                const float w = expf(pp[s * M5_ATTN_PART + 64] - mx);
                o += w * pp[s * M5_ATTN_PART + d];
                l += w * pp[s * M5_ATTN_PART + 65];
            }
            xs[i] = round_dt<T>(o / l);
        }
    }
    __syncthreads();

    const int row0 = (blockIdx.x * 4 + wave) * R;
    if (row0 >= a.N) return;
    const unsigned char* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        wrow[r] = (const unsigned char*)a.W + (int64_t)min(row0 + r, a.N - 1) * a.ldw * sizeof(st);
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;

#pragma unroll 2
    for (int k0 = lane * EPL; k0 < K; k0 += 64 * EPL) {
        Vec16<T> wv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wv[r].load(wrow[r] + (int64_t)k0 * sizeof(st));
        float xv[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e += 4) {
            const float4 t = *reinterpret_cast<const float4*>(xs + k0 + e);
            xv[e] = t.x; xv[e + 1] = t.y; xv[e + 2] = t.z; xv[e + 3] = t.w;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float wf[EPL];
            wv[r].to_float(wf);
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[r] = fmaf(wf[e], xv[e], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);

    // ---- epilogue (lane r2 handles the pair / row r2)
    if constexpr (EPI == M5_GEPI_RESIDUAL) {
        if (lane < R && row0 + lane < a.N) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) if (lane == r) v = acc[r];
            a.xres[row0 + lane] += v;
        }
    } else if constexpr (EPI == M5_GEPI_F32) {
        if (lane < R && row0 + lane < a.N) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) if (lane == r) v = acc[r];
            a.y_f32[row0 + lane] = v;
        }
    } else if constexpr (EPI == M5_GEPI_SWIGLU) {
        static_assert(R % 2 == 0, "pairs");
        if (lane < R / 2 && row0 + 2 * lane + 1 < a.N) {
            float va = 0.f, vb = 0.f;
#pragma unroll
            for (int r = 0; r < R; r += 2) if (lane == r / 2) { va = acc[r]; vb = acc[r + 1]; }
            const float x1 = round_dt<T>(va), x3 = round_dt<T>(vb);
            const float sl = round_dt<T>(silu_f(x1));
            reinterpret_cast<st*>(a.y_dt)[(row0 >> 1) + lane] = T::from_f32(sl * x3);
        }
    } else {   // M5_GEPI_QKV_ROPE
        static_assert(R % 2 == 0, "pairs");
        if (lane < R / 2 && row0 + 2 * lane + 1 < a.N) {
            float va = 0.f, vb = 0.f;
#pragma unroll
            for (int r = 0; r < R; r += 2) if (lane == r / 2) { va = acc[r]; vb = acc[r + 1]; }
            const int n = row0 + 2 * lane;
            const int D = a.dim;
            const int sec = n / D, c = n - sec * D, h = c >> 6, d = c & 63;
            const int pos = a.state[M5_ST_POS];
            const float x0 = round_dt<T>(va), x1 = round_dt<T>(vb);
            float o0 = x0, o1 = x1;
            if (sec < 2) {
                const float cs = a.rope[((int64_t)pos * 32 + (d >> 1)) * 2];
                const float sn = a.rope[((int64_t)pos * 32 + (d >> 1)) * 2 + 1];
                o0 = x0 * cs - x1 * sn;
                o1 = x0 * sn + x1 * cs;
            }
            st* dst;
            if (sec == 0) {
                dst = reinterpret_cast<st*>(a.qbuf) + c;
            } else {
                const int slot = pos % a.window;
                st* base = reinterpret_cast<st*>(sec == 1 ? a.kcache : a.vcache);
                dst = base + ((int64_t)h * a.w_alloc + slot) * 64 + d;
            }
            dst[0] = T::from_f32(o0);
            dst[1] = T::from_f32(o1);
        }
    }
}

template <typename T, int PRO, int EPI, int R>
int launch_gemv(const M5GemvArgs& a, hipStream_t s) {
    const int rows_per_block = 4 * R;
    dim3 grid((a.N + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL((gemv_kernel<T, PRO, EPI, R>), grid, dim3(256), (size_t)a.K * sizeof(float), s, a);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

template <typename T>
int dispatch_gemv(int pro, int epi, const M5GemvArgs& a, hipStream_t s) {
    if (pro == M5_PRO_RMS && epi == M5_GEPI_QKV_ROPE) return launch_gemv<T, M5_PRO_RMS, M5_GEPI_QKV_ROPE, 4>(a, s);
    if (pro == M5_PRO_ATTN && epi == M5_GEPI_RESIDUAL) return launch_gemv<T, M5_PRO_ATTN, M5_GEPI_RESIDUAL, 2>(a, s);
    if (pro == M5_PRO_RMS && epi == M5_GEPI_SWIGLU) return launch_gemv<T, M5_PRO_RMS, M5_GEPI_SWIGLU, 4>(a, s);
    if (pro == M5_PRO_DT && epi == M5_GEPI_RESIDUAL) return launch_gemv<T, M5_PRO_DT, M5_GEPI_RESIDUAL, 2>(a, s);
    if (pro == M5_PRO_RMS && epi == M5_GEPI_F32) return launch_gemv<T, M5_PRO_RMS, M5_GEPI_F32, 4>(a, s);
    if (pro == M5_PRO_DT && epi == M5_GEPI_F32) return launch_gemv<T, M5_PRO_DT, M5_GEPI_F32, 4>(a, s);
    return M5_ERR_UNSUPPORTED;
}


// ----------------------------------------------------------------- GEMV, streaming form
// Same arithmetic as gemv_kernel, restructured for the batch-1 decode regime where each launch
// lives for a few microseconds and what matters is how early the weight stream starts:
//   * the (tiny, L2-resident) activation loads are issued first, then the wave's WHOLE weight
//     share (R rows x NIT x 16 B per lane, non-temporal: streamed once, read by one CU) goes
//     in flight before any prologue arithmetic; the RMSNorm / attention-combine prologue runs
//     under the stream and only its LDS hand-off is waited for;
//   * NW waves x R rows per workgroup are chosen per call site so that the real model's row
//     counts give exactly one workgroup per CU (256), no second partial round.
template <typename T, int PRO, int EPI, int R, int NW, int NIT>
__global__ __launch_bounds__(NW * 64) void gemv_stream_kernel(M5GemvArgs a) {
    using st = typename T::storage;
    constexpr int EPL = 8;                       // 16-bit operands only
    constexpr int K = NIT * 64 * EPL;
    constexpr int NT = NW * 64;
    constexpr int CH = (PRO == M5_PRO_DT) ? 8 : 4;             // elements per 16-byte prologue chunk
    constexpr int NCH = K / CH, JN = (NCH + NT - 1) / NT;       // chunk c = tid + j * NT, j < JN
    static_assert(PRO != M5_PRO_ATTN || NT * 8 == K, "attention combine: 8 outputs per thread");
    __shared__ __attribute__((aligned(16))) st xs[K];            // the activation vector in the operand type (common.h: dot8)
    __shared__ float red[NW];
    __shared__ float wsm[24 * 8 + 24];           // PRO_ATTN: split weights [h][s], then l_tot[h]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {   // workgroups behind the compute grid only prefetch (see M5Prefetch)
        const int ncomp = (a.N + NW * R - 1) / (NW * R);
        if ((int)blockIdx.x >= ncomp) { prefetch_wg(a.pf, blockIdx.x - ncomp, gridDim.x - ncomp, blockIdx.x, NT); return; }
    }
    unsigned long long* dbg = a.dbg;
    ar_stamp(dbg, 0);
    const int done = a.state ? a.state[M5_ST_DONE] : 0;     // consumed only before the epilogue's writes

    // ---- activation loads first (short queue in front of the weight stream)
    float4 xin[JN], nwv[JN];
    uint4 xraw[JN];
    if constexpr (PRO == M5_PRO_RMS) {
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int c = min(tid + j * NT, NCH - 1);
            xin[j] = *reinterpret_cast<const float4*>(a.x_f32 + c * 4);
            nwv[j] = *reinterpret_cast<const float4*>(a.norm_w + c * 4);
        }
    } else if constexpr (PRO == M5_PRO_DT) {
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int c = min(tid + j * NT, NCH - 1);
            xraw[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const st*>(a.x_dt) + c * 8);
        }
    }
    float pm = 0.f, pl = 0.f;
    if constexpr (PRO == M5_PRO_ATTN) {
        if (tid < a.n_heads * 8) {                 // thread = (head, split)
            pm = a.part[(int64_t)tid * M5_ATTN_PART + 64];
            pl = a.part[(int64_t)tid * M5_ATTN_PART + 65];
        }
    }

    // ---- epilogue operands that depend only on the row / position: issued now, consumed after the
    // weight stream, so their L2 / HBM round trip is off the tail of the launch
    const int row0 = (blockIdx.x * NW + wave) * R;
    float xold = 0.f, rcs = 1.f, rsn = 0.f;
    int pos_e = 0;
    if constexpr (EPI == M5_GEPI_RESIDUAL) {
        if (lane < R && row0 + lane < a.N) xold = a.xres[row0 + lane];
    } else if constexpr (EPI == M5_GEPI_QKV_ROPE) {
        pos_e = a.state[M5_ST_POS];
        const int n = row0 + 2 * lane;
        if (lane < R / 2 && n + 1 < a.N && n < 2 * a.dim) {
            const int d = (n % a.dim) & 63;
            rcs = a.rope[((int64_t)pos_e * 32 + (d >> 1)) * 2];
            rsn = a.rope[((int64_t)pos_e * 32 + (d >> 1)) * 2 + 1];
        }
    }

    // ---- the wave's whole weight share in flight
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 wv[R][NIT];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned char* wr = (const unsigned char*)a.W + (int64_t)min(row0 + r, a.N - 1) * a.ldw * 2 + lane * 16;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            wv[r][it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr + it * 1024));
    }

    ar_stamp(dbg, 1);
    // ---- prologue under the stream: activation vector -> LDS (fp32 values already rounded to dtype)
    if constexpr (PRO == M5_PRO_RMS) {
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < JN; ++j)
            if (tid + j * NT < NCH) ss += xin[j].x * xin[j].x + xin[j].y * xin[j].y + xin[j].z * xin[j].z + xin[j].w * xin[j].w;
        const float tot = block_sum<NW>(ss, red);
        const float rstd = rsqrtf(tot / (float)K + a.eps);
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int c = tid + j * NT;
            if (c < NCH) {
                *reinterpret_cast<uint2*>(xs + c * 4) = m5_pack4<T>((xin[j].x * rstd) * nwv[j].x, (xin[j].y * rstd) * nwv[j].y,
                                                                    (xin[j].z * rstd) * nwv[j].z, (xin[j].w * rstd) * nwv[j].w);
            }
        }
    } else if constexpr (PRO == M5_PRO_DT) {
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const int c = tid + j * NT;
            if (c < NCH) *reinterpret_cast<uint4*>(xs + c * 8) = xraw[j];       // already in the operand type
        }
    } else {
        constexpr int PER = 8;
        // merge the 8 split-KV partials of every head: weights per (head, split), then 8 outputs / thread
        const int H = a.n_heads;
        if (tid < H * 8) {
            float mx = pm;
            mx = fmaxf(mx, __shfl_xor(mx, 1));
            mx = fmaxf(mx, __shfl_xor(mx, 2));
            mx = fmaxf(mx, __shfl_xor(mx, 4));
            const float w = expf(pm - mx);
            wsm[tid] = w;
            float l = 0.f;                          // sequential over splits, like the generic kernel
#pragma unroll
            for (int s = 0; s < 8; ++s) l += __shfl(w, (lane & ~7) + s) * __shfl(pl, (lane & ~7) + s);
            if ((tid & 7) == 0) wsm[H * 8 + (tid >> 3)] = l;
        }
        __syncthreads();
        const int i0 = tid * PER, h = i0 >> 6, d = i0 & 63;
        const float* pp = a.part + (int64_t)h * 8 * M5_ATTN_PART + d;
        float4 ov[8][PER / 4];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int e = 0; e < PER / 4; ++e) ov[s][e] = *reinterpret_cast<const float4*>(pp + s * M5_ATTN_PART + 4 * e);
        float o[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) o[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float w = wsm[h * 8 + s];
#pragma unroll
            for (int e = 0; e < PER / 4; ++e) {
                o[4 * e] += w * ov[s][e].x; o[4 * e + 1] += w * ov[s][e].y;
                o[4 * e + 2] += w * ov[s][e].z; o[4 * e + 3] += w * ov[s][e].w;
            }
        }
        const float l = wsm[H * 8 + h];
        {
            const uint2 lo = m5_pack4<T>(o[0] / l, o[1] / l, o[2] / l, o[3] / l), hi = m5_pack4<T>(o[4] / l, o[5] / l, o[6] / l, o[7] / l);
            *reinterpret_cast<uint4*>(xs + i0) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
    __syncthreads();
    ar_stamp(dbg, 2);
    if (row0 >= a.N || done) return;

    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k0 = (it * 64 + lane) * EPL;
        const uint4 xv = *reinterpret_cast<const uint4*>(xs + k0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint4 wr = make_uint4(wv[r][it][0], wv[r][it][1], wv[r][it][2], wv[r][it][3]);
            acc[r] = dot8<T>(wr, xv, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
    ar_stamp(dbg, 3);

    if constexpr (EPI == M5_GEPI_RESIDUAL) {
        if (lane < R && row0 + lane < a.N) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) if (lane == r) v = acc[r];
            a.xres[row0 + lane] = xold + v;
        }
    } else if constexpr (EPI == M5_GEPI_F32) {
        if (lane < R && row0 + lane < a.N) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) if (lane == r) v = acc[r];
            a.y_f32[row0 + lane] = v;
        }
    } else if constexpr (EPI == M5_GEPI_SWIGLU) {
        static_assert(EPI != M5_GEPI_SWIGLU || R % 2 == 0, "pairs");
        if (lane < R / 2 && row0 + 2 * lane + 1 < a.N) {
            float va = 0.f, vb = 0.f;
#pragma unroll
            for (int r = 0; r + 1 < R; r += 2) if (lane == r / 2) { va = acc[r]; vb = acc[r + 1]; }
            const float x1 = round_dt<T>(va), x3 = round_dt<T>(vb);
            const float sl = round_dt<T>(silu_f(x1));
            reinterpret_cast<st*>(a.y_dt)[(row0 >> 1) + lane] = T::from_f32(sl * x3);
        }
    } else {   // M5_GEPI_QKV_ROPE
        static_assert(EPI != M5_GEPI_QKV_ROPE || R % 2 == 0, "pairs");
        if (lane < R / 2 && row0 + 2 * lane + 1 < a.N) {
            float va = 0.f, vb = 0.f;
#pragma unroll
            for (int r = 0; r + 1 < R; r += 2) if (lane == r / 2) { va = acc[r]; vb = acc[r + 1]; }
            const int n = row0 + 2 * lane;
            const int D = a.dim;
            const int sec = n / D, c = n - sec * D, h = c >> 6, d = c & 63;
            const int pos = pos_e;
            const float x0 = round_dt<T>(va), x1 = round_dt<T>(vb);
            float o0 = x0, o1 = x1;
            if (sec < 2) {
                o0 = x0 * rcs - x1 * rsn;
                o1 = x0 * rsn + x1 * rcs;
            }
            st* dst;
            if (sec == 0) {
                dst = reinterpret_cast<st*>(a.qbuf) + c;
            } else {
                const int slot = pos % a.window;
                st* base = reinterpret_cast<st*>(sec == 1 ? a.kcache : a.vcache);
                dst = base + ((int64_t)h * a.w_alloc + slot) * 64 + d;
            }
            dst[0] = T::from_f32(o0);
            dst[1] = T::from_f32(o1);
        }
    }
}

template <typename T, int PRO, int EPI, int R, int NW, int NIT>
int launch_gemv_stream(const M5GemvArgs& a, hipStream_t s) {
    int ncomp = (a.N + NW * R - 1) / (NW * R);
    const bool pf = a.pf.ptr && a.pf.wgs > 0 && a.pf.n_chunks > 0 && (a.pf.chunk_bytes % 64) == 0 && (a.pf.first_chunk % 8) == 0;
    // the prefetchers' residues must line up with the block ids: pad the compute grid's count to a multiple of 8 (no-op here)
    dim3 grid(ncomp + ((pf && ncomp % 8 == 0) ? a.pf.wgs / 8 * 8 : 0));
    hipLaunchKernelGGL((gemv_stream_kernel<T, PRO, EPI, R, NW, NIT>), grid, dim3(NW * 64), 0, s, a);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

// Streaming-form call sites of the real CodecLM geometry (dim 1536, hidden 3584); anything else
// (tiny test models, fp32 parity mode) takes the generic kernel.  Returns 1 when not handled.
template <typename T>
int dispatch_gemv_stream(int pro, int epi, const M5GemvArgs& a, hipStream_t s) {
    const bool al = (((uintptr_t)a.x_f32 | (uintptr_t)a.norm_w | (uintptr_t)a.x_dt | (uintptr_t)a.part) & 15) == 0;
    if (!al || (a.ldw % 8)) return 1;
    if (a.K == 1536) {
        if (pro == M5_PRO_RMS && epi == M5_GEPI_QKV_ROPE) return launch_gemv_stream<T, M5_PRO_RMS, M5_GEPI_QKV_ROPE, 6, 3, 3>(a, s);
        if (pro == M5_PRO_ATTN && epi == M5_GEPI_RESIDUAL && a.nsplit == 8 && a.n_heads == 24)
            return launch_gemv_stream<T, M5_PRO_ATTN, M5_GEPI_RESIDUAL, 2, 3, 3>(a, s);
        if (pro == M5_PRO_RMS && epi == M5_GEPI_SWIGLU) return launch_gemv_stream<T, M5_PRO_RMS, M5_GEPI_SWIGLU, 4, 7, 3>(a, s);
        if (pro == M5_PRO_RMS && epi == M5_GEPI_F32) return launch_gemv_stream<T, M5_PRO_RMS, M5_GEPI_F32, 4, 4, 3>(a, s);
    } else if (a.K == 3584) {
        if (pro == M5_PRO_DT && epi == M5_GEPI_RESIDUAL) return launch_gemv_stream<T, M5_PRO_DT, M5_GEPI_RESIDUAL, 1, 6, 7>(a, s);
    }
    return 1;
}

// ------------------------------------------------------------------- decode attention
// grid (H, nsplit); 4 waves; LPP lanes cover one cached position (16 B each), so a wave
// instruction reads 64/LPP consecutive positions = 1 KiB contiguous of this head's K (or V).
template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(M5AttnDecodeArgs a) {
    using st = typename T::storage;
    constexpr int EPL = T::EPL;
    constexpr int LPP = 64 / EPL;          // lanes per position: 8 (16-bit) or 16 (f32)
    constexpr int PPW = 64 / LPP;          // positions per wave instruction
    __shared__ float sm[4][LPP][EPL + 2];
    {   // batched decode: blockIdx.z = sequence (all strides are 0 for a single sequence)
        const int64_t b = blockIdx.z;
        a.state += b * a.state_bs;
        a.qbuf = reinterpret_cast<const st*>(a.qbuf) + b * a.q_bs;
        a.kcache = reinterpret_cast<const st*>(a.kcache) + b * a.cache_bs;
        a.vcache = reinterpret_cast<const st*>(a.vcache) + b * a.cache_bs;
        a.part += b * a.part_bs;
    }
    if ((int)blockIdx.y >= a.nsplit) {             // appended prefetch workgroups (batch 1): see M5Prefetch
        const int pid = (blockIdx.y - a.nsplit) * gridDim.x + blockIdx.x;
        prefetch_wg(a.pf, pid, (gridDim.y - a.nsplit) * gridDim.x, blockIdx.y * gridDim.x + blockIdx.x, 256);
        return;
    }
    const int done = a.state[M5_ST_DONE];          // consumed before the only global write (a finished sequence just idles)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, split = blockIdx.y;
    const int pos = a.state[M5_ST_POS];
    const int n_valid = min(pos + 1, a.window);
    int chunk = (n_valid + a.nsplit - 1) / a.nsplit;
    chunk = (chunk + PPW * 4 - 1) / (PPW * 4) * (PPW * 4);
    const int start = split * chunk, end = min(n_valid, start + chunk);
    const int sub = lane % LPP, grp = lane / LPP;

    float qv[EPL];
    {
        Vec16<T> q;
        q.load(reinterpret_cast<const st*>(a.qbuf) + h * 64 + sub * EPL);
        q.to_float(qv);
#pragma unroll
        for (int e = 0; e < EPL; ++e) qv[e] *= a.scale;
    }
    const st* Kh = reinterpret_cast<const st*>(a.kcache) + (int64_t)h * a.w_alloc * 64;
    const st* Vh = reinterpret_cast<const st*>(a.vcache) + (int64_t)h * a.w_alloc * 64;

    float m = -INFINITY, l = 0.f, o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = 0.f;

    // The key range of a workgroup is short (<= ~47 positions per wave at the 3000-slot window), and one
    // launch lives for a few microseconds: issue the loads of UN wave-iterations before touching any of
    // them, so the K / V round trips overlap instead of queueing behind each other's softmax updates.
    constexpr int UN = 4;
    for (int base0 = start + wave * PPW; base0 < end; base0 += UN * 4 * PPW) {
        Vec16<T> kv[UN], vv[UN];
        bool okv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int p = base0 + u * 4 * PPW + grp;
            okv[u] = p < end;
            if (okv[u]) {
                kv[u].load(Kh + (int64_t)p * 64 + sub * EPL);
                vv[u].load(Vh + (int64_t)p * 64 + sub * EPL);
            } else {
                kv[u].zero();
                vv[u].zero();
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float kf[EPL], vf[EPL];
            kv[u].to_float(kf);
            vv[u].to_float(vf);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) s = fmaf(qv[e], kf[e], s);
#pragma unroll
            for (int off = 1; off < LPP; off <<= 1) s += __shfl_xor(s, off);
            if (okv[u]) {
                const float mn = fmaxf(m, s);
                const float al = expf(m - mn), pe = expf(s - mn);
                l = l * al + pe;
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[e] = o[e] * al + pe * vf[e];
                m = mn;
            }
        }
    }
    // merge the PPW position groups of the wave
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
        const float mn = fmaxf(m, m2);
        const float w1 = (m == -INFINITY) ? 0.f : expf(m - mn);
        const float w2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
        l = l * w1 + l2 * w2;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float o2 = __shfl_xor(o[e], off);
            o[e] = o[e] * w1 + o2 * w2;
        }
        m = mn;
    }
    if (grp == 0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) sm[wave][sub][e] = o[e];
        sm[wave][sub][EPL] = m;
        sm[wave][sub][EPL + 1] = l;
    }
    __syncthreads();
    if (tid < LPP && !done) {
        float mm = -INFINITY;
        for (int w = 0; w < 4; ++w) mm = fmaxf(mm, sm[w][tid][EPL]);
        float ll = 0.f, oo[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) oo[e] = 0.f;
        for (int w = 0; w < 4; ++w) {
            const float mw = sm[w][tid][EPL];
            const float wt = (mw == -INFINITY) ? 0.f : expf(mw - mm);
            ll += wt * sm[w][tid][EPL + 1];
#pragma unroll
            for (int e = 0; e < EPL; ++e) oo[e] += wt * sm[w][tid][e];
        }
        float* dst = a.part + ((int64_t)h * a.nsplit + split) * M5_ATTN_PART;
#pragma unroll
        for (int e = 0; e < EPL; ++e) dst[tid * EPL + e] = oo[e];
        if (tid == 0) { dst[64] = mm; dst[65] = ll; }
    }
}

// ----------------------------------------------------------------------------- sampler
__device__ inline uint32_t desc_key(float f) {
    // monotone map: larger float -> SMALLER key (so ascending key order = descending value)
    uint32_t u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending-order key
    return ~u;
}
__device__ inline float key_to_float(uint32_t k) {
    uint32_t u = ~k;
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

// in-LDS bitonic sort of n2 (power of two) 64-bit keys, ascending; 1024 threads; ends with a barrier
__device__ inline void bitonic_sort_u64(unsigned long long* sk, int n2, int tid) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += 1024) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const unsigned long long x = sk[i], y = sk[l];
                const bool asc = (i & k) == 0;
                if ((x > y) == asc) { sk[i] = y; sk[l] = x; }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(1024) void sample_kernel(M5SampleArgs a, int V2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* sk = reinterpret_cast<unsigned long long*>(smem);            // V2 sort keys
    float* fs = reinterpret_cast<float*>(smem + (size_t)V2 * 8);                     // V2 floats scratch
    int* cnt = reinterpret_cast<int*>(fs);                                           // alias (used before fs)
    __shared__ float red[16];
    __shared__ int sh_i[4];
    __shared__ float sh_f[2];
    {   // batched decode: blockIdx.x = sequence (all strides are 0 for a single sequence)
        const int64_t b = blockIdx.x;
        a.logits += b * a.logits_bs;
        a.state += b * a.state_bs;
        a.tokens += b * a.tokens_bs;
        if (a.noise) a.noise += b * a.noise_bs;
        if (a.rng) a.rng += b * a.rng_bs;
        a.xres += b * a.xres_bs;
        if (a.eos_table) a.eos_table += b * a.eos_table_bs;
        if (a.n_est_b) a.n_est = a.n_est_b[b];
        if (a.max_len_b) a.max_len = a.max_len_b[b];
    }
    int32_t* st = a.state;
    if (st[M5_ST_DONE]) return;
    const int tid = threadIdx.x;
    const int V = a.V;
    const int n_gen = st[M5_ST_NGEN];
    const int n_tok = st[M5_ST_NTOK];

    // ---- 1. frequency / presence penalty over the last `window` generated ids (samplers.py:20-36)
    const bool pen = n_gen > 1;
    if (pen) {
        for (int i = tid; i < V; i += 1024) cnt[i] = 0;
        __syncthreads();
        const int w = min(n_gen, a.penalty_window);
        for (int j = tid; j < w; j += 1024) atomicAdd(&cnt[(int)a.tokens[n_tok - 1 - j]], 1);
        __syncthreads();
    }
    // ---- 2-5. mask, EOS penalty, temperature; build sort keys
    const float invT = 1.0f / a.temperature;
    for (int i = tid; i < V2; i += 1024) {
        unsigned long long key = 0xffffffffffffffffull;
        if (i < V) {
            float z = a.logits[i];
            if (pen) {
                const int c = cnt[i];
                z = (z - (float)c * a.alpha_frequency) - (c > 0 ? 1.0f : 0.0f) * a.alpha_presence;
            }
            if (i < a.n_text - 1) z = -INFINITY;
            if (a.eos_table && i == a.eos_idx && n_gen <= a.n_est) z = z - a.eos_table[n_gen];
            z = a.div_mode ? z * invT : z / a.temperature;
            key = ((unsigned long long)desc_key(z) << 32) | (unsigned)i;
        }
        sk[i] = key;
    }
    __syncthreads();
    // ---- 6. top-k: threshold = k-th largest, keep everything >= it (samplers.py:70-74).
    // Fast path (small k): a 4-pass radix select finds the threshold key, the survivors are compacted and
    // rank-sorted into sk[0..n) - the same (value desc, index asc) order the full sort produces, so everything
    // downstream (top-p cumulative sums included) sees identical data. Falls back to the full bitonic sort when
    // top-k is off or the tie set at the threshold is large.
    __shared__ int hist[256];
    __shared__ uint32_t sel[2];
    bool selected = false;
    const int cap = min(V2 >> 1, 1024);                                   // tmp lives in the fs scratch (V2 * 4 B)
    if (a.top_k > 0 && min(max(a.top_k, 1), V) <= 256) {
        uint32_t prefix = 0;
        int rank = min(max(a.top_k, 1), V) - 1;
        for (int pass = 3; pass >= 0; --pass) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const int shv = pass * 8;
            for (int i = tid; i < V; i += 1024) {
                const uint32_t hi = (uint32_t)(sk[i] >> 32);
                const bool match = (pass == 3) || ((hi >> (shv + 8)) == prefix);
                if (match) atomicAdd(&hist[(hi >> shv) & 255u], 1);
            }
            __syncthreads();
            if (tid < 64) {
                int hs[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) hs[q] = hist[4 * tid + q];
                const int s4 = hs[0] + hs[1] + hs[2] + hs[3];
                int inc = s4;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int n = __shfl_up(inc, off);
                    if (tid >= off) inc += n;
                }
                int base = inc - s4;                                      // entries in bins below 4*tid
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (rank >= base && rank < base + hs[q]) {
                        sel[0] = (prefix << 8) | (uint32_t)(4 * tid + q);
                        sel[1] = (uint32_t)(rank - base);
                    }
                    base += hs[q];
                }
            }
            __syncthreads();
            prefix = sel[0];
            rank = (int)sel[1];
        }
        const uint32_t thr = prefix;                                      // key of the k-th largest value
        unsigned long long* tmp = reinterpret_cast<unsigned long long*>(fs);
        if (tid == 0) sh_i[2] = 0;
        __syncthreads();
        for (int i = tid; i < V; i += 1024) {
            const unsigned long long key = sk[i];
            if ((uint32_t)(key >> 32) <= thr) {
                const int slot = atomicAdd(&sh_i[2], 1);
                if (slot < cap) tmp[slot] = key;
            }
        }
        __syncthreads();
        const int nk = sh_i[2];
        if (nk <= cap) {                                                  // block-uniform
            for (int j = tid; j < nk; j += 1024) {
                const unsigned long long my = tmp[j];
                int r = 0;
                for (int i = 0; i < nk; ++i) r += (tmp[i] < my) ? 1 : 0;  // keys are unique -> a permutation
                sk[r] = my;
            }
            if (tid == 0) sh_i[0] = nk;
            selected = true;
            __syncthreads();
        }
    }
    if (!selected) {
        // ---- bitonic sort ascending on (desc_key, index): value descending, index ascending
        bitonic_sort_u64(sk, V2, tid);
        if (tid == 0) sh_i[0] = V;
        __syncthreads();
        if (a.top_k > 0) {
            const int k = min(max(a.top_k, 1), V);
            const uint32_t thr = (uint32_t)(sk[k - 1] >> 32);
            for (int j = tid; j < V; j += 1024) {
                const uint32_t hj = (uint32_t)(sk[j] >> 32);
                const uint32_t hn = (j + 1 < V2) ? (uint32_t)(sk[j + 1] >> 32) : 0xffffffffu;
                if (hj == thr && hn != thr) sh_i[0] = j + 1;     // last entry equal to the threshold
            }
            __syncthreads();
        }
    }
    int n_keep = sh_i[0];
    const float v0 = key_to_float((uint32_t)(sk[0] >> 32));
    // ---- 7. top-p over the sorted list (samplers.py:76-91)
    if (a.top_p < 1.0f) {
        float part = 0.f;
        for (int j = tid; j < V2; j += 1024) {
            float e = 0.f;
            if (j < n_keep) e = expf(key_to_float((uint32_t)(sk[j] >> 32)) - v0);
            fs[j] = e;
            part += e;
        }
        const float S = block_sum<16>(part, red);
        // inclusive scan of p_j = e_j / S : contiguous chunk per thread + scan of thread totals
        const int per = V2 >> 10 ? V2 >> 10 : 1;
        const int j0 = tid * per;
        float loc = 0.f;
        if (j0 < V2)
            for (int q = 0; q < per; ++q) { const float pj = fs[j0 + q] / S; loc += pj; fs[j0 + q] = loc; }
        // scan thread totals: wave inclusive scan + wave offsets
        float inc = loc;
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float n = __shfl_up(inc, off);
            if (lane >= off) inc += n;
        }
        __syncthreads();
        if (lane == 63) red[wv] = inc;
        __syncthreads();
        float woff = 0.f;
        for (int w = 0; w < wv; ++w) woff += red[w];
        const float excl = woff + inc - loc;
        if (j0 < V2)
            for (int q = 0; q < per; ++q) fs[j0 + q] += excl;
        __syncthreads();
        // keep j iff j == 0 or cum[j-1] <= top_p ; cum is non-decreasing -> count
        int mycnt = 0;
        for (int j = tid; j < n_keep; j += 1024) mycnt += (j == 0 || !(fs[j - 1] > a.top_p)) ? 1 : 0;
        const int nk2 = (int)(block_sum<16>((float)mycnt, red) + 0.5f);
        n_keep = min(n_keep, nk2);
    }
    __syncthreads();
    float v0f = v0;
    // ---- 8. typical-p (samplers.py:96-122; a no-op at the reference default 1.0): over the kept set,
    // normalized = log_softmax, ent = -sum p log p, keep the tokens whose |-log p - ent| is within the
    // smallest-deviation prefix of cumulative mass `typical_p` (ties with the threshold kept).
    if (a.typical_p <= 0.999f) {
        float* fv = fs + V2;                                                 // values of the kept set
        int* fid = reinterpret_cast<int*>(fs + 2 * V2);                      // ids of the kept set
        float pt = 0.f;
        for (int j = tid; j < n_keep; j += 1024) pt += expf(key_to_float((uint32_t)(sk[j] >> 32)) - v0);
        const float St = block_sum<16>(pt, red);
        const float logSt = logf(St);
        float et = 0.f;
        for (int j = tid; j < n_keep; j += 1024) {
            const float nl = (key_to_float((uint32_t)(sk[j] >> 32)) - v0) - logSt;
            if (nl > -INFINITY) et += nl * expf(nl);            // masked (-inf) entries: 0 * -inf, dropped like nansum does
        }
        const float ent = -block_sum<16>(et, red);
        for (int j = tid; j < V2; j += 1024) {
            unsigned long long key = 0xffffffffffffffffull;
            if (j < n_keep) {
                const float v = key_to_float((uint32_t)(sk[j] >> 32));
                const float nl = (v - v0) - logSt;
                const float shifted = fabsf((-nl) - ent);
                fv[j] = v;
                fid[j] = (int)(sk[j] & 0xffffffffu);
                key = ((unsigned long long)__float_as_uint(shifted) << 32) | (unsigned)j;   // shifted >= 0: bit order = value order
            }
            sk[j] = key;                           // own slot only: read above, written here
        }
        __syncthreads();
        bitonic_sort_u64(sk, V2, tid);             // ascending deviation
        // cumulative probability in that order
        float part_t = 0.f;
        for (int j = tid; j < V2; j += 1024) {
            float e = 0.f;
            if (j < n_keep) e = expf((fv[(int)(sk[j] & 0xffffffffu)] - v0) - logSt);
            fs[j] = e;
            part_t += e;
        }
        __syncthreads();
        const int per = V2 >> 10 ? V2 >> 10 : 1;
        const int j0 = tid * per;
        float loc = 0.f;
        if (j0 < V2)
            for (int q = 0; q < per; ++q) { loc += fs[j0 + q]; fs[j0 + q] = loc; }
        float inc = loc;
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float n = __shfl_up(inc, off);
            if (lane >= off) inc += n;
        }
        __syncthreads();
        if (lane == 63) red[wv] = inc;
        __syncthreads();
        float woff = 0.f;
        for (int w = 0; w < wv; ++w) woff += red[w];
        const float excl = woff + inc - loc;
        if (j0 < V2)
            for (int q = 0; q < per; ++q) fs[j0 + q] += excl;
        __syncthreads();
        int cnt_lt = 0;
        for (int j = tid; j < n_keep; j += 1024) cnt_lt += (fs[j] < a.typical_p) ? 1 : 0;
        int last = (int)(block_sum<16>((float)cnt_lt, red) + 0.5f);
        last = min(last, n_keep - 1);
        const uint32_t thr = (uint32_t)(sk[last] >> 32);
        int cnt_keep = 0;
        for (int j = tid; j < n_keep; j += 1024) cnt_keep += ((uint32_t)(sk[j] >> 32) <= thr) ? 1 : 0;   // sorted: a prefix
        const int n_keep2 = (int)(block_sum<16>((float)cnt_keep, red) + 0.5f);
        // survivors back into (desc_key(value), id) form for the draw; their maximum may have changed
        float mx = -INFINITY;
        unsigned long long mine[8];
        int nm = 0;
        for (int j = tid; j < n_keep2 && nm < 8; j += 1024) {
            const int src = (int)(sk[j] & 0xffffffffu);
            mine[nm++] = ((unsigned long long)desc_key(fv[src]) << 32) | (unsigned)fid[src];
            mx = fmaxf(mx, fv[src]);
        }
        __syncthreads();
        nm = 0;
        for (int j = tid; j < n_keep2 && nm < 8; j += 1024) sk[j] = mine[nm++];
        v0f = block_max<16>(mx, red);
        n_keep = n_keep2;
        __syncthreads();
    }
    // ---- 9. log_softmax over the kept set, p / q, argmax (ar_generate.py:102,115)
    float part = 0.f;
    for (int j = tid; j < n_keep; j += 1024) part += expf(key_to_float((uint32_t)(sk[j] >> 32)) - v0f);
    const float S2 = block_sum<16>(part, red);
    const float logS = logf(S2);
    const float* q = a.noise ? a.noise + (int64_t)n_gen * a.noise_stride : nullptr;
    // noise = NULL: the Exp(1) value of a kept token comes straight from torch's Philox stream for this sampler call (philox.h)
    const unsigned long long nseed = q ? 0ull : a.rng[0], noff = q ? 0ull : a.rng[1] + (unsigned long long)n_gen * a.noise_inc;
    float best = -1.f;
    int besti = 0x7fffffff;
    for (int j = tid; j < n_keep; j += 1024) {
        const float v = key_to_float((uint32_t)(sk[j] >> 32));
        const int id = (int)(sk[j] & 0xffffffffu);
        const float pz = expf((v - v0f) - logS);
        const float qv = q ? q[id] : m5_torch_exponential1(m5_torch_draw_bits(nseed, noff, (unsigned long long)id, a.noise_grid));
        const float sc = pz / qv;
        if (sc > best || (sc == best && id < besti)) { best = sc; besti = id; }
    }
    // block argmax (value desc, index asc)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(besti, off);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    __shared__ float wb[16];
    __shared__ int wi[16];
    if ((tid & 63) == 0) { wb[tid >> 6] = best; wi[tid >> 6] = besti; }
    __syncthreads();
    if (tid == 0) {
        float b = wb[0];
        int bi = wi[0];
        for (int w = 1; w < 16; ++w)
            if (wb[w] > b || (wb[w] == b && wi[w] < bi)) { b = wb[w]; bi = wi[w]; }
        sh_i[1] = bi;
    }
    __syncthreads();
    const int tok = sh_i[1];
    // ---- 10. EOS / append / max_len (ar_generate.py:62,121-157) and next-step embedding
    if (tok == a.eos_idx) {
        if (tid == 0) { st[M5_ST_DONE] = 1; st[M5_ST_LAST] = tok; }
        return;
    }
    for (int i = tid; i < a.dim; i += 1024) a.xres[i] = a.embed[(int64_t)tok * a.dim + i];
    if (tid == 0) {
        a.tokens[n_tok] = tok;
        st[M5_ST_NTOK] = n_tok + 1;
        st[M5_ST_NGEN] = n_gen + 1;
        st[M5_ST_POS] = st[M5_ST_POS] + 1;
        st[M5_ST_LAST] = tok;
        if (n_tok + 1 >= a.max_len) st[M5_ST_DONE] = 1;
    }
}

}  // namespace

extern "C" int m5_ar_gemv(int dtype, int pro, int epi, const M5GemvArgs* a, void* stream) {
    if (!a || !a->W || a->N <= 0 || a->K <= 0) return M5_ERR_ARG;
    const int es = (dtype == M5_F32) ? 4 : 2, al = 16 / es;
    if (a->K % al || a->ldw % al || ((uintptr_t)a->W & 15)) return M5_ERR_ARG;
    if ((size_t)a->K * 4 > 64 * 1024) return M5_ERR_UNSUPPORTED;
    if (pro == M5_PRO_ATTN && (a->K != a->n_heads * 64 || !a->part || a->nsplit <= 0)) return M5_ERR_ARG;
    if (epi == M5_GEPI_QKV_ROPE && (!a->rope || !a->state || !a->kcache || !a->vcache || !a->qbuf || a->N != 3 * a->dim || a->dim % 64)) return M5_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case M5_F32: return dispatch_gemv<F32T>(pro, epi, *a, s);
        case M5_F16: { const int r = dispatch_gemv_stream<F16T>(pro, epi, *a, s); return r == 1 ? dispatch_gemv<F16T>(pro, epi, *a, s) : r; }
        case M5_BF16: { const int r = dispatch_gemv_stream<BF16T>(pro, epi, *a, s); return r == 1 ? dispatch_gemv<BF16T>(pro, epi, *a, s) : r; }
        default: return M5_ERR_ARG;
    }
}

extern "C" int m5_ar_attn_decode(int dtype, const M5AttnDecodeArgs* a, void* stream) {
    if (!a || !a->qbuf || !a->kcache || !a->vcache || !a->part || !a->state || a->n_heads <= 0 || a->nsplit <= 0) return M5_ERR_ARG;
    dim3 grid(a->n_heads, a->nsplit, a->batch > 1 ? a->batch : 1);
    if (a->batch <= 1 && a->pf.ptr && a->pf.wgs > 0 && a->pf.n_chunks > 0 && (a->pf.chunk_bytes % 64) == 0 && (a->pf.first_chunk % 8) == 0 &&
        (a->n_heads % 8) == 0)
        grid.y += (a->pf.wgs + a->n_heads - 1) / a->n_heads;          // whole rows of n_heads prefetch workgroups
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case M5_F32: hipLaunchKernelGGL(attn_decode_kernel<F32T>, grid, dim3(256), 0, s, *a); break;
        case M5_F16: hipLaunchKernelGGL(attn_decode_kernel<F16T>, grid, dim3(256), 0, s, *a); break;
        case M5_BF16: hipLaunchKernelGGL(attn_decode_kernel<BF16T>, grid, dim3(256), 0, s, *a); break;
        default: return M5_ERR_ARG;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}

extern "C" int m5_ar_sample(const M5SampleArgs* a, void* stream) {
    if (!a || !a->logits || !a->state || !a->tokens || !a->embed || !a->xres || a->V <= 1) return M5_ERR_ARG;
    if (!a->noise && (!a->rng || a->noise_grid == 0 || (a->noise_grid % 256) || (a->noise_inc % 4) || a->noise_inc == 0)) return M5_ERR_ARG;
    if (a->V > 8192) return M5_ERR_UNSUPPORTED;
    if (!(a->temperature > 0.f)) return M5_ERR_ARG;
    int V2 = 1024;
    while (V2 < a->V) V2 <<= 1;
    if (a->typical_p <= 0.999f && V2 > 4096) return M5_ERR_UNSUPPORTED;      // typical-p scratch: 20 B per slot
    const size_t sm = (size_t)V2 * (a->typical_p <= 0.999f ? 20 : 12);
    static std::once_flag attr_once;          // thread-safe: several host threads may drive their own streams
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute((const void*)sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 12);   // = 4096 * 24 >= 4096 * 20
    });
    hipLaunchKernelGGL(sample_kernel, dim3(a->batch > 1 ? a->batch : 1), dim3(1024), sm, (hipStream_t)stream, *a, V2);
    M5_CHECK_LAUNCH();
    return M5_OK;
}
