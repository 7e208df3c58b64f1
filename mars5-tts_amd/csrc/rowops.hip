// Row-wise HBM-bound kernels: LayerNorm / RMSNorm (fp32 rows -> GEMM operand dtype),
// embedding gathers with fused sine positional + timestep add, and the AR-prefill
// RoPE + KV-cache write.  One wave per row, lanes stride the row (coalesced 256 B per
// wave instruction); row statistics by wave shuffles only.
#include "common.h"

namespace {

constexpr int MAXI = 32;   // D <= 64*32 = 2048

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, int64_t ldx, const float* gamma, const float* beta,
                                                        float eps, typename T::storage* y, int64_t ldy, int M, int D,
                                                        int n_affine, int64_t affine_stride, int64_t y_affine_stride) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int ni = D >> 6;
    const float* xr = x + (int64_t)row * ldx;
    float v[MAXI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
        if (i < ni) { v[i] = xr[lane + 64 * i]; s += v[i]; }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
        if (i < ni) { const float d = v[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    for (int a = 0; a < n_affine; ++a) {
        const float* g = gamma + a * affine_stride;
        const float* bt = beta + a * affine_stride;
        typename T::storage* yr = y + a * y_affine_stride + (int64_t)row * ldy;
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
            if (i < ni) {
                const int c = lane + 64 * i;
                yr[c] = T::from_f32((v[i] - mean) * rstd * g[c] + bt[c]);
            }
    }
}


// Vector form (D = 256 NV): lane holds NV float4 (16-byte loads, 1 KiB per wave instruction),
// the 16-bit output leaves as 8-byte stores.  Same two-pass statistics as the scalar kernel.
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* x, int64_t ldx, const float* gamma, const float* beta,
                                                            float eps, typename T::storage* y, int64_t ldy, int M,
                                                            int n_affine, int64_t affine_stride, int64_t y_affine_stride, float* mean_out) {
    using st = typename T::storage;
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (int64_t)row * ldx;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (lane + 64 * i) * 4);
    // the affine parameters are requested now, behind the row itself, so their L2 round trip overlaps the two reductions
    // instead of following them.  Several affines of one normalisation (the 7 codebook heads): one per blockIdx.y -- every
    // workgroup is a single pass (re-reading the row from L2 is cheaper than 7 dependent parameter round trips per wave:
    // 18 us per launch for the 450 x 7 head rows of a NAR step before; same arithmetic per (row, affine)).
    float4 g0[NV], b0[NV];
    const int a_first = blockIdx.y;
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            g0[i] = *reinterpret_cast<const float4*>(gamma + a_first * affine_stride + (lane + 64 * i) * 4);
            b0[i] = *reinterpret_cast<const float4*>(beta + a_first * affine_stride + (lane + 64 * i) * 4);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) / (float)D;
    if (mean_out && lane == 0 && blockIdx.y == 0) mean_out[row] = mean;      // m5_layernorm_mean: the row's centre for a deferred LayerNorm chain
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    (void)n_affine;
    {
        st* yr = y + a_first * y_affine_stride + (int64_t)row * ldy;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            const float4 gg = g0[i], bb = b0[i];
            const float o[4] = {v[i].x * rstd * gg.x + bb.x, v[i].y * rstd * gg.y + bb.y,
                                v[i].z * rstd * gg.z + bb.z, v[i].w * rstd * gg.w + bb.w};
            if constexpr (sizeof(st) == 4) {
                *reinterpret_cast<float4*>(yr + c) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                st t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = T::from_f32(o[e]);
                *reinterpret_cast<uint2*>(yr + c) = *reinterpret_cast<const uint2*>(t);
            }
        }
    }
}

// LayerNorm (gamma, beta, eps) followed by a second, affine-free normalisation (eps2) of its fp32 result, in one pass over the
// row: the NAR output heads (reference model.py:236-242,342) apply their own LayerNorm to the decoder's final LayerNorm output;
// the seven heads' statistics are the same, so ONE normalised copy serves all of them once their gamma / beta are folded into
// the head weights (nar_engine.NARModel).  Rows come in n_seq runs of rows_per_seq (x: run s starts x_seq_stride rows
// after run s - 1; y: packed).
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_twice_vec_kernel(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                                                  float eps2, typename T::storage* y, int64_t ldy, int rows_per_seq, int n_seq,
                                                                  int64_t x_seq_stride) {
    using st = typename T::storage;
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_per_seq * n_seq) return;
    const int sq = row / rows_per_seq, r = row - sq * rows_per_seq;
    const float* xr = x + ((int64_t)sq * x_seq_stride + r) * ldx;
    float4 v[NV], g0[NV], b0[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (lane + 64 * i) * 4);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g0[i] = *reinterpret_cast<const float4*>(gamma + (lane + 64 * i) * 4);
        b0[i] = *reinterpret_cast<const float4*>(beta + (lane + 64 * i) * 4);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {          // the first LayerNorm's output, exactly as layernorm_vec_kernel computes it (fp32)
        v[i].x = v[i].x * rstd * g0[i].x + b0[i].x; v[i].y = v[i].y * rstd * g0[i].y + b0[i].y;
        v[i].z = v[i].z * rstd * g0[i].z + b0[i].z; v[i].w = v[i].w * rstd * g0[i].w + b0[i].w;
        s2 += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean2 = wave_sum(s2) / (float)D;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x -= mean2; v[i].y -= mean2; v[i].z -= mean2; v[i].w -= mean2;
        q2 += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd2 = 1.0f / sqrtf(wave_sum(q2) / (float)D + eps2);
    st* yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        const float o[4] = {v[i].x * rstd2, v[i].y * rstd2, v[i].z * rstd2, v[i].w * rstd2};
        if constexpr (sizeof(st) == 4) {
            *reinterpret_cast<float4*>(yr + c) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            st t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = T::from_f32(o[e]);
            *reinterpret_cast<uint2*>(yr + c) = *reinterpret_cast<const uint2*>(t);
        }
    }
}

template <typename T>
bool launch_ln_twice(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, float eps2, void* y, int64_t ldy,
                     int rows_per_seq, int n_seq, int64_t x_seq_stride, int D, hipStream_t s) {
    using st = typename T::storage;
    const int es = sizeof(st);
    if (D % 256 || (ldx % 4) || (ldy * es % (es == 4 ? 16 : 8)) || (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y) & 15)) return false;
    dim3 grid((rows_per_seq * n_seq + 3) / 4);
#define M5_LN2(NV) hipLaunchKernelGGL((layernorm_twice_vec_kernel<T, NV>), grid, dim3(256), 0, s, x, ldx, gamma, beta, eps, eps2, (st*)y, ldy, rows_per_seq, n_seq, x_seq_stride)
    switch (D / 256) {
        case 1: M5_LN2(1); break;
        case 2: M5_LN2(2); break;
        case 4: M5_LN2(4); break;
        case 6: M5_LN2(6); break;
        case 8: M5_LN2(8); break;
        default: return false;
    }
#undef M5_LN2
    return true;
}

template <typename T>
bool launch_ln_vec(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, void* y, int64_t ldy, int M,
                   int D, int n_affine, int64_t affine_stride, int64_t y_affine_stride, hipStream_t s, float* mean_out = nullptr) {
    using st = typename T::storage;
    const int es = sizeof(st);
    if (D % 256 || (ldx % 4) || (ldy * es % (es == 4 ? 16 : 8)) || (affine_stride % 4) || (y_affine_stride * es % (es == 4 ? 16 : 8)) ||
        (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y) & 15))
        return false;
    dim3 grid((M + 3) / 4, n_affine);
#define M5_LNV(NV) hipLaunchKernelGGL((layernorm_vec_kernel<T, NV>), grid, dim3(256), 0, s, x, ldx, gamma, beta, eps, (st*)y, ldy, M, n_affine, affine_stride, y_affine_stride, mean_out)
    switch (D / 256) {
        case 1: M5_LNV(1); break;
        case 2: M5_LNV(2); break;
        case 4: M5_LNV(4); break;
        case 6: M5_LNV(6); break;
        case 8: M5_LNV(8); break;
        default: return false;
    }
#undef M5_LNV
    return true;
}

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* x, int64_t ldx, const float* w, float eps,
                                                      typename T::storage* y, int64_t ldy, int M, int D) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int ni = D >> 6;
    const float* xr = x + (int64_t)row * ldx;
    float v[MAXI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) v[i] = xr[min(lane + 64 * i, D - 1)];     // all loads in flight before the first use
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
        if (i < ni) s += v[i] * v[i];
    const float rstd = rsqrtf(wave_sum(s) / (float)D + eps);
    typename T::storage* yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
        if (i < ni) {
            const int c = lane + 64 * i;
            const float n = v[i] * rstd;          // (x * rsqrt(...)).type_as(x)
            yr[c] = T::from_f32(n * w[c]);        // ... * weight, then cast to the GEMM dtype
        }
}

// Vector form (D = 256 NV): NV 16-byte loads per lane issued back to back, 8-byte stores of the 16-bit output.
// Same arithmetic and summation order per lane as the scalar kernel up to the order of the per-lane partial sums.
template <typename T, int NV>
__global__ __launch_bounds__(256) void rmsnorm_vec_kernel(const float* x, int64_t ldx, const float* w, float eps,
                                                          typename T::storage* y, int64_t ldy, int M) {
    using st = typename T::storage;
    constexpr int D = 256 * NV;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (int64_t)row * ldx;
    float4 v[NV], g[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (lane + 64 * i) * 4);
#pragma unroll
    for (int i = 0; i < NV; ++i) g[i] = *reinterpret_cast<const float4*>(w + (lane + 64 * i) * 4);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    const float rstd = rsqrtf(wave_sum(s) / (float)D + eps);
    st* yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        const float o[4] = {(v[i].x * rstd) * g[i].x, (v[i].y * rstd) * g[i].y, (v[i].z * rstd) * g[i].z, (v[i].w * rstd) * g[i].w};
        if constexpr (sizeof(st) == 4) {
            *reinterpret_cast<float4*>(yr + c) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            st t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = T::from_f32(o[e]);
            *reinterpret_cast<uint2*>(yr + c) = *reinterpret_cast<const uint2*>(t);
        }
    }
}

template <typename T>
bool launch_rms_vec(const float* x, int64_t ldx, const float* w, float eps, void* y, int64_t ldy, int M, int D, hipStream_t s) {
    using st = typename T::storage;
    const int es = sizeof(st);
    if (D % 256 || (ldx % 4) || (ldy * es % (es == 4 ? 16 : 8)) || (((uintptr_t)x | (uintptr_t)w) & 15) || ((uintptr_t)y & (es == 4 ? 15 : 7)))
        return false;
    dim3 grid((M + 3) / 4);
#define M5_RMV(NV) hipLaunchKernelGGL((rmsnorm_vec_kernel<T, NV>), grid, dim3(256), 0, s, x, ldx, w, eps, (st*)y, ldy, M)
    switch (D / 256) {
        case 1: M5_RMV(1); break;
        case 2: M5_RMV(2); break;
        case 4: M5_RMV(4); break;
        case 6: M5_RMV(6); break;
        case 8: M5_RMV(8); break;
        default: return false;
    }
#undef M5_RMV
    return true;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(float* out, int64_t ldo, int R, int D, const float* table,
                                                          const int64_t* idx, const float* alpha, const float* pe,
                                                          const int32_t* pos, const float* add, const int32_t* add_idx) {
    const int r = blockIdx.x;
    const float* src = table + idx[r] * (int64_t)D;
    const float al = alpha ? alpha[0] : 0.f;
    const float* per = pe ? pe + (int64_t)(pos ? pos[r] : r) * D : nullptr;
    const float* ar = add ? add + (int64_t)(add_idx ? add_idx[r] : 0) * D : nullptr;
    for (int c = threadIdx.x; c < D; c += 256) {
        float v = src[c];
        if (per) v = v * 1.0f + al * per[c];
        if (ar) v = v + ar[c];
        out[(int64_t)r * ldo + c] = v;
    }
}

__global__ __launch_bounds__(256) void chunked_embed_kernel(float* out, int64_t ld_rep, int n_rep, int R, int D, int n_q,
                                                            int n_codes, const float* tables, const int64_t* codes,
                                                            const float* lead_row, const float* alpha, const float* pe,
                                                            const float* add, const int32_t* add_index) {
    const int r = blockIdx.x;
    const int lead = lead_row ? 1 : 0;
    const int dq = D / n_q;
    const float al = alpha ? alpha[0] : 0.f;
    const float* ar = add ? add + (int64_t)(add_index ? add_index[0] : 0) * D : nullptr;
    // vector form (round 5; the NAR step's embedding: D = 1024, 128 columns per codebook): four consecutive columns per thread --
    // they share a codebook -- as float4 loads / stores; the same per-element arithmetic as the scalar loop below
    if ((dq & 3) == 0 && (D & 3) == 0 && (ld_rep & 3) == 0 &&
        ((((uintptr_t)out | (uintptr_t)tables | (uintptr_t)pe | (uintptr_t)add | (uintptr_t)lead_row) & 15) == 0)) {
        for (int c = threadIdx.x * 4; c < D; c += 1024) {
            float4 v;
            if (r < lead) {
                v = *reinterpret_cast<const float4*>(lead_row + c);
            } else {
                const int q = c / dq;
                const int64_t code = codes[(int64_t)(r - lead) * n_q + q];
                v = *reinterpret_cast<const float4*>(tables + ((int64_t)q * n_codes + code) * dq + (c - q * dq));
            }
            if (pe) {
                const float4 p4 = *reinterpret_cast<const float4*>(pe + (int64_t)r * D + c);
                v.x = v.x * 1.0f + al * p4.x; v.y = v.y * 1.0f + al * p4.y; v.z = v.z * 1.0f + al * p4.z; v.w = v.w * 1.0f + al * p4.w;
            }
            if (ar) {
                const float4 a4 = *reinterpret_cast<const float4*>(ar + c);
                v.x = v.x + a4.x; v.y = v.y + a4.y; v.z = v.z + a4.z; v.w = v.w + a4.w;
            }
            for (int rep = 0; rep < n_rep; ++rep) *reinterpret_cast<float4*>(out + rep * ld_rep + (int64_t)r * D + c) = v;
        }
        return;
    }
    for (int c = threadIdx.x; c < D; c += 256) {
        float v;
        if (r < lead) {
            v = lead_row[c];
        } else {
            const int q = c / dq;
            const int64_t code = codes[(int64_t)(r - lead) * n_q + q];
            v = tables[((int64_t)q * n_codes + code) * dq + (c - q * dq)];
        }
        if (pe) v = v * 1.0f + al * pe[(int64_t)r * D + c];
        if (ar) v = v + ar[c];
        for (int rep = 0; rep < n_rep; ++rep) out[rep * ld_rep + (int64_t)r * D + c] = v;
    }
}

// one block per row m; thread t handles the (even, odd) pair t of each of q / k, and v.
template <typename T>
__global__ __launch_bounds__(256) void rope_cache_kernel(const typename T::storage* qkv, int M, int H, int pos0,
                                                         const float* rope, typename T::storage* q_out,
                                                         typename T::storage* kc, typename T::storage* vc,
                                                         int64_t cache_hs, int window, typename T::storage* vt,
                                                         int64_t vt_hs, int64_t vt_ds) {
    const int m = blockIdx.x;
    const int D = H * 64;
    const int pos = pos0 + m;
    const int slot = pos % window;
    const typename T::storage* row = qkv + (int64_t)m * 3 * D;
    for (int pr = threadIdx.x; pr < D / 2; pr += 256) {
        const int c = 2 * pr, h = c >> 6, d = c & 63;
        const float cs = rope[((int64_t)pos * 32 + (d >> 1)) * 2], sn = rope[((int64_t)pos * 32 + (d >> 1)) * 2 + 1];
        {
            const float a = T::to_f32(row[c]), b = T::to_f32(row[c + 1]);
            typename T::storage* dst = q_out + ((int64_t)h * M + m) * 64 + d;
            dst[0] = T::from_f32(a * cs - b * sn);
            dst[1] = T::from_f32(a * sn + b * cs);
        }
        {
            const float a = T::to_f32(row[D + c]), b = T::to_f32(row[D + c + 1]);
            typename T::storage* dst = kc + h * cache_hs + (int64_t)slot * 64 + d;
            dst[0] = T::from_f32(a * cs - b * sn);
            dst[1] = T::from_f32(a * sn + b * cs);
        }
        {
            const typename T::storage a = row[2 * D + c], b = row[2 * D + c + 1];
            typename T::storage* dst = vc + h * cache_hs + (int64_t)slot * 64 + d;
            dst[0] = a;
            dst[1] = b;
            vt[h * vt_hs + (int64_t)d * vt_ds + m] = a;
            vt[h * vt_hs + (int64_t)(d + 1) * vt_ds + m] = b;
        }
    }
}

__global__ void add_int_kernel(int32_t* p, int32_t delta) { *p += delta; }

// Silence trim on device (reference mars5/trim.py:110-178 = librosa.effects.trim: centred frames of frame_length samples
// every hop samples over the REFLECT-padded mono signal, frame power in dB relative to the loudest frame; leading /
// trailing frames more than top_db below it are cut).  Kernel 1: one workgroup per frame -> mean power.  Kernel 2: one
// workgroup -> reference power (max), first / last frame above the threshold -> [start, end) in samples.
__global__ __launch_bounds__(256) void frame_power_kernel(const float* y, int n, int frame_length, int hop, float* power) {
    __shared__ float red[4];
    const int f = blockIdx.x, pad = frame_length / 2;
    float s = 0.f;
    for (int i = threadIdx.x; i < frame_length; i += 256) {
        int src = f * hop + i - pad;
        if (src < 0) src = -src;                                  // reflect (no edge repeat), like F.pad(mode="reflect")
        if (src >= n) src = 2 * (n - 1) - src;
        const float v = y[src];
        s += v * v;
    }
    const float tot = block_sum<4>(s, red);
    if (threadIdx.x == 0) power[f] = tot / (float)frame_length;
}
__global__ __launch_bounds__(256) void trim_bounds_kernel(const float* power, int n_frames, int n, int hop, float top_db, int32_t* bounds) {
    __shared__ float red[4];
    __shared__ int lo_s, hi_s;
    float mx = 0.f;
    for (int f = threadIdx.x; f < n_frames; f += 256) mx = fmaxf(mx, power[f]);
    const float ref = block_max<4>(mx, red);
    if (threadIdx.x == 0) { lo_s = 0x7fffffff; hi_s = -1; }
    __syncthreads();
    const float ref_db = 10.0f * log10f(fmaxf(1e-10f, ref));
    int lo = 0x7fffffff, hi = -1;
    for (int f = threadIdx.x; f < n_frames; f += 256) {
        const float db = 10.0f * log10f(fmaxf(1e-10f, power[f])) - ref_db;
        if (db > -top_db) { lo = min(lo, f); hi = max(hi, f); }
    }
    atomicMin(&lo_s, lo);
    atomicMax(&hi_s, hi);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (hi_s < 0) { bounds[0] = 0; bounds[1] = 0; }
        else { bounds[0] = lo_s * hop; bounds[1] = min(n, (hi_s + 1) * hop); }
    }
}

// AR -> NAR hand-off on device (reference inference.py:272-275: (ar_codes - n_text).clamp(0)[first:] through
// speechtok.decode_int, minbpe/codebook.py:88-126): every BPE token id expands to the run of codebook-0 codes it was merged
// from (CSR table off / vals; special tokens expand to nothing).  One workgroup: lengths -> block scan -> scatter.
__global__ __launch_bounds__(1024) void expand_tokens_kernel(const int64_t* tokens, int n, int n_text, const int32_t* off, const int64_t* vals,
                                                             int n_vocab, int64_t* out, int out_cap, int32_t* total) {
    __shared__ int wsum[16];
    __shared__ int blk;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + tid;
        int t = 0, len = 0;
        if (i < n) {
            const int64_t v = tokens[i] - n_text;
            t = (int)(v < 0 ? 0 : v);
            len = (t < n_vocab) ? off[t + 1] - off[t] : 0;
        }
        int inc = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wv; ++w) woff += wsum[w];
        if (tid == 1023) blk = woff + inc;
        const int excl = base + woff + inc - len;
        for (int j = 0; j < len; ++j)
            if (excl + j < out_cap) out[excl + j] = vals[off[t] + j];
        __syncthreads();
        base += blk;
        __syncthreads();
    }
    if (tid == 0) *total = base;
}

}  // namespace

extern "C" int m5_layernorm(int out_dtype, const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                            void* y, int64_t ldy, int M, int D, int n_affine, int64_t affine_stride,
                            int64_t y_affine_stride, void* stream) {
    if (!x || !gamma || !beta || !y || M <= 0 || D <= 0 || n_affine <= 0) return M5_ERR_ARG;
    if (D % 64 || D > 64 * MAXI) return M5_ERR_UNSUPPORTED;
    dim3 grid((M + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
    {
        bool done = false;
        if (out_dtype == M5_F32) done = launch_ln_vec<F32T>(x, ldx, gamma, beta, eps, y, ldy, M, D, n_affine, affine_stride, y_affine_stride, s);
        else if (out_dtype == M5_F16) done = launch_ln_vec<F16T>(x, ldx, gamma, beta, eps, y, ldy, M, D, n_affine, affine_stride, y_affine_stride, s);
        else if (out_dtype == M5_BF16) done = launch_ln_vec<BF16T>(x, ldx, gamma, beta, eps, y, ldy, M, D, n_affine, affine_stride, y_affine_stride, s);
        if (done) { M5_CHECK_LAUNCH(); return M5_OK; }
    }
    switch (out_dtype) {
        case M5_F32: hipLaunchKernelGGL(layernorm_kernel<F32T>, grid, dim3(256), 0, s, x, ldx, gamma, beta, eps, (float*)y, ldy, M, D, n_affine, affine_stride, y_affine_stride); break;
        case M5_F16: hipLaunchKernelGGL(layernorm_kernel<F16T>, grid, dim3(256), 0, s, x, ldx, gamma, beta, eps, (_Float16*)y, ldy, M, D, n_affine, affine_stride, y_affine_stride); break;
        case M5_BF16: hipLaunchKernelGGL(layernorm_kernel<BF16T>, grid, dim3(256), 0, s, x, ldx, gamma, beta, eps, (uint16_t*)y, ldy, M, D, n_affine, affine_stride, y_affine_stride); break;
        default: return M5_ERR_ARG;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}

// m5_layernorm that also leaves each row's mean in mean_out[M] (fp32): the first centre of a deferred-LayerNorm chain
// (M5DeferredLN).  Vector rows only (D % 256 == 0, 16-byte aligned operands), one affine.
extern "C" int m5_layernorm_mean(int out_dtype, const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                 void* y, int64_t ldy, int M, int D, float* mean_out, void* stream) {
    if (!x || !gamma || !beta || !y || !mean_out || M <= 0 || D <= 0) return M5_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    bool done = false;
    if (out_dtype == M5_F32) done = launch_ln_vec<F32T>(x, ldx, gamma, beta, eps, y, ldy, M, D, 1, 0, 0, s, mean_out);
    else if (out_dtype == M5_F16) done = launch_ln_vec<F16T>(x, ldx, gamma, beta, eps, y, ldy, M, D, 1, 0, 0, s, mean_out);
    else if (out_dtype == M5_BF16) done = launch_ln_vec<BF16T>(x, ldx, gamma, beta, eps, y, ldy, M, D, 1, 0, 0, s, mean_out);
    else return M5_ERR_ARG;
    if (!done) return M5_ERR_UNSUPPORTED;
    M5_CHECK_LAUNCH();
    return M5_OK;
}

extern "C" int m5_layernorm_twice(int out_dtype, const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, float eps2,
                                  void* y, int64_t ldy, int rows_per_seq, int n_seq, int64_t x_seq_stride, int D, void* stream) {
    if (!x || !gamma || !beta || !y || rows_per_seq <= 0 || n_seq <= 0 || D <= 0 || x_seq_stride < 0) return M5_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    bool done = false;
    if (out_dtype == M5_F32) done = launch_ln_twice<F32T>(x, ldx, gamma, beta, eps, eps2, y, ldy, rows_per_seq, n_seq, x_seq_stride, D, s);
    else if (out_dtype == M5_F16) done = launch_ln_twice<F16T>(x, ldx, gamma, beta, eps, eps2, y, ldy, rows_per_seq, n_seq, x_seq_stride, D, s);
    else if (out_dtype == M5_BF16) done = launch_ln_twice<BF16T>(x, ldx, gamma, beta, eps, eps2, y, ldy, rows_per_seq, n_seq, x_seq_stride, D, s);
    else return M5_ERR_ARG;
    if (!done) return M5_ERR_UNSUPPORTED;
    M5_CHECK_LAUNCH();
    return M5_OK;
}

extern "C" int m5_rmsnorm(int out_dtype, const float* x, int64_t ldx, const float* w, float eps, void* y, int64_t ldy,
                          int M, int D, void* stream) {
    if (!x || !w || !y || M <= 0 || D <= 0) return M5_ERR_ARG;
    if (D % 64 || D > 64 * MAXI) return M5_ERR_UNSUPPORTED;
    dim3 grid((M + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
    {
        bool done = false;
        if (out_dtype == M5_F32) done = launch_rms_vec<F32T>(x, ldx, w, eps, y, ldy, M, D, s);
        else if (out_dtype == M5_F16) done = launch_rms_vec<F16T>(x, ldx, w, eps, y, ldy, M, D, s);
        else if (out_dtype == M5_BF16) done = launch_rms_vec<BF16T>(x, ldx, w, eps, y, ldy, M, D, s);
        if (done) { M5_CHECK_LAUNCH(); return M5_OK; }
    }
    switch (out_dtype) {
        case M5_F32: hipLaunchKernelGGL(rmsnorm_kernel<F32T>, grid, dim3(256), 0, s, x, ldx, w, eps, (float*)y, ldy, M, D); break;
        case M5_F16: hipLaunchKernelGGL(rmsnorm_kernel<F16T>, grid, dim3(256), 0, s, x, ldx, w, eps, (_Float16*)y, ldy, M, D); break;
        case M5_BF16: hipLaunchKernelGGL(rmsnorm_kernel<BF16T>, grid, dim3(256), 0, s, x, ldx, w, eps, (uint16_t*)y, ldy, M, D); break;
        default: return M5_ERR_ARG;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}

extern "C" int m5_gather_rows(float* out, int64_t ldo, int R, int D, const float* table, const int64_t* idx,
                              const float* alpha, const float* pe, const int32_t* pos, const float* add,
                              const int32_t* add_idx, void* stream) {
    if (!out || !table || !idx || R <= 0 || D <= 0) return M5_ERR_ARG;
    if (pe && !alpha) return M5_ERR_ARG;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, out, ldo, R, D, table, idx, alpha, pe, pos, add, add_idx);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

extern "C" int m5_chunked_embed(float* out, int64_t ld_rep, int n_rep, int R, int D, int n_q, int n_codes,
                                const float* tables, const int64_t* codes, const float* lead_row, const float* alpha,
                                const float* pe, const float* add, const int32_t* add_index, void* stream) {
    if (!out || !tables || R <= 0 || D <= 0 || n_q <= 0 || D % n_q || n_rep <= 0) return M5_ERR_ARG;
    if (!codes && !(lead_row && R == 1)) return M5_ERR_ARG;
    if (pe && !alpha) return M5_ERR_ARG;
    hipLaunchKernelGGL(chunked_embed_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, out, ld_rep, n_rep, R, D, n_q,
                       n_codes, tables, codes, lead_row, alpha, pe, add, add_index);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

extern "C" int m5_rope_cache(int dtype, const void* qkv, int M, int n_heads, int pos0, const float* rope, void* q_out,
                             void* kcache, void* vcache, int64_t cache_hs, int window, void* vt_out, int64_t vt_hs,
                             int64_t vt_ds, void* stream) {
    if (!qkv || !rope || !q_out || !kcache || !vcache || !vt_out || M <= 0 || n_heads <= 0 || window <= 0) return M5_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case M5_F32: hipLaunchKernelGGL(rope_cache_kernel<F32T>, dim3(M), dim3(256), 0, s, (const float*)qkv, M, n_heads, pos0, rope, (float*)q_out, (float*)kcache, (float*)vcache, cache_hs, window, (float*)vt_out, vt_hs, vt_ds); break;
        case M5_F16: hipLaunchKernelGGL(rope_cache_kernel<F16T>, dim3(M), dim3(256), 0, s, (const _Float16*)qkv, M, n_heads, pos0, rope, (_Float16*)q_out, (_Float16*)kcache, (_Float16*)vcache, cache_hs, window, (_Float16*)vt_out, vt_hs, vt_ds); break;
        case M5_BF16: hipLaunchKernelGGL(rope_cache_kernel<BF16T>, dim3(M), dim3(256), 0, s, (const uint16_t*)qkv, M, n_heads, pos0, rope, (uint16_t*)q_out, (uint16_t*)kcache, (uint16_t*)vcache, cache_hs, window, (uint16_t*)vt_out, vt_hs, vt_ds); break;
        default: return M5_ERR_ARG;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}

extern "C" int m5_expand_tokens(const int64_t* tokens, int n, int n_text, const int32_t* off, const int64_t* vals, int n_vocab,
                                int64_t* out, int out_cap, int32_t* total, void* stream) {
    if (!tokens || !off || !vals || !out || !total || n < 0 || n_vocab <= 0 || out_cap < 0) return M5_ERR_ARG;
    hipLaunchKernelGGL(expand_tokens_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, tokens, n, n_text, off, vals, n_vocab, out, out_cap, total);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

extern "C" int m5_trim_bounds(const float* y, int n, int frame_length, int hop, float top_db, float* power, int n_frames,
                              int32_t* bounds, void* stream) {
    if (!y || !power || !bounds || n <= 0 || frame_length <= 0 || hop <= 0) return M5_ERR_ARG;
    if (n <= frame_length / 2 || n_frames != 1 + n / hop || (frame_length % 2)) return M5_ERR_UNSUPPORTED;     // reflect padding needs n > pad
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(frame_power_kernel, dim3(n_frames), dim3(256), 0, s, y, n, frame_length, hop, power);
    hipLaunchKernelGGL(trim_bounds_kernel, dim3(1), dim3(256), 0, s, power, n_frames, n, hop, top_db, bounds);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

extern "C" int m5_add_int(int32_t* p, int32_t delta, void* stream) {
    if (!p) return M5_ERR_ARG;
    hipLaunchKernelGGL(add_int_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p, delta);
    M5_CHECK_LAUNCH();
    return M5_OK;
}
