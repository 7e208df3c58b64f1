// Batched AR decode step (several sequences advance one token per step): the per-sequence pieces
// between the M = B row GEMMs -- RoPE + KV-cache write at each sequence's own position, and the merge of
// the split-KV attention partials.  Both are a few KB of traffic per sequence; the weight stream lives
// in gemm_skinny.hip, the cache scan in ar_decode.hip (attn_decode_kernel, batched over blockIdx.z).
#include "common.h"

namespace {

// one workgroup per sequence; thread t handles the (even, odd) pairs of q / k and the v copy.
// Same arithmetic as rope_cache_kernel (rowops.hip): operands are the dtype-rounded GEMM outputs.
template <typename T>
__global__ __launch_bounds__(256) void rope_cache_batch_kernel(const typename T::storage* qkv, int H, const float* rope,
                                                               const int32_t* state, int state_bs, typename T::storage* qbuf,
                                                               int64_t q_bs, typename T::storage* kc, typename T::storage* vc,
                                                               int64_t cache_bs, int64_t cache_hs, int window) {
    using st = typename T::storage;
    const int64_t b = blockIdx.x;
    const int32_t* stt = state + b * state_bs;
    if (stt[M5_ST_DONE]) return;
    const int D = H * 64;
    const int pos = stt[M5_ST_POS];
    const int slot = pos % window;
    const st* row = qkv + b * 3 * D;
    st* q = qbuf + b * q_bs;
    st* kcb = kc + b * cache_bs;
    st* vcb = vc + b * cache_bs;
    for (int pr = threadIdx.x; pr < D / 2; pr += 256) {
        const int c = 2 * pr, h = c >> 6, d = c & 63;
        const float cs = rope[((int64_t)pos * 32 + (d >> 1)) * 2], sn = rope[((int64_t)pos * 32 + (d >> 1)) * 2 + 1];
        {
            const float x = T::to_f32(row[c]), y = T::to_f32(row[c + 1]);
            q[c] = T::from_f32(x * cs - y * sn);
            q[c + 1] = T::from_f32(x * sn + y * cs);
        }
        {
            const float x = T::to_f32(row[D + c]), y = T::to_f32(row[D + c + 1]);
            st* dst = kcb + h * cache_hs + (int64_t)slot * 64 + d;
            dst[0] = T::from_f32(x * cs - y * sn);
            dst[1] = T::from_f32(x * sn + y * cs);
        }
        {
            st* dst = vcb + h * cache_hs + (int64_t)slot * 64 + d;
            dst[0] = row[2 * D + c];
            dst[1] = row[2 * D + c + 1];
        }
    }
}

// out[b][h*64 + d] = sum_s w_s o_s[d] / sum_s w_s l_s, w_s = exp(m_s - max m): the merge the batch-1
// path does in the Wo GEMV's prologue (ar_decode.hip, M5_PRO_ATTN), sequential over splits.
template <typename T, int NS>
__global__ __launch_bounds__(64) void attn_combine_batch_kernel(const float* part, int64_t part_bs, int H, int nsplit,
                                                                const int32_t* state, int state_bs,
                                                                typename T::storage* out, int64_t out_bs) {
    // grid (H, B), one wave per (sequence, head), lane = d.  NS > 0: nsplit is NS and every partial is loaded before
    // the first use (the merge itself is a handful of flops; the launch is the round trip of its loads).
    const int64_t b = blockIdx.y;
    const int h = blockIdx.x, d = threadIdx.x;
    if (state[b * state_bs + M5_ST_DONE]) return;
    const float* pp = part + b * part_bs + (int64_t)h * nsplit * M5_ATTN_PART;
    float o = 0.f, l = 0.f;
    if constexpr (NS > 0) {
        float pm[NS], pl[NS], po[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            pm[s] = pp[s * M5_ATTN_PART + 64];
            pl[s] = pp[s * M5_ATTN_PART + 65];
            po[s] = pp[s * M5_ATTN_PART + d];
        }
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s) mx = fmaxf(mx, pm[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float w = expf(pm[s] - mx);
            o += w * po[s];
            l += w * pl[s];
        }
    } else {
        float mx = -INFINITY;
        for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, pp[s * M5_ATTN_PART + 64]);
        for (int s = 0; s < nsplit; ++s) {
            const float w = expf(pp[s * M5_ATTN_PART + 64] - mx);
            o += w * pp[s * M5_ATTN_PART + d];
            l += w * pp[s * M5_ATTN_PART + 65];
        }
    }
    out[b * out_bs + h * 64 + d] = T::from_f32(o / l);
}

template <typename T>
void launch_combine(const float* part, int64_t part_bs, int B, int H, int nsplit, const int32_t* state, int state_bs, void* out,
                    int64_t out_bs, hipStream_t s) {
    using st = typename T::storage;
#define M5_CMB(NS) hipLaunchKernelGGL((attn_combine_batch_kernel<T, NS>), dim3(H, B), dim3(64), 0, s, part, part_bs, H, nsplit, state, state_bs, (st*)out, out_bs)
    switch (nsplit) {
        case 8: M5_CMB(8); break;
        case 4: M5_CMB(4); break;
        case 2: M5_CMB(2); break;
        case 1: M5_CMB(1); break;
        default: M5_CMB(0); break;
    }
#undef M5_CMB
}

}  // namespace

extern "C" int m5_ar_rope_cache_batch(int dtype, const void* qkv, int B, int n_heads, const float* rope, const int32_t* state,
                                      int32_t state_bs, void* qbuf, int64_t q_bs, void* kcache, void* vcache, int64_t cache_bs,
                                      int64_t cache_hs, int window, void* stream) {
    if (!qkv || !rope || !state || !qbuf || !kcache || !vcache || B <= 0 || n_heads <= 0 || window <= 0) return M5_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case M5_F32: hipLaunchKernelGGL(rope_cache_batch_kernel<F32T>, dim3(B), dim3(256), 0, s, (const float*)qkv, n_heads, rope, state, state_bs, (float*)qbuf, q_bs, (float*)kcache, (float*)vcache, cache_bs, cache_hs, window); break;
        case M5_F16: hipLaunchKernelGGL(rope_cache_batch_kernel<F16T>, dim3(B), dim3(256), 0, s, (const _Float16*)qkv, n_heads, rope, state, state_bs, (_Float16*)qbuf, q_bs, (_Float16*)kcache, (_Float16*)vcache, cache_bs, cache_hs, window); break;
        case M5_BF16: hipLaunchKernelGGL(rope_cache_batch_kernel<BF16T>, dim3(B), dim3(256), 0, s, (const uint16_t*)qkv, n_heads, rope, state, state_bs, (uint16_t*)qbuf, q_bs, (uint16_t*)kcache, (uint16_t*)vcache, cache_bs, cache_hs, window); break;
        default: return M5_ERR_ARG;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}

extern "C" int m5_ar_attn_combine_batch(int dtype, const float* part, int64_t part_bs, int B, int n_heads, int nsplit,
                                        const int32_t* state, int32_t state_bs, void* out, int64_t out_bs, void* stream) {
    if (!part || !state || !out || B <= 0 || n_heads <= 0 || nsplit <= 0) return M5_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case M5_F32: launch_combine<F32T>(part, part_bs, B, n_heads, nsplit, state, state_bs, out, out_bs, s); break;
        case M5_F16: launch_combine<F16T>(part, part_bs, B, n_heads, nsplit, state, state_bs, out, out_bs, s); break;
        case M5_BF16: launch_combine<BF16T>(part, part_bs, B, n_heads, nsplit, state, state_bs, out, out_bs, s); break;
        default: return M5_ERR_ARG;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}
