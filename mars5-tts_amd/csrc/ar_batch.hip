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
template <typename T>
__global__ __launch_bounds__(256) void attn_combine_batch_kernel(const float* part, int64_t part_bs, int H, int nsplit,
                                                                 const int32_t* state, int state_bs,
                                                                 typename T::storage* out, int64_t out_bs) {
    const int64_t b = blockIdx.x;
    if (state[b * state_bs + M5_ST_DONE]) return;
    const float* pb = part + b * part_bs;
    for (int i = threadIdx.x; i < H * 64; i += 256) {
        const int h = i >> 6, d = i & 63;
        const float* pp = pb + (int64_t)h * nsplit * M5_ATTN_PART;
        float mx = -INFINITY;
        for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, pp[s * M5_ATTN_PART + 64]);
        float o = 0.f, l = 0.f;
        for (int s = 0; s < nsplit; ++s) {
            const float w = expf(pp[s * M5_ATTN_PART + 64] - mx);
            o += w * pp[s * M5_ATTN_PART + d];
            l += w * pp[s * M5_ATTN_PART + 65];
        }
        out[b * out_bs + i] = T::from_f32(o / l);
    }
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
        case M5_F32: hipLaunchKernelGGL(attn_combine_batch_kernel<F32T>, dim3(B), dim3(256), 0, s, part, part_bs, n_heads, nsplit, state, state_bs, (float*)out, out_bs); break;
        case M5_F16: hipLaunchKernelGGL(attn_combine_batch_kernel<F16T>, dim3(B), dim3(256), 0, s, part, part_bs, n_heads, nsplit, state, state_bs, (_Float16*)out, out_bs); break;
        case M5_BF16: hipLaunchKernelGGL(attn_combine_batch_kernel<BF16T>, dim3(B), dim3(256), 0, s, part, part_bs, n_heads, nsplit, state, state_bs, (uint16_t*)out, out_bs); break;
        default: return M5_ERR_ARG;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}
