// Skinny GEMM for the batched AR decode step: C[M x N] = A[M x K] . W[N x K]^T with M <= 32 rows
// (one row per sequence of the batch) -- the batch-B form of the decode GEMV.
//
// The step is HBM-bound on the weights exactly like the GEMV (every weight byte is read once per
// step, now shared by all B sequences), so the kernel is a weight STREAMER that happens to use
// MFMA for the dot products:
//   * one workgroup per 16 weight rows, its 8 waves split K; a wave issues its WHOLE weight share
//     (<= 14 16-byte loads per lane, straight into MFMA operand layout: lane = (row
//     l & 15, k-chunk l >> 4)) before anything else, then the matching activation fragments (L2
//     hits), then runs its <= 14 x MT MFMAs -- no LDS staging, nothing between HBM and the matrix
//     core but registers;
//   * the 8 K-partials are summed through LDS in a fixed order (deterministic), and the epilogue
//     (bias / dtype / fp32 / residual add / SwiGLU on interleaved rows) runs on the reduced tile.
// Rows m >= M read row M - 1 (valid memory) and are not stored.
#include "common.h"

namespace {

typedef float f4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ inline f4_t mfma16(const u32x4& a, const uint4& b, f4_t c);
template <>
__device__ inline f4_t mfma16<F16T>(const u32x4& a, const uint4& b, f4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8_t*>(&a), *reinterpret_cast<const h8_t*>(&b), c, 0, 0, 0);
}
template <>
__device__ inline f4_t mfma16<BF16T>(const u32x4& a, const uint4& b, f4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const b8_t*>(&a), *reinterpret_cast<const b8_t*>(&b), c, 0, 0, 0);
}

// same expression as the wide GEMM's SwiGLU epilogue (gemm16.hip), so prefill and decode agree
__device__ inline float silu_fast(float a) {
    return a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * a));
}

struct SkinnyParams {
    const unsigned char* A; const unsigned char* W; const float* bias; unsigned char* C;
    int64_t lda, ldw, ldc;      // elements
    int M, N, K;
    // SK_EPI_QKV_ROPE (batched decode QKV projection): rotate q, k at each sequence's own position and write
    // q -> qbuf + m*q_bs, k / v -> cache + m*cache_bs + h*cache_hs + slot*64 (slot = pos % window)
    const float* rope; const int32_t* state; int state_bs;
    unsigned char* qbuf; unsigned char* kcache; unsigned char* vcache;
    int64_t q_bs, cache_bs, cache_hs;
    int dim, window;
};
constexpr int SK_EPI_QKV_ROPE = 100;

constexpr int SK_NW = 8;        // waves per workgroup (K split)
constexpr int SK_KSW = 14;      // 32-deep K steps per wave, at most  ->  K <= 3584 (<= 128 VGPRs at MT = 1: two workgroups per CU)

template <typename T, int EPI, int MT, bool NT>
__global__ __launch_bounds__(SK_NW * 64) void skinny_gemm_kernel(SkinnyParams p) {
    using st = typename T::storage;
    __shared__ f4_t red[SK_NW][MT][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int ksteps = p.K >> 5;
    const int per = (ksteps + SK_NW - 1) / SK_NW;
    const int s0 = wave * per;
    const int nst = max(0, min(ksteps, s0 + per) - s0);                  // wave-uniform

    // ---- the wave's whole weight share in flight first (HBM), then the activations (L2)
    const unsigned char* wp = p.W + ((int64_t)(n0 + l15) * p.ldw + (int64_t)s0 * 32 + lg * 8) * 2;
    u32x4 wf[SK_KSW];
#pragma unroll
    for (int s = 0; s < SK_KSW; ++s)
        if (s < nst) {
            if constexpr (NT) wf[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + s * 64));
            else wf[s] = *reinterpret_cast<const u32x4*>(wp + s * 64);
        }
    uint4 xf[MT][SK_KSW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = min(mt * 16 + l15, p.M - 1);
        const unsigned char* xp = p.A + ((int64_t)row * p.lda + (int64_t)s0 * 32 + lg * 8) * 2;
#pragma unroll
        for (int s = 0; s < SK_KSW; ++s)
            if (s < nst) xf[mt][s] = *reinterpret_cast<const uint4*>(xp + s * 64);
    }
    f4_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < SK_KSW; ++s) {
        if (s < nst) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma16<T>(wf[s], xf[mt][s], acc[mt]);
        }
    }
    // acc[mt][r] = sum over this wave's K share of A[m = 16 mt + l15][k] W[n = n0 + 4 lg + r][k]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[wave][mt][lane] = acc[mt];
    __syncthreads();

    if constexpr (EPI == SK_EPI_QKV_ROPE) {
        // thread = (sequence m, pair pi): columns n0 + 2 pi, + 1 are one RoPE pair of one head of q, k or v.
        // The sums are rounded to the operand type first, then rotated -- the arithmetic of EPI_DT followed by
        // rope_cache_kernel (prefill) / rope_cache_batch_kernel.
        const int pi = tid & 7, m = tid >> 3;
        if (m < MT * 16 && m < p.M && !p.state[(int64_t)m * p.state_bs + M5_ST_DONE]) {
            float v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = 2 * pi + h;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < SK_NW; ++w) s += red[w][m >> 4][(n >> 2) * 16 + (m & 15)][n & 3];
                v[h] = round_dt<T>(s + (p.bias ? p.bias[n0 + n] : 0.f));
            }
            const int sec = n0 / p.dim, c = (n0 - sec * p.dim) + 2 * pi, hd = c >> 6, d = c & 63;
            const int pos = p.state[(int64_t)m * p.state_bs + M5_ST_POS];
            const int slot = pos % p.window;
            st* dst;
            if (sec == 0) dst = reinterpret_cast<st*>(p.qbuf) + (int64_t)m * p.q_bs + c;
            else dst = reinterpret_cast<st*>(sec == 1 ? p.kcache : p.vcache) + (int64_t)m * p.cache_bs + hd * p.cache_hs + (int64_t)slot * 64 + d;
            if (sec < 2) {
                const float cs = p.rope[((int64_t)pos * 32 + (d >> 1)) * 2], sn = p.rope[((int64_t)pos * 32 + (d >> 1)) * 2 + 1];
                dst[0] = T::from_f32(v[0] * cs - v[1] * sn);
                dst[1] = T::from_f32(v[0] * sn + v[1] * cs);
            } else {
                dst[0] = T::from_f32(v[0]);
                dst[1] = T::from_f32(v[1]);
            }
        }
    } else if constexpr (EPI == M5_EPI_SWIGLU) {
        // rows of W interleaved (W_i, V_i): n = 2 pi is the gate, n = 2 pi + 1 the value
        const int pi = tid & 7, m = tid >> 3;
        if (m < MT * 16 && m < p.M) {
            float v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = 2 * pi + h;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < SK_NW; ++w) s += red[w][m >> 4][(n >> 2) * 16 + (m & 15)][n & 3];
                v[h] = s + (p.bias ? p.bias[n0 + n] : 0.f);
            }
            const float a = round_dt<T>(v[0]), b = round_dt<T>(v[1]);
            const float sl = round_dt<T>(silu_fast(a));
            reinterpret_cast<st*>(p.C)[(int64_t)m * p.ldc + (n0 >> 1) + pi] = T::from_f32(sl * b);
        }
    } else {
        const int n = tid & 15, m = tid >> 4;
        if (m < MT * 16 && m < p.M) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < SK_NW; ++w) s += red[w][m >> 4][(n >> 2) * 16 + (m & 15)][n & 3];
            const float v = s + (p.bias ? p.bias[n0 + n] : 0.f);
            if constexpr (EPI == M5_EPI_F32) {
                reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + n0 + n] = v;
            } else if constexpr (EPI == M5_EPI_RESIDUAL) {
                float* cp = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n0 + n;
                *cp = *cp + v;
            } else {   // M5_EPI_DT
                reinterpret_cast<st*>(p.C)[(int64_t)m * p.ldc + n0 + n] = T::from_f32(v);
            }
        }
    }
}

template <typename T, int MT, bool NT>
int launch_skinny(int epi, const SkinnyParams& p, hipStream_t s) {
    dim3 grid(p.N / 16), block(SK_NW * 64);
    switch (epi) {
        case M5_EPI_F32: hipLaunchKernelGGL((skinny_gemm_kernel<T, M5_EPI_F32, MT, NT>), grid, block, 0, s, p); break;
        case M5_EPI_DT: hipLaunchKernelGGL((skinny_gemm_kernel<T, M5_EPI_DT, MT, NT>), grid, block, 0, s, p); break;
        case M5_EPI_RESIDUAL: hipLaunchKernelGGL((skinny_gemm_kernel<T, M5_EPI_RESIDUAL, MT, NT>), grid, block, 0, s, p); break;
        case M5_EPI_SWIGLU: hipLaunchKernelGGL((skinny_gemm_kernel<T, M5_EPI_SWIGLU, MT, NT>), grid, block, 0, s, p); break;
        case SK_EPI_QKV_ROPE: hipLaunchKernelGGL((skinny_gemm_kernel<T, SK_EPI_QKV_ROPE, MT, NT>), grid, block, 0, s, p); break;
        default: return M5_ERR_UNSUPPORTED;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}

template <typename T>
int launch_skinny_mt(int epi, const SkinnyParams& p, hipStream_t s) {
    // Non-temporal loads lose ~4 % here (measured, tools/ar_batch_bench.py): a wave instruction covers 64 bytes of
    // each of its 16 rows, so the two halves of a 128-byte line are fetched by consecutive instructions and the
    // line should stay in L2 in between.  M5_SKINNY_NT=1 re-enables them for A/B runs.
    const char* e = m5_tool_env("M5_SKINNY_NT");
    const bool nt = (e && e[0] == '1');
    if (p.M <= 16) return nt ? launch_skinny<T, 1, true>(epi, p, s) : launch_skinny<T, 1, false>(epi, p, s);
    return nt ? launch_skinny<T, 2, true>(epi, p, s) : launch_skinny<T, 2, false>(epi, p, s);
}

}  // namespace

// Does this call fit the skinny kernel?  (m5_gemm asks before choosing a wide-tile configuration.)
bool m5_gemm_skinny_fits(int dtype, int M, int N, int K, int epi, int batch, int64_t lda, int64_t ldw) {
    if (dtype != M5_F16 && dtype != M5_BF16) return false;
    if (batch != 1 || M > 32 || (N % 16) || (K % 32) || K > 32 * SK_KSW * SK_NW) return false;
    if (epi != M5_EPI_F32 && epi != M5_EPI_DT && epi != M5_EPI_RESIDUAL && epi != M5_EPI_SWIGLU) return false;
    if ((lda % 8) || (ldw % 8)) return false;
    const char* e = m5_tool_env("M5_GEMM_SKINNY");                // tuning / A-B only
    return !(e && e[0] == '0');
}

int m5_gemm_skinny_dispatch(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                            void* C, int64_t ldc, int M, int N, int K, int epi, hipStream_t s) {
    SkinnyParams p{};
    p.A = (const unsigned char*)A; p.W = (const unsigned char*)W; p.bias = bias; p.C = (unsigned char*)C;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    if (dtype == M5_F16) return launch_skinny_mt<F16T>(epi, p, s);
    return launch_skinny_mt<BF16T>(epi, p, s);
}

// Batched decode QKV projection with fused RoPE + KV-cache write (include/mars5_hip.h).  16-bit operands run the
// skinny kernel with the rotation in its epilogue; fp32 (parity mode) runs m5_gemm into `qkv_tmp` followed by
// m5_ar_rope_cache_batch -- the same arithmetic in two launches.
extern "C" int m5_ar_qkv_rope_batch(int dtype, const void* xn, int64_t lda, const void* wqkv, int64_t ldw, int B, int n_heads, int K,
                                    const float* rope, const int32_t* state, int32_t state_bs, void* qbuf, int64_t q_bs,
                                    void* kcache, void* vcache, int64_t cache_bs, int64_t cache_hs, int window, void* qkv_tmp,
                                    void* stream) {
    if (!xn || !wqkv || !rope || !state || !qbuf || !kcache || !vcache || B <= 0 || n_heads <= 0 || K <= 0 || window <= 0) return M5_ERR_ARG;
    const int D = n_heads * 64, N = 3 * D;
    if (m5_gemm_skinny_fits(dtype, B, N, K, M5_EPI_DT, 1, lda, ldw) && !(((uintptr_t)xn | (uintptr_t)wqkv) & 15)) {
        SkinnyParams p{};
        p.A = (const unsigned char*)xn; p.W = (const unsigned char*)wqkv; p.lda = lda; p.ldw = ldw; p.M = B; p.N = N; p.K = K;
        p.rope = rope; p.state = state; p.state_bs = state_bs; p.qbuf = (unsigned char*)qbuf; p.kcache = (unsigned char*)kcache;
        p.vcache = (unsigned char*)vcache; p.q_bs = q_bs; p.cache_bs = cache_bs; p.cache_hs = cache_hs; p.dim = D; p.window = window;
        if (dtype == M5_F16) return launch_skinny_mt<F16T>(SK_EPI_QKV_ROPE, p, (hipStream_t)stream);
        return launch_skinny_mt<BF16T>(SK_EPI_QKV_ROPE, p, (hipStream_t)stream);
    }
    if (!qkv_tmp) return M5_ERR_ARG;
    const int st = m5_gemm(dtype, xn, lda, wqkv, ldw, nullptr, qkv_tmp, N, B, N, K, M5_EPI_DT, nullptr, 1, 0, 0, 0, 0, stream);
    if (st != M5_OK) return st;
    return m5_ar_rope_cache_batch(dtype, qkv_tmp, B, n_heads, rope, state, state_bs, qbuf, q_bs, kcache, vcache, cache_bs, cache_hs,
                                  window, stream);
}
