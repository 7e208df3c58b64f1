// Skinny GEMM for the batched AR decode step: C[M x N] = A[M x K] . W[N x K]^T with M <= 32 rows
// (one row per sequence of the batch) -- the batch-B form of the decode GEMV.
//
// The step is HBM-bound on the weights exactly like the GEMV (every weight byte is read once per
// step, now shared by all B sequences), so the kernel is a weight STREAMER that happens to use
// MFMA for the dot products:
//   * one workgroup per 16 weight rows, its 8 waves split K; a wave issues its WHOLE weight share
//     (<= 16 non-temporal 16-byte loads per lane, straight into MFMA operand layout: lane = (row
//     l & 15, k-chunk l >> 4)) before anything else, then the matching activation fragments (L2
//     hits), then runs its <= 16 x MT MFMAs -- no LDS staging, nothing between HBM and the matrix
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
};

constexpr int SK_NW = 8;        // waves per workgroup (K split)
constexpr int SK_KSW = 16;      // 32-deep K steps per wave, at most  ->  K <= 4096

template <typename T, int EPI, int MT>
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
        if (s < nst) wf[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + s * 64));
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

    if constexpr (EPI == M5_EPI_SWIGLU) {
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

template <typename T, int MT>
int launch_skinny(int epi, const SkinnyParams& p, hipStream_t s) {
    dim3 grid(p.N / 16), block(SK_NW * 64);
    switch (epi) {
        case M5_EPI_F32: hipLaunchKernelGGL((skinny_gemm_kernel<T, M5_EPI_F32, MT>), grid, block, 0, s, p); break;
        case M5_EPI_DT: hipLaunchKernelGGL((skinny_gemm_kernel<T, M5_EPI_DT, MT>), grid, block, 0, s, p); break;
        case M5_EPI_RESIDUAL: hipLaunchKernelGGL((skinny_gemm_kernel<T, M5_EPI_RESIDUAL, MT>), grid, block, 0, s, p); break;
        case M5_EPI_SWIGLU: hipLaunchKernelGGL((skinny_gemm_kernel<T, M5_EPI_SWIGLU, MT>), grid, block, 0, s, p); break;
        default: return M5_ERR_UNSUPPORTED;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}

}  // namespace

// Does this call fit the skinny kernel?  (m5_gemm asks before choosing a wide-tile configuration.)
bool m5_gemm_skinny_fits(int dtype, int M, int N, int K, int epi, int batch, int64_t lda, int64_t ldw) {
    if (dtype != M5_F16 && dtype != M5_BF16) return false;
    if (batch != 1 || M > 32 || (N % 16) || (K % 32) || K > 32 * SK_KSW * SK_NW) return false;
    if (epi != M5_EPI_F32 && epi != M5_EPI_DT && epi != M5_EPI_RESIDUAL && epi != M5_EPI_SWIGLU) return false;
    if ((lda % 8) || (ldw % 8)) return false;
    const char* e = getenv("M5_GEMM_SKINNY");                // tuning / A-B only
    return !(e && e[0] == '0');
}

int m5_gemm_skinny_dispatch(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                            void* C, int64_t ldc, int M, int N, int K, int epi, hipStream_t s) {
    SkinnyParams p{(const unsigned char*)A, (const unsigned char*)W, bias, (unsigned char*)C, lda, ldw, ldc, M, N, K};
    if (dtype == M5_F16) return M <= 16 ? launch_skinny<F16T, 1>(epi, p, s) : launch_skinny<F16T, 2>(epi, p, s);
    return M <= 16 ? launch_skinny<BF16T, 1>(epi, p, s) : launch_skinny<BF16T, 2>(epi, p, s);
}
