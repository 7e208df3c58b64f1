// MFMA GEMM  C[M,N] = A[M,K] . W[N,K]^T  for gfx950 (wave64).
//
// Tile 128x128 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA 16x16
// tiles, 64 fp32 accumulators per lane).  Both operands are K-contiguous, so a 128-byte
// K-slab of every row (BK = 64 halves or 32 floats = one full cache line) is staged
// through LDS in rows padded to 144 B (conflict-light ds_read_b128 fragment reads), with
// the next slab prefetched into registers while the current one feeds the matrix cores
// (global -> reg -> LDS split staging).  f16/bf16: v_mfma_f32_16x16x32;  f32 (parity
// mode): v_mfma_f32_16x16x4_f32, bitwise an fmaf chain.
// Fused epilogues: bias, in-place residual, SwiGLU on interleaved rows, head-major
// Q/K/V^T scatter (feeds attention.hip with no transpose pass), bias+SiLU.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 b8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 144;   // LDS row stride in bytes: 128 data + 16 pad

struct GemmParams {
    const unsigned char* A; const unsigned char* W; const float* bias; unsigned char* C;
    int64_t lda, ldw, ldc;           // elements
    int64_t sA, sW, sC, sBias;       // batch strides, elements
    int M, N, K;
    M5QkvScatter sc;
    int sec_kind[3];                 // column section -> 0 q, 1 k, 2 v
};

template <typename T>
__device__ inline f4_t mfma16(const uint4& a, const uint4& b, f4_t c);
template <>
__device__ inline f4_t mfma16<F16T>(const uint4& a, const uint4& b, f4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8_t*>(&a),
                                                  *reinterpret_cast<const h8_t*>(&b), c, 0, 0, 0);
}
template <>
__device__ inline f4_t mfma16<BF16T>(const uint4& a, const uint4& b, f4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const b8_t*>(&a),
                                                   *reinterpret_cast<const b8_t*>(&b), c, 0, 0, 0);
}

// ---- X3: fp32 operands, products on the f16 matrix pipe at fp32-grade accuracy ("f16 x 3").
// Every fp32 operand x is split when its slab is staged into LDS:  hi = f16(x) (11 significant bits, round to nearest),
// lo = f16((x - hi) * 2^11) (the next 11 bits, pre-scaled into f16's normal range), so x = hi + lo 2^-11 up to 2^-22 |x|.
// a w = hi_a hi_w + 2^-11 (hi_a lo_w + lo_a hi_w) + O(2^-22 a w): three v_mfma_f32_16x16x32_f16 per fragment pair -- the f16
// products are exact in fp32 (11 x 11 bits) and accumulate in fp32; the two cross terms share a second accumulator that is
// scaled by 2^-11 once, behind the K loop.  48 MFMAs of 16 pipe cycles per 32-deep slab and wave against 128 MFMAs of 32
// cycles for v_mfma_f32_16x16x4_f32: the matrix pipe does the slab in a fifth of the time, with the operand error (2^-22
// relative, random sign) at the level of fp32 rounding noise of a K = 1024 dot product.  NOT bitwise an fmaf chain: this is
// the fast parity-grade mode (dtype M5_F32X3), the exact kernel stays the reference instrument (M5_F32).
// Range.  f16 spans 2^-14 .. 65504 with full precision; the operands are pre-scaled by exact powers of two (activations x 2^4,
// weights x 2^8, undone behind the K loop) so that the 22-bit representation holds for |a| in 3.8e-6 .. 4094 and |w| in
// 2.4e-7 .. 255 -- LayerNorm outputs, attention outputs, SwiGLU products and weight matrices with room on both sides.  Smaller
// operands lose significance gradually (hi = 0 below the range: the conversions run with f16 denormals flushed, MODE.fp_denorm,
// so no denormal ever reaches the matrix pipe; the value survives in lo with 11 bits, absolute error below 2^-37 for
// activations); LARGER ones overflow to inf / NaN visibly -- use M5_F32 for such data.
constexpr float X3_SA = 16.0f, X3_SW = 256.0f;
__device__ inline void x3_split(const uint4& v, float scale, uint2& hi, uint2& lo) {
    const float x[4] = {__uint_as_float(v.x) * scale, __uint_as_float(v.y) * scale, __uint_as_float(v.z) * scale, __uint_as_float(v.w) * scale};
    _Float16 h[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        h[r] = (_Float16)x[r];
        l[r] = (_Float16)((x[r] - (float)h[r]) * 2048.0f);
    }
    hi = *reinterpret_cast<const uint2*>(h);
    lo = *reinterpret_cast<const uint2*>(l);
}

// TMW = 16-row MFMA tiles per wave along M: 4 (tile 128 x 128) or -- X3 only -- 2 (tile 64 x 128, three workgroups per CU: the
// N = 1024 GEMMs of a 2816-row NAR step are 176 tiles of 128 x 128 on 256 CUs, 352 of 64 x 128).
template <typename T, int EPI, bool X3 = false, int TMW = 4>
__global__ __launch_bounds__(256, X3 ? (TMW == 2 ? 3 : 2) : 1) void gemm_kernel(GemmParams p) {
    using st = typename T::storage;
    constexpr int ES = sizeof(st);
    constexpr int BK = 128 / ES;
    constexpr int BMK = 2 * TMW * 16;             // rows of the workgroup tile (2 waves along M)
    constexpr int NPA = BMK / 32;                 // staging passes over the A rows (32 rows per pass)
    static_assert(!X3 || ES == 4, "X3: fp32 operands");
    static_assert(TMW == 4 || (X3 && TMW == 2), "64-row tiles: X3 only");
    if constexpr (X3) __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11) /* hwreg(HW_REG_MODE, offset 6, width 2): FP_DENORM of f16 / f64 */, 0);   // flush f16 denormals in the conversions
    __shared__ __attribute__((aligned(16))) unsigned char lds[(BMK + 128) * ROWB];
    unsigned char* As = lds;
    unsigned char* Ws = lds + BMK * ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int m0 = blockIdx.y * BMK, n0 = blockIdx.x * BN;
    const int64_t bz = blockIdx.z;
    const unsigned char* A = p.A + bz * p.sA * ES;
    const unsigned char* W = p.W + bz * p.sW * ES;

    // staging assignment: 8 threads cover one 128-byte row slab, 32 rows per pass
    const int chunk = tid & 7, r0 = tid >> 3;
    const unsigned char* ga[NPA];
    const unsigned char* gw[4];
#pragma unroll
    for (int j = 0; j < NPA; ++j) ga[j] = A + (int64_t)min(m0 + r0 + 32 * j, p.M - 1) * p.lda * ES + chunk * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) gw[j] = W + (int64_t)min(n0 + r0 + 32 * j, p.N - 1) * p.ldw * ES + chunk * 16;
    const int lds_st = r0 * ROWB + chunk * 16;

    f4_t acc[TMW][4];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

    f4_t accc[X3 ? TMW : 1][X3 ? 4 : 1];               // X3: the cross terms hi lo + lo hi (scaled by 2^-11 at the end)
    if constexpr (X3) {
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) accc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
    }
    // stage one register slab into LDS.  X3: a row's 32 floats become 32 hi halves (bytes 0..63) + 32 lo halves (64..127)
    auto stage_write = [&](const uint4 (&qa)[NPA], const uint4 (&qw)[4]) {
        if constexpr (X3) {
            uint2 h, l;
#pragma unroll
            for (int j = 0; j < NPA; ++j) {
                x3_split(qa[j], X3_SA, h, l);
                *reinterpret_cast<uint2*>(As + (r0 + 32 * j) * ROWB + chunk * 8) = h;
                *reinterpret_cast<uint2*>(As + (r0 + 32 * j) * ROWB + 64 + chunk * 8) = l;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x3_split(qw[j], X3_SW, h, l);
                *reinterpret_cast<uint2*>(Ws + (r0 + 32 * j) * ROWB + chunk * 8) = h;
                *reinterpret_cast<uint2*>(Ws + (r0 + 32 * j) * ROWB + 64 + chunk * 8) = l;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<uint4*>(As + lds_st + 32 * j * ROWB) = qa[j];
                *reinterpret_cast<uint4*>(Ws + lds_st + 32 * j * ROWB) = qw[j];
            }
        }
    };

    uint4 ra[NPA], rw[4];
    const int nk = p.K / BK;
#pragma unroll
    for (int j = 0; j < NPA; ++j) ra[j] = *reinterpret_cast<const uint4*>(ga[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) rw[j] = *reinterpret_cast<const uint4*>(gw[j]);
    stage_write(ra, rw);
    __syncthreads();

    const unsigned char* a_base = As + (wm * TMW * 16 + l15) * ROWB;
    const unsigned char* w_base = Ws + (wn * 64 + l15) * ROWB;

    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {
            const int64_t koff = (int64_t)(kt + 1) * 128;   // bytes
#pragma unroll
            for (int j = 0; j < NPA; ++j) ra[j] = *reinterpret_cast<const uint4*>(ga[j] + koff);
#pragma unroll
            for (int j = 0; j < 4; ++j) rw[j] = *reinterpret_cast<const uint4*>(gw[j] + koff);
        }
        if constexpr (X3) {
            uint4 ahi[TMW], alo[TMW], whi[4], wlo[4];
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                ahi[i] = *reinterpret_cast<const uint4*>(a_base + i * 16 * ROWB + lg * 16);
                alo[i] = *reinterpret_cast<const uint4*>(a_base + i * 16 * ROWB + 64 + lg * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                whi[i] = *reinterpret_cast<const uint4*>(w_base + i * 16 * ROWB + lg * 16);
                wlo[i] = *reinterpret_cast<const uint4*>(w_base + i * 16 * ROWB + 64 + lg * 16);
            }
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = mfma16<F16T>(ahi[i], whi[j], acc[i][j]);
                    accc[i][j] = mfma16<F16T>(ahi[i], wlo[j], accc[i][j]);
                    accc[i][j] = mfma16<F16T>(alo[i], whi[j], accc[i][j]);
                }
        } else if constexpr (ES == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 af[4], bf[4];
                const int ko = (ks * 32 + lg * 8) * 2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = *reinterpret_cast<const uint4*>(a_base + i * 16 * ROWB + ko);
                    bf[i] = *reinterpret_cast<const uint4*>(w_base + i * 16 * ROWB + ko);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(af[i], bf[j], acc[i][j]);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                float af[4], bf[4];
                const int ko = (kk * 4 + lg) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = *reinterpret_cast<const float*>(a_base + i * 16 * ROWB + ko);
                    bf[i] = *reinterpret_cast<const float*>(w_base + i * 16 * ROWB + ko);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) {
            stage_write(ra, rw);
            __syncthreads();
        }
    }
    if constexpr (X3) {
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = (acc[i][j][r] + accc[i][j][r] * (1.0f / 2048.0f)) * (1.0f / (X3_SA * X3_SW));
    }

    // ---- epilogue: acc[i][j][r] is C[row = m0+wm*TMW*16+i*16+lg*4+r][col = n0+wn*64+j*16+l15]
    const float* bias = p.bias ? p.bias + bz * p.sBias : nullptr;
    unsigned char* Cb = p.C + bz * p.sC * ((EPI == M5_EPI_F32 || EPI == M5_EPI_RESIDUAL) ? 4 : ES);
    const int Dm = p.sc.n_heads * p.sc.head_dim;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + l15;
        const bool cok = col < p.N;
        const float bv = (bias && cok) ? bias[col] : 0.f;
        // QKV scatter: per-column decomposition hoisted out of the row loops
        int kind = 0, hh = 0, dd = 0;
        if constexpr (EPI == M5_EPI_QKV) {
            const int cc = cok ? col : 0;
            kind = p.sec_kind[cc / Dm];
            const int c = cc % Dm;
            hh = c / p.sc.head_dim;
            dd = c % p.sc.head_dim;
        }
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * TMW * 16 + i * 16 + lg * 4 + r;
                const bool ok = cok && row < p.M;
                const float v = acc[i][j][r] + bv;
                if constexpr (EPI == M5_EPI_F32) {
                    if (ok) reinterpret_cast<float*>(Cb)[(int64_t)row * p.ldc + col] = v;
                } else if constexpr (EPI == M5_EPI_RESIDUAL) {
                    if (ok) reinterpret_cast<float*>(Cb)[(int64_t)row * p.ldc + col] += v;
                } else if constexpr (EPI == M5_EPI_DT) {
                    if (ok) reinterpret_cast<st*>(Cb)[(int64_t)row * p.ldc + col] = T::from_f32(v);
                } else if constexpr (EPI == M5_EPI_SILU_DT) {
                    if (ok) reinterpret_cast<st*>(Cb)[(int64_t)row * p.ldc + col] = T::from_f32(silu_f(v));
                } else if constexpr (EPI == M5_EPI_SWIGLU) {
                    const float other = __shfl_xor(v, 1);     // odd column (V_i) of the pair
                    if (ok && !(col & 1)) {
                        const float a = round_dt<T>(v), b = round_dt<T>(other);
                        const float s = round_dt<T>(silu_f(a));
                        reinterpret_cast<st*>(Cb)[(int64_t)row * p.ldc + (col >> 1)] = T::from_f32(s * b);
                    }
                } else if constexpr (EPI == M5_EPI_QKV) {
                    if (ok) {
                        const int b = row / p.sc.rows_per_batch, s = row - b * p.sc.rows_per_batch;
                        const st o = T::from_f32(v);
                        if (kind == 0)
                            reinterpret_cast<st*>(p.sc.q)[b * p.sc.q_bs + hh * p.sc.q_hs + (int64_t)s * p.sc.q_rs + dd] = o;
                        else if (kind == 1)
                            reinterpret_cast<st*>(p.sc.k)[b * p.sc.k_bs + hh * p.sc.k_hs + (int64_t)s * p.sc.k_rs + dd] = o;
                        else
                            reinterpret_cast<st*>(p.sc.vt)[b * p.sc.vt_bs + hh * p.sc.vt_hs + (int64_t)dd * p.sc.vt_ds + s] = o;
                    }
                }
            }
        }
    }
}

template <typename T, bool X3 = false, int TMW = 4>
int launch_gemm(int epi, const GemmParams& p, dim3 grid, hipStream_t s) {
    switch (epi) {
        case M5_EPI_F32: hipLaunchKernelGGL((gemm_kernel<T, M5_EPI_F32, X3, TMW>), grid, dim3(256), 0, s, p); break;
        case M5_EPI_DT: hipLaunchKernelGGL((gemm_kernel<T, M5_EPI_DT, X3, TMW>), grid, dim3(256), 0, s, p); break;
        case M5_EPI_RESIDUAL: hipLaunchKernelGGL((gemm_kernel<T, M5_EPI_RESIDUAL, X3, TMW>), grid, dim3(256), 0, s, p); break;
        case M5_EPI_SWIGLU: hipLaunchKernelGGL((gemm_kernel<T, M5_EPI_SWIGLU, X3, TMW>), grid, dim3(256), 0, s, p); break;
        case M5_EPI_QKV: hipLaunchKernelGGL((gemm_kernel<T, M5_EPI_QKV, X3, TMW>), grid, dim3(256), 0, s, p); break;
        case M5_EPI_SILU_DT: hipLaunchKernelGGL((gemm_kernel<T, M5_EPI_SILU_DT, X3, TMW>), grid, dim3(256), 0, s, p); break;
        default: return M5_ERR_ARG;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}

}  // namespace

int m5_gemm16_dispatch(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                       void* C, int64_t ldc, int M, int N, int K, int epi, const M5QkvScatter* sc, const int* sec_kind,
                       int batch, int64_t sA, int64_t sW, int64_t sC, int64_t sBias, hipStream_t s, const M5DeferredLN* dl = nullptr,
                       const M5RowTiles* rt = nullptr);   // gemm16.hip
bool m5_gemm_skinny_fits(int dtype, int M, int N, int K, int epi, int batch, int64_t lda, int64_t ldw);   // gemm_skinny.hip
int m5_gemm_skinny_dispatch(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                            void* C, int64_t ldc, int M, int N, int K, int epi, hipStream_t s);

static bool use_v1_gemm() {   // M5_GEMM_V1=1: A/B the first-generation register-staged kernel
    static const bool v = [] { const char* e = m5_tool_env("M5_GEMM_V1"); return e && e[0] == '1'; }();
    return v;
}

extern "C" int m5_gemm(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                       void* C, int64_t ldc, int M, int N, int K, int epi, const M5QkvScatter* sc,
                       int batch, int64_t sA, int64_t sW, int64_t sC, int64_t sBias, void* stream) {
    if (!A || !W || M <= 0 || N <= 0 || K <= 0 || batch <= 0) return M5_ERR_ARG;
    if (K % 64 != 0) return M5_ERR_UNSUPPORTED;
    const int es = (dtype == M5_F32 || dtype == M5_F32X3) ? 4 : 2;
    const int al = 16 / es;
    if (lda % al || ldw % al || ((uintptr_t)A & 15) || ((uintptr_t)W & 15)) return M5_ERR_ARG;
    if ((sA % al) || (sW % al)) return M5_ERR_ARG;
    if (epi != M5_EPI_QKV && !C) return M5_ERR_ARG;
    if (epi == M5_EPI_SWIGLU && (N & 1)) return M5_ERR_ARG;
    GemmParams p{};
    p.A = (const unsigned char*)A; p.W = (const unsigned char*)W; p.bias = bias; p.C = (unsigned char*)C;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.sA = sA; p.sW = sW; p.sC = sC; p.sBias = sBias;
    p.M = M; p.N = N; p.K = K;
    if (epi == M5_EPI_QKV) {
        if (!sc || sc->head_dim <= 0 || sc->n_heads <= 0 || sc->rows_per_batch <= 0) return M5_ERR_ARG;
        p.sc = *sc;
        int n = 0;
        if (sc->q) p.sec_kind[n++] = 0;
        if (sc->k) p.sec_kind[n++] = 1;
        if (sc->vt) p.sec_kind[n++] = 2;
        if (n == 0 || N != n * sc->n_heads * sc->head_dim) return M5_ERR_ARG;
    } else {
        p.sc.n_heads = 1; p.sc.head_dim = 1; p.sc.rows_per_batch = 1;
    }
    hipStream_t s = (hipStream_t)stream;
    if (m5_gemm_skinny_fits(dtype, M, N, K, epi, batch, lda, ldw))          // batched decode step: M <= 32 rows
        return m5_gemm_skinny_dispatch(dtype, A, lda, W, ldw, bias, C, ldc, M, N, K, epi, s);
    if (dtype != M5_F32 && dtype != M5_F32X3 && !use_v1_gemm())
        return m5_gemm16_dispatch(dtype, A, lda, W, ldw, bias, C, ldc, M, N, K, epi, epi == M5_EPI_QKV ? &p.sc : nullptr,
                                  p.sec_kind, batch, sA, sW, sC, sBias, s);
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
    switch (dtype) {
        case M5_F32: return launch_gemm<F32T>(epi, p, grid, s);
        case M5_F32X3: {
            // 64-row tiles (three workgroups per CU) when 128-row tiles would leave the machine short of two full rounds of workgroups
            int bm = ((int64_t)grid.x * grid.y * grid.z < 2 * 512) ? 64 : 128;
            if (const char* e = m5_tool_env("M5_X3_BM")) bm = atoi(e);                  // same-process A/B (tools build)
            if (bm == 64) return launch_gemm<F32T, true, 2>(epi, p, dim3((N + BN - 1) / BN, (M + 63) / 64, batch), s);
            return launch_gemm<F32T, true>(epi, p, grid, s);
        }
        case M5_F16: return launch_gemm<F16T>(epi, p, grid, s);
        case M5_BF16: return launch_gemm<BF16T>(epi, p, grid, s);
        default: return M5_ERR_ARG;
    }
}

// m5_gemm with a deferred LayerNorm (M5DeferredLN in include/mars5_hip.h; mode 1 with M5_EPI_RESIDUAL = the producer of the
// rows, mode 2 with M5_EPI_QKV / M5_EPI_SWIGLU = a consumer) and / or over a row-tile list (M5RowTiles).  16-bit operands only.
extern "C" int m5_gemm_ex(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                          void* C, int64_t ldc, int M, int N, int K, int epi, const M5QkvScatter* sc,
                          int batch, int64_t sA, int64_t sW, int64_t sC, int64_t sBias, const M5DeferredLN* dl, const M5RowTiles* rt,
                          void* stream) {
    if (!A || !W || M <= 0 || N <= 0 || K <= 0 || batch <= 0) return M5_ERR_ARG;
    if (dtype != M5_F16 && dtype != M5_BF16) return M5_ERR_UNSUPPORTED;
    if (K % 64 != 0) return M5_ERR_UNSUPPORTED;
    if (lda % 8 || ldw % 8 || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || (sA % 8) || (sW % 8)) return M5_ERR_ARG;
    if (epi != M5_EPI_QKV && !C) return M5_ERR_ARG;
    if (epi == M5_EPI_SWIGLU && (N & 1)) return M5_ERR_ARG;
    M5QkvScatter scv{};
    int sec_kind[3] = {0, 0, 0};
    if (epi == M5_EPI_QKV) {
        if (!sc || sc->head_dim <= 0 || sc->n_heads <= 0 || sc->rows_per_batch <= 0) return M5_ERR_ARG;
        scv = *sc;
        int n = 0;
        if (sc->q) sec_kind[n++] = 0;
        if (sc->k) sec_kind[n++] = 1;
        if (sc->vt) sec_kind[n++] = 2;
        if (n == 0 || N != n * sc->n_heads * sc->head_dim) return M5_ERR_ARG;
    }
    return m5_gemm16_dispatch(dtype, A, lda, W, ldw, bias, C, ldc, M, N, K, epi, epi == M5_EPI_QKV ? &scv : nullptr, sec_kind,
                              batch, sA, sW, sC, sBias, (hipStream_t)stream, dl, rt);
}
