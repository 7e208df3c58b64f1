// 16-bit-operand MFMA GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T  (+ fused epilogue).
//
// What bounds the NAR decoder GEMMs (M ~ 2.8k rows, N 1k-6k, K 1k-3k) is not the matrix
// pipe but where the operand panels come from: a 128x128 tile re-reads its A and W
// K-panels in full, so a launch moves ~30x the unique bytes.  Hence, in this order:
//   * XCD-aware, grouped tile order.  Workgroup b runs on XCD b % 8 (observed dispatch
//     order; only speed depends on it).  The 1-D grid is remapped so that each XCD owns a
//     contiguous run of tiles and walks it down GROUP_M tile-rows before stepping to the
//     next tile-column: the ~64 tiles resident on one XCD at any time form a compact patch
//     whose A / W K-slabs are shared through that XCD's private 4 MiB L2 instead of being
//     refetched over the fabric.
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no
//     VGPR round trip), two 32 KiB stages, one barrier per 64-deep K-step, the next stage
//     in flight under the current stage's MFMAs; two workgroups per CU interleave.
//     The DMA writes lane-linear, so the bank-conflict swizzle (16-byte chunk index XOR
//     row & 7 inside each 128-byte row) is applied to the per-lane SOURCE address and
//     again on the fragment reads (same involution on both sides).
//   * operands swapped on the MFMA (A-operand = W rows, B-operand = A rows): in the C/D
//     layout each lane then owns FOUR CONSECUTIVE OUTPUT COLUMNS of one row, so every
//     epilogue is 16-byte (fp32) / 8-byte (16-bit) vector I/O: bias, in-place residual
//     (all loads issued before the first store), SwiGLU on interleaved rows (the pair is
//     lane-local), bias+SiLU, head-major Q/K scatter.  V^T blocks (s contiguous) flip the
//     operand order back (block-uniform) so the four consecutive elements run along s.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 b8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;

namespace {

constexpr int BM = 128, BN = 128, BKB = 128;      // BKB: K-step in BYTES per row (64 halves)
constexpr int STAGE = (BM + BN) * BKB;            // 32 KiB
constexpr int GROUP_M = 8;

struct Gemm16Params {
    const unsigned char* A; const unsigned char* W; const float* bias; unsigned char* C;
    int64_t lda, ldw, ldc;           // elements
    int64_t sA, sW, sC, sBias;       // batch strides, elements
    int M, N, K;
    int tilesM, tilesN, nblk;
    int vec_c;                       // C rows / batch stride / base allow 16-byte (fp32) or 8-byte (16-bit) vectors
    int vt_vec;                      // V^T scatter may store 4 consecutive s as one 8-byte vector
    M5QkvScatter sc;
    int sec_kind[3];
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

// LDS-DMA, 16 bytes per lane: lane l of the wave writes LDS[lds_base + 16 l .. +16) from its own
// global address.  Issued from inline asm so that hipcc does not count it: with the builtin the
// compiler drains vmcnt(0) before the first ds_read that follows (it cannot prove the DMA's LDS
// write does not alias), which would serialise the prefetch of stage k+1 with the MFMAs of
// stage k.  The kernel waits for its own DMAs explicitly (s_waitcnt vmcnt(0) + barrier).
// M0 (the DMA's LDS base) is compiler-reserved: saved and restored inside the statement.
__device__ inline void glds16(const unsigned char* gsrc, uint32_t lds_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

template <typename T>
__device__ inline uint2 pack4(const float v[4]) {
    using st = typename T::storage;
    st t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = T::from_f32(v[r]);
    return *reinterpret_cast<const uint2*>(t);
}

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm16_kernel(Gemm16Params p) {
    using st = typename T::storage;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- workgroup -> tile: XCD-contiguous runs (bijective for any nblk), grouped order
    int t;
    {
        const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7, j = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tiles = p.tilesM * p.tilesN;
    const int bz = t / tiles;
    t -= bz * tiles;
    int tm, tn;
    {
        const int width = GROUP_M * p.tilesN;
        const int g = t / width, first = g * GROUP_M;
        const int gsz = min(p.tilesM - first, GROUP_M);
        const int w = t - g * width;
        tm = first + w % gsz;
        tn = w / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const unsigned char* A = p.A + (int64_t)bz * p.sA * 2;
    const unsigned char* W = p.W + (int64_t)bz * p.sW * 2;

    // ---- LDS-DMA staging: wave w fills rows [32w, 32w+32) of both operand tiles, 8 rows
    // (1 KiB) per instruction; lane l lands at row 8j + (l >> 3), chunk slot l & 7, and
    // fetches source chunk (l & 7) ^ (l >> 3)  (row & 7 == l >> 3 because 8 | row base).
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const unsigned char* ga[4];
    const unsigned char* gw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ra = min(m0 + wave * 32 + j * 8 + srow, p.M - 1);
        const int rw = min(n0 + wave * 32 + j * 8 + srow, p.N - 1);
        ga[j] = A + (int64_t)ra * p.lda * 2 + schunk * 16;
        gw[j] = W + (int64_t)rw * p.ldw * 2 + schunk * 16;
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto stage_load = [&](int stage, int kt) {
        const uint32_t sa = lds_base + stage * STAGE + wave * 32 * BKB;
        const uint32_t sw = sa + BM * BKB;
        const int64_t koff = (int64_t)kt * BKB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(ga[j] + koff, sa + j * 8 * BKB);
            glds16(gw[j] + koff, sw + j * 8 * BKB);
        }
    };

    // fragment read offsets: row (16 i + l15), chunk (4 ks + lg) ^ (l15 & 7)
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = l15 * BKB + (((ks * 4 + lg) ^ (l15 & 7)) << 4);
    const int a_row0 = wm * 64 * BKB, w_row0 = BM * BKB + wn * 64 * BKB;

    f4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

    // V^T blocks keep (row-major m along registers): block-uniform
    bool vblock = false;
    const int Dm = p.sc.n_heads * p.sc.head_dim;
    if constexpr (EPI == M5_EPI_QKV) vblock = p.sec_kind[min(n0 / Dm, 2)] == 2;

    const int nk = p.K / 64;
    stage_load(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage_load((kt + 1) & 1, kt + 1);
        const unsigned char* sb = lds + (kt & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const uint4*>(sb + a_row0 + i * 16 * BKB + foff[ks]);
                bf[i] = *reinterpret_cast<const uint4*>(sb + w_row0 + i * 16 * BKB + foff[ks]);
            }
            if (EPI == M5_EPI_QKV && vblock) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(af[i], bf[j], acc[i][j]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(bf[j], af[i], acc[i][j]);
            }
        }
    }

    // ---- epilogue.  swapped layout: acc[i][j][r] = C[m0+wm*64+16i+l15][n0+wn*64+16j+4lg+r]
    const float* bias = p.bias ? p.bias + (int64_t)bz * p.sBias : nullptr;
    constexpr bool F32OUT = (EPI == M5_EPI_F32 || EPI == M5_EPI_RESIDUAL);
    unsigned char* Cb = p.C + (int64_t)bz * p.sC * (F32OUT ? 4 : 2);

    if (EPI == M5_EPI_QKV && vblock) {
        // unswapped layout: acc[i][j][r] = C[m0+wm*64+16i+4lg+r][n0+wn*64+16j+l15]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + l15;
            if (col >= p.N) continue;
            const float bv = bias ? bias[col] : 0.f;
            const int c = col % Dm, hh = c / p.sc.head_dim, dd = c % p.sc.head_dim;
            st* vt = reinterpret_cast<st*>(p.sc.vt) + hh * p.sc.vt_hs + (int64_t)dd * p.sc.vt_ds;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + wm * 64 + i * 16 + lg * 4;
                if (row >= p.M) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv;
                const int b = row / p.sc.rows_per_batch, s = row - b * p.sc.rows_per_batch;
                if (p.vt_vec && row + 3 < p.M) {
                    *reinterpret_cast<uint2*>(vt + b * p.sc.vt_bs + s) = pack4<T>(v);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rr = row + r;
                        if (rr < p.M) {
                            const int b2 = rr / p.sc.rows_per_batch, s2 = rr - b2 * p.sc.rows_per_batch;
                            vt[b2 * p.sc.vt_bs + s2] = T::from_f32(v[r]);
                        }
                    }
                }
            }
        }
        return;
    }

    const int nb0 = n0 + wn * 64 + lg * 4;
    float4 oldv[4][4];
    if constexpr (EPI == M5_EPI_RESIDUAL) {
        // issue every read of C before the first write (the compiler cannot reorder a load
        // above a possibly-aliasing store, which would serialise 16 round trips)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + wm * 64 + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = nb0 + j * 16;
                oldv[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < p.M && col < p.N) {
                    const float* cp = reinterpret_cast<const float*>(Cb) + (int64_t)row * p.ldc + col;
                    if (p.vec_c && col + 3 < p.N) {
                        oldv[i][j] = *reinterpret_cast<const float4*>(cp);
                    } else {
                        oldv[i][j].x = cp[0];
                        if (col + 1 < p.N) oldv[i][j].y = cp[1];
                        if (col + 2 < p.N) oldv[i][j].z = cp[2];
                        if (col + 3 < p.N) oldv[i][j].w = cp[3];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = nb0 + j * 16;
        if (col >= p.N) continue;
        const bool full = col + 3 < p.N;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = bias[min(col + r, p.N - 1)];
        }
        int kind = 0, hh = 0, dd = 0;
        if constexpr (EPI == M5_EPI_QKV) {
            kind = p.sec_kind[min(col / Dm, 2)];
            const int c = col % Dm;
            hh = c / p.sc.head_dim;
            dd = c % p.sc.head_dim;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + wm * 64 + i * 16 + l15;
            if (row >= p.M) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[r];
            if constexpr (EPI == M5_EPI_F32 || EPI == M5_EPI_RESIDUAL) {
                float* cp = reinterpret_cast<float*>(Cb) + (int64_t)row * p.ldc + col;
                if constexpr (EPI == M5_EPI_RESIDUAL) {
                    v[0] = oldv[i][j].x + v[0]; v[1] = oldv[i][j].y + v[1];
                    v[2] = oldv[i][j].z + v[2]; v[3] = oldv[i][j].w + v[3];
                }
                if (p.vec_c && full) {
                    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < p.N) cp[r] = v[r];
                }
            } else if constexpr (EPI == M5_EPI_DT || EPI == M5_EPI_SILU_DT) {
                if constexpr (EPI == M5_EPI_SILU_DT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
                }
                st* cp = reinterpret_cast<st*>(Cb) + (int64_t)row * p.ldc + col;
                if (p.vec_c && full) {
                    *reinterpret_cast<uint2*>(cp) = pack4<T>(v);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < p.N) cp[r] = T::from_f32(v[r]);
                }
            } else if constexpr (EPI == M5_EPI_SWIGLU) {
                // rows of W interleaved (W_i, V_i): columns (col, col+1) and (col+2, col+3) are pairs
                st o[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float a = round_dt<T>(v[2 * h]), b = round_dt<T>(v[2 * h + 1]);
                    const float s = round_dt<T>(silu_f(a));
                    o[h] = T::from_f32(s * b);
                }
                st* cp = reinterpret_cast<st*>(Cb) + (int64_t)row * p.ldc + (col >> 1);
                if (full) {
                    *reinterpret_cast<uint32_t*>(cp) = *reinterpret_cast<const uint32_t*>(o);
                } else {
                    if (col + 1 < p.N) cp[0] = o[0];
                }
            } else if constexpr (EPI == M5_EPI_QKV) {
                const int b = row / p.sc.rows_per_batch, s = row - b * p.sc.rows_per_batch;
                st* dst = (kind == 0)
                    ? reinterpret_cast<st*>(p.sc.q) + b * p.sc.q_bs + hh * p.sc.q_hs + (int64_t)s * p.sc.q_rs + dd
                    : reinterpret_cast<st*>(p.sc.k) + b * p.sc.k_bs + hh * p.sc.k_hs + (int64_t)s * p.sc.k_rs + dd;
                *reinterpret_cast<uint2*>(dst) = pack4<T>(v);   // Dm % 4 == 0: a 4-group never straddles N or a head
            }
        }
    }
}

template <typename T>
int launch16(int epi, const Gemm16Params& p, hipStream_t s) {
    const dim3 grid(p.nblk), blk(256);
    switch (epi) {
        case M5_EPI_F32: hipLaunchKernelGGL((gemm16_kernel<T, M5_EPI_F32>), grid, blk, 0, s, p); break;
        case M5_EPI_DT: hipLaunchKernelGGL((gemm16_kernel<T, M5_EPI_DT>), grid, blk, 0, s, p); break;
        case M5_EPI_RESIDUAL: hipLaunchKernelGGL((gemm16_kernel<T, M5_EPI_RESIDUAL>), grid, blk, 0, s, p); break;
        case M5_EPI_SWIGLU: hipLaunchKernelGGL((gemm16_kernel<T, M5_EPI_SWIGLU>), grid, blk, 0, s, p); break;
        case M5_EPI_QKV: hipLaunchKernelGGL((gemm16_kernel<T, M5_EPI_QKV>), grid, blk, 0, s, p); break;
        case M5_EPI_SILU_DT: hipLaunchKernelGGL((gemm16_kernel<T, M5_EPI_SILU_DT>), grid, blk, 0, s, p); break;
        default: return M5_ERR_ARG;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}

}  // namespace

// Called by m5_gemm (gemm.hip) for F16 / BF16 operands after argument validation.
int m5_gemm16_dispatch(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                       void* C, int64_t ldc, int M, int N, int K, int epi, const M5QkvScatter* sc, const int* sec_kind,
                       int batch, int64_t sA, int64_t sW, int64_t sC, int64_t sBias, hipStream_t s) {
    Gemm16Params p{};
    p.A = (const unsigned char*)A; p.W = (const unsigned char*)W; p.bias = bias; p.C = (unsigned char*)C;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.sA = sA; p.sW = sW; p.sC = sC; p.sBias = sBias;
    p.M = M; p.N = N; p.K = K;
    p.tilesM = (M + BM - 1) / BM; p.tilesN = (N + BN - 1) / BN;
    const int64_t nblk = (int64_t)p.tilesM * p.tilesN * batch;
    if (nblk > 0x7fffffff) return M5_ERR_UNSUPPORTED;
    p.nblk = (int)nblk;
    const bool f32out = (epi == M5_EPI_F32 || epi == M5_EPI_RESIDUAL);
    const int cal = f32out ? 15 : 7;
    p.vec_c = (C && (ldc % 4 == 0) && (sC % 4 == 0) && (((uintptr_t)C & cal) == 0)) ? 1 : 0;
    if (sc) {
        p.sc = *sc;
        for (int i = 0; i < 3; ++i) p.sec_kind[i] = sec_kind[i];
        const bool one_batch = M <= sc->rows_per_batch;
        p.vt_vec = (sc->vt && (sc->vt_ds % 4 == 0) && (sc->vt_hs % 4 == 0) && (sc->vt_bs % 4 == 0) &&
                    (((uintptr_t)sc->vt & 7) == 0) && (one_batch || sc->rows_per_batch % 4 == 0)) ? 1 : 0;
        // Q / K scatter stores 4 consecutive d as 8 bytes
        if (sc->head_dim % 4 || (sc->q && ((sc->q_rs % 4) || (sc->q_hs % 4) || (sc->q_bs % 4) || ((uintptr_t)sc->q & 7))) ||
            (sc->k && ((sc->k_rs % 4) || (sc->k_hs % 4) || (sc->k_bs % 4) || ((uintptr_t)sc->k & 7))))
            return M5_ERR_UNSUPPORTED;
    } else {
        p.sc.n_heads = 1; p.sc.head_dim = 1; p.sc.rows_per_batch = 1;
    }
    if (dtype == M5_F16) return launch16<F16T>(epi, p, s);
    return launch16<BF16T>(epi, p, s);
}
