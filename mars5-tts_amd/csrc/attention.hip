// Flash attention for gfx950, head_dim 64, wave64 MFMA.
//
// One workgroup = 4 waves = 64 queries of one (batch, head); each wave owns 16 queries.
// K tiles [64 keys][64 d] and V^T tiles [64 d][64 keys] are staged through LDS (rows
// padded by 16 B) with the next tile prefetched into registers during compute.
//
// The score tile is computed TRANSPOSED (S^T = K.Q^T: A operand = K rows from LDS, B
// operand = Q held in registers for the whole loop), so in the MFMA C layout each lane
// holds, for ITS query (lane & 15), four consecutive keys per 16-key subtile.  That makes
// the softmax row statistics a 2-step cross-group shuffle, and -- the point -- lets the
// probabilities feed the P.V MFMA's A operand straight from registers: the k-slot -> key
// assignment of a lane group is free as long as A (P) and B (V^T) agree, and V^T rows
// give the B operand the same four consecutive keys with one 8/16-byte LDS read.  No LDS
// round trip for P, no V transpose in the kernel (the producing GEMM writes V^T).
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 b8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;

namespace {

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

template <typename T>
__global__ __launch_bounds__(256) void attn_kernel(M5AttnArgs p) {
    using st = typename T::storage;
    constexpr int ES = sizeof(st);
    constexpr int RB = 64 * ES + 16;          // LDS row stride (bytes)
    constexpr int CPR = 64 * ES / 16;         // 16-byte chunks per tile row
    constexpr int NL = 64 * CPR / 256;        // chunks per thread per tile (2 or 4)
    constexpr int EPC = 16 / ES;              // elements per chunk
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 64 * RB];
    unsigned char* Ks = lds;
    unsigned char* Vs = lds + 64 * RB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int q0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    int kl = p.key_len ? p.key_len[b] : p.Sk;
    kl = min(kl, p.Sk);
    int64_t koff = 0, voff = 0;
    if (p.kv_index) {
        const int64_t ix = *p.kv_index;
        koff = ix * p.kv_index_stride_k;
        voff = ix * p.kv_index_stride_v;
    }
    const unsigned char* Kg = (const unsigned char*)p.k + (koff + b * p.k_bs + h * p.k_hs) * ES;
    const unsigned char* Vg = (const unsigned char*)p.vt + (voff + b * p.vt_bs + h * p.vt_hs) * ES;
    const unsigned char* Qg = (const unsigned char*)p.q + (b * p.q_bs + h * p.q_hs) * ES;

    int ntiles = (kl + 63) / 64;
    if (p.causal) ntiles = min(ntiles, q0 / 64 + 1);

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[query l15][its k-slots]
    const int qrow = min(q0 + wave * 16 + l15, p.Sq - 1);
    const int qpos = q0 + wave * 16 + l15;
    uint4 qf16[2];
    float qf32[16];
    if constexpr (ES == 2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf16[ks] = *reinterpret_cast<const uint4*>(Qg + ((int64_t)qrow * p.q_rs + ks * 32 + lg * 8) * 2);
    } else {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            qf32[kk] = *reinterpret_cast<const float*>(Qg + ((int64_t)qrow * p.q_rs + kk * 4 + lg) * 4);
    }

    // ---- tile staging
    uint4 rk[NL], rv[NL];
    auto load_tile = [&](int kt) {
        const int kbase = kt * 64;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int c = tid + 256 * j, row = c / CPR, ch = c % CPR;
            if (kbase + row < p.Sk)
                rk[j] = *reinterpret_cast<const uint4*>(Kg + ((int64_t)(kbase + row) * p.k_rs) * ES + ch * 16);
            else
                rk[j] = make_uint4(0, 0, 0, 0);
            if (kbase + ch * EPC < p.Sk)
                rv[j] = *reinterpret_cast<const uint4*>(Vg + ((int64_t)row * p.vt_ds + kbase) * ES + ch * 16);
            else
                rv[j] = make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int c = tid + 256 * j, row = c / CPR, ch = c % CPR;
            *reinterpret_cast<uint4*>(Ks + row * RB + ch * 16) = rk[j];
            *reinterpret_cast<uint4*>(Vs + row * RB + ch * 16) = rv[j];
        }
    };

    f4_t oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) oacc[i] = f4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    if (ntiles > 0) {
        load_tile(0);
        store_tile();
    }
    __syncthreads();

    for (int kt = 0; kt < ntiles; ++kt) {
        const bool more = kt + 1 < ntiles;
        if (more) load_tile(kt + 1);
        const int kbase = kt * 64;

        // ---- S^T tile: sacc[j][r] = score(key kbase+16j+4lg+r, query l15)
        f4_t sacc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sacc[j] = f4_t{0.f, 0.f, 0.f, 0.f};
            const unsigned char* krow = Ks + (16 * j + l15) * RB;
            if constexpr (ES == 2) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const uint4 a = *reinterpret_cast<const uint4*>(krow + (ks * 32 + lg * 8) * 2);
                    sacc[j] = mfma16<T>(a, qf16[ks], sacc[j]);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const float a = *reinterpret_cast<const float*>(krow + (kk * 4 + lg) * 4);
                    sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qf32[kk], sacc[j], 0, 0, 0);
                }
            }
        }
        // ---- scale, mask, online softmax (statistics per query = per l15 column)
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kidx = kbase + 16 * j + 4 * lg + r;
                const bool vis = kidx < kl && (!p.causal || kidx <= qpos);
                const float s = vis ? sacc[j][r] * p.scale : -INFINITY;
                sacc[j][r] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = expf(m_run - m_use);      // m_run = -inf -> 0
        float ps = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf(sacc[j][r] - m_use);   // masked -> exp(-inf) = 0
                sacc[j][r] = e;
                ps += e;
            }
        ps += __shfl_xor(ps, 16);
        ps += __shfl_xor(ps, 32);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        // ---- rescale O: its rows are queries 4lg+r, whose alpha lives in lane 4lg+r
        float ar[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * lg + r);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[dt][r] *= ar[r];
        // ---- O += P.V   (A = P from registers, B = V^T rows from LDS)
        if constexpr (ES == 2) {
            uint4 pa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                st tmp[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    tmp[r] = T::from_f32(sacc[2 * i][r]);
                    tmp[4 + r] = T::from_f32(sacc[2 * i + 1][r]);
                }
                pa[i] = *reinterpret_cast<const uint4*>(tmp);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const unsigned char* vrow = Vs + (16 * dt + l15) * RB;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint2 lo = *reinterpret_cast<const uint2*>(vrow + (16 * (2 * i) + 4 * lg) * 2);
                    const uint2 hi = *reinterpret_cast<const uint2*>(vrow + (16 * (2 * i + 1) + 4 * lg) * 2);
                    const uint4 bv = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    oacc[dt] = mfma16<T>(pa[i], bv, oacc[dt]);
                }
            }
        } else {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const unsigned char* vrow = Vs + (16 * dt + l15) * RB;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 v4 = *reinterpret_cast<const float4*>(vrow + (16 * j + 4 * lg) * 4);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sacc[j][0], v4.x, oacc[dt], 0, 0, 0);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sacc[j][1], v4.y, oacc[dt], 0, 0, 0);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sacc[j][2], v4.z, oacc[dt], 0, 0, 0);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sacc[j][3], v4.w, oacc[dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- normalise and store: oacc[dt][r] = O[query q0+wave*16+4lg+r][d = 16dt+l15]
    float lr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) lr[r] = __shfl(l_run, 4 * lg + r);
    st* Og = reinterpret_cast<st*>(p.o) + b * p.o_bs + h * 64;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qr = q0 + wave * 16 + 4 * lg + r;
        if (qr < p.Sq) {
            const float inv = 1.0f / lr[r];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) Og[(int64_t)qr * p.o_rs + 16 * dt + l15] = T::from_f32(oacc[dt][r] * inv);
        }
    }
}


// ---- X3: fp32 Q / K / V^T / O in memory, both products on the f16 matrix pipe as three split-operand terms (gemm.hip "X3":
// x = hi + lo 2^-11 with hi = f16(x s), lo = f16((x s - hi) 2^11), s a power of two that is undone exactly; products
// hi hi + 2^-11 (hi lo + lo hi), fp32 accumulation; operand error 2^-22).  Same tiling, key order, masking and softmax
// arithmetic (expf, running max / sum in fp32) as attn_kernel<F32T>; per 64-key tile 48 f16 MFMAs of 16 pipe cycles instead
// of 128 fp32 MFMAs of 32.  Ranges: |q|, |k|, |v| < 4094 (scale 2^4); the probabilities (<= 1) are split at scale 2^10.
constexpr float AX_S = 16.0f, AX_SP = 1024.0f;
__device__ inline void ax_split4(const float* x, float scale, _Float16* h, _Float16* l) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float xs = x[r] * scale;
        h[r] = (_Float16)xs;
        l[r] = (_Float16)((xs - (float)h[r]) * 2048.0f);
    }
}

__global__ __launch_bounds__(256, 2) void attn_x3_kernel(M5AttnArgs p) {
    constexpr int RB = 256 + 16;              // LDS row: 64 hi halves | 64 lo halves | pad
    __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11) /* hwreg(HW_REG_MODE, 6, 2): FP_DENORM of f16 / f64 */, 0);   // f16 denormals flushed in the conversions
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 64 * RB];
    unsigned char* Ks = lds;
    unsigned char* Vs = lds + 64 * RB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int q0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    int kl = p.key_len ? p.key_len[b] : p.Sk;
    kl = min(kl, p.Sk);
    int64_t koff = 0, voff = 0;
    if (p.kv_index) {
        const int64_t ix = *p.kv_index;
        koff = ix * p.kv_index_stride_k;
        voff = ix * p.kv_index_stride_v;
    }
    const float* Kg = (const float*)p.k + (koff + b * p.k_bs + h * p.k_hs);
    const float* Vg = (const float*)p.vt + (voff + b * p.vt_bs + h * p.vt_hs);
    const float* Qg = (const float*)p.q + (b * p.q_bs + h * p.q_hs);

    int ntiles = (kl + 63) / 64;
    if (p.causal) ntiles = min(ntiles, q0 / 64 + 1);

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[query l15][d = 32 ks + 8 lg .. +7], split once
    const int qrow = min(q0 + wave * 16 + l15, p.Sq - 1);
    const int qpos = q0 + wave * 16 + l15;
    uint4 qhi[2], qlo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const float4 a = *reinterpret_cast<const float4*>(Qg + (int64_t)qrow * p.q_rs + ks * 32 + lg * 8);
        const float4 c = *reinterpret_cast<const float4*>(Qg + (int64_t)qrow * p.q_rs + ks * 32 + lg * 8 + 4);
        const float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        _Float16 hh[8], ll[8];
        ax_split4(x, AX_S, hh, ll);
        ax_split4(x + 4, AX_S, hh + 4, ll + 4);
        qhi[ks] = *reinterpret_cast<const uint4*>(hh);
        qlo[ks] = *reinterpret_cast<const uint4*>(ll);
    }

    // ---- tile staging: fp32 rows of 64 values = 16 chunks of 4; 1024 chunks per tile, 4 per thread and operand
    float4 rk[4], rv[4];
    auto load_tile = [&](int kt) {
        const int kbase = kt * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = tid + 256 * j, row = c >> 4, ch = c & 15;
            rk[j] = (kbase + row < p.Sk) ? *reinterpret_cast<const float4*>(Kg + (int64_t)(kbase + row) * p.k_rs + ch * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            rv[j] = (kbase + ch * 4 < p.Sk) ? *reinterpret_cast<const float4*>(Vg + (int64_t)row * p.vt_ds + kbase + ch * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = tid + 256 * j, row = c >> 4, ch = c & 15;
            _Float16 hh[4], ll[4];
            const float xk[4] = {rk[j].x, rk[j].y, rk[j].z, rk[j].w};
            ax_split4(xk, AX_S, hh, ll);
            *reinterpret_cast<uint2*>(Ks + row * RB + ch * 8) = *reinterpret_cast<const uint2*>(hh);
            *reinterpret_cast<uint2*>(Ks + row * RB + 128 + ch * 8) = *reinterpret_cast<const uint2*>(ll);
            const float xv[4] = {rv[j].x, rv[j].y, rv[j].z, rv[j].w};
            ax_split4(xv, AX_S, hh, ll);
            *reinterpret_cast<uint2*>(Vs + row * RB + ch * 8) = *reinterpret_cast<const uint2*>(hh);
            *reinterpret_cast<uint2*>(Vs + row * RB + 128 + ch * 8) = *reinterpret_cast<const uint2*>(ll);
        }
    };

    f4_t oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) oacc[i] = f4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    if (ntiles > 0) {
        load_tile(0);
        store_tile();
    }
    __syncthreads();

    for (int kt = 0; kt < ntiles; ++kt) {
        const bool more = kt + 1 < ntiles;
        if (more) load_tile(kt + 1);
        const int kbase = kt * 64;
        // ---- S^T tile: sacc[j][r] = score(key kbase+16j+4lg+r, query l15)
        f4_t sacc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f4_t sm = f4_t{0.f, 0.f, 0.f, 0.f}, sc = f4_t{0.f, 0.f, 0.f, 0.f};
            const unsigned char* krow = Ks + (16 * j + l15) * RB;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 ah = *reinterpret_cast<const uint4*>(krow + (ks * 32 + lg * 8) * 2);
                const uint4 al = *reinterpret_cast<const uint4*>(krow + 128 + (ks * 32 + lg * 8) * 2);
                sm = mfma16<F16T>(ah, qhi[ks], sm);
                sc = mfma16<F16T>(ah, qlo[ks], sc);
                sc = mfma16<F16T>(al, qhi[ks], sc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[j][r] = (sm[r] + sc[r] * (1.0f / 2048.0f)) * (1.0f / (AX_S * AX_S));
        }
        // ---- scale, mask, online softmax (statistics per query = per l15 column): attn_kernel<F32T>'s arithmetic
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kidx = kbase + 16 * j + 4 * lg + r;
                const bool vis = kidx < kl && (!p.causal || kidx <= qpos);
                const float s = vis ? sacc[j][r] * p.scale : -INFINITY;
                sacc[j][r] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = expf(m_run - m_use);      // m_run = -inf -> 0
        float ps = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf(sacc[j][r] - m_use);   // masked -> exp(-inf) = 0
                sacc[j][r] = e;
                ps += e;
            }
        ps += __shfl_xor(ps, 16);
        ps += __shfl_xor(ps, 32);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        float ar[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * lg + r);
        // ---- O = O alpha + P.V   (A = P split in registers, B = V^T hi / lo rows from LDS); the tile's product is formed in
        // its own accumulators and joins the running O in fp32
        uint4 ph[2], pl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float x[8] = {sacc[2 * i][0], sacc[2 * i][1], sacc[2 * i][2], sacc[2 * i][3],
                                sacc[2 * i + 1][0], sacc[2 * i + 1][1], sacc[2 * i + 1][2], sacc[2 * i + 1][3]};
            _Float16 hh[8], ll[8];
            ax_split4(x, AX_SP, hh, ll);
            ax_split4(x + 4, AX_SP, hh + 4, ll + 4);
            ph[i] = *reinterpret_cast<const uint4*>(hh);
            pl[i] = *reinterpret_cast<const uint4*>(ll);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const unsigned char* vrow = Vs + (16 * dt + l15) * RB;
            f4_t tm = f4_t{0.f, 0.f, 0.f, 0.f}, tc = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint2 h0 = *reinterpret_cast<const uint2*>(vrow + (16 * (2 * i) + 4 * lg) * 2);
                const uint2 h1 = *reinterpret_cast<const uint2*>(vrow + (16 * (2 * i + 1) + 4 * lg) * 2);
                const uint2 l0 = *reinterpret_cast<const uint2*>(vrow + 128 + (16 * (2 * i) + 4 * lg) * 2);
                const uint2 l1 = *reinterpret_cast<const uint2*>(vrow + 128 + (16 * (2 * i + 1) + 4 * lg) * 2);
                const uint4 vh = make_uint4(h0.x, h0.y, h1.x, h1.y), vl = make_uint4(l0.x, l0.y, l1.x, l1.y);
                tm = mfma16<F16T>(ph[i], vh, tm);
                tc = mfma16<F16T>(ph[i], vl, tc);
                tc = mfma16<F16T>(pl[i], vh, tc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                oacc[dt][r] = oacc[dt][r] * ar[r] + (tm[r] + tc[r] * (1.0f / 2048.0f)) * (1.0f / (AX_SP * AX_S));
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    // ---- normalise and store: oacc[dt][r] = O[query q0+wave*16+4lg+r][d = 16dt+l15]
    float lr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) lr[r] = __shfl(l_run, 4 * lg + r);
    float* Og = reinterpret_cast<float*>(p.o) + b * p.o_bs + h * 64;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qr = q0 + wave * 16 + 4 * lg + r;
        if (qr < p.Sq) {
            const float inv = 1.0f / lr[r];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) Og[(int64_t)qr * p.o_rs + 16 * dt + l15] = oacc[dt][r] * inv;
        }
    }
}

}  // namespace

int m5_attention16_dispatch(int dtype, const M5AttnArgs* a, hipStream_t s);   // attention16.hip; returns 1 when not handled

static bool use_v1_attn() {   // M5_ATTN_V1=1: A/B the first-generation kernel
    static const bool v = [] { const char* e = m5_tool_env("M5_ATTN_V1"); return e && e[0] == '1'; }();
    return v;
}

extern "C" int m5_attention(int dtype, const M5AttnArgs* a, void* stream) {
    if (!a || !a->q || !a->k || !a->vt || !a->o || a->B <= 0 || a->H <= 0 || a->Sq <= 0 || a->Sk <= 0) return M5_ERR_ARG;
    const int es = (dtype == M5_F32 || dtype == M5_F32X3) ? 4 : 2, al = 16 / es;
    if (a->q_rs % al || a->k_rs % al || a->vt_ds % al || a->q_hs % al || a->k_hs % al || a->vt_hs % al ||
        a->q_bs % al || a->k_bs % al || a->vt_bs % al || a->kv_index_stride_k % al || a->kv_index_stride_v % al)
        return M5_ERR_ARG;
    if (((uintptr_t)a->q & 15) || ((uintptr_t)a->k & 15) || ((uintptr_t)a->vt & 15)) return M5_ERR_ARG;
    if (a->vt_ds < ((a->Sk + 63) / 64) * 64) return M5_ERR_ARG;   // V^T rows are read in whole 64-key tiles
    dim3 grid((a->Sq + 63) / 64, a->H, a->B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype != M5_F32 && dtype != M5_F32X3 && !use_v1_attn()) {
        const int r = m5_attention16_dispatch(dtype, a, s);
        if (r != 1) return r;
    }
    switch (dtype) {
        case M5_F32: hipLaunchKernelGGL(attn_kernel<F32T>, grid, dim3(256), 0, s, *a); break;
        case M5_F32X3: hipLaunchKernelGGL(attn_x3_kernel, grid, dim3(256), 0, s, *a); break;
        case M5_F16: hipLaunchKernelGGL(attn_kernel<F16T>, grid, dim3(256), 0, s, *a); break;
        case M5_BF16: hipLaunchKernelGGL(attn_kernel<BF16T>, grid, dim3(256), 0, s, *a); break;
        default: return M5_ERR_ARG;
    }
    M5_CHECK_LAUNCH();
    return M5_OK;
}
