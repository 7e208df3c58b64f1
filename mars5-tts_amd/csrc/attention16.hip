// Flash attention for 16-bit operands on gfx950, head_dim 64: 32x32x16 MFMA, LDS-DMA staging.
//
// One workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries for the
// whole key loop.  Per 64-key tile, K [64 keys][64 d] and V^T [64 d][64 keys] arrive by LDS-DMA
// (16 x 1 KiB instructions, 3 stages in flight, one barrier per tile; 48 KiB per workgroup so
// three workgroups share a CU and cover each other's softmax VALU with MFMAs).
//
// Everything is computed TRANSPOSED so that the query index is the MFMA column (lane & 31) in both
// products and all softmax state is lane-local -- no cross-lane traffic for the running max /
// sum / rescale except one lane^32 exchange of the row max per tile:
//   S^T[key][q] = K . Q^T       A = K rows from LDS (ds_read_b128), B = Q held in registers
//   O^T[d][q]  += V^T . P^T     A = V^T rows from LDS (ds_read_b128), B = P^T straight from the
//                               S^T accumulators (bf16/f16 pack, no LDS round trip)
// The C/D row of a 32x32 MFMA that a lane holds in registers 8m..8m+7 is {4h..4h+3} u {8+4h..}
// (+16m), h = lane >> 5; K rows are therefore fetched with row bits 2 and 3 swapped, which makes
// those 8 registers the scores of 8 CONSECUTIVE keys 16m + 8h .. +7 -- exactly one 16-byte
// k-slot group of the P^T operand and one ds_read_b128 of the V^T row.
// LDS image of both tiles: [64 rows][128 B], 16-byte chunk index XOR ((row >> 1) & 7), applied on
// the DMA source address and on the fragment reads (conflict-free for ds_read_b128's lane groups).
#include "common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 b8_t;
typedef __attribute__((ext_vector_type(16))) float f16_t;

namespace {

template <typename T>
__device__ inline f16_t mfma32(const uint4& a, const uint4& b, f16_t c);
template <>
__device__ inline f16_t mfma32<F16T>(const uint4& a, const uint4& b, f16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8_t*>(&a),
                                                  *reinterpret_cast<const h8_t*>(&b), c, 0, 0, 0);
}
template <>
__device__ inline f16_t mfma32<BF16T>(const uint4& a, const uint4& b, f16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const b8_t*>(&a),
                                                   *reinterpret_cast<const b8_t*>(&b), c, 0, 0, 0);
}

// LDS-DMA of 16 bytes per lane (see gemm16.hip: issued from asm so hipcc does not drain it).
__device__ inline void glds16(const unsigned char* gsrc, uint32_t lds_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}

constexpr int NWAVE = 4, QB = 32 * NWAVE, KT = 64;
constexpr int TILE_B = 64 * 128;                 // one operand tile: 64 rows x 128 bytes
constexpr int STAGE_B = 2 * TILE_B;              // K tile + V^T tile
constexpr int NST = 3;

template <typename T>
__global__ __launch_bounds__(NWAVE * 64, 3) void attn16_kernel(M5AttnArgs p) {
    using st = typename T::storage;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE_B];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int q0 = blockIdx.x * QB, h = blockIdx.y, b = blockIdx.z;
    int kl = p.key_len ? p.key_len[b] : p.Sk;
    kl = min(kl, p.Sk);
    int64_t koff = 0, voff = 0;
    if (p.kv_index) {
        const int64_t ix = *p.kv_index;
        koff = ix * p.kv_index_stride_k;
        voff = ix * p.kv_index_stride_v;
    }
    const unsigned char* Kg = (const unsigned char*)p.k + (koff + b * p.k_bs + h * p.k_hs) * 2;
    const unsigned char* Vg = (const unsigned char*)p.vt + (voff + b * p.vt_bs + h * p.vt_hs) * 2;
    const unsigned char* Qg = (const unsigned char*)p.q + (b * p.q_bs + h * p.q_hs) * 2;

    int ntiles = (kl + KT - 1) / KT;
    if (p.causal) ntiles = min(ntiles, (min(q0 + QB, p.Sq) - 1) / KT + 1);

    // ---- Q fragments (B operand of S^T): lane holds Q[query l31][d = 16 ds + 8 hh .. +8]
    const int qpos = q0 + wave * 32 + l31;
    const int qrow = min(qpos, p.Sq - 1);
    uint4 qf[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds)
        qf[ds] = *reinterpret_cast<const uint4*>(Qg + ((int64_t)qrow * p.q_rs + ds * 16 + hh * 8) * 2);

    // ---- DMA assignment: instruction q = 4 wave + j; q < 8: K rows 8q..8q+7, else V^T rows 8(q-8)..
    // lane l -> row 8q' + (l >> 3), chunk slot l & 7, source chunk slot ^ ((row >> 1) & 7)
    const int srow = lane >> 3;
    const unsigned char* gsrc[4];
    int64_t gstep[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = wave * 4 + j;                              // wave-uniform
        const int row = (q & 7) * 8 + srow;
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        if (q < 8) {       // K tile: rows are keys (clamped to valid memory; masked by index later)
            gsrc[j] = Kg + chunk * 16;                           // + min(kbase + row, Sk-1) * k_rs * 2 per tile
            gstep[j] = row;
        } else {           // V^T tile: rows are d, the tile advances along the row
            gsrc[j] = Vg + (int64_t)row * p.vt_ds * 2 + chunk * 16;
            gstep[j] = -1;
        }
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto stage_load = [&](int stage, int kt) {
        const int kbase = kt * KT;
        const uint32_t sb = lds_base + stage * STAGE_B + wave * 4 * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned char* src = (gstep[j] >= 0)
                ? gsrc[j] + (int64_t)min(kbase + (int)gstep[j], p.Sk - 1) * p.k_rs * 2
                : gsrc[j] + (int64_t)kbase * 2;
            glds16(src, sb + j * 1024);
        }
    };

    // ---- fragment read offsets (bytes inside a tile)
    // K: A row i = l31 -> tile row pi(l31) (bits 2,3 swapped) + 32 kb, chunk 2 ds + hh
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    int koffs[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const int row = krow + 32 * kb;
            koffs[kb][ds] = row * 128 + (((2 * ds + hh) ^ ((row >> 1) & 7)) << 4);
        }
    // V^T: A row = d = l31 + 32 db, chunk 4 kb + 2 m + hh  (keys 32 kb + 16 m + 8 hh .. +7)
    int voffs[2][4];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = l31 + 32 * db;
            voffs[db][c] = TILE_B + row * 128 + (((2 * c + hh) ^ ((row >> 1) & 7)) << 4);
        }

    f16_t oacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;          // scores in the exp2 domain

    if (ntiles > 0) stage_load(0, 0);
    if (ntiles > 1) stage_load(1, 1);
    int slot = 0;
    for (int kt = 0; kt < ntiles; ++kt) {
        if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile kt landed, kt+1 may be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 < ntiles) stage_load(slot == 0 ? 2 : slot - 1, kt + 2);
        const unsigned char* sb = lds + slot * STAGE_B;
        slot = (slot == 2) ? 0 : slot + 1;
        const int kbase = kt * KT;

        // ---- S^T = K . Q^T : sacc[kb][8 m + j] = score(key kbase + 32 kb + 16 m + 8 hh + j, query l31)
        f16_t sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const uint4 a = *reinterpret_cast<const uint4*>(sb + koffs[kb][ds]);
                sacc[kb] = mfma32<T>(a, qf[ds], sacc[kb]);
            }
        }
        // ---- scale, mask (only tiles that touch the key limit or the causal diagonal)
        const bool need_mask = (kbase + KT > kl) || (p.causal && kbase + KT - 1 > q0 + wave * 32);
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = sacc[kb][r] * sc2;
                if (need_mask) {
                    const int kidx = kbase + 32 * kb + 16 * (r >> 3) + 8 * hh + (r & 7);
                    const bool vis = kidx < kl && (!p.causal || kidx <= qpos);
                    s = vis ? s : -INFINITY;
                }
                sacc[kb][r] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);      // m_run = -inf -> 0
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(sacc[kb][r] - m_use);   // masked -> 0
                sacc[kb][r] = e;
                ps += e;
            }
        l_run = l_run * alpha + ps;                  // per half-lane partial sum; halves merged at the end
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        }
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                st tmp[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) tmp[j] = T::from_f32(sacc[kb][8 * m + j]);
                const uint4 pb = *reinterpret_cast<const uint4*>(tmp);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const uint4 a = *reinterpret_cast<const uint4*>(sb + voffs[db][2 * kb + m]);
                    oacc[db] = mfma32<T>(a, pb, oacc[db]);
                }
            }
    }

    // ---- normalise and store: oacc[db][r] = O[query l31][d = 32 db + (r&3) + 8 (r>>2) + 4 hh]
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (qpos < p.Sq) {
        const float inv = 1.0f / l_tot;
        st* Og = reinterpret_cast<st*>(p.o) + b * p.o_bs + (int64_t)qpos * p.o_rs + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                st t4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) t4[r] = T::from_f32(oacc[db][4 * g + r] * inv);
                *reinterpret_cast<uint2*>(Og + 32 * db + 8 * g + 4 * hh) = *reinterpret_cast<const uint2*>(t4);
            }
    }
}

}  // namespace

// Called by m5_attention (attention.hip) for F16 / BF16 operands after argument validation.
int m5_attention16_dispatch(int dtype, const M5AttnArgs* a, hipStream_t s) {
    if ((a->o_rs % 4) || (a->o_bs % 4) || ((uintptr_t)a->o & 7)) return 1;      // 8-byte output vectors
    if ((a->k_rs % 8) || (a->q_rs % 8) || (a->vt_ds % 8)) return 1;
    dim3 grid((a->Sq + QB - 1) / QB, a->H, a->B);
    if (dtype == M5_F16) hipLaunchKernelGGL(attn16_kernel<F16T>, grid, dim3(NWAVE * 64), 0, s, *a);
    else hipLaunchKernelGGL(attn16_kernel<BF16T>, grid, dim3(NWAVE * 64), 0, s, *a);
    M5_CHECK_LAUNCH();
    return M5_OK;
}
