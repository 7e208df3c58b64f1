// Flash attention for 16-bit operands on gfx950, head_dim 64: 32x32x16 MFMA, LDS-DMA staging.
//
// One workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries for the
// whole key loop.  Per 64-key tile, K [64 keys][64 d] and V^T [64 d][64 keys] arrive by LDS-DMA
// (16 x 1 KiB instructions, 3 LDS stages, one barrier per tile; 48 KiB per workgroup).
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
//
// KH = 2 (round 3): the 64 keys of a tile are split over TWO waves per 32-query group (keys 32 kh .. +31 of every tile:
// one 32x32 score block and one k-half of the PV product each); a workgroup is then 2 query groups x 2 key halves = 64
// queries, and the halves' (m, l, O) merge once through LDS after the key loop.  Same LDS reads and MFMAs per query, twice
// the waves: at one utterance (352 query blocks of 128 on 256 CUs = 1.4 waves per SIMD) a wave's MFMA, softmax-VALU and
// ds_read phases have no other wave to overlap with; with 704 workgroups of 64 queries there are 2.75.  Measured: no gain
// there (-2 %), a loss at high occupancy (twice the operand DMA per query) -- a tools-only variant (M5_ATTN_KH=2).
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 b8_t;
typedef __attribute__((ext_vector_type(16))) float f16_t;
typedef __attribute__((ext_vector_type(2))) float f2_t;

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

constexpr int KT = 64;
constexpr int TILE_B = 64 * 128;                 // one operand tile: 64 rows x 128 bytes
constexpr int STAGE_B = 2 * TILE_B;              // K tile + V^T tile
constexpr int NST = 3;

// VARIANT: 0 = product kernel; 1..3 = timing ablations (WRONG results; M5_ATTN_VARIANT, tools only):
// 1 no per-tile DMA, 2 no per-tile barrier, 3 no softmax VALU.  4 (round-6 probe, correct results): s_setprio 1 around the
// two MFMA clusters of a tile (cdna_hip_programming.md T5) -- measured 1.5-4 % SLOWER (profiles/r6a_attn_setprio_negative.txt).
template <typename T, int NWAVE, int VARIANT, int KH = 1>
__global__ __launch_bounds__(NWAVE * 64, NWAVE <= 4 ? 2 : 1) void attn16_kernel(M5AttnArgs p) {
    using st = typename T::storage;
    static_assert(KH == 1 || (KH == 2 && NWAVE % 2 == 0), "key halves");
    constexpr int NQG = NWAVE / KH;                 // 32-query groups per workgroup
    constexpr int NKB = 2 / KH;                     // 32-key blocks of a tile per wave
    constexpr int QB = 32 * NQG;
    constexpr int NJ = 16 / NWAVE;                  // DMA instructions per wave per tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE_B];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int qg = (KH == 1) ? wave : (wave % NQG), kh = (KH == 1) ? 0 : (wave / NQG);      // wave-uniform
    const int q0 = blockIdx.x * QB, h = blockIdx.y, b = blockIdx.z;
    if (p.q_len && q0 >= p.q_len[b]) return;        // a query block of padding rows (workgroup-uniform: before any barrier)
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
    const int qpos = q0 + qg * 32 + l31;
    const int qrow = min(qpos, p.Sq - 1);
    uint4 qf[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds)
        qf[ds] = *reinterpret_cast<const uint4*>(Qg + ((int64_t)qrow * p.q_rs + ds * 16 + hh * 8) * 2);

    // ---- DMA assignment: instruction q = 4 wave + j; q < 8: K rows 8q..8q+7, else V^T rows 8(q-8)..
    // lane l -> row 8q' + (l >> 3), chunk slot l & 7, source chunk slot ^ ((row >> 1) & 7).
    // Tiles are loaded in order, so each lane just advances a running source pointer (K: 64 rows,
    // V^T: 128 bytes along the row); K rows past Sk - 1 (last tile only) read row Sk - 1 instead
    // (valid memory; those keys are masked by index).
    const int srow = lane >> 3;
    const unsigned char* gcur[NJ];
    const unsigned char* klast[NJ];
    int64_t gstep[NJ];
    int krow_of[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = wave * NJ + j;                              // wave-uniform
        const int row = (q & 7) * 8 + srow;
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        if (q < 8) {
            gcur[j] = Kg + (int64_t)row * p.k_rs * 2 + chunk * 16;
            klast[j] = Kg + (int64_t)(p.Sk - 1) * p.k_rs * 2 + chunk * 16;
            gstep[j] = (int64_t)KT * p.k_rs * 2;
            krow_of[j] = row;
        } else {
            gcur[j] = Vg + (int64_t)row * p.vt_ds * 2 + chunk * 16;
            klast[j] = gcur[j];
            gstep[j] = KT * 2;
            krow_of[j] = -(1 << 30);                             // never past the end
        }
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto stage_load = [&](int stage, int kt) {                   // called with kt = 0, 1, 2, ... in order
        const int over = kt * KT - p.Sk;                         // row r is past the end iff r + over >= 0
        const uint32_t sb = lds_base + stage * STAGE_B + wave * NJ * 1024;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            glds16((krow_of[j] + over >= 0) ? klast[j] : gcur[j], sb + j * 1024);
            gcur[j] += gstep[j];
        }
    };

    // ---- fragment read offsets (bytes inside a tile)
    // K: A row i = l31 -> tile row pi(l31) (bits 2,3 swapped) + 32 kb, chunk 2 ds + hh
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    int koffs[NKB][4];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const int row = krow + 32 * (kb + kh);               // (KH == 2: the wave's own key block kh)
            koffs[kb][ds] = row * 128 + (((2 * ds + hh) ^ ((row >> 1) & 7)) << 4);
        }
    // V^T: A row = d = l31 + 32 db, chunk 4 kb + 2 m + hh  (keys 32 kb + 16 m + 8 hh .. +7)
    int voffs[2][2 * NKB];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int c = 0; c < 2 * NKB; ++c) {
            const int row = l31 + 32 * db;
            voffs[db][c] = TILE_B + row * 128 + (((2 * (c + 2 * kh) + hh) ^ ((row >> 1) & 7)) << 4);
        }

    f16_t oacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;          // scores in the exp2 domain

    // S^T = K . Q^T of one tile: s[kb][8 m + j] = score(key 64 kt + 32 kb + 16 m + 8 hh + j, query l31)
    auto qk_tile = [&](const unsigned char* sb, f16_t (&s)[NKB]) {
        if (VARIANT == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const uint4 a = *reinterpret_cast<const uint4*>(sb + koffs[kb][ds]);
                s[kb] = mfma32<T>(a, qf[ds], s[kb]);
            }
        }
        if (VARIANT == 4) __builtin_amdgcn_s_setprio(0);
    };
    // softmax of tile kt (scores in `s`, overwritten by the probabilities) and O^T += V^T . P^T
    auto softmax_pv = [&](const unsigned char* sb, int kt, f16_t (&s)[NKB]) {
        const int kbase = kt * KT;
        // mask (only tiles that touch the key limit or the causal diagonal); online softmax in the
        // exp2 domain with the score scale folded into the exponent's fma:
        //   p = exp2(s * sc2 - m),  m = running max of s * sc2   (sc2 > 0, so max commutes)
        // VALU per lane and tile: 16 v_max3 + 16 v_pk_fma + 32 v_exp + 16 v_pk_add + 16 cvt_pk.
        const bool need_mask = (kbase + KT > kl) || (p.causal && kbase + KT - 1 > q0 + qg * 32);
        if (need_mask) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kidx = kbase + 32 * (kb + kh) + 16 * (r >> 3) + 8 * hh + (r & 7);
                    const bool vis = kidx < kl && (!p.causal || kidx <= qpos);
                    s[kb][r] = vis ? s[kb][r] : -INFINITY;
                }
        }
        if (VARIANT == 3) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    st tmp[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) tmp[j] = T::from_f32(s[kb][8 * m + j]);
                    const uint4 pb = *reinterpret_cast<const uint4*>(tmp);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const uint4 a = *reinterpret_cast<const uint4*>(sb + voffs[db][2 * kb + m]);
                        oacc[db] = mfma32<T>(a, pb, oacc[db]);
                    }
                }
            l_run = 1.f;
            return;
        }
        float mx = fmaxf(s[0][0], s[NKB - 1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[NKB - 1][r]);      // -> v_max3_f32
        mx = fmaxf(mx, lane_xor32(mx)) * sc2;
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);      // m_run = -inf -> 0
        const f2_t sc2v = {sc2, sc2}, mneg = {-m_use, -m_use};
        f2_t psum = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f2_t sv = {s[kb][r], s[kb][r + 1]};
                const f2_t arg = __builtin_elementwise_fma(sv, sc2v, mneg);          // masked: -inf
                f2_t e;
                e[0] = __builtin_amdgcn_exp2f(arg[0]);
                e[1] = __builtin_amdgcn_exp2f(arg[1]);
                s[kb][r] = e[0];
                s[kb][r + 1] = e[1];
                psum += e;
            }
        l_run = l_run * alpha + (psum[0] + psum[1]);   // per half-lane partial sum; halves merged at the end
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        }
        if (VARIANT == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                st tmp[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) tmp[j] = T::from_f32(s[kb][8 * m + j]);
                const uint4 pb = *reinterpret_cast<const uint4*>(tmp);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const uint4 a = *reinterpret_cast<const uint4*>(sb + voffs[db][2 * kb + m]);
                    oacc[db] = mfma32<T>(a, pb, oacc[db]);
                }
            }
        if (VARIANT == 4) __builtin_amdgcn_s_setprio(0);
    };

    // ---- key loop, software-pipelined across tiles: the S^T MFMAs of tile kt+1 are issued BEFORE
    // the softmax VALU work of tile kt, so one wave keeps the matrix pipe and the VALU busy at the
    // same time (there are only ~1.4 waves per SIMD at NAR sizes: no other wave would cover it).
    // Two score register sets ping-pong (loop unrolled by two, no register moves).
    // Per tile: wait tile kt+1 -> barrier -> DMA tile kt+2 into the slot tile kt-1 used.
    if (ntiles > 0) stage_load(0, 0);
    if (ntiles > 1) stage_load(1, 1);
    f16_t sA[NKB], sB[NKB];
    if (ntiles > 0) {
        if (ntiles > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NJ) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        qk_tile(lds, sA);
    }
    int slot = 0;                                                // LDS slot of tile kt
    auto step = [&](int kt, f16_t (&cur)[NKB], f16_t (&nxt)[NKB]) {
        const int s1 = (slot == 2) ? 0 : slot + 1;               // slot of tile kt+1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // tile kt+1 (the only DMA in flight) landed
        if (VARIANT != 2) __syncthreads();                       // ... for every wave; tile kt-1 fully consumed
        if (VARIANT != 1 && kt + 2 < ntiles) stage_load(slot == 0 ? 2 : slot - 1, kt + 2);
        if (kt + 1 < ntiles) qk_tile(lds + s1 * STAGE_B, nxt);
        softmax_pv(lds + slot * STAGE_B, kt, cur);
        slot = s1;
    };
    for (int kt = 0; kt < ntiles; kt += 2) {
        step(kt, sA, sB);
        if (kt + 1 < ntiles) step(kt + 1, sB, sA);
    }

    // ---- KH == 2: the upper key half hands (m, l, O) to the lower one's same lane through LDS (the stage buffers are free)
    if constexpr (KH == 2) {
        float* mg = reinterpret_cast<float*>(lds) + qg * 34 * 64 + lane;         // [qg][34 values][64 lanes]: conflict-free
        __syncthreads();                                                         // every wave has read its last tile
        if (kh == 1) {
            mg[0] = m_run;
            mg[64] = l_run;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) mg[(2 + 16 * db + r) * 64] = oacc[db][r];
        }
        __syncthreads();
        if (kh == 1) return;
        const float m2 = mg[0], l2 = mg[64];
        const float mn = fmaxf(m_run, m2);
        const float mu = (mn == -INFINITY) ? 0.f : mn;
        const float a1 = __builtin_amdgcn_exp2f(m_run - mu), a2 = __builtin_amdgcn_exp2f(m2 - mu);      // -inf -> 0
        l_run = l_run * a1 + l2 * a2;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[db][r] = oacc[db][r] * a1 + mg[(2 + 16 * db + r) * 64] * a2;
    }
    // ---- normalise and store: oacc[db][r] = O[query l31][d = 32 db + (r&3) + 8 (r>>2) + 4 hh]
    const float l_tot = l_run + lane_xor32(l_run);
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


// ---------------------------------------------------------------------------------------------------------------------
// attn16s_kernel (round 6): the SAME arithmetic as attn16_kernel<T, 4, 0, 1> -- same fragments, same fma / exp2 / sum order,
// bit-identical output -- with the instruction stream of a key tile laid out by hand (sched_barrier between small groups)
// instead of leaving three big blocks to the scheduler.  What the compiler made of the kernel above: the 8 S^T MFMAs of tile
// t+1 back to back (the wave sits at each until the matrix pipe frees: ~256 cycles in which it issues nothing else), then the
// 17-deep max chain, then exp / PV, then a 16-deep dependent v_pk_add chain with s_nops AFTER the last PV MFMA, then the
// barrier and ~45 instructions of DMA issue in front of the next S^T block -- ~500 cycles per tile with the matrix pipe idle
// and nothing overlapping the VALU.  Here: one MFMA every 4-8 other instructions through the whole tile --
//   phase 1: the 8 S^T MFMAs of tile t+1 interleaved with tile t's max chain, the lane^32 exchange, alpha, the DMA issue of
//            tile t+2 and the V^T fragment reads of tile t;
//   phase 2: per 8-key block {8 fma, 8 exp2, 8 row-sum adds, 4 cvt_pk} then its 2 PV MFMAs (the row sums ride with the exps:
//            no dependent chain at the end of the tile);
// the DMA of all tiles but the last takes a path without the per-row clamp, m0 is saved once per four pieces.
// FM = 1: the fma / add of the softmax as single v_fma_f32 / v_add_f32 (asm) -- MI355X_MICROARCH.md prices packed f32 VALU
// beside MFMAs at +22..26 cycles per instruction over the two scalar ones; FM = 0 leaves the choice to the compiler.
__device__ inline void glds16x4(const unsigned char* g0, const unsigned char* g1, const unsigned char* g2, const unsigned char* g3,
                                uint32_t lds_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(lds_base) : "memory", "scc");
}
template <int FM>
__device__ inline float sm_fma(float a, float b, float c) {
    if constexpr (FM == 1) { float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
    else return __builtin_fmaf(a, b, c);
}
template <int FM>
__device__ inline float sm_add(float a, float b) {
    if constexpr (FM == 1) { float d; asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
    else return a + b;
}
#define M5_SB() __builtin_amdgcn_sched_barrier(0)
// two fp32 values -> one dword of two operand-type values, round to nearest even (one v_cvt_pk_*: the same values as two T::from_f32)
template <typename T>
__device__ inline uint32_t pack2(float a, float b);
template <>
__device__ inline uint32_t pack2<BF16T>(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    const f2_t v = {a, b};
    const bf2 r = __builtin_convertvector(v, bf2);
    return *reinterpret_cast<const uint32_t*>(&r);
}
template <>
__device__ inline uint32_t pack2<F16T>(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 hf2;
    const f2_t v = {a, b};
    const hf2 r = __builtin_convertvector(v, hf2);
    return *reinterpret_cast<const uint32_t*>(&r);
}

// ABL (tools, timing only, WRONG results unless noted): 1 no per-tile DMA, 2 no per-tile barrier, 3 no softmax fma / exp / sum, 4 no S^T MFMAs,
// 5 no PV MFMAs, 6 (correct) the four DMA pieces of a tile issued one per 8-key block of phase 2, 7 = 1 + 2
template <typename T, int FM, int ABL = 0>
__global__ __launch_bounds__(256, 2) void attn16s_kernel(M5AttnArgs p) {
    using st = typename T::storage;
    constexpr int NJ = 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE_B];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int q0 = blockIdx.x * 128, h = blockIdx.y, b = blockIdx.z;
    if (p.q_len && q0 >= p.q_len[b]) return;
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
    if (p.causal) ntiles = min(ntiles, (min(q0 + 128, p.Sq) - 1) / KT + 1);

    const int qpos = q0 + wave * 32 + l31;
    const int qrow = min(qpos, p.Sq - 1);
    uint4 qf[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds)
        qf[ds] = *reinterpret_cast<const uint4*>(Qg + ((int64_t)qrow * p.q_rs + ds * 16 + hh * 8) * 2);

    // DMA assignment: as attn16_kernel (instruction q = 4 wave + j: waves 0, 1 bring the K rows, waves 2, 3 the V^T rows)
    const int srow = lane >> 3;
    const unsigned char* gcur[NJ];
    const unsigned char* klast[NJ];
    int krow_of[NJ];
    const bool kwave = wave < 2;                                  // wave-uniform
    const int64_t gstep = kwave ? (int64_t)KT * p.k_rs * 2 : (int64_t)KT * 2;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = wave * NJ + j;
        const int row = (q & 7) * 8 + srow;
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        if (kwave) {
            gcur[j] = Kg + (int64_t)row * p.k_rs * 2 + chunk * 16;
            klast[j] = Kg + (int64_t)(p.Sk - 1) * p.k_rs * 2 + chunk * 16;
            krow_of[j] = row;
        } else {
            gcur[j] = Vg + (int64_t)row * p.vt_ds * 2 + chunk * 16;
            klast[j] = gcur[j];
            krow_of[j] = -(1 << 30);
        }
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto stage_load = [&](int stage, int kt) {                   // called with kt = 0, 1, 2, ... in order
        const uint32_t sb = lds_base + stage * STAGE_B + wave * NJ * 1024;
        if ((kt + 1) * KT <= p.Sk) {                              // no row of this tile is past the end (wave-uniform)
            glds16x4(gcur[0], gcur[1], gcur[2], gcur[3], sb);
        } else {
            const int over = kt * KT - p.Sk;
            glds16x4((krow_of[0] + over >= 0) ? klast[0] : gcur[0], (krow_of[1] + over >= 0) ? klast[1] : gcur[1],
                     (krow_of[2] + over >= 0) ? klast[2] : gcur[2], (krow_of[3] + over >= 0) ? klast[3] : gcur[3], sb);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) gcur[j] += gstep;
    };

    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    int koffs[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const int row = krow + 32 * kb;
            koffs[kb][ds] = row * 128 + (((2 * ds + hh) ^ ((row >> 1) & 7)) << 4);
        }
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
    const float sc2 = p.scale * 1.4426950408889634f;

    auto ld = [&](const unsigned char* sb, int off) { return *reinterpret_cast<const uint4*>(sb + off); };
    auto qk = [&](const uint4& a, const uint4& bq, f16_t c) -> f16_t {
        if constexpr (ABL == 4) { c[0] += __uint_as_float(a.x); return c; }
        else return mfma32<T>(a, bq, c);
    };
    auto mask_tile = [&](int kt, f16_t (&s)[2]) {
        const int kbase = kt * KT;
        const bool need_mask = (kbase + KT > kl) || (p.causal && kbase + KT - 1 > q0 + wave * 32);
        if (need_mask) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kidx = kbase + 32 * kb + 16 * (r >> 3) + 8 * hh + (r & 7);
                    const bool vis = kidx < kl && (!p.causal || kidx <= qpos);
                    s[kb][r] = vis ? s[kb][r] : -INFINITY;
                }
        }
    };
    // tile statistics once the 32 scores' maximum `mx` of this lane is known: m_new, alpha, -m_use
    float alpha, mneg, m_new;
    auto stats = [&](float mx) {
        mx = fmaxf(mx, lane_xor32(mx)) * sc2;
        m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        mneg = -m_use;
    };
    auto rescale = [&]() {
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        }
    };
    float ps0, ps1;                                               // row-sum partials (even / odd registers, attn16_kernel's order)
    // one 8-key block of P^T: scores -> probabilities (kept in s for nothing: only the packed operand is used)
    auto p_block = [&](f16_t (&s)[2], int kb, int m) -> uint4 {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (ABL == 3) { w[j] = pack2<T>(s[kb][8 * m + 2 * j], s[kb][8 * m + 2 * j + 1]); continue; }
            const float e0 = __builtin_amdgcn_exp2f(sm_fma<FM>(s[kb][8 * m + 2 * j], sc2, mneg));
            const float e1 = __builtin_amdgcn_exp2f(sm_fma<FM>(s[kb][8 * m + 2 * j + 1], sc2, mneg));
            ps0 = sm_add<FM>(ps0, e0);
            ps1 = sm_add<FM>(ps1, e1);
            w[j] = pack2<T>(e0, e1);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    };
    uint32_t dma_sb = 0;                                          // ABL 6: LDS base of the tile being fetched (0 = none)
    bool dma_slow = false;
    int dma_over = 0;
    auto pv_phase = [&](const unsigned char* sbc, f16_t (&cur)[2], uint4 (&va)[2][4]) {
        ps0 = 0.f;
        ps1 = 0.f;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const uint4 pb = p_block(cur, blk >> 1, blk & 1);
            M5_SB();
            if constexpr (ABL == 6) {
                if (dma_sb) {
                    glds16((dma_slow && krow_of[blk] + dma_over >= 0) ? klast[blk] : gcur[blk], dma_sb + blk * 1024);
                    gcur[blk] += gstep;
                }
                M5_SB();
            }
            if constexpr (ABL != 5) {
                oacc[0] = mfma32<T>(va[0][blk], pb, oacc[0]);
                oacc[1] = mfma32<T>(va[1][blk], pb, oacc[1]);
            } else {
                oacc[0][blk] += __uint_as_float(pb.x);
            }
            M5_SB();
        }
        l_run = l_run * alpha + (ps0 + ps1);
        m_run = m_new;
    };

    if (ntiles > 0) stage_load(0, 0);
    if (ntiles > 1) stage_load(1, 1);
    f16_t sA[2], sB[2];
    if (ntiles > 0) {
        if (ntiles > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NJ) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sA[kb][r] = 0.f;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) sA[kb] = mfma32<T>(ld(lds, koffs[kb][ds]), qf[ds], sA[kb]);
        }
    }
    int slot = 0;
    // tile kt (scores in cur) with a tile kt+1 behind it
    auto full = [&](int kt, f16_t (&cur)[2], f16_t (&nxt)[2]) {
        const int s1 = (slot == 2) ? 0 : slot + 1;
        const unsigned char* sbc = lds + slot * STAGE_B;
        const unsigned char* sbn = lds + s1 * STAGE_B;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // tile kt+1 (the only DMA in flight) landed
        if constexpr (ABL != 2 && ABL != 7) __builtin_amdgcn_s_barrier();       // ... for every wave; tile kt-1 fully consumed
        M5_SB();
        uint4 ka[2][4], va[2][4];
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) ka[kb][ds] = ld(sbn, koffs[kb][ds]);
        M5_SB();
        mask_tile(kt, cur);
        float mx = fmaxf(cur[0][0], cur[1][0]);
#pragma unroll
        for (int r = 1; r < 8; ++r) mx = fmaxf(fmaxf(mx, cur[0][r]), cur[1][r]);
        M5_SB();
#pragma unroll
        for (int r = 0; r < 16; ++r) { nxt[0][r] = 0.f; nxt[1][r] = 0.f; }
        nxt[0] = qk(ka[0][0], qf[0], nxt[0]);
        M5_SB();
#pragma unroll
        for (int r = 8; r < 12; ++r) mx = fmaxf(fmaxf(mx, cur[0][r]), cur[1][r]);
        M5_SB();
        nxt[1] = qk(ka[1][0], qf[0], nxt[1]);
        M5_SB();
#pragma unroll
        for (int r = 12; r < 16; ++r) mx = fmaxf(fmaxf(mx, cur[0][r]), cur[1][r]);
        M5_SB();
        nxt[0] = qk(ka[0][1], qf[1], nxt[0]);
        M5_SB();
#pragma unroll
        for (int c = 0; c < 2; ++c) { va[0][c] = ld(sbc, voffs[0][c]); va[1][c] = ld(sbc, voffs[1][c]); }
        M5_SB();
        nxt[1] = qk(ka[1][1], qf[1], nxt[1]);
        M5_SB();
        stats(mx);
        M5_SB();
        nxt[0] = qk(ka[0][2], qf[2], nxt[0]);
        M5_SB();
        if constexpr (ABL == 6) {
            dma_sb = (kt + 2 < ntiles) ? lds_base + (slot == 0 ? 2 : slot - 1) * STAGE_B + wave * NJ * 1024 : 0;
            dma_slow = (kt + 3) * KT > p.Sk;
            dma_over = (kt + 2) * KT - p.Sk;
        } else if constexpr (ABL != 1 && ABL != 7) {
            if (kt + 2 < ntiles) stage_load(slot == 0 ? 2 : slot - 1, kt + 2);       // into the slot tile kt-1 used
        }
        M5_SB();
        nxt[1] = qk(ka[1][2], qf[2], nxt[1]);
        M5_SB();
#pragma unroll
        for (int c = 2; c < 4; ++c) { va[0][c] = ld(sbc, voffs[0][c]); va[1][c] = ld(sbc, voffs[1][c]); }
        M5_SB();
        nxt[0] = qk(ka[0][3], qf[3], nxt[0]);
        M5_SB();
        rescale();
        M5_SB();
        nxt[1] = qk(ka[1][3], qf[3], nxt[1]);
        M5_SB();
        pv_phase(sbc, cur, va);
        slot = s1;
    };
    // the last tile
    auto last = [&](int kt, f16_t (&cur)[2]) {
        const unsigned char* sbc = lds + slot * STAGE_B;
        dma_sb = 0;
        uint4 va[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { va[0][c] = ld(sbc, voffs[0][c]); va[1][c] = ld(sbc, voffs[1][c]); }
        mask_tile(kt, cur);
        float mx = fmaxf(cur[0][0], cur[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, cur[0][r]), cur[1][r]);
        stats(mx);
        rescale();
        pv_phase(sbc, cur, va);
    };
    if (ntiles > 0) {
        int kt = 0;
        for (; kt + 2 < ntiles; kt += 2) {
            full(kt, sA, sB);
            full(kt + 1, sB, sA);
        }
        if (kt + 1 < ntiles) {
            full(kt, sA, sB);
            last(kt + 1, sB);
        } else {
            last(kt, sA);
        }
    }

    const float l_tot = l_run + lane_xor32(l_run);
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
    // M5_ATTN_NW / M5_ATTN_VARIANT: tuning + ablation hooks (tools/attn_bench.py); product = 4 waves, variant 0
    static const int nw = [] { const char* e = m5_tool_env("M5_ATTN_NW"); return e ? atoi(e) : 4; }();
    static const int var = [] { const char* e = m5_tool_env("M5_ATTN_VARIANT"); return e ? atoi(e) : 0; }();
#define M5_A16(TT, NWV, VV) hipLaunchKernelGGL((attn16_kernel<TT, NWV, VV>), dim3((a->Sq + 32 * NWV - 1) / (32 * NWV), a->H, a->B), dim3(NWV * 64), 0, s, *a)
#define M5_A16K(TT) hipLaunchKernelGGL((attn16_kernel<TT, 4, 0, 2>), dim3((a->Sq + 63) / 64, a->H, a->B), dim3(256), 0, s, *a)
    // Key-half split (KH = 2, see the header): tools only (M5_ATTN_KH=2).  Measured (profiles/r3x_attn_key_half_split_ab.txt):
    // -2 % at the NAR shape of one utterance, -25 % on the small one-off problems (AR prefill, speaker encoder: 3 us each),
    // +16 % at a batched group's 16 x 2240 rows and +10 % at the 5399-row long form -- so the product keeps one form.
#ifdef M5_TOOLS
    static const int sched = [] { const char* e = m5_tool_env("M5_ATTN_SCHED"); return e ? atoi(e) : 0; }();
    static const int khe = [] { const char* e = m5_tool_env("M5_ATTN_KH"); return e ? atoi(e) : 0; }();
#define M5_A16S(TT, FMV) hipLaunchKernelGGL((attn16s_kernel<TT, FMV>), dim3((a->Sq + 127) / 128, a->H, a->B), dim3(256), 0, s, *a)
#define M5_A16SA(AB) hipLaunchKernelGGL((attn16s_kernel<BF16T, 1, AB>), dim3((a->Sq + 127) / 128, a->H, a->B), dim3(256), 0, s, *a)
    static const int abl = [] { const char* e = m5_tool_env("M5_ATTN_ABL"); return e ? atoi(e) : 0; }();
    if (sched == 2 && abl >= 1 && abl <= 7 && dtype == M5_BF16) {
        switch (abl) {
            case 1: M5_A16SA(1); break;
            case 2: M5_A16SA(2); break;
            case 3: M5_A16SA(3); break;
            case 4: M5_A16SA(4); break;
            case 5: M5_A16SA(5); break;
            case 6: M5_A16SA(6); break;
            default: M5_A16SA(7); break;
        }
    } else
    if (sched >= 1 && sched <= 2 && nw == 4 && var == 0) {
        if (dtype == M5_F16) { if (sched == 2) M5_A16S(F16T, 1); else M5_A16S(F16T, 0); }
        else { if (sched == 2) M5_A16S(BF16T, 1); else M5_A16S(BF16T, 0); }
    } else
    if (khe == 2 && nw == 4 && var == 0) {
        if (dtype == M5_F16) M5_A16K(F16T);
        else M5_A16K(BF16T);
    } else
#endif
    if (dtype == M5_F16) {
        M5_A16(F16T, 4, 0);
#ifdef M5_TOOLS
    } else if (nw == 2) {
        M5_A16(BF16T, 2, 0);
    } else if (nw == 8) {
        M5_A16(BF16T, 8, 0);
    } else if (var >= 1 && var <= 4) {          // timing ablations (1-3: results are WRONG by construction) and probes
        switch (var) {
            case 1: M5_A16(BF16T, 4, 1); break;
            case 2: M5_A16(BF16T, 4, 2); break;
            case 4: M5_A16(BF16T, 4, 4); break;
            default: M5_A16(BF16T, 4, 3); break;
        }
#endif
    } else {
        (void)nw; (void)var;
        M5_A16(BF16T, 4, 0);
    }
#undef M5_A16
#undef M5_A16K
    M5_CHECK_LAUNCH();
    return M5_OK;
}
