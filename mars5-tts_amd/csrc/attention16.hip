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

// ---------------------------------------------------------------------------------------------------------------------
// attn16w_kernel (round 6): attn16_kernel<T, 4, 0, 1>'s arithmetic -- same fragments, same fma / exp2 / row-sum order, bit-identical
// output (tools/attn_bench.py csum) -- with the operand traffic taken off the computing waves and the tile loop laid out by hand.
//
// Why.  Ablations of the tile loop (profiles/r6k_attn_tile_loop_ablation.txt): dropping the four LDS-DMA instructions a wave issues
// per key tile -- 4 KB of the 16 -- buys as much as dropping the whole softmax (-17 % at the NAR shape, -22 % at a batched one): a
// `global_load_lds_dwordx4` holds the issuing wave for 60-185 cycles wherever it is placed (MI355X_MICROARCH.md, "LDS-DMA piece issue
// cost"; moving the pieces into the softmax phase was slower still), and the per-tile barrier makes every wave pay the slowest one's.
// Instruction rates (tools/probes/valu_rate.hip, profiles/r6j_valu_instruction_rates.txt): v_exp_f32 occupies the VALU 8.6 cycles,
// v_max3 / v_cvt_pk / v_pk_* 4.7, v_fma / v_add 2.5; a lone wave issues one instruction per ~5.3 cycles -- the loop is issue-bound
// per wave and the matrix pipe (512 of ~1900 cycles per tile) idles unless MFMAs are spread between the VALU work.
//
// What.  * A loader wave (the AR decode step's wave 7, csrc/ar_mega.hip): wave NCW alone requests K and V^T tiles -- scalar bases + two
// lane-constant 32-bit offsets (saddr form: no per-piece VALU), m0 saved once per tile -- K three tiles and V^T two tiles ahead of
// their use into two 3-slot rings (the same 48 KB), and joins the tile barrier once the next step's operands have landed (counted
// vmcnt: the newest group stays in flight).  Waves 0..NCW-1 never touch the VM counter inside the loop.
// * The tile loop as one hand-ordered stream (sched_barrier between small groups): the 8 S^T MFMAs of tile t+1 spread between tile
// t's max chain, lane^32 exchange, alpha and the first 8-key block's exponentials; each PV MFMA followed by half of the next block's
// {fma, exp2, row-sum add, cvt_pk} -- no MFMA waits on the matrix pipe behind another, no dependent add chain at the end of the tile
// (what the compiler made of attn16_kernel: 8 S^T MFMAs back to back, then the max chain, then exp / PV, then 16 dependent v_pk_add
// with s_nops, then barrier + 45 instructions of DMA issue: ~500 cycles per tile with the matrix pipe idle).
// * The softmax's fma / add stay compiler-visible instructions: an inline-asm v_add_f32 that consumes a v_exp_f32 result does NOT
// get the s_nop the hazard recogniser inserts between a transcendental and its VALU user on gfx950 (found the hard way: wrong row
// sums in one of three inlined copies of the step).
// the 8 pieces of one tile: scalar bases b0..b7 (rows 8q ..), lane offsets `ve` / `vo` for even / odd q, m0 saved once
__device__ inline void glds16_tile(uint32_t ve, uint32_t vo, const unsigned char* b0, const unsigned char* b1, const unsigned char* b2,
                                   const unsigned char* b3, const unsigned char* b4, const unsigned char* b5, const unsigned char* b6,
                                   const unsigned char* b7, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %11\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %4\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %6\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %7\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %8\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %9\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %10\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(ve), "v"(vo), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(b4), "s"(b5), "s"(b6), "s"(b7), "s"(lds_addr)
                 : "memory", "scc");
}
// s_barrier between compiler-level memory barriers (the builtin alone does not keep LDS reads on their side of it)
__device__ inline void tile_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
constexpr int RING_B = 3 * TILE_B;               // K ring at 0, V^T ring behind it

// NCW computing waves (32 query rows each) + the loader = wave NCW.  NCW = 3: 4-wave workgroups of 96 rows, two per CU at two waves
// per SIMD (256 VGPRs each).  NCW = 4: 5-wave workgroups of 128 rows; two per CU need three waves on a SIMD = 168 VGPRs.
template <typename T, int NCW, int DBG = 0>
__device__ __forceinline__ void attn16w_body(const M5AttnArgs& p);
template <typename T, int NCW, int WPE>
__global__ __launch_bounds__((NCW + 1) * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void attn16w_kernel(M5AttnArgs p) { attn16w_body<T, NCW>(p); }
template <typename T, int NCW, int DBG>
__device__ __forceinline__ void attn16w_body(const M5AttnArgs& p) {
    constexpr int QB = 32 * NCW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * RING_B];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int q0 = blockIdx.x * QB, h = blockIdx.y, b = blockIdx.z;
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
    int ntiles = (kl + KT - 1) / KT;
    if (p.causal) ntiles = min(ntiles, (min(q0 + QB, p.Sq) - 1) / KT + 1);
    if (ntiles <= 0) ntiles = 0;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

    if (wave == NCW) {
        // ---------------- loader ----------------
        // piece q (0..7) of a tile = its rows 8q .. 8q+7: lane -> row 8q + (lane >> 3), 16-byte chunk (lane & 7) ^ ((row >> 1) & 7);
        // (row >> 1) & 7 = (4 (q & 1) + (lane >> 4)) & 7: two lane-constant offsets per operand (q even / odd)
        const int srow = lane >> 3;
        uint32_t kvo[2], vvo[2];
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int chunk = (lane & 7) ^ ((4 * par + (srow >> 1)) & 7);
            kvo[par] = (uint32_t)(srow * p.k_rs * 2 + chunk * 16);
            vvo[par] = (uint32_t)((int64_t)srow * p.vt_ds * 2 + chunk * 16);
        }
        const int64_t k8 = (int64_t)8 * p.k_rs * 2, v8 = (int64_t)8 * p.vt_ds * 2;
        auto load_k = [&](int t) {
            const uint32_t sb = lds_base + (t % 3) * TILE_B;
            if ((t + 1) * KT <= p.Sk) {
                const unsigned char* base = Kg + (int64_t)t * KT * p.k_rs * 2;
                glds16_tile(kvo[0], kvo[1], base, base + k8, base + 2 * k8, base + 3 * k8, base + 4 * k8, base + 5 * k8, base + 6 * k8,
                            base + 7 * k8, sb);
            } else {                                              // rows past Sk - 1 read row Sk - 1 (valid memory; masked by index)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int row = min(t * KT + 8 * q + srow, p.Sk - 1);
                    const int chunk = (lane & 7) ^ ((4 * (q & 1) + (srow >> 1)) & 7);
                    glds16(Kg + (int64_t)row * p.k_rs * 2 + chunk * 16, sb + q * 1024);
                }
            }
        };
        auto load_v = [&](int t) {
            const uint32_t sb = lds_base + RING_B + (t % 3) * TILE_B;
            const unsigned char* base = Vg + (int64_t)t * KT * 2;
            glds16_tile(vvo[0], vvo[1], base, base + v8, base + 2 * v8, base + 3 * v8, base + 4 * v8, base + 5 * v8, base + 6 * v8,
                        base + 7 * v8, sb);
        };
        if (ntiles == 0) return;
        // prologue groups: {K0} {K1, V0} {K2, V1}; step kt then issues {K(kt+3), V(kt+2)}
        load_k(0);
        if (ntiles > 1) load_k(1);
        load_v(0);
        if (ntiles > 2) load_k(2);
        if (ntiles > 1) load_v(1);
        // K0 landed (everything younger may stay in flight when all three groups are whole)
        if (ntiles > 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tile_barrier();
        for (int kt = 0; kt + 1 < ntiles; ++kt) {
            // top of step kt: K(kt+1) and V(kt) landed; the group issued during step kt-1 ({K(kt+2), V(kt+1)}, whole iff kt+2 < ntiles)
            // may stay in flight
            if (kt + 2 < ntiles) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tile_barrier();
            if (kt + 3 < ntiles) load_k(kt + 3);                 // into the slot of K(kt): consumed in step kt-1
            if (kt + 2 < ntiles) load_v(kt + 2);                 // into the slot of V(kt-1)
        }
        // the last tile's step has no barrier; its V^T tile was waited for above when ntiles > 1 ...
        return;
    }

    // ---------------- compute waves ----------------
    using st = typename T::storage;
    const unsigned char* Qg = (const unsigned char*)p.q + (b * p.q_bs + h * p.q_hs) * 2;
    const int qpos = q0 + wave * 32 + l31;
    const int qrow = min(qpos, p.Sq - 1);
    uint4 qf[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds)
        qf[ds] = *reinterpret_cast<const uint4*>(Qg + ((int64_t)qrow * p.q_rs + ds * 16 + hh * 8) * 2);

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
            voffs[db][c] = RING_B + row * 128 + (((2 * c + hh) ^ ((row >> 1) & 7)) << 4);
        }

    f16_t oacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;

    auto ld = [&](const unsigned char* sb, int off) { return *reinterpret_cast<const uint4*>(sb + off); };
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
    float ps0, ps1;
    // half (two dwords = four keys) of one 8-key block of P^T
    auto p_half = [&](f16_t (&s)[2], int blk, int half, uint32_t (&w)[4]) {
        const int kb = blk >> 1, m = blk & 1;
#pragma unroll
        for (int j = 2 * half; j < 2 * half + 2; ++j) {
            const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][8 * m + 2 * j], sc2, mneg));
            const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][8 * m + 2 * j + 1], sc2, mneg));
            ps0 += e0;
            ps1 += e1;
            w[j] = pack2<T>(e0, e1);
        }
    };
    // blocks 1..3 and the PV MFMAs of all four; block 0's operand arrives in w0 (built under the S^T MFMAs)
    auto pv_phase = [&](f16_t (&cur)[2], uint4 (&va)[2][4], uint32_t (&w0)[4]) {
        uint32_t wa[4], wb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wa[j] = w0[j];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            uint32_t (&wc)[4] = (blk & 1) ? wb : wa;
            uint32_t (&wn)[4] = (blk & 1) ? wa : wb;
            const uint4 pb = make_uint4(wc[0], wc[1], wc[2], wc[3]);
            oacc[0] = mfma32<T>(va[0][blk], pb, oacc[0]);
            M5_SB();
            if (blk < 3) p_half(cur, blk + 1, 0, wn);
            M5_SB();
            oacc[1] = mfma32<T>(va[1][blk], pb, oacc[1]);
            M5_SB();
            if (blk < 3) p_half(cur, blk + 1, 1, wn);
            M5_SB();
        }
        l_run = l_run * alpha + (ps0 + ps1);
        m_run = m_new;
    };

    f16_t sA[2], sB[2];
    if (ntiles > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the Q fragments (the only VM operations of a computing wave)
        tile_barrier();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sA[kb][r] = 0.f;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) sA[kb] = mfma32<T>(ld(lds, koffs[kb][ds]), qf[ds], sA[kb]);
        }
    }
    int slot = 0;                                                // ring slot of tile kt (both rings)
    auto full = [&](int kt, f16_t (&cur)[2], f16_t (&nxt)[2]) {
        const int s1 = (slot == 2) ? 0 : slot + 1;
        const unsigned char* sbc = lds + slot * TILE_B;          // + voffs: V^T(kt)
        const unsigned char* sbn = lds + s1 * TILE_B;            // + koffs: K(kt+1)
        tile_barrier();                            // K(kt+1), V^T(kt) landed (the loader waited); tile kt-1 fully consumed
        M5_SB();
        // fragment registers: at most two quads (32 VGPRs) live at a time -- K d-slices 0, 1 -> K d-slices 2, 3 -> V^T keys 0..31 -> 32..63
        // (the whole kernel has to fit 168 VGPRs: three waves per SIMD = two 5-wave workgroups per CU)
        uint4 ka[2][4], va[2][4];
        uint32_t w0[4];
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) { ka[0][ds] = ld(sbn, koffs[0][ds]); ka[1][ds] = ld(sbn, koffs[1][ds]); }
        M5_SB();
        mask_tile(kt, cur);
        float mx = fmaxf(cur[0][0], cur[1][0]);
#pragma unroll
        for (int r = 1; r < 8; ++r) mx = fmaxf(fmaxf(mx, cur[0][r]), cur[1][r]);
        M5_SB();
#pragma unroll
        for (int r = 0; r < 16; ++r) { nxt[0][r] = 0.f; nxt[1][r] = 0.f; }
        nxt[0] = mfma32<T>(ka[0][0], qf[0], nxt[0]);
        M5_SB();
#pragma unroll
        for (int ds = 2; ds < 4; ++ds) { ka[0][ds] = ld(sbn, koffs[0][ds]); ka[1][ds] = ld(sbn, koffs[1][ds]); }
#pragma unroll
        for (int r = 8; r < 12; ++r) mx = fmaxf(fmaxf(mx, cur[0][r]), cur[1][r]);
        M5_SB();
        nxt[1] = mfma32<T>(ka[1][0], qf[0], nxt[1]);
        M5_SB();
#pragma unroll
        for (int r = 12; r < 16; ++r) mx = fmaxf(fmaxf(mx, cur[0][r]), cur[1][r]);
        M5_SB();
        nxt[0] = mfma32<T>(ka[0][1], qf[1], nxt[0]);
        M5_SB();
        stats(mx);
        M5_SB();
        nxt[1] = mfma32<T>(ka[1][1], qf[1], nxt[1]);
        M5_SB();
#pragma unroll
        for (int c = 0; c < 2; ++c) { va[0][c] = ld(sbc, voffs[0][c]); va[1][c] = ld(sbc, voffs[1][c]); }
        ps0 = 0.f;
        ps1 = 0.f;
        p_half(cur, 0, 0, w0);
        M5_SB();
        nxt[0] = mfma32<T>(ka[0][2], qf[2], nxt[0]);
        M5_SB();
        p_half(cur, 0, 1, w0);
        M5_SB();
        nxt[1] = mfma32<T>(ka[1][2], qf[2], nxt[1]);
        M5_SB();
        rescale();
        M5_SB();
        nxt[0] = mfma32<T>(ka[0][3], qf[3], nxt[0]);
        nxt[1] = mfma32<T>(ka[1][3], qf[3], nxt[1]);
        M5_SB();
#pragma unroll
        for (int c = 2; c < 4; ++c) { va[0][c] = ld(sbc, voffs[0][c]); va[1][c] = ld(sbc, voffs[1][c]); }
        M5_SB();
        pv_phase(cur, va, w0);
        slot = s1;
    };
    auto last = [&](int kt, f16_t (&cur)[2]) {
        const unsigned char* sbc = lds + slot * TILE_B;
        uint4 va[2][4];
        uint32_t w0[4];
        if (kt == 0) {                                           // a single tile: its V^T was not waited for by any step barrier
            // (the loader waited vmcnt(0) before the prologue barrier when ntiles <= 2)
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { va[0][c] = ld(sbc, voffs[0][c]); va[1][c] = ld(sbc, voffs[1][c]); }
        mask_tile(kt, cur);
        float mx = fmaxf(cur[0][0], cur[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, cur[0][r]), cur[1][r]);
        stats(mx);
        rescale();
        ps0 = 0.f;
        ps1 = 0.f;
        p_half(cur, 0, 0, w0);
        p_half(cur, 0, 1, w0);
        pv_phase(cur, va, w0);
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
    // Which form (the two produce the same bits -- tools/attn_bench.py csum, tests/test_gpu_kernels.py::test_attention_forms_bit_identical --
    // so this is a scheduling decision only): the loader-wave kernel with 96-row workgroups while attn16_kernel's grid would leave
    // the chip at <= 2 workgroups per CU (one utterance's NAR step: 352 workgroups -> 38.6 vs 33.4 us; speaker encoder / prefill
    // shapes -13 %), attn16_kernel's 128-row workgroups, three per CU, beyond (16 x 2240 rows: 433 vs 472 us;
    // profiles/r6o_attn_loader_wave.txt).
    const int64_t wgs128 = (int64_t)((a->Sq + 127) / 128) * a->H * a->B;
    bool loader_form = wgs128 <= 512;
#define M5_A16W(TT, NCWV, WPEV) hipLaunchKernelGGL((attn16w_kernel<TT, NCWV, WPEV>), dim3((a->Sq + 32 * NCWV - 1) / (32 * NCWV), a->H, a->B), dim3(64 * (NCWV + 1)), 0, s, *a)
#ifdef M5_TOOLS
    // M5_ATTN_SCHED: 0 = attn16_kernel always, 3 = loader form always, 5 = 4 computing waves + loader (one workgroup per CU), unset = product rule
    static const int sched = [] { const char* e = m5_tool_env("M5_ATTN_SCHED"); return e ? atoi(e) : -1; }();
    static const int khe = [] { const char* e = m5_tool_env("M5_ATTN_KH"); return e ? atoi(e) : 0; }();
    if (sched == 0 || nw != 4 || var != 0 || khe == 2) loader_form = false;
    if (sched == 3) loader_form = true;
    if (sched == 5 && nw == 4 && var == 0) {
        if (dtype == M5_F16) M5_A16W(F16T, 4, 2); else M5_A16W(BF16T, 4, 2);
    } else
#endif
    if (loader_form) {
        if (dtype == M5_F16) M5_A16W(F16T, 3, 2); else M5_A16W(BF16T, 3, 2);
    } else
#ifdef M5_TOOLS
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
#undef M5_A16W
    M5_CHECK_LAUNCH();
    return M5_OK;
}
