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
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 b8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;

namespace {

constexpr int GROUP_M = 8;

struct Gemm16Params {
    const unsigned char* A; const unsigned char* W; const float* bias; unsigned char* C;
    int64_t lda, ldw, ldc;           // elements
    int64_t sA, sW, sC, sBias;       // batch strides, elements
    int M, N, K;
    int tilesM, tilesN, nblk, group_m;
    unsigned long long* dbg;         // diagnostics: workgroup 0 records {shader clock, 100 MHz wall clock} at entry / exit
    int abl;                         // tools build only (M5_GEMM_ABL): timing ablations with WRONG results -- 1: no fragment reads / MFMAs,
                                     // 2: no operand DMA after the prologue stages, 3: neither (barriers + epilogue only)
    int qkv_stage;                   // QKV scatter: alignment / shape allow the LDS-staged 16-byte-chunk epilogue
    int vec16;                       // 16-bit outputs: rows / batch stride / base / width allow 16-byte row chunks (LDS-staged epilogue)
    int vec_c;                       // C rows / batch stride / base allow 16-byte (fp32) or 8-byte (16-bit) vectors
    int vt_vec;                      // V^T scatter may store 4 consecutive s as one 8-byte vector
    int fast_c;                      // fp32 C: vectors allowed AND N a multiple of 4 AND bias 16-byte aligned: the epilogue's loads / stores
                                     // are whole float4s with ONE row predicate (the general path spends ~1000 issue slots per wave on
                                     // per-element bounds branches: ~2 us per launch at one wave per SIMD)
    M5QkvScatter sc;
    int sec_kind[3];
    // EPI_RESIDUAL_LN: LayerNorm of the updated rows fused behind the residual add (see the epilogue)
    const float* ln_g; const float* ln_b; float ln_eps;
    unsigned char* xn; int64_t ld_xn;            // normalised rows, operand type
    float* ln_part;                              // [tilesM][BM][tilesN] 8-byte words {mean, M2 | launch tag}
    const int* ln_tag_step; int ln_tag;          // launch tag = 1 + (*ln_tag_step * 64 + ln_tag) % 255
    unsigned int* ln_err;                        // += 1 if a wait timed out (never hangs)
    // EPI_Q_CROSS: cross-attention against a short pre-projected memory fused behind its Q projection (see the epilogue)
    const int64_t* xa_tab;                       // per sequence: {K base, V^T base, Le, Lep, step stride of K, of V^T} (elements)
    const int* xa_step; int xa_rows_per_seq;     // memory block of step *xa_step; rows per sequence in A / out
    unsigned char* xa_out; int64_t xa_ld_out;    // attention output [M][n_heads * 64], operand type
    float xa_scale;
    // DLN: LayerNorm DEFERRED into the consuming GEMM.  LN(x) W^T + b = r (xt W'^T - d s) + b' with
    // xt = T(x - cen) a second output of the residual GEMM that produced x (DLN = 1: it also writes per (row, column tile)
    // partial sums {sum, sum of squares} of x - cen), W' = T(W diag gamma), s = row sums of W', b' = b + W beta, and per row
    // d = mean(x - cen), r = rsqrt(var + eps) from the partials (DLN = 2: the consumer; it also leaves cen + d for the next
    // producer of that row).  The LayerNorm launch, its read of x and the normalised copy disappear.
    unsigned char* dl_xt; int64_t dl_ld_xt;      // producer: centred operand-type copy of the updated rows
    float* dl_part; int dl_np;                   // [rows][dl_np] {sum, sumsq}: producer writes entry tn (dl_np = its tilesN), consumer reads all
    const float* dl_cen_in; float* dl_cen_out;   // producer: per-row centre = cen_in + delta (nullptr: 0); column tile 0 leaves it in cen_out
    float* dl_delta;                             // consumer, column tile 0: d of the row (producer: read)
    const float* dl_s; int64_t dl_s_bs;          // consumer: per output column, sum_k W'[n][k] (fp32); batch stride
    float dl_eps, dl_inv_n;                      // consumer: LayerNorm eps, 1 / feature count
    int dl_rows_bs;                              // rows per batch entry in dl_part / dl_cen_* / dl_xt (batched launches)
    // Row-tile list (M5RowTiles): only the listed BM-row tiles exist in the grid.  A batch of sequences of different lengths
    // lives in one padded layout (rows_per_seq rows each); the tiles that hold nothing but padding are not launched, and
    // the XCD-contiguous tile order runs over the compacted list (balanced whatever the lengths are).
    const int* rt_map; int rt_tpb;               // entry e -> batch e / rt_tpb, row tile e % rt_tpb (rt_tpb = 0: flat, row tile e)
    const int* rt_len; int rt_rps;               // real rows per sequence (nullptr: none) and rows per sequence of the padded layout:
                                                 // a deferred-LayerNorm consumer gives the pad rows of its tiles d = r = 0
    const M5RowTiles* rt_host;                   // host side only: the caller's lists (launch16 picks the one of its tile height)
};

// Row-tile list of tile height BM for this launch (host): false = the caller gave lists but none fits (unsupported)
template <int BM>
bool rt_apply(Gemm16Params& p, int batch) {
    const M5RowTiles* rt = p.rt_host;
    if (!rt) return true;
    constexpr int idx = BM == 96 ? 0 : (BM == 128 ? 1 : (BM == 192 ? 2 : -1));
    if constexpr (idx < 0) return false;
    else {
        if (!rt->map[idx] || rt->n[idx] <= 0 || rt->rows_per_seq <= 0 || rt->rows_per_seq % BM) return false;
        p.rt_map = rt->map[idx];
        p.tilesM = rt->n[idx];
        p.rt_tpb = batch > 1 ? rt->rows_per_seq / BM : 0;
        p.rt_len = rt->seq_len;
        p.rt_rps = rt->rows_per_seq;
        return true;
    }
}
constexpr int EPI_RESIDUAL_LN = 101;
constexpr int EPI_Q_CROSS = 102;
constexpr int EPI_SOFTMAX_HEADS = 103;           // C (16-bit) = per-head softmax over the wave's TN*16 columns of acc + bias (xattn_absorb.hip)
constexpr int XA_MAX_KT = 4;                     // fused cross-attention: at most 4 key tiles of 16 (memory length <= 64)

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

// silu with the hardware exp2 / reciprocal (16-bit operand paths; results are rounded to the
// operand type right after).  The epilogue of a K = 1024 SwiGLU GEMM otherwise spends as many
// issue cycles in libm expf + IEEE division as the main loop spends in MFMAs.
__device__ inline float silu_fast(float a) {
    return a * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * a));
}

template <typename T>
__device__ inline uint2 pack4(const float v[4]) {
    using st = typename T::storage;
    st t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = T::from_f32(v[r]);
    return *reinterpret_cast<const uint2*>(t);
}

// s_waitcnt vmcnt(y * n) for the wave-uniform y in [0, Y] (y younger stages of n DMA instructions each may stay in flight)
template <int Y, int JN>
__device__ inline void wait_younger(int y, bool full_share) {
    if constexpr (Y > 0) {
        if (y == Y) {
            if (full_share) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Y * JN) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Y * (JN - 1)) : "memory");
            return;
        }
        wait_younger<Y - 1, JN>(y, full_share);
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// Workgroup tile = (WM x WN) waves, each wave (TM x TN) MFMA tiles of 16x16 (fp32 accumulators
// 4 TM TN per lane); BKB = K-step in BYTES per row (128: 64 halves, 64: 32 halves); NSTAGE LDS
// stages (NSTAGE-1 K-steps of DMA in flight ahead of the MFMAs); OCC = workgroups per CU the
// LDS / register budget is sized for.
//
// What bounds these GEMMs on MI355X is the rate at which one CU can pull operand bytes
// (measured ~45-50 GB/s per CU through L2 -> LDS-DMA, independent of pipeline depth, row
// stride or K phase; profiles/): time ~ max over CUs of the bytes that CU stages.  So besides
// the 128x128 configuration (2 workgroups per CU) there are "region" configurations that cut the
// output into ~240-256 large rectangles, ONE per CU (192x384, 192x192, 96x128): the bytes staged
// per flop drop 1.5-2x and every CU gets the same amount, instead of 128x128 tiles in 2.06 rounds.
// PF ("prefetched fragments", round 3): the K loop keeps TWO fragment register sets and places its one barrier per K-step
// between the two 32-deep MFMA blocks: block 0 of K-step kt runs on fragments read during block 1 of K-step kt-1, block 1
// on fragments read during block 0 -- no MFMA ever waits for an LDS read or sits right behind the barrier.  Measured on
// the 4-wave (one wave per SIMD) 96x128 region kernel: without it a K-step costs ~800 cycles with the DMAs removed against
// 408 cycles of MFMA issue (profiles/r3a_gemm_ablation.txt); hipBLASLt's 128x96 kernel runs these shapes 1.5-1.8x faster
// (profiles/r3a_blas_yardstick.txt).  The K-step's fragments are all in registers at the barrier, so the slot just read
// is refilled one K-step earlier than in the plain loop (NSTAGE K-steps of DMA in flight instead of NSTAGE-1).
// FAST (round 3): the instantiation for aligned, vectorisable operands -- every shape of the NAR / AR engines.  The general
// kernel carries a scalar / per-element fallback next to each vector path of its prologue and epilogues; those fallbacks are
// never taken on the engines' shapes but sit BETWEEN the pieces of the hot path, which a workgroup walks once, instruction-
// cache cold (the step alternates between a dozen kernels): measured, the identical slow path became 6.5 us per launch slower
// when the kernel merely grew from 2100 to 2500 instructions, and the "fixed" cost of a launch is ~10 us against 3.6 us for
// hipBLASLt's kernels (profiles/r3m_*).  FAST compiles the fallbacks out: straight-line code, a fraction of the size.
constexpr int DL_MAX_NP = 8;                      // deferred LayerNorm: at most 8 column tiles of partials per row (D <= 8 x 128)
// LW (round 6, "loader waves"): the workgroup carries NW extra waves -- wave NW + w issues compute wave w's share of the LDS-DMA
// pieces (and nothing else), waits for them with the counted vmcnt and joins the K loop's barriers; the computing waves only
// read fragments and issue MFMAs (their VM counter holds the epilogue's preloads alone).  Why: a `global_load_lds_dwordx4`
// costs the wave that issues it 60-185 cycles among MFMAs and LDS reads (MI355X_MICROARCH.md; tools/probes/lds_dma_rate.hip:
// a wave that does nothing else issues one per 45 cycles, a CU takes 59 B/clk from L2 with >= 4 waves issuing), the 96x128
// region kernel stages 7 pieces per wave and K-step beside 24 MFMAs (384 cycles of matrix pipe): its K-step takes ~1040
// cycles = 27 B/clk/CU, ~775 with the DMA removed (profiles/r3a_gemm_ablation.txt).  A loader wave beside each computing
// wave on its SIMD issues them in parallel (different instruction class, different wave).  (Five stages behind the loader
// waves change nothing, profiles/r6x_*: at ~31 B/clk/CU the K loop runs at the rate the L2s deliver 240 different tile pairs.)
template <typename T, int EPI, int WM, int WN, int TM, int TN, int BKB, int NSTAGE, int OCC, bool PF = false, bool FAST = false, int DLN = 0,
          bool LW = false>
__global__ __launch_bounds__(WM * WN * 64 * (LW ? 2 : 1), (OCC * WM * WN * (LW ? 2 : 1) + 3) / 4) void gemm16_kernel(Gemm16Params p) {
    using st = typename T::storage;
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NW = WM * WN;
    static_assert(!LW || (PF && DLN != 2), "loader waves: prefetched-fragment loop, no consumer-side LDS-DMA prologue");
    constexpr int STAGE = (BM + BN) * BKB;
    // DLN = 2 (consumer): behind the stages, the tile's BM x np partial pairs (LDS-DMA'd in front of the operand stages) and
    // the per-row {d, r} derived from them.  DLN = 1 (producer): the per-wave row sums that the WN waves of a row exchange.
    constexpr int DL_PART = (DLN == 2) ? BM * DL_MAX_NP * 8 : 0;     // consumer: [BM][np] pairs ({d, r} of a row later overwrite its first pair)
    constexpr int DL_BYTES = (DLN == 2) ? DL_PART + 2 * BN * 4 : ((DLN == 1) ? NW * TM * 16 * 8 : 0);   // + bias' and s of the tile's columns
    static_assert(DLN == 0 || FAST, "deferred LayerNorm: FAST instantiations only");
    static_assert(DLN != 1 || EPI == M5_EPI_RESIDUAL, "deferred LayerNorm producer = residual epilogue");
    static_assert(DLN != 2 || EPI == M5_EPI_QKV || EPI == M5_EPI_SWIGLU || EPI == EPI_SOFTMAX_HEADS, "deferred LayerNorm consumers");
    constexpr int RPI = 1024 / BKB;                    // rows per 1 KiB DMA instruction (8 or 16)
    constexpr int NQ = (BM + BN) / RPI;                // DMA instructions per stage; instruction q -> wave q % NW
    constexpr int JN = (NQ + NW - 1) / NW;
    constexpr int KS = BKB / 64;                       // MFMA k-steps (32 halves) per stage
    static_assert(BM % RPI == 0 && BN % RPI == 0, "stage tiling");
    static_assert(NSTAGE * STAGE + DL_BYTES <= 160 * 1024 && NSTAGE >= 2 && NSTAGE <= 10, "LDS budget");
    static_assert((NSTAGE - 2) * JN < 64, "vmcnt range");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NSTAGE * STAGE + DL_BYTES];
    unsigned char* const dl_lds = lds + NSTAGE * STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_raw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = LW && wave_raw >= NW;          // wave-uniform; a loader wave takes computing wave (wave_raw - NW)'s DMA share
    const int wave = loader ? wave_raw - NW : wave_raw;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lg = lane >> 4;
#ifdef M5_TOOLS
    const bool dbg_on = p.dbg && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1);
#else
    constexpr bool dbg_on = false;                     // phase clocks exist in the tools build only
#endif
    unsigned long long* dbg = p.dbg + (blockIdx.x == 0 ? 0 : 8);
    if (dbg_on) { dbg[0] = clock64(); dbg[1] = wall_clock64(); }

    // ---- workgroup -> tile: XCD-contiguous runs (bijective for any nblk), grouped order
    int t;
    {
        const int b = blockIdx.x, q = p.nblk >> 3, r = p.nblk & 7, xcd = b & 7, j = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tiles = p.tilesM * p.tilesN;
    int bz = t / tiles;
    t -= bz * tiles;
    int tm, tn;
    {
        const int width = p.group_m * p.tilesN;
        const int g = t / width, first = g * p.group_m;
        const int gsz = min(p.tilesM - first, p.group_m);
        const int w = t - g * width;
        tm = first + w % gsz;
        tn = w / gsz;
    }
    if (p.rt_map) {                                    // row-tile list: tm indexes the compacted list (p.tilesM = its length)
        const int e = __builtin_amdgcn_readfirstlane(p.rt_map[tm]);
        if (p.rt_tpb) { bz = e / p.rt_tpb; tm = e - bz * p.rt_tpb; }
        else tm = e;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const unsigned char* A = p.A + (int64_t)bz * p.sA * 2;
    const unsigned char* W = p.W + (int64_t)bz * p.sW * 2;

    // ---- LDS-DMA staging: instruction q fills rows [q RPI, (q+1) RPI) of the stage image (A tile
    // rows first, W tile rows after), 1 KiB each.  The DMA writes lane-linear, so the bank swizzle
    // lives in the per-lane SOURCE chunk and (same involution) in the fragment reads:
    //   BKB = 128: lane l -> row l >> 3, slot l & 7,  source chunk slot ^ (row & 7)
    //   BKB =  64: lane l -> row l >> 2, slot l & 3,  source chunk slot ^ ((-(row >> 2)) & 3)
    // (conflict-free for ds_read_b128's four 16-lane groups in both layouts: SQ_LDS_BANK_CONFLICT = 0).
    const int srow = (BKB == 128) ? (lane >> 3) : (lane >> 2);
    const int schunk = (BKB == 128) ? ((lane & 7) ^ srow) : ((lane & 3) ^ ((-(srow >> 2)) & 3));
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // DMA pieces in the scalar-base form: global_load_lds_dwordx4 voffset, s[base:base+1].  The base is the tile's first A / W
    // row at the K-step (wave-uniform: two scalar adds per stage), the 32-bit per-lane offset is tile-local (row clamp folded
    // in), so neither 64-bit per-lane pointers nor their per-K-step vector adds exist, and M0 is written without a save /
    // restore (nothing else in this kernel uses M0): three instructions per piece.  (The batch index comes out of an integer
    // division, which hipcc evaluates on the vector ALU: readfirstlane makes the bases provably uniform.)
    auto uniform_ptr = [](const unsigned char* q) {
        const uint64_t u = (uint64_t)(uintptr_t)q;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        return (const unsigned char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    const unsigned char* Au = uniform_ptr(A + (int64_t)min(m0, p.M - 1) * p.lda * 2);
    const unsigned char* Wu = uniform_ptr(W + (int64_t)min(n0, p.N - 1) * p.ldw * 2);
    uint32_t voff[JN];
    bool piece_a[JN];                                   // wave-uniform: piece j of this wave is rows of A (else of W)
#pragma unroll
    for (int j = 0; j < JN; ++j) {
        const int q = wave + NW * j;                   // wave-uniform
        const int r = q * RPI + srow;                  // row of the stage image
        piece_a[j] = q * RPI < BM;
        if (r < BM) voff[j] = (uint32_t)((int64_t)min(r, max(p.M - 1 - m0, 0)) * p.lda * 2 + schunk * 16);
        else voff[j] = (uint32_t)((int64_t)min(r - BM, max(p.N - 1 - n0, 0)) * p.ldw * 2 + schunk * 16);
    }
    auto piece = [&](int j, const unsigned char* sA, const unsigned char* sW, uint32_t sbase) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voff[j]), "s"(piece_a[j] ? sA : sW), "s"(sbase + j * NW * 1024) : "memory");
    };
    auto stage_load = [&](int stage, int kt) {
        const uint32_t sbase = lds_base + stage * STAGE + wave * 1024;
        const unsigned char* sA = Au + (int64_t)kt * BKB;
        const unsigned char* sW = Wu + (int64_t)kt * BKB;
#pragma unroll
        for (int j = 0; j < JN; ++j)
            if (NQ % NW == 0 || wave + NW * j < NQ) piece(j, sA, sW, sbase);
    };

    // fragment read offsets: row (16 i + l15), k-chunk (4 ks + lg) swizzled like the source
    int foff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        foff[ks] = l15 * BKB + ((BKB == 128) ? (((ks * 4 + lg) ^ (l15 & 7)) << 4) : ((lg ^ ((-(l15 >> 2)) & 3)) << 4));
    const int a_row0 = wm * TM * 16 * BKB, w_row0 = BM * BKB + wn * TN * 16 * BKB;

    f4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

    // A wave whose columns lie in the V section keeps the unswapped operand order (4 consecutive
    // ROWS per lane, for the V^T scatter).  Wave-uniform: the host only picks configurations whose
    // per-wave column span (16 TN) divides the section width Dm.
    bool vblock = false;
    const int Dm = p.sc.n_heads * p.sc.head_dim;
    // (readfirstlane: the tile index comes out of vector-ALU divisions, so hipcc would treat the branch as divergent and run
    // BOTH operand orders under EXEC masks -- which MFMA ignores)
    if constexpr (EPI == M5_EPI_QKV) vblock = __builtin_amdgcn_readfirstlane(p.sec_kind[min((n0 + wn * TN * 16) / Dm, 2)]) == 2;

    // in-place residual: the old C values of this wave's tile are requested BEFORE the K loop (they come
    // from HBM: 11.5 MB of fp32 per 2816 x 1024 launch) and consumed after it -- the epilogue then only writes
    constexpr bool PRELOAD_C = (EPI == M5_EPI_RESIDUAL || EPI == EPI_RESIDUAL_LN) && (TM * TN <= 16);
    float4 oldpre[PRELOAD_C ? TM : 1][PRELOAD_C ? TN : 1];
    auto preload_c = [&]() {
        if constexpr (PRELOAD_C) {
            const float* Cr = reinterpret_cast<const float*>(p.C) + (int64_t)bz * p.sC;
            if (FAST || p.fast_c) {                          // rows clamped (a row past M is never stored), columns by whole float4s
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float* rp = Cr + (int64_t)min(m0 + wm * TM * 16 + i * 16 + l15, p.M - 1) * p.ldc;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = min(n0 + wn * TN * 16 + j * 16 + lg * 4, p.N - 4);
                        oldpre[i][j] = *reinterpret_cast<const float4*>(rp + col);
                    }
                }
                return;
            }
            if constexpr (!FAST) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = m0 + wm * TM * 16 + i * 16 + l15;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * TN * 16 + j * 16 + lg * 4;
                    oldpre[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < p.M && col < p.N) {
                        const float* cp = Cr + (int64_t)row * p.ldc + col;
                        if (p.vec_c && col + 3 < p.N) {
                            oldpre[i][j] = *reinterpret_cast<const float4*>(cp);
                        } else {
                            oldpre[i][j].x = cp[0];
                            if (col + 1 < p.N) oldpre[i][j].y = cp[1];
                            if (col + 2 < p.N) oldpre[i][j].z = cp[2];
                            if (col + 3 < p.N) oldpre[i][j].w = cp[3];
                        }
                    }
                }
            }
            }
        }
    };

    // EPI_Q_CROSS: this wave's 64 columns are ONE head; the memory of a sequence is short (<= 64 keys) and already
    // projected, so its K / V^T fragments for this head are requested now (L2 hits, they land during the K loop).
    // Fragment layouts (16x16x32 MFMA, lane = (l15, lg)): K as the first operand of S^T = K q^T: row = key 16 t + l15,
    // d-chunk 4 ks + lg.  V^T as the first operand of O^T = V^T P^T: row = d 16 dt + l15, and its 8 k-slots of key step
    // sp are keys {32 sp + 4 lg + 0..3, 32 sp + 16 + 4 lg + 0..3} -- exactly the keys whose probabilities lane
    // (l15, lg) holds after S^T (rows 4 lg + r of key tiles 2 sp and 2 sp + 1), so P never leaves its registers.
    uint4 xk[EPI == EPI_Q_CROSS ? XA_MAX_KT : 1][2], xv[EPI == EPI_Q_CROSS ? 4 : 1][EPI == EPI_Q_CROSS ? XA_MAX_KT / 2 : 1];
    int xa_seq = -1, xa_le = 0;
    auto xa_load = [&](int seq) {
        const int64_t* tb = p.xa_tab + (int64_t)seq * 6;
        const int le = (int)tb[2];
        const int64_t lep = tb[3];
        const int64_t stp = *p.xa_step;
        const int hh = (n0 + wn * 64) >> 6;
        const unsigned char* Kh = reinterpret_cast<const unsigned char*>(tb[0]) + (stp * tb[4] + (int64_t)hh * le * 64) * 2;
        const unsigned char* Vh = reinterpret_cast<const unsigned char*>(tb[1]) + (stp * tb[5] + (int64_t)hh * 64 * lep) * 2;
        const int nkt = (le + 15) >> 4;
#pragma unroll
        for (int t = 0; t < XA_MAX_KT; ++t) {
            const int key = min(t * 16 + l15, le - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                xk[t][ks] = (t < nkt) ? *reinterpret_cast<const uint4*>(Kh + ((int64_t)key * 64 + (ks * 4 + lg) * 8) * 2) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int sp = 0; sp < XA_MAX_KT / 2; ++sp) {
                uint4 u = make_uint4(0, 0, 0, 0);
                if (2 * sp < nkt) {
                    const unsigned char* vr = Vh + ((int64_t)(dt * 16 + l15) * lep + sp * 32 + lg * 4) * 2;
                    const uint2 a0 = *reinterpret_cast<const uint2*>(vr);
                    const uint2 a1 = (2 * sp + 1 < nkt) ? *reinterpret_cast<const uint2*>(vr + 32) : make_uint2(0, 0);
                    u = make_uint4(a0.x, a0.y, a1.x, a1.y);
                }
                xv[dt][sp] = u;
            }
        xa_seq = seq;
        xa_le = le;
    };
    if constexpr (EPI == EPI_Q_CROSS) xa_load(min(m0 + wm * TM * 16, p.M - 1) / p.xa_rows_per_seq);

    // bias of this lane's columns (swapped layout: 4 consecutive columns of each of the TN tiles), requested BEFORE the K
    // loop: loaded behind it, the epilogue opened with 16 dependent scalar loads and a vmcnt(0) (~1 us per launch)
    const float* bias = p.bias ? p.bias + (int64_t)bz * p.sBias : nullptr;
    // ... where the register budget allows it (the 16-wave 192x384 SwiGLU region kernel has 128 VGPRs per wave: 24 more live
    // values across the K loop spilled 95 registers to scratch there)
    constexpr int WAVES_PER_SIMD = (OCC * WM * WN + 3) / 4;
    constexpr bool HOIST_B = TM * TN * 4 * (PRELOAD_C ? 2 : 1) + (TM + TN) * 4 * (PF ? 2 : 1) + TN * 4 + 48 <= 512 / WAVES_PER_SIMD;
    float bvh[TN][4];
    auto load_bias = [&]() {
        if (FAST && !bias) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) bvh[j][r] = 0.f;
            return;
        }
        if (FAST || (bias && ((reinterpret_cast<uintptr_t>(bias) & 15) == 0) && (p.N & 3) == 0 && p.N >= 4)) {    // whole float4s, column clamped
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float4 t4 = *reinterpret_cast<const float4*>(bias + min(n0 + wn * TN * 16 + j * 16 + lg * 4, p.N - 4));
                bvh[j][0] = t4.x; bvh[j][1] = t4.y; bvh[j][2] = t4.z; bvh[j][3] = t4.w;
            }
            return;
        }
        if constexpr (!FAST) {
        const bool bvec = bias && ((reinterpret_cast<uintptr_t>(bias) & 15) == 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * TN * 16 + j * 16 + lg * 4;
            if (bvec && col + 3 < p.N) {
                const float4 t4 = *reinterpret_cast<const float4*>(bias + col);
                bvh[j][0] = t4.x; bvh[j][1] = t4.y; bvh[j][2] = t4.z; bvh[j][3] = t4.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) bvh[j][r] = bias ? bias[min(col + r, p.N - 1)] : 0.f;
            }
        }
        }
    };
    // (hipcc cannot sink these loads below the loop: the DMA statements inside it clobber "memory")
    // bias of column tile j as the epilogues consume it: the hoisted registers, or -- no budget for them -- loaded on the spot
    auto bias_j = [&](int j, float (&b4)[4]) {
        if constexpr (DLN == 2) {            // the tile's bias' columns were LDS-DMA'd in front of the operand stages
            const float4 t4 = *reinterpret_cast<const float4*>(dl_lds + DL_PART + (wn * TN * 16 + j * 16 + lg * 4) * 4);
            b4[0] = t4.x; b4[1] = t4.y; b4[2] = t4.z; b4[3] = t4.w;
        } else if constexpr (HOIST_B) {
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = bvh[j][r];
        } else {
            const int col = n0 + wn * TN * 16 + j * 16 + lg * 4;
            if constexpr (FAST) {
                const float4 t4 = bias ? *reinterpret_cast<const float4*>(bias + min(col, p.N - 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
                b4[0] = t4.x; b4[1] = t4.y; b4[2] = t4.z; b4[3] = t4.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) b4[r] = bias ? bias[min(col + r, p.N - 1)] : 0.f;
            }
        }
    };

    const int nk = p.K * 2 / BKB;
    const bool full_share = (NQ % NW == 0) || (wave < NQ % NW);     // this wave issues JN (else JN - 1) DMAs per stage
    // ---- deferred LayerNorm.  Rows of dl_part / dl_cen / dl_xt are numbered over the whole (batched) problem.
    const int dl_row0 = (DLN != 0) ? bz * p.dl_rows_bs + m0 : 0;
    if constexpr (DLN == 2) {
        // consumer: this tile's rows x np partial pairs as ONE flat LDS-DMA copy (rows are contiguous in dl_part), requested in
        // FRONT of the operand stages: the VM counter retires in order, so they have landed when K-step 0 has -- no registers,
        // no exposed latency.  Rows past M repeat the last 16 bytes (finite junk, never stored).
        const uint32_t row_b = (uint32_t)p.dl_np * 8;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane(min(BM, p.M - m0)) * row_b;
        const unsigned char* pb = uniform_ptr(reinterpret_cast<const unsigned char*>(p.dl_part) + (int64_t)dl_row0 * row_b);
        constexpr int NPQ = (BM * DL_MAX_NP * 8 + 1023) / 1024;
#pragma unroll
        for (int q0 = 0; q0 < NPQ; q0 += NW) {
            const int q = q0 + wave;                                   // wave-uniform
            if (q < NPQ && (uint32_t)q * 1024 < total) {
                const uint32_t off = min((uint32_t)q * 1024 + (uint32_t)lane * 16, total - 16);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(off), "s"(pb), "s"(lds_base + NSTAGE * STAGE + q * 1024) : "memory");
            }
        }
        // ... and the tile's BN columns of bias' and of s (row sums of the folded weights): the epilogue reads them from LDS
        // instead of opening with a dependent global round trip.  Columns past N repeat the last four.
        constexpr int NPB = (BN * 4 + 1023) / 1024;                     // pieces per vector
        const unsigned char* vb[2] = {uniform_ptr(reinterpret_cast<const unsigned char*>(p.bias + (int64_t)bz * p.sBias)),
                                      uniform_ptr(reinterpret_cast<const unsigned char*>(p.dl_s + (int64_t)bz * p.dl_s_bs))};
#pragma unroll
        for (int q0 = 0; q0 < 2 * NPB; q0 += NW) {
            const int q = q0 + wave;                                   // wave-uniform: vector q / NPB, piece q % NPB
            if (q < 2 * NPB) {
                const int which = q / NPB, pq = q - which * NPB;
                const uint32_t o = (uint32_t)pq * 1024 + (uint32_t)lane * 16;
                if (o < BN * 4) {
                    const uint32_t off = (uint32_t)min(n0 + (int)(o >> 2), p.N - 4) * 4;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                                 :: "v"(off), "s"(which ? vb[1] : vb[0]), "s"(lds_base + NSTAGE * STAGE + DL_PART + which * BN * 4 + pq * 1024) : "memory");
                }
            }
        }
    }
    float dl_cen[(DLN == 1) ? TM : 1];
    if constexpr (DLN == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
        {
            // centre of the row = the previous producer's centre + the d the consumer in between measured (= the row's mean then)
            const int gr = bz * p.dl_rows_bs + min(m0 + wm * TM * 16 + i * 16 + l15, p.M - 1);
            dl_cen[i] = (p.dl_cen_in ? p.dl_cen_in[gr] : 0.f) + (p.dl_delta ? p.dl_delta[gr] : 0.f);
        }
    }
    auto mfma_block = [&](const uint4 (&af)[TM], const uint4 (&bf)[TN]) {
        if (EPI == M5_EPI_QKV && vblock) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma16<T>(af[i], bf[j], acc[i][j]);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma16<T>(bf[j], af[i], acc[i][j]);
        }
    };
    if constexpr (PF) {
        static_assert(KS == 2, "prefetched-fragment loop: two 32-deep MFMA blocks per K-step");
        static_assert(NQ % NW == 0, "prefetched-fragment loop: every wave issues JN DMAs per stage (constant vmcnt counts)");
        static_assert((NSTAGE - 1) * JN < 64 && TM * TN >= JN, "vmcnt range; one DMA piece behind each of the first JN MFMAs");
        auto read_frags = [&](const unsigned char* sb, int ks, uint4 (&af)[TM], uint4 (&bf)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(sb + a_row0 + i * 16 * BKB + foff[ks]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const uint4*>(sb + w_row0 + j * 16 * BKB + foff[ks]);
        };
        if (!LW || loader) {
#pragma unroll
        for (int sgi = 0; sgi < NSTAGE; ++sgi)
            if (sgi < nk) {
#pragma unroll
                for (int j = 0; j < JN; ++j) piece(j, Au + (int64_t)sgi * BKB, Wu + (int64_t)sgi * BKB, lds_base + sgi * STAGE + wave * 1024);
            }
        }
        if constexpr (LW) {
            if (loader) {
                // the loader's K loop: the computing waves' barriers, one for one (kstep below), the counted waits and the refills
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSTAGE - 1) * JN) : "memory");       // K-step 0 has landed (all NSTAGE stages are whole
                if (nk < NSTAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // ... unless the problem is shorter)
                __syncthreads();
                int slot = 0, kt = 0;
                auto lstep = [&](int kt_, auto y_tag, auto refill_tag) {
                    constexpr int Y = decltype(y_tag)::value;
                    constexpr bool REFILL = decltype(refill_tag)::value;
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Y * JN) : "memory");
                    __syncthreads();
                    if constexpr (REFILL) {
                        const unsigned char* sA = Au + (int64_t)(kt_ + NSTAGE) * BKB;
                        const unsigned char* sW = Wu + (int64_t)(kt_ + NSTAGE) * BKB;
                        const uint32_t sbase = lds_base + slot * STAGE + wave * 1024;
#pragma unroll
                        for (int j = 0; j < JN; ++j) piece(j, sA, sW, sbase);
                    }
                    slot = (slot + 1 == NSTAGE) ? 0 : slot + 1;
                };
                for (; kt + NSTAGE < nk; ++kt) lstep(kt, std::integral_constant<int, NSTAGE - 2>{}, std::true_type{});
                auto ltail = [&](auto r_tag) {
                    constexpr int R = decltype(r_tag)::value;
                    if (nk - 1 - kt == R) { lstep(kt, std::integral_constant<int, (R - 1 < NSTAGE - 2 ? R - 1 : NSTAGE - 2)>{}, std::false_type{}); ++kt; }
                };
                if constexpr (NSTAGE >= 5) ltail(std::integral_constant<int, 4>{});
                if constexpr (NSTAGE >= 4) ltail(std::integral_constant<int, 3>{});
                if constexpr (NSTAGE >= 3) ltail(std::integral_constant<int, 2>{});
                ltail(std::integral_constant<int, 1>{});
                return;
            }
        }
        // old C and bias are requested BEHIND the operand stages (the VM counter retires in order: in front of them, the first
        // MFMA would wait for 11.5 MB of cold fp32 residual).  These loads are younger than K-step 0, so the counted wait
        // below over-waits a little on the first K-step only -- never under-waits.
        preload_c();
        if constexpr (HOIST_B && DLN != 2) load_bias();
        if constexpr (!LW) wait_younger<NSTAGE - 1, JN>(min(NSTAGE - 1, nk - 1), true);             // K-step 0 has landed
        __syncthreads();
        uint4 af0[TM], bf0[TN], af1[TM], bf1[TN];
        read_frags(lds, 0, af0, bf0);
        int slot = 0;                                      // stage slot of K-step kt
        // One K-step with a successor: block 0 under the reads of block 1; then K-step kt+1 has landed once at most Y younger
        // stages remain in flight, and at the barrier every wave holds ALL fragments of K-step kt in registers (lgkmcnt(0)),
        // so its slot is refilled at once (REFILL: K-step kt+NSTAGE exists), one piece behind each of the first JN MFMAs of
        // block 1, which runs under the reads of the next block 0.  The issue order is pinned (sched_barrier): left alone,
        // hipcc sinks the fragment reads to just in front of their first use and the MFMAs wait for LDS again.
        // The steady-state loop is ONE basic block (constant wait, unconditional refill): with control flow between the two
        // MFMA blocks hipcc rotates the 48 accumulator registers through copies every iteration.
        auto kstep = [&](int kt, auto y_tag, auto refill_tag) {
            constexpr int Y = decltype(y_tag)::value;
            constexpr bool REFILL = decltype(refill_tag)::value;
            read_frags(lds + slot * STAGE, 1, af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            const int nslot = (slot + 1 == NSTAGE) ? 0 : slot + 1;
            if constexpr (!LW) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Y * JN) : "memory");
            __syncthreads();
            read_frags(lds + nslot * STAGE, 0, af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* sA = Au + (int64_t)(kt + NSTAGE) * BKB;
            const unsigned char* sW = Wu + (int64_t)(kt + NSTAGE) * BKB;
            const uint32_t sbase = lds_base + slot * STAGE + wave * 1024;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (EPI == M5_EPI_QKV && vblock) acc[i][j] = mfma16<T>(af1[i], bf1[j], acc[i][j]);
                    else acc[i][j] = mfma16<T>(bf1[j], af1[i], acc[i][j]);
                    if constexpr (REFILL && !LW) {
                        if (i * TN + j < JN) {
                            piece(i * TN + j, sA, sW, sbase);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            slot = nslot;
        };
        int kt = 0;
#ifdef M5_TOOLS
        if (p.abl & 2) {
            for (; kt + NSTAGE < nk; ++kt) kstep(kt, std::integral_constant<int, NSTAGE - 2>{}, std::false_type{});
        }
#endif
        for (; kt + NSTAGE < nk; ++kt) kstep(kt, std::integral_constant<int, NSTAGE - 2>{}, std::true_type{});
        // the last min(nk, NSTAGE) - 1 K-steps with a successor: r K-steps follow, min(NSTAGE-2, r-1) of them still in flight
        auto tail = [&](auto r_tag) {
            constexpr int R = decltype(r_tag)::value;
            if (nk - 1 - kt == R) { kstep(kt, std::integral_constant<int, (R - 1 < NSTAGE - 2 ? R - 1 : NSTAGE - 2)>{}, std::false_type{}); ++kt; }
        };
        static_assert(NSTAGE <= 5, "tail unrolled for up to 4 K-steps");
        if constexpr (NSTAGE >= 5) tail(std::integral_constant<int, 4>{});
        if constexpr (NSTAGE >= 4) tail(std::integral_constant<int, 3>{});
        if constexpr (NSTAGE >= 3) tail(std::integral_constant<int, 2>{});
        tail(std::integral_constant<int, 1>{});
        read_frags(lds + slot * STAGE, 1, af1, bf1);       // last K-step: nothing left to wait for or to refill
        mfma_block(af0, bf0);
        mfma_block(af1, bf1);
    } else {
    preload_c();
    if constexpr (HOIST_B && DLN != 2) load_bias();
#pragma unroll
    for (int sgi = 0; sgi < NSTAGE - 1; ++sgi)
        if (sgi < nk) stage_load(sgi, sgi);
    int slot = 0;                                      // stage slot of K-step kt
    for (int kt = 0; kt < nk; ++kt) {
        // K-step kt has landed once at most min(NSTAGE-2, nk-1-kt) younger stages remain in flight
        // (vmcnt is per wave: a wave that issues JN - 1 instructions per stage counts with JN - 1)
        const int younger = min(NSTAGE - 2, nk - 1 - kt);
        wait_younger<NSTAGE - 2, JN>(younger, full_share);
        __syncthreads();
        // the slot read during K-step kt-1 is free now: refill it with K-step kt+NSTAGE-1
#ifdef M5_TOOLS
        if (kt + NSTAGE - 1 < nk && !(p.abl & 2)) stage_load(slot == 0 ? NSTAGE - 1 : slot - 1, kt + NSTAGE - 1);
#else
        if (kt + NSTAGE - 1 < nk) stage_load(slot == 0 ? NSTAGE - 1 : slot - 1, kt + NSTAGE - 1);
#endif
        const unsigned char* sb = lds + slot * STAGE;
        slot = (slot + 1 == NSTAGE) ? 0 : slot + 1;
#ifdef M5_TOOLS
        if (p.abl & 1) continue;
#endif
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(sb + a_row0 + i * 16 * BKB + foff[ks]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const uint4*>(sb + w_row0 + j * 16 * BKB + foff[ks]);
            mfma_block(af, bf);
        }
    }
    }
    if (dbg_on) { dbg[2] = clock64(); dbg[3] = wall_clock64(); }
    if constexpr (DLN == 2) {
        // per row {d = mean(x - cen), r = 1 / sqrt(var + eps)} from the np partial pairs (summed in a fixed order: every column
        // tile of a row derives the same two numbers); column tile 0 leaves the row's new centre for the next producer.
        // (visible to the epilogues behind their __syncthreads(); the partials are older than every K-step's barrier)
        const int np = p.dl_np;
        // Row-tile lists: the tile's rows past its sequence's own length are padding that another tile height's producer may
        // never have written (stale partials -> r up to 1 / sqrt(eps)): they get d = r = 0, i.e. the bias alone -- finite.
        int real_rows = BM;
        if (p.rt_len) {
            const int sq = p.rt_tpb ? bz : m0 / p.rt_rps;               // (a tile never straddles sequences: rows_per_seq % BM == 0)
            real_rows = p.rt_len[sq] - (p.rt_tpb ? m0 : m0 - sq * p.rt_rps);
        }
        for (int r = tid; r < BM; r += NW * 64) {
            float2* pr = reinterpret_cast<float2*>(dl_lds + r * np * 8);
            float s1 = 0.f, s2 = 0.f;
            for (int n = 0; n < np; ++n) { const float2 v = pr[n]; s1 += v.x; s2 += v.y; }
            float d = s1 * p.dl_inv_n;
            const float var = fmaxf(s2 * p.dl_inv_n - d * d, 0.f);
            float rs = 1.0f / sqrtf(var + p.dl_eps);
            if (r >= real_rows) { d = 0.f; rs = 0.f; }
            pr[0] = make_float2(d, rs);                                  // in place: this thread is the row's only reader
            if (tn == 0 && p.dl_delta && m0 + r < p.M) p.dl_delta[dl_row0 + r] = d;          // a store, no load: nothing here waits on memory
        }
    }
    // {d, r} of local row lr / the folded weights' row sums of the tile's columns lcol .. lcol + 3 (tile-local column; both in
    // LDS, read behind the epilogue's barrier)
    auto dl_row = [&](int lr) -> float2 { return *reinterpret_cast<const float2*>(dl_lds + lr * p.dl_np * 8); };
    auto dl_s4 = [&](int lcol, float (&s4)[4]) {
        const float4 t4 = *reinterpret_cast<const float4*>(dl_lds + DL_PART + BN * 4 + lcol * 4);
        s4[0] = t4.x; s4[1] = t4.y; s4[2] = t4.z; s4[3] = t4.w;
    };
    // ---- epilogue.  swapped layout: acc[i][j][r] = C[mw + 16 i + l15][nw + 16 j + 4 lg + r]
    constexpr bool F32OUT = (EPI == M5_EPI_F32 || EPI == M5_EPI_RESIDUAL || EPI == EPI_RESIDUAL_LN);
    unsigned char* Cb = p.C + (int64_t)bz * p.sC * (F32OUT ? 4 : 2);
    const int mw = m0 + wm * TM * 16, nw = n0 + wn * TN * 16;

    // ---- Q projection + cross-attention against a short memory (NAR decoder: 39-60 keys per sequence).  The wave owns
    // 48 queries x one head of q: q goes to LDS in the operand type (the same rounding the separate kernels apply),
    // then per 16-query tile S^T = K q^T (2 MFMAs per key tile), softmax over the keys (4 per lane and key tile, the
    // rest across the 4 lane groups), O^T = V^T P^T with P taken from the S^T registers, and the 16 x 64 output tile
    // is stored as 8-byte runs of d.  The separate attention launch (10 us for ~0.2 us of work) and the q round trip
    // disappear.  A 16-query tile never straddles sequences (rows per sequence is a multiple of 16); a 48-row wave
    // tile may, in which case the fragments are re-read for the tile that belongs to the next sequence.
    if constexpr (EPI == EPI_Q_CROSS) {
        static_assert(TN == 4, "one head per wave");
        constexpr int RBQ = 128 + 16;
        __syncthreads();                                      // the stage buffers are free
        unsigned char* wsq = lds + wave * (TM * 16 * RBQ);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float bv4[4];
            bias_j(j, bv4);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float vq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) vq[r] = acc[i][j][r] + bv4[r];
                *reinterpret_cast<uint2*>(wsq + (i * 16 + l15) * RBQ + (j * 16 + lg * 4) * 2) = pack4<T>(vq);
            }
        }
        __builtin_amdgcn_wave_barrier();
        st* Out = reinterpret_cast<st*>(p.xa_out);
        const int hh = (n0 + wn * 64) >> 6;
        const float sl2 = p.xa_scale * 1.4426950408889634f;   // softmax in the exp2 domain (v_exp_f32), like attn16
        if (dbg_on) { dbg[6] = wall_clock64(); }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row0 = mw + i * 16;
            if (row0 >= p.M) break;                            // wave-uniform
            const int seq = row0 / p.xa_rows_per_seq;
            if (seq != xa_seq) xa_load(seq);                   // wave-uniform, rare (tile on a sequence boundary)
            const int le = xa_le, nkt = (le + 15) >> 4;
            uint4 qf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(wsq + (i * 16 + l15) * RBQ + (ks * 4 + lg) * 16);
            f4_t sc[XA_MAX_KT];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < XA_MAX_KT; ++t) {
                sc[t] = f4_t{0.f, 0.f, 0.f, 0.f};
                if (t < nkt) {
                    sc[t] = mfma16<T>(xk[t][0], qf[0], sc[t]);
                    sc[t] = mfma16<T>(xk[t][1], qf[1], sc[t]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {                  // sc[t][r] = q_{l15} . k_{16 t + 4 lg + r}, in log2 units
                    const float vs = (t * 16 + lg * 4 + r < le) ? sc[t][r] * sl2 : -INFINITY;
                    sc[t][r] = vs;
                    mx = fmaxf(mx, vs);
                }
            }
            mx = fmaxf(mx, lane_xor16(mx));
            mx = fmaxf(mx, lane_xor32(mx));
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < XA_MAX_KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(sc[t][r] - mx); sc[t][r] = e; sum += e; }
            sum += lane_xor16(sum);
            sum += lane_xor32(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
            f4_t o[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sp = 0; sp < XA_MAX_KT / 2; ++sp) {
                if (2 * sp < nkt) {
                    float p0[4] = {sc[2 * sp][0], sc[2 * sp][1], sc[2 * sp][2], sc[2 * sp][3]};
                    float p1[4] = {sc[2 * sp + 1][0], sc[2 * sp + 1][1], sc[2 * sp + 1][2], sc[2 * sp + 1][3]};
                    const uint2 u0 = pack4<T>(p0), u1 = pack4<T>(p1);
                    const uint4 pf = make_uint4(u0.x, u0.y, u1.x, u1.y);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16<T>(xv[dt][sp], pf, o[dt]);
                }
            }
            const int row = row0 + l15;                        // o[dt][r] = O[query l15][d = 16 dt + 4 lg + r]
            if (row < p.M) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    float ov[4] = {o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
                    *reinterpret_cast<uint2*>(Out + (int64_t)row * p.xa_ld_out + hh * 64 + dt * 16 + lg * 4) = pack4<T>(ov);
                }
            }
        }
        if (dbg_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[4] = clock64(); dbg[5] = wall_clock64(); }
        return;
    }

    // ---- residual add + LayerNorm of the updated rows (NAR decoder: every LayerNorm but the first follows a
    // residual GEMM).  A row is spread over the tilesN workgroups of its row tile, so they exchange per-tile
    // (mean, M2) partials through memory and meet at a per-row-tile counter: each then normalises the values it
    // still holds in registers and writes its 128 columns of xn -- the separate LayerNorm launch (6 us, a read of
    // x and a round trip of xn) disappears.  Requirements checked by the host: the whole grid is co-resident
    // (one workgroup per CU, nblk <= CUs), N = tilesN * BN exactly, batch 1.  The partials travel as agent-scope
    // relaxed 8-byte atomics (write-through / L2-bypassing on this multi-XCD part) -- no release fence, which would
    // write back the whole L2.  Waits are bounded (error flag, no hang).
    if constexpr (EPI == EPI_RESIDUAL_LN) {
        static_assert(PRELOAD_C && WN == 2, "fused LayerNorm epilogue: two waves per row, preloaded residual");
        float v[TM][TN][4];
        float* Cf = reinterpret_cast<float*>(Cb);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = nw + j * 16 + lg * 4;
            float bv4[4];
            bias_j(j, bv4);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float4 o = oldpre[i][j];
                v[i][j][0] = o.x + (acc[i][j][0] + bv4[0]); v[i][j][1] = o.y + (acc[i][j][1] + bv4[1]);
                v[i][j][2] = o.z + (acc[i][j][2] + bv4[2]); v[i][j][3] = o.w + (acc[i][j][3] + bv4[3]);
                const int row = mw + i * 16 + l15;
                if (row < p.M) *reinterpret_cast<float4*>(Cf + (int64_t)row * p.ldc + col) = make_float4(v[i][j][0], v[i][j][1], v[i][j][2], v[i][j][3]);
            }
        }
        // per-row (mean, M2) over this wave's 64 columns: lane sums, then the 4 lane groups
        float mean_w[TM], m2_w[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) sm += (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
            sm += lane_xor16(sm);
            sm += lane_xor32(sm);
            mean_w[i] = sm * (1.0f / (TN * 16));
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = v[i][j][r] - mean_w[i]; q += d * d; }
            q += lane_xor16(q);
            q += lane_xor32(q);
            m2_w[i] = q;
        }
        __syncthreads();                                      // the stage buffers are free
        float* sw = reinterpret_cast<float*>(lds);            // [NW][TM*16][2], then [BM][2] combined at +NW*TM*32
        if (lg == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                sw[((wave * TM + i) * 16 + l15) * 2] = mean_w[i];
                sw[((wave * TM + i) * 16 + l15) * 2 + 1] = m2_w[i];
            }
        }
        __syncthreads();
        // exchange: one 8-byte word per (row, column tile) = {mean, M2 with the launch tag in its low 8 mantissa bits};
        // a reader polls the tilesN words of its row until all carry this launch's tag.  Every launch overwrites every
        // word it later reads and consecutive launches on one scratch carry different tags, so a word is either the
        // previous launch's (other tag) or this launch's -- no counters, no store acknowledgement on the critical path.
        const unsigned tag = 1u + (unsigned)(((p.ln_tag_step ? *p.ln_tag_step : 0) * 64 + p.ln_tag) % 255);
        // Layout [row tile][column tile][row]: a workgroup's 96 words are contiguous (768 bytes, written by two store
        // instructions, no other writer in their lines).  (The first version interleaved the 8 column tiles of a row in
        // one 64-byte segment, the pattern that cost csrc/ar_mega.hip 0.3 us per writer; here the re-layout measured the
        // same, 3.30 vs 3.21 ms per step without the fusion: what this exchange pays for is the start skew of the 8
        // workgroups of a row tile plus the round trip, not the stores.)
        unsigned long long* words = reinterpret_cast<unsigned long long*>(p.ln_part) + ((int64_t)tm * p.tilesN) * BM;
        float* cmb = sw + NW * TM * 32;                       // [BM][2]: mean, rstd of the whole row
        if (tid < BM) {                                        // row tid of the tile: merge its two waves (64 columns each)
            const int wmr = tid / (TM * 16), lr = tid - wmr * (TM * 16);
            const float ma = sw[((wmr * WN + 0) * TM * 16 + lr) * 2], qa = sw[((wmr * WN + 0) * TM * 16 + lr) * 2 + 1];
            const float mb = sw[((wmr * WN + 1) * TM * 16 + lr) * 2], qb = sw[((wmr * WN + 1) * TM * 16 + lr) * 2 + 1];
            const float mt = 0.5f * (ma + mb);
            const float qt = (qa + qb) + (float)(TN * 16) * ((ma - mt) * (ma - mt) + (mb - mt) * (mb - mt));
            const unsigned long long wv = ((unsigned long long)__float_as_uint(mt) << 32) | ((__float_as_uint(qt) & 0xffffff00u) | tag);
            unsigned long long* rw = words + tid;
            __hip_atomic_store(rw + tn * BM, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float mj[16], qj[16];
            const int nt = p.tilesN;
            int spins = 0;
            bool all;
            do {
                all = true;
#pragma unroll
                for (int n = 0; n < 16; ++n) {
                    if (n < nt) {
                        const unsigned long long u = __hip_atomic_load(rw + n * BM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        all = all && ((unsigned)(u & 0xffu) == tag);
                        mj[n] = __uint_as_float((unsigned)(u >> 32));
                        qj[n] = __uint_as_float((unsigned)(u & 0xffffff00u));
                    }
                }
                if (!all) {
                    if (++spins > (1 << 20)) { atomicAdd(p.ln_err, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            } while (!all);
            float msum = 0.f;
#pragma unroll
            for (int n = 0; n < 16; ++n) if (n < nt) msum += mj[n];
            const float mean = msum / (float)nt;
            float q = 0.f, dev = 0.f;
#pragma unroll
            for (int n = 0; n < 16; ++n) if (n < nt) { q += qj[n]; dev += (mj[n] - mean) * (mj[n] - mean); }
            const float var = (q + (float)BN * dev) / (float)p.N;
            cmb[tid * 2] = mean;
            cmb[tid * 2 + 1] = 1.0f / sqrtf(var + p.ln_eps);
        }
        __syncthreads();
        st* Xn = reinterpret_cast<st*>(p.xn);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int lrow = wm * TM * 16 + i * 16 + l15, row = m0 + lrow;
            const float mean = cmb[lrow * 2], rstd = cmb[lrow * 2 + 1];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = nw + j * 16 + lg * 4;
                const float4 g = *reinterpret_cast<const float4*>(p.ln_g + col), bb = *reinterpret_cast<const float4*>(p.ln_b + col);
                float o[4] = {(v[i][j][0] - mean) * rstd * g.x + bb.x, (v[i][j][1] - mean) * rstd * g.y + bb.y,
                              (v[i][j][2] - mean) * rstd * g.z + bb.z, (v[i][j][3] - mean) * rstd * g.w + bb.w};
                if (row < p.M) *reinterpret_cast<uint2*>(Xn + (int64_t)row * p.ld_xn + col) = pack4<T>(o);
            }
        }
        if (dbg_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[4] = clock64(); dbg[5] = wall_clock64(); }
        return;
    }

    // ---- Q / K / V^T scatter, one head (64 columns) per wave: stage through LDS, store 16-byte chunks.
    // Q, K: rows of 128 bytes ([s][64 d]) go out whole.  V^T: the unswapped accumulator layout gives each
    // lane 4 consecutive s of one d; the tile is staged transposed ([d][s]) and leaves as 16-byte runs of s.
    if constexpr (EPI == M5_EPI_QKV && TN == 4) {
        constexpr int RBQ = 128 + 16;                         // Q/K stage row: 64 d
        constexpr int RBV = TM * 32 + 16;                     // V^T stage row: TM*16 s
        constexpr int WSZ = (TM * 16 * RBQ > 64 * RBV) ? TM * 16 * RBQ : 64 * RBV;
        static_assert(!FAST || NW * WSZ <= NSTAGE * STAGE, "FAST QKV: the staged scatter must fit the stage buffers");
        if constexpr (NW * WSZ <= NSTAGE * STAGE) {
            if (FAST || p.qkv_stage) {
                if constexpr (DLN == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS only: the d store stays in flight
                else __syncthreads();
                unsigned char* ws = lds + wave * WSZ;
                const int ncol0 = n0 + wn * 64;                // this wave's first column: one (section, head)
                if (ncol0 >= p.N) return;                      // (whole-head granularity: N is a multiple of 64)
                const int kind = p.sec_kind[min(ncol0 / Dm, 2)];
                const int hh = (ncol0 % Dm) >> 6;
                const float* biasp = p.bias ? p.bias + (int64_t)bz * p.sBias : nullptr;
                const int mrow0 = m0 + wm * TM * 16;
                if (!vblock) {
                    float2 drow[(DLN == 2) ? TM : 1];
                    if constexpr (DLN == 2) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) drow[i] = dl_row(wm * TM * 16 + i * 16 + l15);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float bv4[4], sv4[4] = {0.f, 0.f, 0.f, 0.f};
                        bias_j(j, bv4);
                        if constexpr (DLN == 2) dl_s4(wn * 64 + j * 16 + lg * 4, sv4);
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            float v[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if constexpr (DLN == 2) v[r] = drow[i].y * (acc[i][j][r] - drow[i].x * sv4[r]) + bv4[r];
                                else v[r] = acc[i][j][r] + bv4[r];
                            }
                            *reinterpret_cast<uint2*>(ws + (i * 16 + l15) * RBQ + (j * 16 + lg * 4) * 2) = pack4<T>(v);
                        }
                    }
                    st* base = (kind == 0) ? reinterpret_cast<st*>(p.sc.q) + hh * p.sc.q_hs : reinterpret_cast<st*>(p.sc.k) + hh * p.sc.k_hs;
                    const int64_t bs = (kind == 0) ? p.sc.q_bs : p.sc.k_bs, rs = (kind == 0) ? p.sc.q_rs : p.sc.k_rs;
                    const int rr = lane >> 3, ch = lane & 7;
#pragma unroll
                    for (int pass = 0; pass < TM * 2; ++pass) {
                        const int r = pass * 8 + rr, row = mrow0 + r;
                        if (row < p.M) {
                            const int b = row / p.sc.rows_per_batch, sq = row - b * p.sc.rows_per_batch;
                            const uint4 val = *reinterpret_cast<const uint4*>(ws + r * RBQ + ch * 16);
                            *reinterpret_cast<uint4*>(base + b * bs + (int64_t)sq * rs + ch * 8) = val;
                        }
                    }
                } else {
                    // unswapped: acc[i][j][r] = C[mrow0 + 16 i + 4 lg + r][ncol0 + 16 j + l15]
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float bvv, svv = 0.f;
                        if constexpr (DLN == 2) {
                            bvv = reinterpret_cast<const float*>(dl_lds + DL_PART)[wn * 64 + j * 16 + l15];
                            svv = reinterpret_cast<const float*>(dl_lds + DL_PART + BN * 4)[wn * 64 + j * 16 + l15];
                        } else {
                            bvv = biasp ? biasp[min(ncol0 + j * 16 + l15, p.N - 1)] : 0.f;
                        }
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            float v[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if constexpr (DLN == 2) {
                                    const float2 dr = dl_row(wm * TM * 16 + i * 16 + lg * 4 + r);
                                    v[r] = dr.y * (acc[i][j][r] - dr.x * svv) + bvv;
                                } else {
                                    v[r] = acc[i][j][r] + bvv;
                                }
                            }
                            *reinterpret_cast<uint2*>(ws + (j * 16 + l15) * RBV + (i * 16 + lg * 4) * 2) = pack4<T>(v);
                        }
                    }
                    constexpr int CPRV = TM * 2, RPPV = 64 / CPRV;      // 16-byte chunks (8 s) per d-row, d-rows per pass
                    const int rr = lane / CPRV, ch = lane - rr * CPRV;
                    st* base = reinterpret_cast<st*>(p.sc.vt) + hh * p.sc.vt_hs;
                    const int row = mrow0 + ch * 8;                      // 8 consecutive rows: same batch (rows_per_batch % 8 == 0)
                    const int b = row / p.sc.rows_per_batch, sq = row - b * p.sc.rows_per_batch;
#pragma unroll
                    for (int pass = 0; pass < (64 + RPPV - 1) / RPPV; ++pass) {
                        const int d = pass * RPPV + rr;
                        if (rr < RPPV && d < 64 && row < p.M) {
                            const uint4 val = *reinterpret_cast<const uint4*>(ws + d * RBV + ch * 16);
                            *reinterpret_cast<uint4*>(base + b * p.sc.vt_bs + (int64_t)d * p.sc.vt_ds + sq) = val;
                        }
                    }
                }
                if (dbg_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[4] = clock64(); dbg[5] = wall_clock64(); }
                return;
            }
        }
    }

    if (EPI == M5_EPI_QKV && vblock) {
        // unswapped layout: acc[i][j][r] = C[mw + 16 i + 4 lg + r][nw + 16 j + l15]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = nw + j * 16 + l15;
            if (col >= p.N) continue;
            const float bv = bias ? bias[col] : 0.f;
            const int c = col % Dm, hh = c / p.sc.head_dim, dd = c % p.sc.head_dim;
            st* vt = reinterpret_cast<st*>(p.sc.vt) + hh * p.sc.vt_hs + (int64_t)dd * p.sc.vt_ds;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = mw + i * 16 + lg * 4;
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

    // ---- absorbed cross-attention scores (xattn_absorb.hip): the wave's TN*16 columns are ONE head's (padded) key block;
    // acc + c is the scaled score, its softmax over those columns leaves as 16-bit probabilities (the A operand of the
    // P.B GEMM).  Padded keys carry c = -1e30 -> probability exactly 0.  A row's values sit in the 4 lane groups of a lane's
    // l15 (4 consecutive columns per 16-column tile each): two xor-shuffles finish max and sum.
    if constexpr (EPI == EPI_SOFTMAX_HEADS) {
        constexpr int RB = TN * 32, RBS = RB + 16, CPR = RB / 16, RPP = 64 / CPR;
        static_assert(NW * TM * 16 * RBS <= NSTAGE * STAGE, "the output tile is staged in the (dead) K-loop stages");
        if constexpr (DLN == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
        unsigned char* ws = lds + wave * (TM * 16 * RBS);
        float bvs[TN][4], svs[(DLN == 2) ? TN : 1][4];
#pragma unroll
        for (int j = 0; j < TN; ++j) bias_j(j, bvs[j]);
        if constexpr (DLN == 2) {
#pragma unroll
            for (int j = 0; j < TN; ++j) dl_s4(wn * TN * 16 + j * 16 + lg * 4, svs[j]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float v[TN][4];
            float mx = -INFINITY;
            float2 dr = make_float2(0.f, 1.f);
            if constexpr (DLN == 2) dr = dl_row(wm * TM * 16 + i * 16 + l15);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = acc[i][j][r];
                    if constexpr (DLN == 2) a = dr.y * (a - dr.x * svs[j][r]);
                    v[j][r] = (a + bvs[j][r]) * 1.4426950408889634f;
                    mx = fmaxf(mx, v[j][r]);
                }
            mx = fmaxf(mx, lane_xor16(mx));
            mx = fmaxf(mx, lane_xor32(mx));
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[j][r] = __builtin_amdgcn_exp2f(v[j][r] - mx); sum += v[j][r]; }
            sum += lane_xor16(sum);
            sum += lane_xor32(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float o[4] = {v[j][0] * inv, v[j][1] * inv, v[j][2] * inv, v[j][3] * inv};
                *reinterpret_cast<uint2*>(ws + (i * 16 + l15) * RBS + (j * 16 + lg * 4) * 2) = pack4<T>(o);
            }
        }
        const int rr = lane / CPR, ch = lane - rr * CPR;
        st* Cd = reinterpret_cast<st*>(p.C) + (int64_t)bz * p.sC;
        const int mrow0 = m0 + wm * TM * 16, ocol = nw + ch * 8;
#pragma unroll
        for (int pass = 0; pass < (TM * 16 + RPP - 1) / RPP; ++pass) {
            const int r = pass * RPP + rr;
            if (rr < RPP && r < TM * 16 && mrow0 + r < p.M && ocol < p.N) {
                const uint4 val = *reinterpret_cast<const uint4*>(ws + r * RBS + ch * 16);
                *reinterpret_cast<uint4*>(Cd + (int64_t)(mrow0 + r) * p.ldc + ocol) = val;
            }
        }
        return;
    }

    // ---- 16-bit outputs (plain / SiLU / SwiGLU): stage the wave's output tile through LDS and store
    // whole 16-byte row chunks.  Straight from the accumulator layout a SwiGLU store instruction writes
    // 16 rows x 16 bytes (4 bytes per lane); measured 10 us of a 47 us K = 1024 SwiGLU launch went into
    // that epilogue.  The stage buffers are dead after the last K-step, each wave owns a private slice.
    if constexpr (EPI == M5_EPI_DT || EPI == M5_EPI_SILU_DT || EPI == M5_EPI_SWIGLU) {
        constexpr int OUTC = (EPI == M5_EPI_SWIGLU) ? TN * 8 : TN * 16;   // output columns of the wave tile
        constexpr int RB = OUTC * 2;                                      // bytes per output row
        constexpr int RBS = RB + 16;                                      // padded LDS row stride
        constexpr int CPR = RB / 16, RPP = 64 / (CPR > 0 ? CPR : 1);      // 16-byte chunks per row, rows per pass
        static_assert(!FAST || (RB % 16 == 0 && NW * TM * 16 * RBS <= NSTAGE * STAGE), "FAST 16-bit epilogue: staged stores must fit");
        if constexpr (RB % 16 == 0 && NW * TM * 16 * RBS <= NSTAGE * STAGE) {
            if (FAST || p.vec16) {
                if constexpr (DLN == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else __syncthreads();                                     // every wave is done reading the stages
                unsigned char* ws = lds + wave * (TM * 16 * RBS);
                float2 drow[(DLN == 2) ? TM : 1];
                if constexpr (DLN == 2) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) drow[i] = dl_row(wm * TM * 16 + i * 16 + l15);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float bv4[4], sv4[4] = {0.f, 0.f, 0.f, 0.f};
                    bias_j(j, bv4);
                    if constexpr (DLN == 2) dl_s4(wn * TN * 16 + j * 16 + lg * 4, sv4);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (DLN == 2) v[r] = drow[i].y * (acc[i][j][r] - drow[i].x * sv4[r]) + bv4[r];
                            else v[r] = acc[i][j][r] + bv4[r];
                        }
                        unsigned char* dst = ws + (i * 16 + l15) * RBS;
                        if constexpr (EPI == M5_EPI_SWIGLU) {
                            st o[2];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const float a = round_dt<T>(v[2 * h]), b = round_dt<T>(v[2 * h + 1]);
                                const float sl = round_dt<T>(silu_fast(a));
                                o[h] = T::from_f32(sl * b);
                            }
                            *reinterpret_cast<uint32_t*>(dst + (j * 8 + lg * 2) * 2) = *reinterpret_cast<const uint32_t*>(o);
                        } else {
                            if constexpr (EPI == M5_EPI_SILU_DT) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = silu_fast(v[r]);
                            }
                            *reinterpret_cast<uint2*>(dst + (j * 16 + lg * 4) * 2) = pack4<T>(v);
                        }
                    }
                }
                // read back 16-byte chunks: lane -> (row = pass * RPP + lane / CPR, chunk = lane % CPR)
                const int rr = lane / CPR, ch = lane - rr * CPR;
                const int nout = (EPI == M5_EPI_SWIGLU) ? p.N / 2 : p.N;
                const int ocol = ((EPI == M5_EPI_SWIGLU) ? (n0 + wn * TN * 16) / 2 : (n0 + wn * TN * 16)) + ch * 8;
                st* Cd = reinterpret_cast<st*>(p.C) + (int64_t)bz * p.sC;
                const int mrow0 = m0 + wm * TM * 16;
#pragma unroll
                for (int pass = 0; pass < (TM * 16 + RPP - 1) / RPP; ++pass) {
                    const int r = pass * RPP + rr;
                    if (rr < RPP && r < TM * 16 && mrow0 + r < p.M && ocol < nout) {
                        const uint4 val = *reinterpret_cast<const uint4*>(ws + r * RBS + ch * 16);
                        *reinterpret_cast<uint4*>(Cd + (int64_t)(mrow0 + r) * p.ldc + ocol) = val;
                    }
                }
                if (dbg_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[4] = clock64(); dbg[5] = wall_clock64(); }
                return;
            }
        }
    }

    // per-column-group constants (this lane's 4 consecutive columns of each of the TN tiles)
    float bv[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) bias_j(j, bv[j]);
    if constexpr (EPI == M5_EPI_F32 || EPI == M5_EPI_RESIDUAL) {
        static_assert(!FAST || EPI == M5_EPI_F32 || PRELOAD_C, "FAST residual epilogue needs the preloaded C tile");
        if ((FAST || p.fast_c) && (EPI == M5_EPI_F32 || PRELOAD_C)) {
            float* Cf = reinterpret_cast<float*>(Cb);
            if constexpr (DLN == 1) {
                // ---- deferred-LayerNorm producer.  Order: ONE wait for the preloaded C tile, all arithmetic in registers, the
                // centred copy + row sums out through LDS, and the fp32 C tile LAST as twelve back-to-back stores.  (Loads and
                // stores share the VM counter and retire out of order with respect to each other, so a store issued between two
                // uses of preloaded values makes hipcc wait for the earlier STORES too: the plain epilogue pays about one store
                // round trip for that, and anything behind it would queue up behind the whole tile.)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        asm volatile("" : "+v"(oldpre[i][j].x), "+v"(oldpre[i][j].y), "+v"(oldpre[i][j].z), "+v"(oldpre[i][j].w));
                float ps1[TM], ps2[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ps1[i] = 0.f; ps2[i] = 0.f;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float4 o;
                        o.x = oldpre[i][j].x + (acc[i][j][0] + bv[j][0]); o.y = oldpre[i][j].y + (acc[i][j][1] + bv[j][1]);
                        o.z = oldpre[i][j].z + (acc[i][j][2] + bv[j][2]); o.w = oldpre[i][j].w + (acc[i][j][3] + bv[j][3]);
                        oldpre[i][j] = o;                                         // the updated row values (stored at the end)
                        const float xc[4] = {o.x - dl_cen[i], o.y - dl_cen[i], o.z - dl_cen[i], o.w - dl_cen[i]};
                        ps1[i] += (xc[0] + xc[1]) + (xc[2] + xc[3]);
                        ps2[i] += (xc[0] * xc[0] + xc[1] * xc[1]) + (xc[2] * xc[2] + xc[3] * xc[3]);
                        acc[i][j] = f4_t{xc[0], xc[1], xc[2], xc[3]};             // the deferred LayerNorm's operand, before rounding
                    }
                }
                // the centred copy leaves through the (dead) stage buffers as whole 16-byte row chunks (straight from the
                // accumulator layout a store instruction writes 16 rows x 32 bytes: measured +4 us per launch)
                constexpr int RBX = TN * 32 + 16;                                 // padded LDS row: TN*16 columns of 2 bytes
                static_assert(NW * TM * 16 * RBX <= NSTAGE * STAGE, "the centred copy is staged in the K-loop stages");
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done reading the stages (LDS only: nothing to drain)
                unsigned char* wsx = lds + wave * (TM * 16 * RBX);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float xv[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        *reinterpret_cast<uint2*>(wsx + (i * 16 + l15) * RBX + (j * 16 + lg * 4) * 2) = pack4<T>(xv);
                    }
                // row sums over this wave's columns (the 4 lane groups of a row), then over the WN waves of the row through LDS,
                // in a fixed order; one {sum, sumsq} pair per (row, column tile) goes out
                float2* xs = reinterpret_cast<float2*>(dl_lds);                   // [NW][TM * 16]
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float s1 = ps1[i], s2 = ps2[i];
                    s1 += lane_xor16(s1); s1 += lane_xor32(s1);
                    s2 += lane_xor16(s2); s2 += lane_xor32(s2);
                    if (lg == 0) xs[wave * TM * 16 + i * 16 + l15] = make_float2(s1, s2);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                {
                    constexpr int CPRX = TN * 2, RPPX = 64 / CPRX;               // 16-byte chunks per row, rows per pass
                    const int rr = lane / CPRX, ch = lane - rr * CPRX;
                    st* Xt = reinterpret_cast<st*>(p.dl_xt) + (int64_t)bz * p.dl_rows_bs * p.dl_ld_xt;
                    const int ocol = nw + ch * 8;
                    uint4 xq[(TM * 16 + RPPX - 1) / RPPX];
#pragma unroll
                    for (int pass = 0; pass < (TM * 16 + RPPX - 1) / RPPX; ++pass)
                        xq[pass] = *reinterpret_cast<const uint4*>(wsx + min(pass * RPPX + rr, TM * 16 - 1) * RBX + ch * 16);
#pragma unroll
                    for (int pass = 0; pass < (TM * 16 + RPPX - 1) / RPPX; ++pass) {
                        const int r = pass * RPPX + rr;
                        if (rr < RPPX && r < TM * 16 && mw + r < p.M) *reinterpret_cast<uint4*>(Xt + (int64_t)(mw + r) * p.dl_ld_xt + ocol) = xq[pass];
                    }
                }
                for (int r = tid; r < BM; r += NW * 64) {
                    const int wmr = r / (TM * 16), lr = r - wmr * (TM * 16);
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int w = 0; w < WN; ++w) { const float2 v = xs[(wmr * WN + w) * TM * 16 + lr]; s1 += v.x; s2 += v.y; }
                    if (m0 + r < p.M) reinterpret_cast<float2*>(p.dl_part)[(int64_t)(dl_row0 + r) * p.dl_np + tn] = make_float2(s1, s2);
                }
                // the row's centre for the NEXT producer of these rows (column tile 0, one lane per row)
                if (tn == 0 && wn == 0 && lg == 0 && p.dl_cen_out) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = mw + i * 16 + l15;
                        if (row < p.M) p.dl_cen_out[bz * p.dl_rows_bs + row] = dl_cen[i];
                    }
                }
                // (round 6: the fp32 tile FIRST and the LDS staging of the centred copy under its drain measured 1 % SLOWER per
                // step, 2.89 -> 2.92 ms, profiles/r6b_dln_store_first_ab_negative.txt: the tile goes last)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = mw + i * 16 + l15;
                    float* rp = Cf + (int64_t)row * p.ldc;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (row < p.M) *reinterpret_cast<float4*>(rp + nw + j * 16 + lg * 4) = oldpre[i][j];
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = mw + i * 16 + l15;
                float* rp = Cf + (int64_t)row * p.ldc;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = nw + j * 16 + lg * 4;
                    float4 o;
                    o.x = acc[i][j][0] + bv[j][0]; o.y = acc[i][j][1] + bv[j][1]; o.z = acc[i][j][2] + bv[j][2]; o.w = acc[i][j][3] + bv[j][3];
                    if constexpr (EPI == M5_EPI_RESIDUAL) {
                        o.x = oldpre[i][j].x + o.x; o.y = oldpre[i][j].y + o.y; o.z = oldpre[i][j].z + o.z; o.w = oldpre[i][j].w + o.w;
                    }
                    if (row < p.M && col < p.N) *reinterpret_cast<float4*>(rp + col) = o;
                }
            }
            if (dbg_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[4] = clock64(); dbg[5] = wall_clock64(); }
            return;
        }
    }
    // in-place residual: row i+1's reads of C are issued BEFORE row i's writes (the compiler cannot
    // hoist a load above a possibly-aliasing store, which would serialise TM x TN round trips)
    auto load_old = [&](int i, float4* o) {
        const int row = mw + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = nw + j * 16 + lg * 4;
            o[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.M && col < p.N) {
                const float* cp = reinterpret_cast<const float*>(Cb) + (int64_t)row * p.ldc + col;
                if (p.vec_c && col + 3 < p.N) {
                    o[j] = *reinterpret_cast<const float4*>(cp);
                } else {
                    o[j].x = cp[0];
                    if (col + 1 < p.N) o[j].y = cp[1];
                    if (col + 2 < p.N) o[j].z = cp[2];
                    if (col + 3 < p.N) o[j].w = cp[3];
                }
            }
        }
    };
    float4 oldc[TN], oldn[TN];
    if constexpr (EPI == M5_EPI_RESIDUAL && !PRELOAD_C) load_old(0, oldc);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if constexpr (EPI == M5_EPI_RESIDUAL) {
            if constexpr (PRELOAD_C) {
#pragma unroll
                for (int j = 0; j < TN; ++j) oldc[j] = oldpre[i][j];
            } else {
                if (i + 1 < TM) load_old(i + 1, oldn);
            }
        }
        const int row = mw + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = nw + j * 16 + lg * 4;
            if (row >= p.M || col >= p.N) continue;
            const bool full = col + 3 < p.N;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[j][r];
            if constexpr (EPI == M5_EPI_F32 || EPI == M5_EPI_RESIDUAL) {
                float* cp = reinterpret_cast<float*>(Cb) + (int64_t)row * p.ldc + col;
                if constexpr (EPI == M5_EPI_RESIDUAL) {
                    v[0] = oldc[j].x + v[0]; v[1] = oldc[j].y + v[1];
                    v[2] = oldc[j].z + v[2]; v[3] = oldc[j].w + v[3];
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
                    for (int r = 0; r < 4; ++r) v[r] = silu_fast(v[r]);
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
                    const float sl = round_dt<T>(silu_fast(a));
                    o[h] = T::from_f32(sl * b);
                }
                st* cp = reinterpret_cast<st*>(Cb) + (int64_t)row * p.ldc + (col >> 1);
                if (full) {
                    *reinterpret_cast<uint32_t*>(cp) = *reinterpret_cast<const uint32_t*>(o);
                } else {
                    if (col + 1 < p.N) cp[0] = o[0];
                }
            } else if constexpr (EPI == M5_EPI_QKV) {
                const int kind = p.sec_kind[min(col / Dm, 2)];
                const int c = col % Dm, hh = c / p.sc.head_dim, dd = c % p.sc.head_dim;
                const int b = row / p.sc.rows_per_batch, s = row - b * p.sc.rows_per_batch;
                st* dst = (kind == 0)
                    ? reinterpret_cast<st*>(p.sc.q) + b * p.sc.q_bs + hh * p.sc.q_hs + (int64_t)s * p.sc.q_rs + dd
                    : reinterpret_cast<st*>(p.sc.k) + b * p.sc.k_bs + hh * p.sc.k_hs + (int64_t)s * p.sc.k_rs + dd;
                *reinterpret_cast<uint2*>(dst) = pack4<T>(v);   // Dm % 4 == 0: a 4-group never straddles N or a head
            }
        }
        if constexpr (EPI == M5_EPI_RESIDUAL && !PRELOAD_C) {
#pragma unroll
            for (int j = 0; j < TN; ++j) oldc[j] = oldn[j];
        }
    }
    if (dbg_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg[4] = clock64(); dbg[5] = wall_clock64(); }
}

// Is the FAST instantiation (vector paths only) of epilogue E compilable for this tiling?  (the staged epilogues need their
// output tile to fit the dead stage buffers; the residual one needs the preloaded C tile)
constexpr bool fast_ok(int E, int WM, int WN, int TM, int TN, int BKB, int NSTAGE) {
    const int NW = WM * WN, STAGE = (WM * TM * 16 + WN * TN * 16) * BKB;
    if (E == M5_EPI_F32) return true;
    if (E == M5_EPI_RESIDUAL) return TM * TN <= 16;
    if (E == M5_EPI_DT || E == M5_EPI_SILU_DT) return NW * TM * 16 * (TN * 32 + 16) <= NSTAGE * STAGE;
    if (E == M5_EPI_SWIGLU) return NW * TM * 16 * (TN * 16 + 16) <= NSTAGE * STAGE;
    if (E == M5_EPI_QKV) {
        const int RBQ = 128 + 16, RBV = TM * 32 + 16;
        const int WSZ = (TM * 16 * RBQ > 64 * RBV) ? TM * 16 * RBQ : 64 * RBV;
        return TN == 4 && NW * WSZ <= NSTAGE * STAGE;
    }
    return false;
}

// Deferred-LayerNorm instantiations (DLN = 1: residual producer; DLN = 2: QKV / SwiGLU consumers) exist for the tilings the
// engines' shapes use (DLNC: bit 0 = producer, bit 1 = consumers); anything else answers M5_ERR_UNSUPPORTED.
template <typename T, int WM, int WN, int TM, int TN, int BKB, int NSTAGE, int OCC, bool PF = false, int DLNC = 0, int ONLY = -1, bool LW = false>
int launch16(int epi, Gemm16Params& p, int batch, hipStream_t s, bool fast = false, int dln = 0) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    static_assert(!LW || ONLY == M5_EPI_RESIDUAL, "loader waves: instantiated for the residual epilogue only");
    if (LW && !fast) return M5_ERR_UNSUPPORTED;
    if (ONLY >= 0 && epi != ONLY) return M5_ERR_UNSUPPORTED;        // a tiling instantiated for one epilogue only (compile time)
    if (dln) {
        if (!fast) return M5_ERR_UNSUPPORTED;
        p.tilesM = (p.M + BM - 1) / BM; p.tilesN = (p.N + BN - 1) / BN;
        if (!rt_apply<BM>(p, batch)) return M5_ERR_UNSUPPORTED;
        const int64_t nb = (int64_t)p.tilesM * p.tilesN * (p.rt_map ? 1 : batch);
        if (nb > 0x7fffffff) return M5_ERR_UNSUPPORTED;
        p.nblk = (int)nb;
        p.group_m = max(1, GROUP_M * 128 / BM);
        const dim3 grid(p.nblk), blk(WM * WN * 64 * (LW ? 2 : 1));
        if constexpr ((DLNC & 1) != 0 && fast_ok(M5_EPI_RESIDUAL, WM, WN, TM, TN, BKB, NSTAGE)) {
            if (dln == 1 && epi == M5_EPI_RESIDUAL) {
                if (p.N % BN || p.dl_np != p.tilesN) return M5_ERR_UNSUPPORTED;       // one partial pair per (row, column tile)
                hipLaunchKernelGGL((gemm16_kernel<T, M5_EPI_RESIDUAL, WM, WN, TM, TN, BKB, NSTAGE, OCC, PF, true, 1, LW>), grid, blk, 0, s, p);
                M5_CHECK_LAUNCH();
                return M5_OK;
            }
        }
        if constexpr ((DLNC & 2) != 0) {
            if constexpr (fast_ok(M5_EPI_QKV, WM, WN, TM, TN, BKB, NSTAGE)) {
                if (dln == 2 && epi == M5_EPI_QKV) {
                    hipLaunchKernelGGL((gemm16_kernel<T, M5_EPI_QKV, WM, WN, TM, TN, BKB, NSTAGE, OCC, PF, true, 2>), grid, blk, 0, s, p);
                    M5_CHECK_LAUNCH();
                    return M5_OK;
                }
            }
            if constexpr (fast_ok(M5_EPI_SWIGLU, WM, WN, TM, TN, BKB, NSTAGE)) {
                if (dln == 2 && epi == M5_EPI_SWIGLU) {
                    hipLaunchKernelGGL((gemm16_kernel<T, M5_EPI_SWIGLU, WM, WN, TM, TN, BKB, NSTAGE, OCC, PF, true, 2>), grid, blk, 0, s, p);
                    M5_CHECK_LAUNCH();
                    return M5_OK;
                }
            }
        }
        return M5_ERR_UNSUPPORTED;
    }
    p.tilesM = (p.M + BM - 1) / BM; p.tilesN = (p.N + BN - 1) / BN;
    if (!rt_apply<BM>(p, batch)) return M5_ERR_UNSUPPORTED;
    const int64_t nblk = (int64_t)p.tilesM * p.tilesN * (p.rt_map ? 1 : batch);
    if (nblk > 0x7fffffff) return M5_ERR_UNSUPPORTED;
    p.nblk = (int)nblk;
    p.group_m = max(1, GROUP_M * 128 / BM);
    const dim3 grid(p.nblk), blk(WM * WN * 64 * (LW ? 2 : 1));
    if constexpr (LW) {
        if constexpr (fast_ok(M5_EPI_RESIDUAL, WM, WN, TM, TN, BKB, NSTAGE)) {
            hipLaunchKernelGGL((gemm16_kernel<T, M5_EPI_RESIDUAL, WM, WN, TM, TN, BKB, NSTAGE, OCC, PF, true, 0, true>), grid, blk, 0, s, p);
            M5_CHECK_LAUNCH();
            return M5_OK;
        }
        return M5_ERR_UNSUPPORTED;
    } else {
#define M5_G16(E)                                                                                                          \
    do {                                                                                                                   \
        if constexpr (fast_ok(E, WM, WN, TM, TN, BKB, NSTAGE)) {                                                           \
            if (fast) { hipLaunchKernelGGL((gemm16_kernel<T, E, WM, WN, TM, TN, BKB, NSTAGE, OCC, PF, true>), grid, blk, 0, s, p); break; } \
        }                                                                                                                  \
        hipLaunchKernelGGL((gemm16_kernel<T, E, WM, WN, TM, TN, BKB, NSTAGE, OCC, PF, false>), grid, blk, 0, s, p);        \
    } while (0)
    if constexpr (ONLY == M5_EPI_RESIDUAL) {
        M5_G16(M5_EPI_RESIDUAL);
    } else {
    switch (epi) {
        case M5_EPI_F32: M5_G16(M5_EPI_F32); break;
        case M5_EPI_DT: M5_G16(M5_EPI_DT); break;
        case M5_EPI_RESIDUAL: M5_G16(M5_EPI_RESIDUAL); break;
        case M5_EPI_SWIGLU: M5_G16(M5_EPI_SWIGLU); break;
        case M5_EPI_QKV: M5_G16(M5_EPI_QKV); break;
        case M5_EPI_SILU_DT: M5_G16(M5_EPI_SILU_DT); break;
        default: return M5_ERR_ARG;
    }
    }
#undef M5_G16
    M5_CHECK_LAUNCH();
    return M5_OK;
    }
}

// Tile configurations.  M5_GEMM_CFG=<n> forces one (tuning sweeps); otherwise pick_config().
struct CfgInfo { int bm, bn, tn, occ; float t_fix_us, t_iter_us; int only_epi; };   // t_iter: per 64-deep K-step, measured (profiles/)
static const CfgInfo kCfg[] = {
    {128, 128, 4, 2, 8.f, 0.72f, -1},             // 0: 128x128 tile, 4 waves, 2 stages, 2 WG/CU
    {192, 384, 6, 1, 8.f, 2.20f, -1},             // 1: region 192x384, 8 waves (2x4 of 96x96), 2 stages
    {192, 192, 4, 1, 7.f, 1.20f, -1},             // 2: region 192x192, 12 waves (4x3 of 48x64), 3 stages
    { 96, 128, 4, 1, 8.f, 0.64f, -2},             // 3: region  96x128, 4 waves (2x2 of 48x64), 4 stages (sweeps: superseded by 7)
    { 96, 128, 2, 2, 8.f, 0.66f, -1},             // 4: tile    96x128, 8 waves (2x4 of 48x32), 2 stages, 2 WG/CU
    {192, 384, 6, 1, 8.f, 2.10f, M5_EPI_SWIGLU},  // 5: region 192x384, 16 waves (4x4 of 48x96), 2 stages (the QKV / residual
                                                  //    epilogues spill at 128 VGPRs: SwiGLU only)
    {192, 192, 4, 1, 7.f, 1.40f, -2},             // 6: region 192x192, 6 waves (2x3 of 96x64), 3 stages (sweeps: superseded by 2)
    { 96, 128, 4, 1, 8.f, 0.48f, -1},             // 7: = 3 with the prefetched-fragment K loop (PF); 0.45-0.48 us per K-step measured hot
    { 96, 128, 4, 1, 8.f, 0.48f, -2},             // 8: = 7 with 5 stages (140 KB of LDS: five K-steps of DMA in flight)
    {192, 384, 6, 1, 8.f, 2.10f, -2},             // 9 (tools build, round-4 probe): = 5 with 32-deep K-steps and 4 stages (same 144 KB: three
                                                  //    half-steps of DMA in flight instead of one whole step)
    {192, 192, 4, 1, 7.f, 1.20f, -2},             // 10 (tools build): = 2 with 32-deep K-steps and 6 stages
    { 96, 128, 2, 1, 8.f, 0.48f, -2},             // 11 (tools build, round-6 probe): region 96x128 with EIGHT waves (2x4 of 48x32), 4 stages, one workgroup
                                                  //    per CU -- two waves per SIMD for the residual class.  Measured SLOWER than cfg 7 on every shape
                                                  //    (profiles/r6a_gemm_8wave_residual_region_negative.txt): 16.8 -> 18.4, 14.6 -> 16.0, 31.3 -> 37.1 us
    { 96, 128, 4, 1, 8.f, 0.40f, -2},             // 12 (round 6): = 7 with LOADER WAVES (gemm16_kernel LW): 4 computing + 4 loader waves, residual epilogue only
};
constexpr int kNumCfg = sizeof(kCfg) / sizeof(kCfg[0]);

template <typename T>
int launch_cfg(int cfg, int epi, Gemm16Params& p, int batch, hipStream_t s, bool fast, int dln = 0) {
    switch (cfg) {
        case 0: return launch16<T, 2, 2, 4, 4, 128, 2, 2, false, 3>(epi, p, batch, s, fast, dln);
        case 1: return launch16<T, 2, 4, 6, 6, 128, 2, 1, false, 2>(epi, p, batch, s, fast, dln);
        case 2: return launch16<T, 4, 3, 3, 4, 128, 3, 1, false, 2>(epi, p, batch, s, fast, dln);
        case 3: return launch16<T, 2, 2, 3, 4, 128, 4, 1>(epi, p, batch, s, fast, dln);
        case 4: return launch16<T, 2, 4, 3, 2, 128, 2, 2>(epi, p, batch, s, fast, dln);
        case 5: return launch16<T, 4, 4, 3, 6, 128, 2, 1, false, 2>(epi, p, batch, s, fast, dln);
        case 6: return launch16<T, 2, 3, 6, 4, 128, 3, 1>(epi, p, batch, s, fast, dln);
        case 7: return launch16<T, 2, 2, 3, 4, 128, 4, 1, true, 1>(epi, p, batch, s, fast, dln);
        case 8: return launch16<T, 2, 2, 3, 4, 128, 5, 1, true>(epi, p, batch, s, fast, dln);
        case 12: return launch16<T, 2, 2, 3, 4, 128, 4, 1, true, 1, M5_EPI_RESIDUAL, true>(epi, p, batch, s, fast, dln);
#ifdef M5_TOOLS
        case 11: return launch16<T, 2, 4, 3, 2, 128, 4, 1, false, 1, M5_EPI_RESIDUAL>(epi, p, batch, s, fast, dln);
        case 9: return launch16<T, 4, 4, 3, 6, 64, 4, 1>(epi, p, batch, s, fast, dln);
        case 10: return launch16<T, 4, 3, 3, 4, 64, 6, 1>(epi, p, batch, s, fast, dln);
#endif
        default: return M5_ERR_ARG;
    }
}

// Cheapest configuration under: time = rounds x (fixed + K-steps x per-step time), rounds =
// ceil(workgroups / (256 CUs x workgroups per CU)).  `span_div`: for QKV scatters with a V section
// the per-wave column span must divide the section width.
int pick_config(int M, int N, int K, int batch, int span_div, int epi, int dln = 0) {
    int best = 0;
    float best_t = 1e30f;
    for (int c = 0; c < kNumCfg; ++c) {
        const CfgInfo& f = kCfg[c];
        if (f.only_epi == -2 || (f.only_epi >= 0 && f.only_epi != epi)) continue;
        if (dln == 1 && c != 0 && c != 7) continue;                // deferred-LayerNorm producer: 128-column tiles with the preloaded C tile
        if (dln == 2 && c != 0 && c != 1 && c != 2 && c != 5) continue;
        if (span_div && (span_div % (f.tn * 16))) continue;
        if (epi == M5_EPI_QKV && f.tn != 4) continue;       // one head per wave: only those tilings have the LDS-staged 16-byte scatter
        const int64_t wg = (int64_t)((M + f.bm - 1) / f.bm) * ((N + f.bn - 1) / f.bn) * batch;
        const int64_t slots = 256 * f.occ;
        const int64_t rounds = (wg + slots - 1) / slots;
        // a partially filled last round of an occ-2 configuration runs its workgroups alone on their CUs
        float t = (float)rounds * (f.t_fix_us + (float)(K / 64) * f.t_iter_us);
        // many rounds (batched NAR groups, M >= 10k rows): the tail round no longer matters and most of a workgroup's
        // fixed cost hides under its successor / co-resident workgroup (measured at M = 35840, profiles/r2t_gemm_configs.txt)
        if (wg >= 4 * slots) t = (float)wg / (float)slots * ((f.occ == 1 ? 0.2f : 0.7f) * f.t_fix_us + (float)(K / 64) * f.t_iter_us);
        if (epi == M5_EPI_QKV && (c == 3 || c == 7)) t += 2.5f * (float)rounds;     // measured: its 4-wave scatter epilogue is the slowest
        if (epi == M5_EPI_QKV && c == 0 && wg >= 4 * slots) t *= 1.12f;             // measured at M = 35,840 / 98,304 (profiles/r4q_large_m_gemm_config_sweep.txt):
                                                                                    // 128x128 is 4-5 % behind the 192x192 region for the scatter epilogue
        if (t < best_t) { best_t = t; best = c; }
    }
    return best;
}

unsigned long long* g_gemm_dbg = nullptr;

int num_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
}

}  // namespace

// deferred-LayerNorm arguments -> kernel parameters (shared by m5_gemm16_dispatch and m5_xattn_scores); M5_OK or an error
static int dl_fill(Gemm16Params& p, const M5DeferredLN* dl, int epi_kind /* 1 producer, 2 consumer */, int N) {
    if (dl->mode != epi_kind) return M5_ERR_ARG;
    if (!dl->part || dl->np <= 0 || dl->np > DL_MAX_NP || (dl->np & 1) || dl->rows_bs <= 0) return M5_ERR_UNSUPPORTED;
    if (((uintptr_t)dl->part & 15)) return M5_ERR_ARG;
    p.dl_part = dl->part; p.dl_np = dl->np; p.dl_cen_in = dl->cen_in; p.dl_cen_out = dl->cen_out; p.dl_delta = dl->delta; p.dl_rows_bs = dl->rows_bs;
    if (epi_kind == 1) {
        if (!dl->xt || (dl->ld_xt % 8) || ((uintptr_t)dl->xt & 15)) return M5_ERR_ARG;      // 16-byte row chunks
        p.dl_xt = (unsigned char*)dl->xt; p.dl_ld_xt = dl->ld_xt;
    } else {
        if (!dl->s || ((uintptr_t)dl->s & 15) || (dl->s_bs % 4) || dl->n_feat <= 0 || !(dl->eps > 0.f) || (N % 4) || N < 4) return M5_ERR_ARG;
        if (!p.bias || ((uintptr_t)p.bias & 15) || (p.sBias % 4)) return M5_ERR_ARG;           // b' = b + W beta always exists; it travels by LDS-DMA
        p.dl_s = dl->s; p.dl_s_bs = dl->s_bs; p.dl_eps = dl->eps; p.dl_inv_n = 1.0f / (float)dl->n_feat;
    }
    return M5_OK;
}

#ifdef M5_TOOLS    // tools library only: measured slower inside the NAR step; kept as an A/B instrument
// Q projection with the cross-attention it feeds fused into its epilogue (include/mars5_hip.h).  Eligible shapes only
// (M5_ERR_UNSUPPORTED otherwise; the caller then runs m5_gemm(EPI_QKV) + m5_attention): 16-bit operands, head_dim 64,
// memory length <= 64, rows per sequence a multiple of 16.
extern "C" int m5_gemm_q_cross_attn(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                    int M, int n_heads, int K, const int64_t* mem_table, int max_le, int rows_per_seq,
                                    const int32_t* step, float scale, void* out, int64_t ld_out, void* stream) {
    if (!A || !W || !mem_table || !step || !out || M <= 0 || n_heads <= 0 || K <= 0 || rows_per_seq <= 0) return M5_ERR_ARG;
    if (dtype != M5_F16 && dtype != M5_BF16) return M5_ERR_UNSUPPORTED;
    if (max_le <= 0 || max_le > 16 * XA_MAX_KT || (rows_per_seq % 16) || (K % 64) || (n_heads % 2)) return M5_ERR_UNSUPPORTED;
    if ((lda % 8) || (ldw % 8) || (ld_out % 4) || (((uintptr_t)A | (uintptr_t)W) & 15) || ((uintptr_t)out & 7)) return M5_ERR_UNSUPPORTED;
    // The host (mars5_tts_amd/ops.py) only calls this when M5_GEMM_XATTN=1: it is a tested opt-in.  Measured (tools/xattn_clock.py, tools/nar_step_bench.py): back to back on L2-hot data
    // the fused launch takes 20.3 us against 24.8 us for projection + attention launches, but inside the NAR step (memory
    // block of the step cold in HBM, weights streaming) the step gets 65 us SLOWER (3.76 vs 3.69 ms): the cold K / V^T
    // fragments sit in front of the operand DMAs in the in-order load queue, and the region-96x128 tiling this epilogue
    // needs (one head per wave) is slower for this GEMM than the 8-wave tiling the plain projection uses.
    constexpr int BM = 96, BN = 128;
    Gemm16Params p{};
    p.A = (const unsigned char*)A; p.W = (const unsigned char*)W; p.bias = bias;
    p.lda = lda; p.ldw = ldw; p.M = M; p.N = n_heads * 64; p.K = K;
    p.sc.n_heads = 1; p.sc.head_dim = 1; p.sc.rows_per_batch = 1;
    p.dbg = g_gemm_dbg;
    p.xa_tab = mem_table; p.xa_step = step; p.xa_rows_per_seq = rows_per_seq; p.xa_out = (unsigned char*)out; p.xa_ld_out = ld_out;
    p.xa_scale = scale;
    p.tilesM = (M + BM - 1) / BM; p.tilesN = p.N / BN;
    const int64_t nblk = (int64_t)p.tilesM * p.tilesN;
    if (nblk > 0x7fffffff) return M5_ERR_UNSUPPORTED;
    p.nblk = (int)nblk; p.group_m = max(1, GROUP_M * 128 / BM);
    const dim3 grid(p.nblk), blk(256);
    if (dtype == M5_F16) hipLaunchKernelGGL((gemm16_kernel<F16T, EPI_Q_CROSS, 2, 2, 3, 4, 128, 4, 1>), grid, blk, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm16_kernel<BF16T, EPI_Q_CROSS, 2, 2, 3, 4, 128, 4, 1>), grid, blk, 0, (hipStream_t)stream, p);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

#endif  // M5_TOOLS

// Scores + per-head softmax of the absorbed cross-attention (include/mars5_hip.h, xattn_absorb.hip): P[b] = softmax_heads(
// X[b] A[b]^T + c[b]) for `batch` sequences, head blocks of Lp = 48 or 64 columns (one per wave), 96-row tiles.
static int xattn_scores_impl(int dtype, const void* X, int64_t ldx, int64_t sX, const void* A, int64_t sA_tab, const float* c, int64_t sc_tab,
                             void* P, int64_t ldp, int64_t sP, int M, int n_heads, int Lp, int K, int batch, const M5DeferredLN* dl,
                             const M5RowTiles* rt, void* stream) {
    if (!X || !A || !c || !P || M <= 0 || n_heads <= 0 || K <= 0 || batch <= 0) return M5_ERR_ARG;
    if (dtype != M5_F16 && dtype != M5_BF16) return M5_ERR_UNSUPPORTED;
    if ((Lp != 48 && Lp != 64) || (n_heads % 2) || (K % 64)) return M5_ERR_UNSUPPORTED;
    if ((ldx % 8) || (sX % 8) || (sA_tab % 8) || (ldp % 8) || (sP % 8) || (((uintptr_t)X | (uintptr_t)A | (uintptr_t)P) & 15)) return M5_ERR_ARG;
    Gemm16Params p{};
    p.A = (const unsigned char*)X; p.W = (const unsigned char*)A; p.bias = c; p.C = (unsigned char*)P;
    p.lda = ldx; p.ldw = K; p.ldc = ldp; p.sA = sX; p.sW = sA_tab; p.sC = sP; p.sBias = sc_tab;
    p.M = M; p.N = n_heads * Lp; p.K = K;
    p.sc.n_heads = 1; p.sc.head_dim = 1; p.sc.rows_per_batch = 1;
    p.vec16 = 1; p.vec_c = 1;
    if (dl) {
        const int rc = dl_fill(p, dl, 2, n_heads * Lp);
        if (rc != M5_OK) return rc;
    }
    const int BN = 2 * Lp;
    // tile shape: 96 x 2Lp with 4 waves and 4 stages (one workgroup per CU), or -- tools build, M5_XATTN_CFG=1 / 2 -- 64 x 2Lp /
    // 128 x 2Lp with 8 waves, 2 stages and two workgroups per CU
    int cfg = 0;
    if (const char* ce = m5_tool_env("M5_XATTN_CFG")) cfg = atoi(ce);
    if (const char* pf = m5_tool_env("M5_GEMM_PF")) { if (atoi(pf) == 0 && cfg == 0) cfg = 3; }         // same-process A/B (tools build)
    if (ldx >= (1ll << 22) || K >= (1 << 22)) return M5_ERR_UNSUPPORTED;          // tile-local DMA offsets are 32-bit
    const bool off32 = true;
    const int BM = cfg == 1 ? 64 : (cfg == 2 ? 128 : 96);
    p.tilesM = (M + BM - 1) / BM; p.tilesN = p.N / BN;
    p.rt_host = rt;
    if (rt && (BM != 96 || !rt_apply<96>(p, batch))) return M5_ERR_UNSUPPORTED;
    const int64_t nblk = (int64_t)p.tilesM * p.tilesN * (p.rt_map ? 1 : batch);
    if (nblk > 0x7fffffff) return M5_ERR_UNSUPPORTED;
    p.nblk = (int)nblk; p.group_m = max(1, GROUP_M * 128 / BM);
    const dim3 grid(p.nblk);
    hipStream_t s = (hipStream_t)stream;
    const bool xs_fast = (((uintptr_t)c & 15) == 0) && (sc_tab % 4 == 0) && m5_tool_env("M5_GEMM_FAST") == nullptr;
    if (dl) {
        if (!xs_fast || cfg != 0) return M5_ERR_UNSUPPORTED;
#define M5_XSD(TT, TNv) hipLaunchKernelGGL((gemm16_kernel<TT, EPI_SOFTMAX_HEADS, 2, 2, 3, TNv, 128, 4, 1, true, true, 2>), grid, dim3(256), 0, s, p)
        if (dtype == M5_F16) { if (Lp == 48) M5_XSD(F16T, 3); else M5_XSD(F16T, 4); }
        else { if (Lp == 48) M5_XSD(BF16T, 3); else M5_XSD(BF16T, 4); }
#undef M5_XSD
        M5_CHECK_LAUNCH();
        return M5_OK;
    }
#define M5_XS(TT, WMv, TMv, TNv, NSv, OCv, PFv) do { if (xs_fast) hipLaunchKernelGGL((gemm16_kernel<TT, EPI_SOFTMAX_HEADS, WMv, 2, TMv, TNv, 128, NSv, OCv, PFv, true>), grid, dim3(WMv * 128), 0, s, p); \
        else hipLaunchKernelGGL((gemm16_kernel<TT, EPI_SOFTMAX_HEADS, WMv, 2, TMv, TNv, 128, NSv, OCv, PFv, false>), grid, dim3(WMv * 128), 0, s, p); } while (0)
#ifdef M5_TOOLS
#define M5_XS_CFG(TT, TNv) do { if (cfg == 1) M5_XS(TT, 4, 1, TNv, 2, 2, false); else if (cfg == 2) M5_XS(TT, 4, 2, TNv, 2, 2, false); else if (cfg == 3 || !off32) M5_XS(TT, 2, 3, TNv, 4, 1, false); else M5_XS(TT, 2, 3, TNv, 4, 1, true); } while (0)
#else
#define M5_XS_CFG(TT, TNv) do { if (off32) M5_XS(TT, 2, 3, TNv, 4, 1, true); else M5_XS(TT, 2, 3, TNv, 4, 1, false); } while (0)
#endif
    if (dtype == M5_F16) { if (Lp == 48) M5_XS_CFG(F16T, 3); else M5_XS_CFG(F16T, 4); }
    else { if (Lp == 48) M5_XS_CFG(BF16T, 3); else M5_XS_CFG(BF16T, 4); }
#undef M5_XS_CFG
#undef M5_XS
    M5_CHECK_LAUNCH();
    return M5_OK;
}

extern "C" int m5_xattn_scores(int dtype, const void* X, int64_t ldx, int64_t sX, const void* A, int64_t sA_tab, const float* c, int64_t sc_tab,
                               void* P, int64_t ldp, int64_t sP, int M, int n_heads, int Lp, int K, int batch, void* stream) {
    return xattn_scores_impl(dtype, X, ldx, sX, A, sA_tab, c, sc_tab, P, ldp, sP, M, n_heads, Lp, K, batch, nullptr, nullptr, stream);
}

// ... optionally with the LayerNorm that produced X deferred into the epilogue (M5DeferredLN, mode 2: X is the centred copy, A
// the operands built from the gamma-folded query weights) and / or over a row-tile list (M5RowTiles).
extern "C" int m5_xattn_scores_ex(int dtype, const void* X, int64_t ldx, int64_t sX, const void* A, int64_t sA_tab, const float* c, int64_t sc_tab,
                                  void* P, int64_t ldp, int64_t sP, int M, int n_heads, int Lp, int K, int batch, const M5DeferredLN* dl,
                                  const M5RowTiles* rt, void* stream) {
    return xattn_scores_impl(dtype, X, ldx, sX, A, sA_tab, c, sc_tab, P, ldp, sP, M, n_heads, Lp, K, batch, dl, rt, stream);
}

#ifdef M5_TOOLS    // tools library only: measured slower than GEMM + LayerNorm launches
// Residual GEMM with the following LayerNorm fused into its epilogue (include/mars5_hip.h).  Eligible shapes only
// (M5_ERR_UNSUPPORTED otherwise; the caller then runs m5_gemm + m5_layernorm): 16-bit operands, batch 1, region 96x128
// tiling with the whole grid co-resident (one workgroup per CU), N a multiple of 128 and <= 2048.
extern "C" int m5_gemm_residual_ln(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                   float* C, int64_t ldc, int M, int N, int K, const float* ln_gamma, const float* ln_beta,
                                   float ln_eps, void* xn, int64_t ld_xn, void* scratch, int64_t scratch_bytes,
                                   const int32_t* tag_step, int tag, void* stream) {
    if (!A || !W || !C || !ln_gamma || !ln_beta || !xn || !scratch || M <= 0 || N <= 0 || K <= 0) return M5_ERR_ARG;
    if (dtype != M5_F16 && dtype != M5_BF16) return M5_ERR_UNSUPPORTED;
    constexpr int BM = 96, BN = 128;
    const int tilesM = (M + BM - 1) / BM, tilesN = N / BN;
    if ((N % BN) || tilesN > 16 || (K % 64) || (int64_t)tilesM * tilesN > num_cus()) return M5_ERR_UNSUPPORTED;
    if ((lda % 8) || (ldw % 8) || (ldc % 4) || (ld_xn % 4) || (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) ||
        ((uintptr_t)xn & 7)) return M5_ERR_UNSUPPORTED;
    const int64_t need = 256 + (int64_t)tilesM * tilesN * BM * 8;
    if (scratch_bytes < need || ((uintptr_t)scratch & 15)) return M5_ERR_ARG;
    // The host (mars5_tts_amd/ops.py) only calls this when M5_GEMM_LN=1 (a tested opt-in): measured SLOWER than GEMM + LayerNorm launches on MI355X (3.60 vs 3.50 ms per NAR step,
    // tools/nar_step_bench.py): the exchange between the 8 workgroups of a row tile (write-through store, L2-bypassing
    // polls, plus their start skew) costs ~8 us per launch against the 6 us LayerNorm launch it removes.  Kept as a
    // tested opt-in and as the record of that measurement.
    Gemm16Params p{};
    p.A = (const unsigned char*)A; p.W = (const unsigned char*)W; p.bias = bias; p.C = (unsigned char*)C;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    p.sc.n_heads = 1; p.sc.head_dim = 1; p.sc.rows_per_batch = 1;
    p.vec_c = 1;
    p.dbg = g_gemm_dbg;
    p.ln_g = ln_gamma; p.ln_b = ln_beta; p.ln_eps = ln_eps; p.xn = (unsigned char*)xn; p.ld_xn = ld_xn;
    p.ln_err = (unsigned int*)scratch;                                  // [0]: timeouts; exchange words from byte 256
    p.ln_part = (float*)((unsigned char*)scratch + 256);
    p.ln_tag_step = tag_step; p.ln_tag = tag;
    p.tilesM = tilesM; p.tilesN = tilesN; p.nblk = tilesM * tilesN; p.group_m = max(1, GROUP_M * 128 / BM);
    const dim3 grid(p.nblk), blk(256);
    if (dtype == M5_F16) hipLaunchKernelGGL((gemm16_kernel<F16T, EPI_RESIDUAL_LN, 2, 2, 3, 4, 128, 4, 1>), grid, blk, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm16_kernel<BF16T, EPI_RESIDUAL_LN, 2, 2, 3, 4, 128, 4, 1>), grid, blk, 0, (hipStream_t)stream, p);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

extern "C" int m5_debug_gemm_clock(unsigned long long* buf) {   // diagnostics (tools/gemm_clock.py); nullptr disables
    g_gemm_dbg = buf;
    return M5_OK;
}
#endif

// Called by m5_gemm (gemm.hip) for F16 / BF16 operands after argument validation.
int m5_gemm16_dispatch(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                       void* C, int64_t ldc, int M, int N, int K, int epi, const M5QkvScatter* sc, const int* sec_kind,
                       int batch, int64_t sA, int64_t sW, int64_t sC, int64_t sBias, hipStream_t s, const M5DeferredLN* dl,
                       const M5RowTiles* rt) {
    Gemm16Params p{};
    p.rt_host = rt;
    p.A = (const unsigned char*)A; p.W = (const unsigned char*)W; p.bias = bias; p.C = (unsigned char*)C;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.sA = sA; p.sW = sW; p.sC = sC; p.sBias = sBias;
    int dln = 0;
    if (dl) {
        dln = (epi == M5_EPI_RESIDUAL) ? 1 : 2;
        if (epi != M5_EPI_RESIDUAL && epi != M5_EPI_QKV && epi != M5_EPI_SWIGLU) return M5_ERR_UNSUPPORTED;
        const int rc = dl_fill(p, dl, dln, N);
        if (rc != M5_OK) return rc;
    }
    p.M = M; p.N = N; p.K = K;
    p.dbg = g_gemm_dbg;
    const bool f32out = (epi == M5_EPI_F32 || epi == M5_EPI_RESIDUAL);
    const int cal = f32out ? 15 : 7;
    {
        const int nout = (epi == M5_EPI_SWIGLU) ? N / 2 : N;
        p.vec16 = (!f32out && C && (ldc % 8 == 0) && (sC % 8 == 0) && (((uintptr_t)C & 15) == 0) && (nout % 8 == 0)) ? 1 : 0;
    }
    p.vec_c = (C && (ldc % 4 == 0) && (sC % 4 == 0) && (((uintptr_t)C & cal) == 0)) ? 1 : 0;
    p.fast_c = (f32out && p.vec_c && (N % 4 == 0) && N >= 4 && (!bias || ((((uintptr_t)bias & 15) == 0) && (sBias % 4 == 0)))) ? 1 : 0;
    if (sc) {
        p.sc = *sc;
        for (int i = 0; i < 3; ++i) p.sec_kind[i] = sec_kind[i];
        const bool one_batch = M <= sc->rows_per_batch;
        p.vt_vec = (sc->vt && (sc->vt_ds % 4 == 0) && (sc->vt_hs % 4 == 0) && (sc->vt_bs % 4 == 0) &&
                    (((uintptr_t)sc->vt & 7) == 0) && (one_batch || sc->rows_per_batch % 4 == 0)) ? 1 : 0;
        {
            auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
            const bool ok_q = !sc->q || (al16(sc->q) && sc->q_rs % 8 == 0 && sc->q_hs % 8 == 0 && sc->q_bs % 8 == 0 && sc->q_rs >= 64);
            const bool ok_k = !sc->k || (al16(sc->k) && sc->k_rs % 8 == 0 && sc->k_hs % 8 == 0 && sc->k_bs % 8 == 0 && sc->k_rs >= 64);
            const bool ok_v = !sc->vt || (al16(sc->vt) && sc->vt_ds % 8 == 0 && sc->vt_hs % 8 == 0 && sc->vt_bs % 8 == 0);
            p.qkv_stage = (sc->head_dim == 64 && ok_q && ok_k && ok_v && (M % 8 == 0) &&
                           (one_batch || sc->rows_per_batch % 8 == 0)) ? 1 : 0;
        }
        // Q / K scatter stores 4 consecutive d as 8 bytes
        if (sc->head_dim % 4 || (sc->q && ((sc->q_rs % 4) || (sc->q_hs % 4) || (sc->q_bs % 4) || ((uintptr_t)sc->q & 7))) ||
            (sc->k && ((sc->k_rs % 4) || (sc->k_hs % 4) || (sc->k_bs % 4) || ((uintptr_t)sc->k & 7))))
            return M5_ERR_UNSUPPORTED;
    } else {
        p.sc.n_heads = 1; p.sc.head_dim = 1; p.sc.rows_per_batch = 1;
    }
    if (const char* ab = m5_tool_env("M5_GEMM_ABL")) p.abl = atoi(ab);
    if (const char* fc = m5_tool_env("M5_GEMM_FASTC")) { if (atoi(fc) == 0) p.fast_c = 0; }
    const char* fe = m5_tool_env("M5_GEMM_CFG");             // tuning sweeps (tools build only); read per call on purpose
    int forced = (fe && fe[0]) ? atoi(fe) : -1;
    if (forced < 0) {                                        // per-epilogue override: M5_GEMM_CFG_E<epi>=<n>
        char nm[24];
        snprintf(nm, sizeof nm, "M5_GEMM_CFG_E%d", epi);
        const char* fe2 = m5_tool_env(nm);
        if (fe2 && fe2[0]) forced = atoi(fe2);
    }
    const int span_div = (sc && sc->vt) ? sc->n_heads * sc->head_dim : 0;
    int cfg = forced >= 0 ? forced : pick_config(M, N, K, batch, span_div, epi, dln);
    if (cfg < 0 || cfg >= kNumCfg || (span_div && (span_div % (kCfg[cfg].tn * 16)))) cfg = 0;
    if (lda >= (1ll << 22) || ldw >= (1ll << 22)) return M5_ERR_UNSUPPORTED;       // tile-local DMA offsets (rows x leading dimension) are 32-bit
    const bool off32 = true;
    if (const char* pf = m5_tool_env("M5_GEMM_PF")) { if (atoi(pf) == 0 && cfg == 7) cfg = 3; }      // same-process A/B (tools build)
    if (cfg == 7 && !off32) cfg = 3;
    // (PF on the multi-wave configurations 2 and 0 measured SLOWER -- their co-resident waves already cover the LDS latency:
    // QKV 26.7 -> 37.0 us, 34.8 -> 38.6 us; profiles/r3b_gemm_pf_ab.txt -- and was removed)
    // FAST instantiation: every vector path of this epilogue applies (the engines' shapes always qualify)
    const bool bias_ok = (N % 4 == 0) && N >= 4 && (!bias || ((((uintptr_t)bias & 15) == 0) && (sBias % 4 == 0)));
    bool fast = bias_ok && (K % 64 == 0);
    if (epi == M5_EPI_F32 || epi == M5_EPI_RESIDUAL) fast = fast && p.fast_c;
    else if (epi == M5_EPI_QKV) fast = fast && p.qkv_stage;
    else fast = fast && p.vec16;
    if (const char* fk = m5_tool_env("M5_GEMM_FAST")) { if (atoi(fk) == 0) fast = false; }      // same-process A/B (tools build)
    if (dln && !fast) return M5_ERR_UNSUPPORTED;
    // the residual class on its 96x128 region: loader waves (cfg 12 = cfg 7 + 4 DMA-only waves; same MFMA order, same bits) whenever
    // the FAST instantiation applies -- linear2 31.3 -> 28.8 us, out-proj 16.6 -> 16.2, M = 35,840 linear2 341 -> 330 (profiles/r6u_*)
    if (cfg == 7 && forced < 0 && fast && epi == M5_EPI_RESIDUAL && dln != 2) cfg = 12;
    if (dtype == M5_F16) return launch_cfg<F16T>(cfg, epi, p, batch, s, fast, dln);
    return launch_cfg<BF16T>(cfg, epi, p, batch, s, fast, dln);
}
