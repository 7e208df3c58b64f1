// Cross-attention against a SHORT memory with the projections absorbed into the memory (NAR decoder, reference
// mars5/model.py:179-203: nn.MultiheadAttention(tgt, memory) of every decoder layer; the text memory has Le ~ 20-60 rows
// while the target has ~1.4k).
//
// Per head h:  S_h = (x Wq_h^T + bq_h) K_h^T / 8 = x (K_h Wq_h / 8)^T + (K_h bq_h / 8)      =: x A_h^T + c_h
//              y   = sum_h softmax(S_h) V_h Wo_h^T + bo = sum_h P_h (V_h Wo_h^T) + bo       =: P B + bo
// with Wq_h = rows h*64..h*64+63 of the query projection and Wo_h = columns h*64.. of the output projection.  A (H*Lp x D)
// and B^T (D x H*Lp) depend on the memory only, so they are built once per reverse step for all layers by ONE launch of
// this kernel; the decoder layer then runs two GEMMs (x A^T with a per-head softmax epilogue, P B with the residual
// epilogue) of N = K = H*Lp = 768 (Lp = 48) instead of q-projection (N = 1024) + attention launch + out-projection
// (K = 1024): one launch and a quarter of the flops less per layer, and no q / attention-output round trip.
// Exact in exact arithmetic; in 16-bit operands the rounding points move from q and softmax(S) V to A and B (same count,
// same magnitude) -- bounded against the oracle by tests/test_gpu_parity16.py.  fp32 parity mode keeps the unfused order.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) _Float16 h8a_t;
typedef __attribute__((ext_vector_type(8))) __bf16 b8a_t;
typedef __attribute__((ext_vector_type(4))) float f4a_t;

namespace {

template <typename T>
__device__ inline f4a_t mfma_a(const uint4& a, const uint4& b, f4a_t c);
template <>
__device__ inline f4a_t mfma_a<F16T>(const uint4& a, const uint4& b, f4a_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8a_t*>(&a), *reinterpret_cast<const h8a_t*>(&b), c, 0, 0, 0);
}
template <>
__device__ inline f4a_t mfma_a<BF16T>(const uint4& a, const uint4& b, f4a_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const b8a_t*>(&a), *reinterpret_cast<const b8a_t*>(&b), c, 0, 0, 0);
}

// grid (H, n_seq * n_layers, 2 * NPART): blockIdx.z & 1 = 0 builds A and c, 1 builds B^T, of head blockIdx.x for (layer,
// sequence) blockIdx.y; blockIdx.z >> 1 = which NPART-th of the model dimension.  256 threads = 4 waves; wave w owns a
// quarter of the workgroup's columns.  (The loop is latency-bound -- global fragments, two MFMAs, a store -- so the
// columns are spread over 4x more workgroups than rows of work would need: 66 -> ~35 us per reverse step, round 3.)
// MFMA 16x16x32 with the gemm16 operand order: mfma(Pfrag, Qfrag) -> acc[r] = C[row of Q = l15][row of P = 4 lg + r].
template <typename T, int JT>      // JT = Lp / 16 key tiles (3 or 4)
__global__ __launch_bounds__(256) void absorb_kernel(const int64_t* tab_seq, const int64_t* tab_layer, int n_seq, int D, const int32_t* step,
                                                     float scale) {
    using st = typename T::storage;
    constexpr int Lp = JT * 16;
    const int h = blockIdx.x, ls = blockIdx.y, which = blockIdx.z & 1, part = blockIdx.z >> 1, npart = gridDim.z >> 1;
    const int layer = ls / n_seq;
    const int64_t* ts = tab_seq + (int64_t)ls * 8;
    const int64_t* tl = tab_layer + (int64_t)layer * 4;
    const int le = (int)ts[2];
    const int64_t stp = *step;
    const int H = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
    const int nq = D / (4 * npart);                         // model-dimension columns per wave
    const int col0 = part * (D / npart);                    // first column of this workgroup
    // output staging (round 5): with 64 columns per wave (the NAR geometry) a wave's output block is staged in its own LDS slice
    // and stored as 16-byte row chunks; SROW = LDS row stride in elements (64 + 8: the 8-byte writes of a 16-lane group spread over banks)
    constexpr int SROW = 72;
    __shared__ __attribute__((aligned(16))) unsigned char stage_lds[4 * 64 * SROW * 2];
    unsigned char* const stg = stage_lds + wave * (64 * SROW * 2);
    const bool staged = nq == 64 && (((uintptr_t)ts[4] | (uintptr_t)ts[6]) & 15) == 0 && (D & 7) == 0;
    // memory rows of this head at this step: [Le][64], 16-bit
    const st* mem = reinterpret_cast<const st*>(which == 0 ? ts[0] : ts[1]) + stp * ts[3] + (int64_t)h * le * 64;
    uint4 mf[JT][2];                                        // fragments of the memory rows: row 16 jt + l15, k chunk 4 ks + lg
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int j = jt * 16 + l15;
            mf[jt][ks] = (j < le) ? *reinterpret_cast<const uint4*>(mem + (int64_t)j * 64 + (ks * 4 + lg) * 8) : make_uint4(0, 0, 0, 0);
        }
    if (which == 0) {
        // A[h Lp + j][n] = scale * sum_d K[j][d] WqT[h][n][d]      (WqT: [H][D][64], d contiguous)
        const st* wq = reinterpret_cast<const st*>(tl[0]) + (int64_t)h * D * 64;
        st* A = reinterpret_cast<st*>(ts[4]) + (int64_t)h * Lp * D;
        // 4 column tiles per trip: their 8 weight fragments are requested before the first MFMA (the loop is latency-bound)
        for (int nt0 = 0; nt0 < nq / 16; nt0 += 4) {
            uint4 wf[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    wf[u][ks] = *reinterpret_cast<const uint4*>(wq + (int64_t)(col0 + wave * nq + min(nt0 + u, nq / 16 - 1) * 16 + l15) * 64 + (ks * 4 + lg) * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (nt0 + u >= nq / 16) break;
                const int n0 = col0 + wave * nq + (nt0 + u) * 16;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    f4a_t acc = {0.f, 0.f, 0.f, 0.f};
                    acc = mfma_a<T>(wf[u][0], mf[jt][0], acc);      // acc[r] = C[j = 16 jt + l15][n = n0 + 4 lg + r]
                    acc = mfma_a<T>(wf[u][1], mf[jt][1], acc);
                    st o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = T::from_f32(acc[r] * scale);
                    if (staged) *reinterpret_cast<uint2*>(stg + ((jt * 16 + l15) * SROW + u * 16 + lg * 4) * 2) = *reinterpret_cast<const uint2*>(o);
                    else *reinterpret_cast<uint2*>(A + (int64_t)(jt * 16 + l15) * D + n0 + lg * 4) = *reinterpret_cast<const uint2*>(o);
                }
            }
        }
        if (staged) {
            // the wave's Lp rows x 64 columns leave as whole 128-byte rows, 16 bytes per lane (straight from the accumulator layout a
            // store instruction wrote 16 rows x 32 bytes: 100 MB per step in 32-byte pieces)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int pass = 0; pass < Lp / 8; ++pass) {
                const int row = pass * 8 + (lane >> 3), ch = lane & 7;
                const uint4 v = *reinterpret_cast<const uint4*>(stg + (row * SROW + ch * 8) * 2);
                *reinterpret_cast<uint4*>(A + (int64_t)row * D + col0 + wave * nq + ch * 8) = v;
            }
        }
        // c[h Lp + j] = scale * K[j] . bq[h 64 ..]  (fp32);  padded keys: -1e30 (their softmax weight is exactly 0)
        if (part == 0 && threadIdx.x < Lp) {
            const int j = threadIdx.x;
            float cv = -1e30f;
            if (j < le) {
                const float* bq0 = reinterpret_cast<const float*>(tl[2]);       // nullptr: no query bias
                float s = 0.f;
                for (int d = 0; d < 64; ++d) s = fmaf(T::to_f32(mem[(int64_t)j * 64 + d]), bq0 ? bq0[h * 64 + d] : 0.f, s);
                cv = s * scale;
            }
            reinterpret_cast<float*>(ts[5])[h * Lp + j] = cv;
            // deferred LayerNorm (M5DeferredLN): the scores GEMM also needs s[n] = sum_k A[n][k] = scale K[j] . (row sums of Wq_h)
            // -- tab_layer[3] = fp32 [D] row sums of the (gamma-folded, operand-rounded) query weights, tab_seq[7] = s out
            if (ts[7] && tl[3]) {
                float sv = 0.f;
                if (j < le) {
                    const float* wsum = reinterpret_cast<const float*>(tl[3]);
                    float a = 0.f;
                    for (int d = 0; d < 64; ++d) a = fmaf(T::to_f32(mem[(int64_t)j * 64 + d]), wsum[h * 64 + d], a);
                    sv = a * scale;
                }
                reinterpret_cast<float*>(ts[7])[h * Lp + j] = sv;
            }
        }
    } else {
        // Bt[n][h Lp + j] = sum_d Wo[n][h 64 + d] V[j][d]
        const st* wo = reinterpret_cast<const st*>(tl[1]) + h * 64;
        st* Bt = reinterpret_cast<st*>(ts[6]) + h * Lp;
        const int ldb = H * Lp;
        for (int nt0 = 0; nt0 < nq / 16; nt0 += 4) {
            uint4 wf[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    wf[u][ks] = *reinterpret_cast<const uint4*>(wo + (int64_t)(col0 + wave * nq + min(nt0 + u, nq / 16 - 1) * 16 + l15) * D + (ks * 4 + lg) * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (nt0 + u >= nq / 16) break;
                const int n0 = col0 + wave * nq + (nt0 + u) * 16;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    f4a_t acc = {0.f, 0.f, 0.f, 0.f};
                    acc = mfma_a<T>(mf[jt][0], wf[u][0], acc);      // acc[r] = C[n = n0 + l15][j = 16 jt + 4 lg + r]
                    acc = mfma_a<T>(mf[jt][1], wf[u][1], acc);
                    st o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = T::from_f32(acc[r]);
                    if (staged) *reinterpret_cast<uint2*>(stg + ((u * 16 + l15) * SROW + jt * 16 + lg * 4) * 2) = *reinterpret_cast<const uint2*>(o);
                    else *reinterpret_cast<uint2*>(Bt + (int64_t)(n0 + l15) * ldb + jt * 16 + lg * 4) = *reinterpret_cast<const uint2*>(o);
                }
            }
        }
        if (staged) {
            // the wave's 64 rows x Lp columns leave as 16-byte chunks of whole Lp-column row segments
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            constexpr int CPR = Lp / 8;                         // 16-byte chunks per row
#pragma unroll
            for (int pass = 0; pass < CPR; ++pass) {
                const int c = pass * 64 + lane, row = c / CPR, ch = c - row * CPR;
                const uint4 v = *reinterpret_cast<const uint4*>(stg + (row * SROW + ch * 8) * 2);
                *reinterpret_cast<uint4*>(Bt + (int64_t)(col0 + wave * nq + row) * ldb + ch * 8) = v;
            }
        }
    }
}

}  // namespace

extern "C" int m5_xattn_absorb(int dtype, const int64_t* tab_seq, const int64_t* tab_layer, int n_layers, int n_seq, int n_heads,
                               int D, int Lp, const int32_t* step, float scale, void* stream) {
    if (!tab_seq || !tab_layer || !step || n_layers <= 0 || n_seq <= 0 || n_heads <= 0) return M5_ERR_ARG;
    if ((dtype != M5_F16 && dtype != M5_BF16) || (Lp != 48 && Lp != 64) || (D % 64)) return M5_ERR_UNSUPPORTED;
    const int npart = (D % 1024 == 0) ? 4 : ((D % 512 == 0) ? 2 : 1);        // 64 columns per wave
    const dim3 grid(n_heads, n_seq * n_layers, 2 * npart);
    hipStream_t s = (hipStream_t)stream;
#define M5_ABS(TT, JT) hipLaunchKernelGGL((absorb_kernel<TT, JT>), grid, dim3(256), 0, s, tab_seq, tab_layer, n_seq, D, step, scale)
    if (dtype == M5_F16) { if (Lp == 48) M5_ABS(F16T, 3); else M5_ABS(F16T, 4); }
    else { if (Lp == 48) M5_ABS(BF16T, 3); else M5_ABS(BF16T, 4); }
#undef M5_ABS
    M5_CHECK_LAUNCH();
    return M5_OK;
}
