// NAR reverse-diffusion step tail (reference diffuser.py:364-393 + :467-468) as ONE kernel.
//
// One wave per (frame s, codebook q) row of K = 1025 classes, 17 classes per lane held in
// registers for the whole chain: CFG mix -> /temperature -> log_softmax -> q_posterior
// (two log_add_exp) -> logsumexp normalise -> Gumbel-argmax with the caller's uniforms;
// known rows (m = 1) take the q_sample branch instead and never touch the logits.
// HBM-bound: reads 2 logit rows + 1 uniform row per unknown row, 1 uniform row per known
// row, writes one int64.  The reference materialises ~10 (1,S,8,1025) fp32 temporaries.
// Compiled with -ffp-contract=off: every mul/add rounds separately like the reference's
// chain of elementwise ATen ops.
#include "common.h"

namespace {

constexpr int MAXC = 17;   // K <= 64*17 = 1088

__device__ inline float lae(float a, float b) {   // diffuser.py:22-24
    const float mx = fmaxf(a, b);
    return mx + logf(expf(a - mx) + expf(b - mx));
}
__device__ inline float gumbel(float u) {          // diffuser.py:225-226
    return -logf(fmaxf(-logf(fmaxf(u, 1e-7f)), 1e-7f));
}

__global__ __launch_bounds__(256) void nar_sample_kernel(M5NarSampleArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.S * a.n_q) return;
    const int s = row / a.n_q, q = row - s * a.n_q;
    const int K = a.K;
    const float* cst = a.consts + (int64_t)a.step[0] * M5_NAR_CONSTS;
    const int t = (int)cst[6];
    const int64_t xk = a.x_known[row];
    const bool known = a.m[row] != 0;
    int64_t result;

    if (known) {
        if (t == 0) {
            result = xk;                                          // diffuser.py:387-388
        } else {
            const float* u2 = a.u2 + (int64_t)row * K;
            const float c4 = cst[4], c5 = cst[5];
            // log_add_exp(one_hot_log + c4, c5) takes only two values per row: evaluate each once
            // (same arithmetic per element as the reference, just not 1025 times)
            const float q_hit = lae(0.f + c4, c5), q_miss = lae(a.log_eps + c4, c5);
            float best = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int k = lane + 64 * i;
                if (k < K) {
                    const float v = gumbel(u2[k]) + ((k == (int)xk) ? q_hit : q_miss);
                    if (v > best) { best = v; bi = k; }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ob = __shfl_xor(best, off);
                const int oi = __shfl_xor(bi, off);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            result = bi;
        }
    } else {
        // unknown rows only exist for codebooks >= 1 past the prompt (m covers the rest)
        const int64_t lrow = (int64_t)(s - a.row_offset) * a.ld_row + (int64_t)(q - 1) * a.ld_q;
        const float* zc = a.logits_c + lrow;
        const float* zu = a.logits_u ? a.logits_u + lrow : nullptr;
        const float* u1 = a.u1 + (int64_t)row * K;
        const int xt = (int)a.x[row];
        const float w = a.guidance_w, w1 = 1.0f - a.guidance_w;
        const float invT = 1.0f / a.temperature;
        float z[MAXC];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int k = lane + 64 * i;
            z[i] = -INFINITY;
            if (k < K) {
                float v = zc[k];
                if (zu) {                                          // diffuser.py:360-364
                    const float ca = w * v;
                    const float cb = w1 * zu[k];
                    v = ca + cb;
                }
                v = a.div_mode ? v * invT : v / a.temperature;     // :366
                z[i] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = wave_max(mx);
        float se = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
            if (lane + 64 * i < K) se += expf(z[i] - mx);
        const float lse0 = logf(wave_sum(se));
        const float c0 = cst[0], c1 = cst[1], c2 = cst[2], c3 = cst[3];
        const float p_hit = lae(0.f + c2, c3), p_miss = lae(a.log_eps + c2, c3);   // two values per row, as above
        float mu = -INFINITY;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int k = lane + 64 * i;
            if (k < K) {
                const float l0 = (z[i] - mx) - lse0;               // log_softmax (:367)
                const float ev = (t == 0) ? l0 : lae(l0 + c0, c1); // q_pred(t-1) / where(t==0) (:187-193)
                const float un = ev + ((k == xt) ? p_hit : p_miss); // + q_pred_one_timestep(index_to_log_onehot(x_t)) (:368, :199)
                z[i] = un;
                mu = fmaxf(mu, un);
            }
        }
        mu = wave_max(mu);
        float su = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
            if (lane + 64 * i < K) su += expf(z[i] - mu);
        const float lse = logf(wave_sum(su)) + mu;                  // torch.logsumexp (:204)
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int k = lane + 64 * i;
            if (k < K) {
                const float v = gumbel(u1[k]) + (z[i] - lse);       // log_sample_categorical (:219-228)
                if (v > best) { best = v; bi = k; }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        result = bi;
    }
    // L0 override while t > q0_override_steps (diffuser.py:467-468); x_quant0 == x_known[...,0]
    if (q == 0 && t > a.q0_override_steps) result = xk;
    if (lane == 0) a.x[row] = result;
}

}  // namespace

extern "C" int m5_nar_sample(const M5NarSampleArgs* a, void* stream) {
    if (!a || !a->logits_c || !a->x || !a->x_known || !a->m || !a->u1 || !a->u2 || !a->consts || !a->step) return M5_ERR_ARG;
    if (a->K <= 0 || a->K > 64 * MAXC || a->S <= 0 || a->n_q <= 1 || a->row_offset < 0 || a->row_offset > a->S) return M5_ERR_ARG;
    if (!(a->temperature > 0.f)) return M5_ERR_ARG;
    if (a->guidance_w != 1.0f && !a->logits_u) return M5_ERR_ARG;
    M5NarSampleArgs b = *a;
    if (b.guidance_w == 1.0f) b.logits_u = nullptr;
    const int rows = a->S * a->n_q;
    hipLaunchKernelGGL(nar_sample_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, b);
    M5_CHECK_LAUNCH();
    return M5_OK;
}
