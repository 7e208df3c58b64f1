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

__device__ inline float lae(float a, float b) {   // diffuser.py:22-24: max + log(exp(a - max) + exp(b - max))
    // One of the two exponentials is exp(+0) = 1 exactly, and fp32 addition commutes, so exp(a - mx) + exp(b - mx) ==
    // 1 + exp(min - mx) bit for bit: one libm expf per call instead of two (the kernel is bound by its transcendentals).
    // (a = b = -inf: the reference computes exp(NaN) and returns NaN; min - mx is NaN here too.)
    const float mx = fmaxf(a, b), mn = fminf(a, b);
    return mx + logf(1.0f + expf(mn - mx));
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

    if (q == 0 && t > a.q0_override_steps) {
        result = xk;             // L0 override (diffuser.py:467-468): whatever this row would draw is replaced, so it draws nothing
    } else if (known) {
        if (t == 0) {
            result = xk;                                          // diffuser.py:387-388
        } else {
            const float* u2 = a.u2 + (int64_t)row * K;
            const float c4 = cst[4], c5 = cst[5];
            // log_add_exp(one_hot_log + c4, c5) takes only two values per row: evaluate each once
            // (same arithmetic per element as the reference, just not 1025 times)
            const float q_hit = lae(0.f + c4, c5), q_miss = lae(a.log_eps + c4, c5);
            // v_k = gumbel(u_k) + (k == x_known ? q_hit : q_miss), arg-max with the first index on ties.  Among the K - 1 "miss"
            // classes the constant is the same and gumbel is non-decreasing in u, so only each lane's LARGEST u (first index on
            // equal u) can win: 64 candidates + the hit class get the two logf, not 1025 (this branch is 71 % of the rows and the
            // kernel is bound by its transcendentals).  Exactness: a non-candidate could only win by TYING in v with a larger u
            // at a lower index; for u >= 0.9 adjacent floats u are >= 6 ulps apart in v (|dv/du| >= 10, libm logf <= 1 ulp), so
            // distinct u give distinct v there.  The row's largest u is below 0.9 with probability 0.9^1025 ~ 1e-47; the plain
            // loop over every class is kept for that case (wave-uniform branch).
            float ub = -1.f, uh = 0.f;
            int ui = 0x7fffffff;
            bool has_hit = false;
            float uu[MAXC];
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int k = lane + 64 * i;
                uu[i] = (k < K) ? u2[k] : -1.f;
            }
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int k = lane + 64 * i;
                if (k < K) {
                    if (k == (int)xk) { has_hit = true; uh = uu[i]; }
                    else if (uu[i] > ub) { ub = uu[i]; ui = k; }        // ascending k: the first index of equal u stays
                }
            }
            float best = -INFINITY;
            int bi = 0x7fffffff;
            if (wave_max(ub) >= 0.9f) {
                if (ui != 0x7fffffff) { best = gumbel(ub) + q_miss; bi = ui; }
                if (has_hit) {
                    const float vh = gumbel(uh) + q_hit;
                    if (vh > best || (vh == best && (int)xk < bi)) { best = vh; bi = (int)xk; }
                }
            } else {
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int k = lane + 64 * i;
                    if (k < K) {
                        const float v = gumbel(uu[i]) + ((k == (int)xk) ? q_hit : q_miss);
                        if (v > best) { best = v; bi = k; }
                    }
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
        // the row's uniforms are requested with the logits (they are consumed four reductions later: requested there, the wave
        // sat through a dependent HBM round trip in the middle of the chain)
        float uu[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) uu[i] = (lane + 64 * i < K) ? u1[lane + 64 * i] : 0.5f;
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
                const float v = gumbel(uu[i]) + (z[i] - lse);       // log_sample_categorical (:219-228)
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

// ---- The step's uniforms, generated in this library exactly as `torch.rand` generates them on this device (reference
// diffuser.py:219-228, 380-390: `torch.rand_like(logits)` once for log_sample_categorical of the model branch, once for q_sample
// of the known branch), so the two ATen `uniform_` launches, their 2 x 4 n bytes and the eager launch gaps around them leave
// the loop and the WHOLE reverse step (forward + uniforms + posterior / sample + step counter) is one captured hipGraph.
// torch (ATen/native/cuda/DistributionTemplates.h, distribution_nullary_kernel; rocrand's Philox4x32-10 behind hiprand):
//   G = 256 * min(CUs * (maxThreadsPerMultiProcessor / 256), ceil(n / 256)) threads, thread idx starts Philox(seed,
//   subsequence idx, offset) and per loop iteration `it` takes ONE 4-output call for elements
//   e = it * 4 G + ii * G + idx, ii = 0..3: counter = {offset / 4 + it (64 bit), idx (64 bit)}, key = seed;
//   u = 2^-32 + float(x) * 2^-32 (rocrand uniform_distribution: (0, 1]), then 1.0 -> 0.0 (uniform_kernel's bound reversal).
// A draw advances the generator by inc = (ceil-div(n, 4 G)) * 4.  Reverse step i makes draw 1 at offset0 + 2 i inc and --
// unless t = 0 -- draw 2 at offset0 + (2 i + 1) inc (only the LAST step has t = 0); {seed, offset0} live in DEVICE memory
// (rng[0..1]), so a captured step graph serves any generator state the host binds a run to.
// The posterior / sample kernel reads draw 1 on the rows it samples from the model (m = 0) and draw 2 on the known rows, so
// ONE merged buffer is written: out[e] = m[row(e)] ? u2[e] : u1[e]  (m = NULL: draw 1 everywhere = torch.rand itself).
__global__ __launch_bounds__(256) void nar_uniform_kernel(M5NarUniformArgs a) {
    const unsigned G = a.grid_threads;
    const unsigned it = blockIdx.y, idx = blockIdx.x * 256u + threadIdx.x;         // grid = (G / 256, iterations): no division
    const long long e0 = (long long)it * 4 * G + idx;
    if (e0 >= a.n) return;
    const int step = a.step ? a.step[0] : 0;
    const bool second = a.m && !(a.consts && (int)a.consts[(long long)step * M5_NAR_CONSTS + 6] == 0);      // t = 0: one draw only
    const unsigned long long seed = a.rng[0], off1 = a.rng[1] + (unsigned long long)step * 2ull * a.inc;
    const uint2 key = make_uint2((unsigned)seed, (unsigned)(seed >> 32));
    // which draw each of this thread's four elements takes: a wave's 64 threads cover 64 consecutive elements of four rows, so
    // most waves need only ONE of the two Philox calls (prompt frames: all known; generated frames: known for codebook 0 only) --
    // the calls are skipped wave-uniformly (twenty 32 x 32 -> 64-bit multiplies each: they are this kernel's time)
    bool known[4] = {false, false, false, false}, need1 = false, need2 = false;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const long long e = e0 + (long long)ii * G;
        if (e < a.n) {
            if (second) {
                // row = e / K by the host's multiply-shift (exact for e < 2^32, checked there), or a plain division
                const unsigned row = a.k_magic ? (unsigned)(((unsigned long long)(unsigned)e * a.k_magic) >> (32 + a.k_shift)) : (unsigned)(e / a.K);
                known[ii] = a.m[row] != 0;
            }
            need2 = need2 || known[ii];
            need1 = need1 || !known[ii];
        }
    }
    uint4 r1 = make_uint4(0u, 0u, 0u, 0u), r2 = r1;
    if (__ballot(need1) != 0) {
        const unsigned long long c1 = off1 / 4 + it;
        r1 = m5_philox4x32_10(make_uint4((unsigned)c1, (unsigned)(c1 >> 32), idx, 0u), key);
    }
    if (__ballot(need2) != 0) {
        const unsigned long long c2 = (off1 + a.inc) / 4 + it;
        r2 = m5_philox4x32_10(make_uint4((unsigned)c2, (unsigned)(c2 >> 32), idx, 0u), key);
    }
    const unsigned v1[4] = {r1.x, r1.y, r1.z, r1.w}, v2[4] = {r2.x, r2.y, r2.z, r2.w};
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const long long e = e0 + (long long)ii * G;
        if (e < a.n) {
            const unsigned v = known[ii] ? v2[ii] : v1[ii];
            a.out[e] = a.transform == 1 ? m5_torch_exponential1(v) : m5_torch_uniform(v);
        }
    }
}

}  // namespace

extern "C" int m5_nar_uniforms(const M5NarUniformArgs* a, void* stream) {
    if (!a || !a->out || !a->rng || a->n <= 0 || a->n >= (1ll << 32) || a->grid_threads == 0 || (a->grid_threads % 256) || (a->inc % 4)) return M5_ERR_ARG;
    if (a->m && (a->K <= 0 || !a->consts || !a->step)) return M5_ERR_ARG;
    if (a->transform != 0 && a->transform != 1) return M5_ERR_ARG;
    if (a->inc == 0) return M5_ERR_ARG;           // (two draws at the same offset would be identical)
    if (a->m && a->k_magic) {
        // the multiply-shift must equal e / K for EVERY e < n: with m K = 2^(32+s) + r, r >= 0, (e m) >> (32+s) = floor(e / K + e r / (K 2^(32+s)))
        // and the floor is unchanged iff e r < 2^(32+s) -- the closed-form bound, not a sample of row boundaries (a wrong magic would
        // index m[] out of bounds)
        if (a->k_shift > 31) return M5_ERR_ARG;
        const unsigned __int128 one = (unsigned __int128)1 << (32 + a->k_shift), mk = (unsigned __int128)a->k_magic * (unsigned)a->K;
        if (mk < one || (mk - one) * (unsigned __int128)(a->n - 1) >= one) return M5_ERR_ARG;
    }
    const long long G = a->grid_threads;
    const long long iters = (a->n + 4 * G - 1) / (4 * G);
    if (iters > 65535) return M5_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(nar_uniform_kernel, dim3((unsigned)(G / 256), (unsigned)iters), dim3(256), 0, (hipStream_t)stream, *a);
    M5_CHECK_LAUNCH();
    return M5_OK;
}

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
