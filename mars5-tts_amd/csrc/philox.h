// torch's device random numbers, reproduced bit for bit (csrc/nar_sample.hip has the launch geometry): Philox4x32-10 (rocrand's
// engine behind hiprand / torch), rocrand's uint -> (0, 1] float map, and the two ATen transforms this library needs.
// Device code under hipcc; the same text compiles as host code for tests/test_philox_cpu.py (Random123 known-answer vectors and
// the draw geometry against a Python restatement), which supplies uint2 / uint4 / make_uint2 / make_uint4 itself.
#pragma once
#ifdef __HIPCC__
#define M5_RNG_FN __device__ inline
#else
#include <math.h>
#define M5_RNG_FN static inline
#endif
M5_RNG_FN uint4 m5_philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long m0 = (unsigned long long)0xD2511F53u * c.x, m1 = (unsigned long long)0xCD9E8D57u * c.z;
        c = make_uint4((unsigned)(m1 >> 32) ^ c.y ^ k.x, (unsigned)m1, (unsigned)(m0 >> 32) ^ c.w ^ k.y, (unsigned)m0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
M5_RNG_FN float m5_curand_uniform(unsigned v) {              // rocrand_device::detail::uniform_distribution: (0, 1]
    return 2.3283064e-10f + ((float)v * 2.3283064e-10f);
}
M5_RNG_FN float m5_torch_uniform(unsigned v) {               // at::native uniform_kernel(0, 1): the bound reversed to [0, 1)
    const float u = m5_curand_uniform(v);
    return u == 1.0f ? 0.0f : u;
}
M5_RNG_FN float m5_torch_exponential1(unsigned v) {          // at::transformation::exponential(u, lambda = 1) on the device
    const float u = m5_curand_uniform(v);
    const float lg = (u >= 1.0f - 5.9604645e-08f) ? -5.9604645e-08f : logf(u);      // eps / 2 = 2^-24
    return (-1.0f / 1.0f) * lg;
}
// element e of a torch draw of n values made with generator state (seed, offset): which Philox call and which of its outputs
M5_RNG_FN unsigned m5_torch_draw_bits(unsigned long long seed, unsigned long long offset, unsigned long long e, unsigned grid_threads) {
    const unsigned long long it = e / (4ull * grid_threads), rem = e - it * 4ull * grid_threads;
    const unsigned ii = (unsigned)(rem / grid_threads), idx = (unsigned)(rem - (unsigned long long)ii * grid_threads);
    const unsigned long long c = offset / 4 + it;
    const uint4 r = m5_philox4x32_10(make_uint4((unsigned)c, (unsigned)(c >> 32), idx, 0u), make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
    return ii == 0 ? r.x : (ii == 1 ? r.y : (ii == 2 ? r.z : r.w));
}
